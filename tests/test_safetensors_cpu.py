"""safetensors -> VarBuilder (infer_image.rs:35-40).  Round trip of a checkpoint-shaped file on the CPU."""
import numpy as np


def test_varbuilder_from_safetensors_roundtrip(tmp_path):
    from safetensors.numpy import save_file
    import candle_birefnet_amd as cb
    cfg = cb.BiRefNetConfig()
    cfg.swin.depths = [1, 1, 1, 1]
    spec = cb.birefnet_weight_spec(cfg)
    small = [(n, s, k) for n, s, k in spec if int(np.prod(s)) <= 20000][:40]
    w = cb.synth_weights(small, seed=7)
    p = str(tmp_path / "model.safetensors")
    save_file(w, p)
    vb = cb.VarBuilder.from_safetensors(p)
    for n, s, _ in small:
        head, _, leaf = n.rpartition(".")
        b = vb
        for part in head.split("."):
            b = b.pp(part)
        np.testing.assert_array_equal(b.get(s, leaf), w[n])
