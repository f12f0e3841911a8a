import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import candle_birefnet_amd as cb
import torch_ref as R
for dm in ("reference_cpu", "deformable"):
    cfg = cb.BiRefNetConfig(deform_mode=dm); cfg.swin.depths = [2, 2, 2, 2]
    w = cb.synth_weights(cb.birefnet_weight_spec(cfg), seed=42)
    for (B, H, W) in [(1, 64, 64), (2, 352, 544), (3, 96, 32)]:
        x = cb.synth_input(B, H, W)
        ref, parts = R.forward_logits(x, w, cfg, torch.float64, return_parts=True)
        for mode in ("f32", "f32_split3", "f32_split2", "bf16"):
            try:
                m = cb.BiRefNet.new(cfg, cb.VarBuilder.from_tensors(w), compute=mode)
                feats = m.backbone.forward(x)
                e1 = max(float(np.abs(a - b.numpy()).max() / max(1.0, float(b.abs().max()))) for a, b in zip(feats, parts["f"]))
                x4s = m.squeeze_module.forward(parts["x4"].float().numpy())
                e2 = float(np.abs(x4s - parts["x4s"].numpy()).max() / max(1.0, float(parts["x4s"].abs().max())))
                out = m.decoder.forward(x, *[parts[k].float().numpy() for k in ("x1", "x2", "x3", "x4s")])
                e3 = float(np.abs(out - ref.numpy()).max())
                full = float(np.abs(m.forward_logits(x) - ref.numpy()).max())
                print(dm, B, H, W, mode, f"backbone {e1:.1e} squeeze {e2:.1e} decoder {e3:.1e} full {full:.1e}", flush=True)
                m.close()
            except Exception as e:
                print(dm, B, H, W, mode, "FAIL", str(e)[:200], flush=True)
