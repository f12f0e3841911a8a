import sys, numpy as np, torch
sys.path.insert(0, ".")
import candle_birefnet_amd as cb
for dm in ("reference_cpu", "deformable"):
    cfg = cb.BiRefNetConfig(deform_mode=dm)
    w = cb.synth_weights(cb.birefnet_weight_spec(cfg), seed=42)
    ms = {mode: cb.BiRefNet.new(cfg, cb.VarBuilder.from_tensors(w), compute=mode) for mode in ("f32", "f32_split3", "f32_split2", "bf16")}
    for (H, W, B) in [(32, 32, 1), (64, 32, 3), (96, 160, 2), (224, 224, 7), (352, 544, 1), (512, 512, 16), (1056, 1056, 2), (2048, 1024, 1), (2080, 2080, 1)]:
        x = torch.from_numpy(cb.synth_input(B, H, W)).cuda()
        out = []
        ys = {}
        for k, m in ms.items():
            try:
                ys[k] = m.forward_logits(x).float().cpu().numpy()
            except Exception as e:
                out.append(f"FAIL[{k}] {str(e)[:200]}")
        if "f32" in ys:
            for k in ys:
                if k != "f32": out.append(f"{k}-f32 {np.abs(ys[k]-ys['f32']).max():.2e}")
        print(dm, H, W, B, "finite", all(np.isfinite(v).all() for v in ys.values()), " ".join(out), flush=True)
    for m in ms.values(): m.close()
