import sys, numpy as np, torch
sys.path.insert(0, ".")
import candle_birefnet_amd as cb
for dm in ("reference_cpu", "deformable"):
    cfg = cb.BiRefNetConfig(deform_mode=dm)
    w = cb.synth_weights(cb.birefnet_weight_spec(cfg), seed=42)
    ms = {mode: cb.BiRefNet.new(cfg, cb.VarBuilder.from_tensors(w), compute=mode) for mode in ("f32_split3", "bf16")}
    for (H, W, B) in [(608, 608, 1), (736, 864, 2), (992, 1120, 1), (1184, 1056, 1), (672, 672, 5), (800, 1312, 4), (1504, 1504, 1), (416, 1760, 3)]:
        x = torch.from_numpy(cb.synth_input(B, H, W)).cuda()
        try:
            ys = {k: m.forward_logits(x).float().cpu().numpy() for k, m in ms.items()}
        except Exception as e:
            print("FAIL", dm, H, W, B, str(e)[:300], flush=True); continue
        d = float(np.abs(ys["bf16"] - ys["f32_split3"]).max())
        print(dm, H, W, B, "finite", bool(np.isfinite(ys["bf16"]).all() and np.isfinite(ys["f32_split3"]).all()), "bf16 vs f32_split3 max diff %.3e" % d, flush=True)
    for m in ms.values(): m.close()
