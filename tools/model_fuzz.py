"""random (batch, H, W) through BiRefNet.forward_logits in modes bf16 / f16 / f32_split2 / f32_split3 / f32_half2 against mode f32: refused launches, NaNs,
disagreement beyond the modes' bounds.  tools/model_fuzz.py [n] [seed] [deform_mode]"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_birefnet_amd as cb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dm = sys.argv[3] if len(sys.argv) > 3 else "reference_cpu"
cfg = cb.BiRefNetConfig(deform_mode=dm)
w = cb.synth_weights(cb.birefnet_weight_spec(cfg), seed=42)
BOUND = {"f32_split3": 1e-4, "f32_half2": 1e-4, "f32_split2": 2e-4, "bf16": 2.4e-2, "f16": 3e-3}
ms = {mode: cb.BiRefNet.new(cfg, cb.VarBuilder.from_tensors(w), compute=mode) for mode in ("f32", "f32_split3", "f32_half2", "f32_split2", "bf16", "f16")}
bad = 0
for it in range(n):
    H, W = 32 * int(rng.integers(1, 45)), 32 * int(rng.integers(1, 45))
    B = int(rng.integers(1, max(2, min(9, (1536 * 1536) // (H * W) + 1))))
    x = torch.from_numpy(cb.synth_input(B, H, W)).cuda()
    try:
        ref = ms["f32"].forward_logits(x).float().cpu().numpy()
    except Exception as e:
        bad += 1; print("FAIL f32", B, H, W, str(e)[:200], flush=True); continue
    for mode in ("f32_split3", "f32_half2", "f32_split2", "bf16", "f16"):
        try:
            y = ms[mode].forward_logits(x).float().cpu().numpy()
            d = float(np.abs(y - ref).max())
            if not np.isfinite(y).all() or d > BOUND[mode]:
                bad += 1; print("BAD ", mode, B, H, W, f"diff {d:.2e}", flush=True)
        except Exception as e:
            bad += 1; print("FAIL", mode, B, H, W, str(e)[:200], flush=True)
print(f"{n} cases [{dm}], {bad} problems")
