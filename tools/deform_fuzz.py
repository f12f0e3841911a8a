"""random DeformableConv2d geometries (channels, taps, stride, padding, map sizes, batch) in both deform modes and every compute mode
against the fp64 torch restatement (mode bf16: looser bound, bf16 operands).  tools/deform_fuzz.py [n] [seed]"""
import sys, os, numpy as np, torch
import torch.nn.functional as F
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import candle_birefnet_amd as cb
from candle_birefnet_amd import ops
import torch_ref as R
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
TOL = {"f32": 2e-4, "f32_split3": 2e-4, "f32_half2": 2e-4, "f32_split2": 5e-4, "bf16": 3e-2, "f16": 4e-3}
bad = 0
for it in range(n):
    C = int(rng.choice([32, 64, 96, 128])); O = int(rng.choice([8, 32, 64, 100, 128, 256])); k = int(rng.choice([1, 3, 3, 7])); s = int(rng.choice([1, 1, 2]))
    pad = int(rng.choice([0, k // 2])); B = int(rng.integers(1, 4)); H = int(rng.integers(max(k, 3), 40)); W = int(rng.integers(max(k, 3), 40))
    if H + 2 * pad < k or W + 2 * pad < k: continue
    g = lambda *sh, std=1.0: (rng.standard_normal(sh) * std).astype(np.float32)
    t = {"offset_conv.weight": g(2 * k * k, C, k, k, std=1.5 * (C * k * k) ** -0.5), "offset_conv.bias": g(2 * k * k, std=0.3),
         "modulator_conv.weight": g(k * k, C, k, k, std=(C * k * k) ** -0.5), "modulator_conv.bias": g(k * k, std=0.1),
         "regular_conv.weight": g(O, C, k, k, std=(C * k * k) ** -0.5), "regular_conv.bias": g(O, std=0.1)}
    x = g(B, C, H, W)
    xt = torch.from_numpy(x).double(); td = {m: torch.from_numpy(a).double() for m, a in t.items()}
    for dm in ("reference_cpu", "deformable"):
        if dm == "reference_cpu":
            ref = F.conv2d(xt, td["regular_conv.weight"], td["regular_conv.bias"], stride=s, padding=pad)
        else:
            off = F.conv2d(xt, td["offset_conv.weight"], td["offset_conv.bias"], stride=s, padding=pad)
            msk = 2.0 / (torch.exp(-F.conv2d(xt, td["modulator_conv.weight"], td["modulator_conv.bias"], stride=s, padding=pad)) + 1.0)
            ref = R.deform_conv2d(xt, off, msk, td["regular_conv.weight"], td["regular_conv.bias"], s, pad)
        ref = ref.numpy()
        layer = cb.DeformableConv2d.new(C, O, k, s, pad, cb.VarBuilder.from_tensors(t), mode=dm)
        desc = f"{dm} B{B} C{C} O{O} k{k} s{s} p{pad} {H}x{W}"
        for mode in ("f32", "f32_split3", "f32_half2", "f32_split2", "bf16", "f16"):
            ops.set_compute(mode)
            try:
                y = np.asarray(layer.forward(x), np.float64)
                err = float(np.abs(y - ref).max() / max(1.0, np.abs(ref).max()))
                if y.shape != ref.shape or not np.isfinite(y).all() or err > TOL[mode]:
                    bad += 1; print("BAD ", mode, desc, f"err {err:.2e}", flush=True)
            except Exception as e:
                bad += 1; print("FAIL", mode, desc, str(e)[:200], flush=True)
        ops.set_compute("f32")
print(f"{n} cases, {bad} problems")
