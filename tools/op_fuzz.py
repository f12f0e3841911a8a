"""random shapes through the op-level conv2d / linear entry points in every compute mode against torch fp64 on the operands the mode
multiplies (bf16-rounded in mode bf16): launch refusals, NaNs and errors beyond the mode's bound are printed.  tools/op_fuzz.py [n] [seed]"""
import sys, os, numpy as np, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from candle_birefnet_amd import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
def r16(a, mode="bf16"): return torch.from_numpy(a).to(torch.float16 if mode == "f16" else torch.bfloat16).to(torch.float64)
TOL = {"f32": 2e-5, "f32_split3": 2e-5, "f32_half2": 2e-5, "f32_split2": 2e-4, "bf16": 1.2e-2, "f16": 1.5e-3}
bad = 0
for it in range(n):
    kind = "conv" if rng.random() < 0.7 else "linear"
    act = [None, "relu", "gelu_erf"][rng.integers(3)]
    if kind == "conv":
        B = int(rng.integers(1, 4)); Cin = int(rng.choice([3, 16, 31, 32, 48, 64, 65, 96, 128, 130, 192, 256, 480]))
        H, W = int(rng.integers(5, 72)), int(rng.integers(5, 72)); O = int(rng.choice([1, 3, 16, 17, 63, 64, 96, 130, 192, 256, 300, 384, 512]))
        k = int(rng.choice([1, 3, 5, 7])); s = int(rng.choice([1, 1, 2])); dil = int(rng.choice([1, 1, 2])) if k == 3 else 1
        pad = int(rng.choice([0, dil * (k // 2)]))
        if H + 2 * pad < dil * (k - 1) + 1 or W + 2 * pad < dil * (k - 1) + 1: continue
        x = rng.standard_normal((B, Cin, H, W), dtype=np.float32); w = (rng.standard_normal((O, Cin, k, k), dtype=np.float32) * (Cin * k * k) ** -0.5)
        b = rng.standard_normal(O, dtype=np.float32) * 0.1 if rng.random() < 0.7 else None
        desc = f"conv B{B} Cin{Cin} {H}x{W} O{O} k{k} s{s} p{pad} d{dil} act={act} bias={b is not None}"
    else:
        M = int(rng.choice([1, 7, 64, 100, 144, 1000, 4097, 20000, 40000, 70001])); K = int(rng.choice([32, 64, 96, 192, 384, 768, 1000 // 32 * 32, 3072])); N = int(rng.choice([1, 16, 64, 100, 192, 384, 576, 768, 1000]))
        x = rng.standard_normal((M, K), dtype=np.float32); w = rng.standard_normal((N, K), dtype=np.float32) * K ** -0.5
        b = rng.standard_normal(N, dtype=np.float32) * 0.1 if rng.random() < 0.7 else None
        res = rng.standard_normal((M, N), dtype=np.float32) if rng.random() < 0.4 else None
        desc = f"linear M{M} K{K} N{N} act={act} bias={b is not None} res={res is not None}"
    for mode in ("f32", "f32_split3", "f32_half2", "f32_split2", "bf16", "f16"):
        ops.set_compute(mode)
        try:
            if kind == "conv":
                y = np.asarray(ops.conv2d(x, w, b, stride=s, padding=pad, dilation=dil, act=act), np.float64)
                xr, wr = (r16(x, mode), r16(w, mode)) if mode in ("bf16", "f16") else (torch.from_numpy(x).double(), torch.from_numpy(w).double())
                ref = F.conv2d(xr, wr, None if b is None else torch.from_numpy(b).double(), stride=s, padding=pad, dilation=dil)
            else:
                y = np.asarray(ops.linear(x, w, b, act=act, residual=res), np.float64)
                xr, wr = (r16(x, mode), r16(w, mode)) if mode in ("bf16", "f16") else (torch.from_numpy(x).double(), torch.from_numpy(w).double())
                ref = xr @ wr.T + (0 if b is None else torch.from_numpy(b).double())
            ref = F.gelu(ref) if act == "gelu_erf" else F.relu(ref) if act == "relu" else ref
            if kind == "linear" and res is not None: ref = ref + torch.from_numpy(res).double()
            ref = ref.numpy()
            err = float(np.abs(y - ref).max() / max(1.0, np.abs(ref).max()))
            if not np.isfinite(y).all() or err > TOL[mode]:
                bad += 1; print("BAD ", mode, desc, f"err {err:.2e}", flush=True)
        except Exception as e:
            bad += 1; print("FAIL", mode, desc, str(e)[:220], flush=True)
    ops.set_compute("f32")
print(f"{n} cases, {bad} problems")
