"""fp32-parity error of the split GEMM modes on random data (op-level linear, vs fp64)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_birefnet_amd as cb
from candle_birefnet_amd import ops
rng = np.random.default_rng(0)
for (M, N, K) in [(5120, 3072, 768), (640, 256, 768), (5120, 768, 3072)]:
    x = rng.standard_normal((M, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    ref = x.astype(np.float64) @ w.astype(np.float64).T + b
    for mode in ("f32", "f32_split3", "f32_split2", "bf16"):
        ops.set_compute(mode)
        y = ops.linear(torch.from_numpy(x).cuda(), w, b).cpu().numpy().astype(np.float64)
        e = np.abs(y - ref)
        print(f"{M}x{N}x{K} {mode:14s} max abs {e.max():.3e} rms {np.sqrt((e**2).mean()):.3e}  (ref rms {np.sqrt((ref**2).mean()):.2f})", flush=True)
ops.set_compute("f32")
