"""Yardstick only (never on the product path): the vendor library's bf16 GEMM (torch.matmul -> hipBLASLt / rocBLAS) on the shapes of the
path at batch 8, plain C = A W^T with bf16 output, random data, back-to-back launches."""
import time
import torch
SHAPES = [(40960, 2304, 768), (40960, 768, 768), (40960, 3072, 768), (40960, 768, 3072), (655360, 768, 192), (655360, 192, 768),
          (163840, 1536, 384), (163840, 384, 1536), (10240, 6144, 1536), (10240, 1536, 6144)]
for M, N, K in SHAPES:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(3):
        c = a @ w.t()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        c = a @ w.t()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{M:7d} x {N:5d} x {K:5d}   {dt * 1e6:8.1f} us   {2.0 * M * N * K / dt / 1e12:7.1f} TF/s", flush=True)
