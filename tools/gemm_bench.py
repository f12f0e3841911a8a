#!/usr/bin/env python3
"""GEMM tile / split-K sweep on the shapes of the path (tuning aid; uses brn_gemm_microbench)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_birefnet_amd as cb

SHAPES = [(4096, 3072, 768), (4096, 768, 3072), (4096, 2304, 768), (4096, 768, 768), (1024, 768, 3072), (1024, 3072, 768), (1024, 2304, 768),
          (1024, 768, 768), (65536, 768, 192), (65536, 192, 768), (65536, 576, 192), (16384, 1536, 384), (16384, 384, 1536), (1024, 64, 51840),
          (1024, 1536, 6144), (256, 1536, 6144), (65536, 256, 3136), (65536, 64, 4320)]


def run(M, N, K, cfg=-1, sk=1, iters=20):
    ms = C.c_float(0)
    cb._ffi.check(cb._ffi.lib.brn_gemm_microbench(M, N, K, cfg, sk, iters, 0, C.byref(ms)))
    return ms.value


if __name__ == "__main__":
    print(f"{'M':>6} {'N':>5} {'K':>6} | plan     | 128x128  | 128x64   | 64x64    | 128² 8w24 | 256x128  | 128² 8w42 (TF/s)")
    for M, N, K in SHAPES:
        fl = 2.0 * M * N * K / 1e9
        cells = [run(M, N, K)] + [run(M, N, K, c, 1) for c in (0, 1, 2, 3, 4, 5)]
        print(f"{M:6d} {N:5d} {K:6d} | " + " | ".join(f"{fl / ms:8.1f}" for ms in cells), flush=True)
