#!/usr/bin/env python3
"""Ablation of the bf16-storage GEMM (diag build): which part of a launch costs what.  BRN_GEMM_ABLATE bits: 1 no A loads, 2 no W loads,
4 no fragment reads / MFMA, 8 no epilogue."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_birefnet_amd as cb

shapes = [(40960, 2304, 768)]
cfgs = [0, 2]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
if os.environ.get("BRN_SWEEP_CFGS"):
    cfgs = [int(c) for c in os.environ["BRN_SWEEP_CFGS"].split(",")]
for (M, N, K) in shapes:
    for cfg in cfgs:
        row = []
        for abl in (0, 8, 16, 7, 7 + 16, 15, 3, 3 + 8, 4, 4 + 8, 1, 2):
            os.environ["BRN_GEMM_ABLATE"] = str(abl)
            ms = C.c_float(0)
            st = cb._ffi.lib.brn_gemm_microbench(M, N, K, 4000 + cfg, 1, 10, 0, C.byref(ms))
            row.append(f"abl{abl}: {ms.value * 1e3:7.1f}us" if st == 0 else f"abl{abl}: err")
        print(f"{M}x{N}x{K} cfg{cfg}  " + " | ".join(row), flush=True)
