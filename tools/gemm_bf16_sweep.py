#!/usr/bin/env python3
"""Tile / stage sweep of the bf16-storage GEMM (kernels/gemm_bf16.hip) on the shapes of the path at B=8 (BASELINE configs[2]).
Needs the diag build:  make -C candle_birefnet_amd/csrc diag;  BRN_LIB_PATH=candle_birefnet_amd/libbirefnet_hip_diag.so python tools/gemm_bf16_sweep.py"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_birefnet_amd as cb

SHAPES = [(40960, 2304, 768), (40960, 768, 768), (40960, 3072, 768), (40960, 768, 3072),      # stage 2 (18 blocks)
          (655360, 576, 192), (655360, 192, 192), (655360, 768, 192), (655360, 192, 768),     # stage 0
          (163840, 1152, 384), (163840, 1536, 384), (163840, 384, 1536),                      # stage 1
          (10240, 4608, 1536), (10240, 6144, 1536), (10240, 1536, 6144),                      # stage 3
          (5120, 2304, 768), (5120, 3072, 768), (5120, 768, 3072)]                            # stage 2 at B=1
CFGS = {0: "128x128", 1: "128x64", 2: "256x256", 3: "256x192", 12: "256x128", 19: "256x256 4w", -1: "plan"}
if os.environ.get("BRN_SWEEP_CFGS"):
    CFGS = {int(c): CFGS.get(int(c), str(c)) for c in os.environ["BRN_SWEEP_CFGS"].split(",")}
if len(sys.argv) > 1:
    SHAPES = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for (M, N, K) in SHAPES:
    row = []
    for cfg, name in CFGS.items():
        ms = C.c_float(0)
        code = 4999 if cfg < 0 else 4000 + cfg
        st = cb._ffi.lib.brn_gemm_microbench(M, N, K, code, 1, 10, 0, C.byref(ms))
        if st != 0:
            row.append(f"{name}: err")
            continue
        row.append(f"{name}: {2.0 * M * N * K / ms.value / 1e9:7.1f}")
    print(f"{M:7d} x {N:5d} x {K:5d}  TF/s  " + " | ".join(row), flush=True)
