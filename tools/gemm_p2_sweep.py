#!/usr/bin/env python3
"""warp-specialised split2 GEMM: fp32 A split while staging (cfg 2006) vs A already in the P2 layout (cfg 2009)"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_birefnet_amd as cb
def run(M, N, K, cfg=-1, sk=1, iters=20):
    ms = C.c_float(0)
    cb._ffi.check(cb._ffi.lib.brn_gemm_microbench(M, N, K, cfg, sk, iters, 0, C.byref(ms)))
    return ms.value
print("shape                 | fp32 A: TF/s-eq, us | P2 A: TF/s-eq, us")
for M, N, K in [(5120,3072,768),(5120,2304,768),(5120,768,3072),(5120,768,768),(81920,768,192),(81920,576,192),(81920,192,768),(20480,1536,384),(20480,384,1536),(1280,6144,1536)]:
    fl = 2.0*M*N*K/1e9
    a, b = run(M,N,K,2006), run(M,N,K,2009)
    print(f"{M:6d} {N:5d} {K:6d} | {fl/a:6.1f} {a*1e3:6.1f} | {fl/b:6.1f} {b*1e3:6.1f}", flush=True)
