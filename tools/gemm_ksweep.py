#!/usr/bin/env python3
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_birefnet_amd as cb
def run(M, N, K, cfg=-1, sk=1, iters=10):
    ms = C.c_float(0)
    cb._ffi.check(cb._ffi.lib.brn_gemm_microbench(M, N, K, cfg, sk, iters, 0, C.byref(ms)))
    return ms.value
print("shape                 | 128x128 | 64x64 | 256x128 | 128x128 8w  (TF/s, ms)")
for M, N, K in [(4096,3072,768),(4096,3072,3072),(4096,3072,12288),(8192,8192,4096),(16384,3072,768),(65536,3072,768),(4096,4096,4096),(2048,2048,16384)]:
    fl = 2.0*M*N*K/1e9
    r = [run(M,N,K,c) for c in (0,2,4,5)]
    print(f"{M:6d} {N:5d} {K:6d} | " + " | ".join(f"{fl/ms:6.1f} {ms:7.3f}" for ms in r), flush=True)
