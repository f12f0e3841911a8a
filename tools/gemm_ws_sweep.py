#!/usr/bin/env python3
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_birefnet_amd as cb
import numpy as np
def run(M, N, K, cfg=-1, sk=1, iters=10):
    ms = C.c_float(0)
    cb._ffi.check(cb._ffi.lib.brn_gemm_microbench(M, N, K, cfg, sk, iters, 0, C.byref(ms)))
    return ms.value
print("shape                 | f32 plan | s3 ws | s2 ws | s1 ws  (fp32-equivalent TF/s)")
for M, N, K in [(5120,3072,768),(5120,768,3072),(5120,2304,768),(5120,768,768),(81920,768,192),(20480,1536,384),(65536,256,3136),(8192,8192,4096)]:
    fl = 2.0*M*N*K/1e9
    r = [run(M,N,K,-1), run(M,N,K,3006), run(M,N,K,2006), run(M,N,K,1006)]
    print(f"{M:6d} {N:5d} {K:6d} | " + " | ".join(f"{fl/ms:7.1f}" for ms in r), flush=True)
