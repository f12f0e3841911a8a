#!/usr/bin/env python3
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_birefnet_amd as cb
def run(M, N, K, cfg=-1, sk=1, iters=10):
    ms = C.c_float(0)
    cb._ffi.check(cb._ffi.lib.brn_gemm_microbench(M, N, K, cfg, sk, iters, 0, C.byref(ms)))
    return ms.value
print("shape (split2)         | plan | ws sk1 | ws sk2 | ws sk3 | ws sk4 | 128x64 sk1 | 128x64 sk2 | 64x64 sk1 | 64x64 sk2  (TF/s-eq)")
for M, N, K in [(5120,768,3072),(5120,768,768),(1280,1536,6144),(1280,6144,1536),(1280,4608,1536),(1280,1536,1536),(20480,384,1536),(20480,384,384),(81920,192,768),(81920,192,192),(5120,3072,768)]:
    fl = 2.0*M*N*K/1e9
    r = [run(M,N,K,2999)] + [run(M,N,K,2006,s) for s in (1,2,3,4)] + [run(M,N,K,2001,s) for s in (1,2)] + [run(M,N,K,2002,s) for s in (1,2)]
    print(f"{M:6d} {N:5d} {K:6d} | " + " | ".join(f"{fl/ms:6.1f}" for ms in r), flush=True)
