#!/usr/bin/env python3
"""tall-K, N = 64 convs of the decoder's coarse levels: 64x64 tiles with more K slices"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_birefnet_amd as cb
def run(M, N, K, cfg=-1, sk=1, iters=20):
    ms = C.c_float(0)
    cb._ffi.check(cb._ffi.lib.brn_gemm_microbench(M, N, K, cfg, sk, iters, 0, C.byref(ms)))
    return ms.value
sks = (16, 32, 48, 64, 96, 128, 192)
print("shape (split2, us)     | plan | 64x64 sk " + " ".join(f"{s:5d}" for s in sks) + " | 128x64 sk 16 32 64")
for M, N, K in [(1024,64,51840),(1024,64,31104),(1024,64,27648),(4096,64,17280),(4096,64,6912),(1024,16,13824),(4096,16,6912),(1024,256,3136),(1024,256,576),(1024,3072,576),(1024,1536,576),(1024,384,576)]:
    r = [run(M,N,K,2999)] + [run(M,N,K,2002,s) for s in sks] + [run(M,N,K,2001,s) for s in (16,32,64)]
    print(f"{M:6d} {N:5d} {K:6d} | " + " ".join(f"{ms*1e3:6.1f}" for ms in r), flush=True)
