"""split-K sweep of the 3-plane warp-specialised GEMM on the batch-1 stage-2 shapes (diag build; BRN_LIB_PATH=...diag.so)"""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_birefnet_amd as cb
def run(M, N, K, cfg=-1, sk=1, iters=20):
    ms = C.c_float(0)
    cb._ffi.check(cb._ffi.lib.brn_gemm_microbench(M, N, K, cfg, sk, iters, 0, C.byref(ms)))
    return ms.value
print("shape (split3)         | plan | ws sk1 | sk2 | sk3 | sk4   (TF/s-eq)")
for M, N, K in [(5120,768,3072),(5120,768,768),(5120,2304,768),(5120,3072,768),(1280,1536,6144),(1280,6144,1536),(1280,4608,1536),(1280,1536,1536),(20480,384,1536),(81920,192,768)]:
    fl = 2.0*M*N*K/1e9
    r = [run(M,N,K,3999)] + [run(M,N,K,3006,s) for s in (1,2,3,4)]
    print(f"{M:6d} {N:5d} {K:6d} | " + " | ".join(f"{fl/ms:6.1f}" for ms in r), flush=True)
