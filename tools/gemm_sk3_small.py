"""split-K sweep of the 3-plane warp-specialised GEMM on the SMALL-M shapes of the batch-1 path (decoder convs on 32x32 / 64x64 maps, stage-3
GEMMs); diag build (BRN_LIB_PATH=...diag.so).  Dense stand-ins for the conv shapes (same M, N, K)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_birefnet_amd as cb
def run(M, N, K, cfg=-1, sk=1, iters=20):
    ms = C.c_float(0)
    cb._ffi.check(cb._ffi.lib.brn_gemm_microbench(M, N, K, cfg, sk, iters, 0, C.byref(ms)))
    return ms.value
print("shape (split3)            | plan   | ws sk1 | sk2 | sk4 | sk8 | sk12 | sk16   (us)")
for M, N, K in [(1024,256,3136),(4096,256,3136),(1024,64,51840),(4096,64,17280),(16384,64,8640),(1280,1536,1536),(1280,1536,6144),(1280,6144,1536),(1024,256,576),(4096,192,576)]:
    r = [run(M,N,K,3999)] + [run(M,N,K,3006,s) for s in (1,2,4,8,12,16)]
    print(f"{M:6d} {N:5d} {K:6d} | " + " | ".join(f"{ms*1e3:7.1f}" for ms in r), flush=True)
