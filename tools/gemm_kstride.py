"""Does the row stride (K) matter?  Same tile counts, K around 768: a power-of-two-ish stride that maps rows to few L2 channels would show as a jump."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_birefnet_amd as cb
def run(M, N, K, cfg, iters=20):
    ms = C.c_float(0)
    cb._ffi.check(cb._ffi.lib.brn_gemm_microbench(M, N, K, cfg, 1, iters, 0, C.byref(ms)))
    return ms.value
print("K     | split2 ws us, TF-eq | f32 cfg0 us, TF")
for K in (704, 736, 768, 800, 832, 864, 896, 1024, 1056, 1536, 1568):
    M, N = 5120, 3072
    fl = 2.0 * M * N * K / 1e9
    a = run(M, N, K, 2006); b = run(M, N, K, 0)
    print(f"{K:5d} | {a*1e3:7.1f} {fl/a:7.1f} | {b*1e3:7.1f} {fl/b:7.1f}", flush=True)
