"""What the fused epilogues cost: bias+GELU(erf) and residual add on the big stage-3 GEMMs (split2 ws kernel)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_birefnet_amd as cb
def run(M, N, K, cfg=2999, iters=30):
    ms = C.c_float(0)
    cb._ffi.check(cb._ffi.lib.brn_gemm_microbench(M, N, K, cfg, 1, iters, 0, C.byref(ms)))
    return ms.value * 1e3
for M, N, K in [(5120, 3072, 768), (5120, 768, 3072), (5120, 2304, 768), (81920, 768, 192)]:
    os.environ.pop("BRN_GEMM_ACT", None); os.environ.pop("BRN_GEMM_RES", None)
    base = run(M, N, K)
    os.environ["BRN_GEMM_ACT"] = "0"; b0 = run(M, N, K)
    os.environ["BRN_GEMM_ACT"] = "2"; g = run(M, N, K)
    os.environ.pop("BRN_GEMM_ACT"); os.environ["BRN_GEMM_RES"] = "1"; r = run(M, N, K)
    print(f"{M}x{N}x{K}: plain {base:.1f} us | +bias {b0:.1f} | +bias+gelu {g:.1f} | +residual {r:.1f}", flush=True)
