import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_birefnet_amd as cb
out=(C.c_float*2)()
print("consumer-loop probe (fragments from LDS, no global memory, no barriers): bf16 MFMA TF/s (peak 2450)")
for wgs in (256, 512, 1024):
    row=[]
    for cfg in (111, 121, 112, 122, 113, 123):
        cb._ffi.check(cb._ffi.lib.brn_gemm_microbench(wgs, 2000, 2, cfg, 1, 5, 0, C.cast(out, C.POINTER(C.c_float))))
        row.append(f"{out[0]:7.0f}")
    print(f"wgs {wgs:5d} | NP1 v0 {row[0]} v1 {row[1]} | NP2 v0 {row[2]} v1 {row[3]} | NP3 v0 {row[4]} v1 {row[5]}", flush=True)
