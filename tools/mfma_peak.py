import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_birefnet_amd as cb
out=(C.c_float*2)()
for cfg, name in ((100, 'f32 32x32x2 (4 acc)'), (101, 'bf16 32x32x16 (4 acc)'), (102, 'bf16 32x32x16 (1 dependent chain)')):
    for wgs in (256, 512, 1024):
        cb._ffi.check(cb._ffi.lib.brn_gemm_microbench(wgs, 20000, 2, cfg, 1, 5, 0, C.cast(out, C.POINTER(C.c_float))))
        print(f'{name:36s} wgs {wgs:5d}  TF/s {out[0]:8.1f}  clock MHz {out[1]:7.1f}', flush=True)
