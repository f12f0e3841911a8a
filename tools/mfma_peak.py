import ctypes as C, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_birefnet_amd as cb
out=(C.c_float*2)()
for wgs in (256, 512, 1024, 2048):
    cb._ffi.check(cb._ffi.lib.brn_gemm_microbench(wgs, 20000, 2, 100, 1, 5, 0, C.cast(out, C.POINTER(C.c_float))))
    print('wgs', wgs, 'TF/s', out[0], 'clock MHz', out[1])
