#!/usr/bin/env python3
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_birefnet_amd as cb
M, N, K, cfg = (int(v) for v in sys.argv[1:5])
ms = C.c_float(0)
cb._ffi.check(cb._ffi.lib.brn_gemm_microbench(M, N, K, cfg, 1, 5, 0, C.byref(ms)))
print(ms.value)
