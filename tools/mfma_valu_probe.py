"""How MFMA and plain VALU share a SIMD (two waves per SIMD): specialised waves vs every wave doing both."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_birefnet_amd as cb
out=(C.c_float*2)()
names = {0: "MFMA only (waves 0-3: 24 MFMA/iter)", 1: "VALU only (waves 4-7: 72 VALU/iter)", 2: "specialised: 0-3 MFMA, 4-7 VALU",
         3: "every wave 12 MFMA + 36 VALU, interleaved", 4: "every wave 12 MFMA + 36 VALU, VALU in a block",
         5: "waves 0-3: 24 MFMA + 16 ds_read_b128", 6: "waves 4-7: 72 VALU + 4 write2st64_b64 + 4 write_b128 + wait", 7: "5 and 6 together"}
for wgs in (256, 512):
    for mode in range(8):
        cb._ffi.check(cb._ffi.lib.brn_gemm_microbench(wgs, 4000, 2, 130 + mode, 1, 5, 0, C.cast(out, C.POINTER(C.c_float))))
        print(f"wgs {wgs} mode {mode} {names[mode]:48s}: {out[0]:8.1f} us  ({out[0]*1e-6/4000*2.0e9:.0f} cycles/iter @2.0GHz)", flush=True)
