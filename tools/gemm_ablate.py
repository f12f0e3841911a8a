#!/usr/bin/env python3
import ctypes as C, sys, os, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import candle_birefnet_amd as cb
    M, N, K = 5120, 3072, 768
    out = []
    for cfg in (1006, 2006, 3006, 2000):
        ms = C.c_float(0)
        cb._ffi.check(cb._ffi.lib.brn_gemm_microbench(M, N, K, cfg, 1, 20, 0, C.byref(ms)))
        out.append(f"{ms.value*1e3:7.1f}")
    print(" | ".join(out)); sys.exit(0)
print("ablate (1=no gload, 2=no lds_store, 4=no compute) | s1 ws | s2 ws | s3 ws | s2 128²  (us per launch, 5120x3072x768)")
for abl in (0, 3, 7, 19, 23):
    env = dict(os.environ, BRN_GEMM_ABLATE=str(abl))
    r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
    print(f"abl={abl}: {r.stdout.strip()} {r.stderr.strip()[-200:] if r.returncode else ''}", flush=True)
