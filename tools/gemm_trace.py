"""Per-workgroup cycle stamps of gemm_split_ws_kernel (BRN_GEMM_TRACE): where a 128x128 tile's time goes."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candle_birefnet_amd as cb

def run(M, N, K, cfg, abl, path):
    os.environ["BRN_GEMM_ABLATE"] = str(abl)
    os.environ["BRN_GEMM_TRACE"] = path
    out = (C.c_float * 2)()
    cb._ffi.check(cb._ffi.lib.brn_gemm_microbench(M, N, K, cfg, 1, 20, 0, C.cast(out, C.POINTER(C.c_float))))
    full = np.fromfile(path, dtype=np.uint64).reshape(-1, 256).astype(np.int64)
    tr = full[:, :16]
    c, pr = tr[:, :8], tr[:, 8:]
    t0c = c[:, 0].min(); t0w = c[:, 1].min()
    wall = (np.maximum(c[:, 6], pr[:, 5]).max() - t0w) / 100.0   # us (100 MHz)
    ghz = (c[:, 5] - c[:, 0]).astype(float) / np.maximum(1, (c[:, 6] - c[:, 1])) / 10.0
    print(f"\n=== {M}x{N}x{K} cfg {cfg} abl {abl}: {out[0]*1e3:.1f} us/launch (events), traced span {wall:.1f} us, clock ~{np.median(ghz):.2f} GHz, {len(tr)} WGs")
    f = np.median(ghz) * 1e3   # cycles per us
    start = (c[:, 1] - t0w) / 100.0
    pro = (c[:, 3] - c[:, 0]) / f; loop = (c[:, 4] - c[:, 3]) / f; epi = (c[:, 5] - c[:, 4]) / f
    ploop = (pr[:, 4] - pr[:, 3]) / f
    for name, v in (("start", start), ("prologue", pro), ("consumer loop", loop), ("epilogue", epi), ("producer loop", ploop)):
        print(f"  {name:14s} us: min {v.min():7.2f}  p10 {np.percentile(v,10):7.2f}  med {np.median(v):7.2f}  p90 {np.percentile(v,90):7.2f}  max {v.max():7.2f}")
    # rounds: WGs by start time
    order = np.argsort(start)
    first = order[:512]; rest = order[512:]
    if len(rest):
        print(f"  first-round WGs: loop med {np.median(loop[first]):.2f} us | later WGs ({len(rest)}): start med {np.median(start[rest]):.2f}, loop med {np.median(loop[rest]):.2f}")
    # per-K-tile stamps (cycles): producer P0 start, P1 after lds_store, P2 after gload issue; consumer C0 start, C1 after MFMA issue
    P = full[:, 16:16 + 96].reshape(-1, 24, 4); Cn = full[:, 128:128 + 96].reshape(-1, 24, 4)
    sel = order[:512]
    ts = slice(4, 20)
    p_vm = (P[sel, ts, 3] - P[sel, ts, 0]); p_store = (P[sel, ts, 1] - P[sel, ts, 3]); p_gl = (P[sel, ts, 2] - P[sel, ts, 1]); p_wait = (P[sel, 5:21, 0] - P[sel, ts, 2])
    c_work = (Cn[sel, ts, 1] - Cn[sel, ts, 0]); c_wait = (Cn[sel, 5:21, 0] - Cn[sel, ts, 1])
    print(f"  first-round per K-tile cycles (median): producer vmcnt wait {np.median(p_vm):.0f}, lds_store {np.median(p_store):.0f}, gload issue {np.median(p_gl):.0f}, barrier wait {np.median(p_wait):.0f} | consumer work {np.median(c_work):.0f}, barrier wait {np.median(c_wait):.0f} | period {np.median(Cn[sel, 5:21, 0] - Cn[sel, ts, 0]):.0f}")
    hw = c[:, 2]
    cu = ((hw >> 32) & 0xf) * 1000 + ((hw >> 13) & 0x7) * 100 + ((hw >> 8) & 0xf)   # xcc, se, cu
    ncu = len(np.unique(cu))
    print(f"  distinct (xcc,se,cu) seen: {ncu}")

if __name__ == "__main__":
    os.makedirs("gpurun_out", exist_ok=True)
    abls = [int(a) for a in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 3, 19]
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 2006
    for abl in abls:
        run(5120, 3072, 768, cfg, abl, "gpurun_out/trace.bin")
