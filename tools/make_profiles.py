#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the small CSV / JSON summaries committed under profiles/.

  python tools/make_profiles.py --tag r01b --stats gpurun_out/prof_stats --fetch gpurun_out/pmc_fetch --write gpurun_out/pmc_write \
                                [--launches gpurun_out/launches.csv] [--bench gpurun_out/bench.json] [--forwards 4]

stats : rocprofv3 --kernel-trace --stats            -> <tag>_kernel_stats_*.csv (per kernel: calls, total / avg / min / max ns, %)
fetch : rocprofv3 --pmc FETCH_SIZE --kernel-trace   -> <tag>_pmc_hbm_*.csv (per kernel KB) + <tag>_gemm_traffic.json
write : rocprofv3 --pmc WRITE_SIZE --kernel-trace      (separate passes, MI355X_MICROARCH.md "HBM"; FETCH_SIZE is doubled:
        gfx950 tallies 128-B read requests at 64 B; WRITE_SIZE is taken as is)
"""
import argparse, collections, csv, glob, json, os, re, shutil


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    if not hits:
        raise SystemExit(f"no *{suffix} under {d}")
    return hits


def short(name):
    return re.sub(r"\(.*$", "", name).strip()


def kernel_stats(d, out, forwards):
    rows = []
    for f in find(d, "kernel_trace.csv"):
        rows += list(csv.DictReader(open(f)))
    agg = collections.OrderedDict()
    for r in rows:
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        a = agg.setdefault(r["Kernel_Name"], [0, 0, 1 << 62, 0])
        a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
    tot = sum(a[1] for a in agg.values())
    with open(out, "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["Name", f"Calls({forwards} forwards)", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, a[0], a[1], f"{a[1] / a[0]:.1f}", f"{100.0 * a[1] / tot:.2f}", a[2], a[3]])
    return tot, agg


def counter(d, name):
    per = collections.OrderedDict()
    for f in find(d, "counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name:
                continue
            a = per.setdefault(short(r["Kernel_Name"]), [set(), 0.0])
            a[0].add(r["Dispatch_Id"]); a[1] += float(r["Counter_Value"])
    return per


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", required=True)
    ap.add_argument("--suffix", default="b1_1024_split2")
    ap.add_argument("--stats"); ap.add_argument("--fetch"); ap.add_argument("--write")
    ap.add_argument("--launches"); ap.add_argument("--bench")
    ap.add_argument("--forwards", type=int, default=4)
    a = ap.parse_args()
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
    if a.stats:
        out = os.path.join(root, f"{a.tag}_kernel_stats_{a.suffix}.csv")
        tot, agg = kernel_stats(a.stats, out, a.forwards)
        print("wrote", out, f"({tot / 1e6 / a.forwards:.3f} ms of kernels per forward)")
    if a.fetch and a.write:
        fe, wr = counter(a.fetch, "FETCH_SIZE"), counter(a.write, "WRITE_SIZE")
        out = os.path.join(root, f"{a.tag}_pmc_hbm_{a.suffix}.csv")
        gem = {"gemm_family_dispatches": 0, "hbm_read_gb_x2corrected": 0.0, "hbm_write_gb": 0.0}
        with open(out, "w", newline="") as fo:
            w = csv.writer(fo)
            w.writerow(["kernel", "dispatches(1 forward)", "FETCH_SIZE_KB_raw", "WRITE_SIZE_KB", "read_GB_corrected(x2)", "write_GB"])
            for k in sorted(fe, key=lambda k: -fe[k][1]):
                n = len(fe[k][0]); f_kb = fe[k][1]; w_kb = wr.get(k, [set(), 0.0])[1]
                rgb, wgb = 2 * f_kb * 1024 / 1e9, w_kb * 1024 / 1e9
                w.writerow([k, n, f"{f_kb:.1f}", f"{w_kb:.1f}", f"{rgb:.3f}", f"{wgb:.3f}"])
                if ("gemm" in k or "patch_embed" in k) and "reduce" not in k:      # (patch_embed_ln_kernel is bracketed as the gather GEMM it replaces)
                    gem["gemm_family_dispatches"] += n; gem["hbm_read_gb_x2corrected"] += rgb; gem["hbm_write_gb"] += wgb
        json.dump(gem, open(os.path.join(root, f"{a.tag}_gemm_traffic_{a.suffix}.json"), "w"))
        print("wrote", out, gem)
    if a.launches:
        shutil.copy(a.launches, os.path.join(root, f"{a.tag}_launches_{a.suffix}.csv"))
    if a.bench:
        shutil.copy(a.bench, os.path.join(root, f"{a.tag}_bench_{a.suffix}.json"))


if __name__ == "__main__":
    main()
