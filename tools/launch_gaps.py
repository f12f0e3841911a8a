#!/usr/bin/env python3
"""idle time between consecutive kernels of one forward, from a rocprofv3 --kernel-trace CSV (one stream: BRN_BRANCH_STREAMS=0 BRN_SPLIT_STREAMS=1):
tools/launch_gaps.py <dir with *_kernel_trace.csv> <forwards in the trace>"""
import csv, glob, sys, os
d, nf = sys.argv[1], int(sys.argv[2])
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)) if "copyBuffer" not in r["Kernel_Name"] and "fillBuffer" not in r["Kernel_Name"]))
# the forwards are the last nf equal-length runs of the trace: split by the longest gaps
n = len(rows) // nf * nf
per = len(rows) // nf
rows = rows[len(rows) - per * nf:]
for k in range(nf):
    r = rows[k * per:(k + 1) * per]
    busy = sum(e - s for s, e, _ in r)
    span = r[-1][1] - r[0][0]
    gaps = [r[i + 1][0] - r[i][1] for i in range(len(r) - 1)]
    pos = [g for g in gaps if g > 0]
    print(f"forward {k}: {len(r)} kernels, span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, idle {sum(pos) / 1e6:.3f} ms in {len(pos)} gaps (median {sorted(pos)[len(pos) // 2] / 1e3:.2f} us)")
