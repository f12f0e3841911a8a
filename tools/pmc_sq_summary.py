#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --pmc run (SQ / GRBM counters) joined with its kernel trace.

  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS \
            SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d DIR -- python3 bench.py ...
  python tools/pmc_sq_summary.py DIR [out.csv]

MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_32x32x16_bf16) summed over the SIMDs; SQ_WAVE_CYCLES /
SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; GRBM_GUI_ACTIVE is summed over the 8 XCDs.
mfma_util = MFMA busy cycles / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)."""
import collections, csv, glob, os, re, sys

d = sys.argv[1]
rows = []
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    rows += list(csv.DictReader(open(f)))
trace = {}
for f in sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        trace[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])


def short(n):
    n = re.sub(r"\(.*$", "", n).strip()
    n = n.replace("void brn::", "").replace("brn::", "")
    return n


agg = collections.OrderedDict()
for r in rows:
    k = short(r["Kernel_Name"])
    a = agg.setdefault(k, {"disp": set(), "ns": 0, "c": collections.Counter(), "vgpr": r.get("VGPR_Count", ""), "lds": r.get("LDS_Block_Size", "")})
    if r["Dispatch_Id"] not in a["disp"]:
        a["disp"].add(r["Dispatch_Id"])
        a["ns"] += trace.get(r["Dispatch_Id"], 0)
    a["c"][r["Counter_Name"]] += float(r["Counter_Value"])
names = sorted({r["Counter_Name"] for r in rows})
out = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
out.writerow(["kernel", "dispatches", "total_us", "vgpr", "lds_bytes"] + names + ["mfma_util", "wait_any_frac", "wait_inst_lds_frac", "active_inst_frac"])
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
    c = a["c"]
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    util = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 1024.0) if gui else 0.0
    wc = c.get("SQ_WAVE_CYCLES", 0.0)
    fr = lambda n: (c.get(n, 0.0) / wc) if wc else 0.0
    out.writerow([k, len(a["disp"]), f"{a['ns'] / 1e3:.1f}", a["vgpr"], a["lds"]] + [f"{c.get(n, 0.0):.0f}" for n in names] +
                 [f"{util:.3f}", f"{fr('SQ_WAIT_ANY'):.3f}", f"{fr('SQ_WAIT_INST_LDS'):.3f}", f"{fr('SQ_ACTIVE_INST_ANY'):.3f}"])
