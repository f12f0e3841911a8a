#!/usr/bin/env python3
"""VGPR / SGPR / LDS / occupancy of every kernel of the product library, from hipcc -Rpass-analysis=kernel-resource-usage with the
Makefile's own flags -> profiles/<tag>_kernel_resources.csv"""
import csv, glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "candle_birefnet_amd", "csrc")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
mk = open(os.path.join(CSRC, "Makefile")).read()
flags = [f for f in re.search(r"^CXXFLAGS\s*=\s*(.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950").split() if f != "-fPIC"]
rows = []
for f in sorted(glob.glob(os.path.join(CSRC, "kernels", "*.hip"))):
    extra = ["-ffp-contract=off"] if f.endswith("imageproc.hip") else []
    p = subprocess.run(["/opt/rocm/bin/hipcc", *flags, *extra, "-c", f, "-o", "/dev/null", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage"],
                       capture_output=True, text=True)
    cur = None
    for line in p.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = {"file": os.path.basename(f), "kernel": re.sub(r"\(.*$", "", name).replace("void brn::", "").replace("brn::", "")}
            rows.append(cur)
            continue
        for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r" SGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                         ("occupancy_waves_per_simd", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds_bytes", r"LDS Size \[bytes/block\]: (\d+)"),
                         ("vgpr_spill", r"VGPRs Spill: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = m.group(1)
out = os.path.join(ROOT, "profiles", f"{tag}_kernel_resources.csv")
cols = ["file", "kernel", "vgpr", "agpr", "sgpr", "scratch", "vgpr_spill", "lds_bytes", "occupancy_waves_per_simd"]
with open(out, "w", newline="") as fo:
    w = csv.DictWriter(fo, fieldnames=cols)
    w.writeheader()
    for r in rows:
        w.writerow({c: r.get(c, "") for c in cols})
print("wrote", out, len(rows), "kernels")
