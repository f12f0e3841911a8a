#!/bin/bash
# round 4: the rewritten GAP kernels: tests that run an ASPP / decoder block / whole model, then the per-launch tables
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -k "aspp or decblk or golden or kats or pieces or c3_batch8" > gpurun_out/r04_gap_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r04_gap_tests.log
[ $rc -eq 0 ] || { tail -40 gpurun_out/r04_gap_tests.log; exit $rc; }
bash tools/dump_launches.sh c2 gpurun_out/r04_launches_c2_gap.csv --also ""
bash tools/dump_launches.sh c3 gpurun_out/r04_launches_c3_gap.csv --also ""
python - <<'P'
import csv
for f in ('gpurun_out/r04_launches_c2_gap.csv','gpurun_out/r04_launches_c3_gap.csv'):
    rows=list(csv.DictReader(open(f)))
    print(f, [ (r['ms'], r['gbytes']) for r in rows if r['family']=='elementwise' and r['region']=='1'])
P
