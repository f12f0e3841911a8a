#!/bin/bash
# round 4: mode f32_half2 — its tests, then c2 timed in it with the other fp32 modes beside it
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s -k "half2" > gpurun_out/r04_half2_tests.log 2>&1; rc=$?
grep -E "relative-to-max|half2\]|passed|failed|Error|error" gpurun_out/r04_half2_tests.log | tail -60
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py --compute f32_half2 --steps 30 --warmup 10 --other-configs off --also f32_split3,f32_split2 > gpurun_out/r04_bench_half2.json 2> gpurun_out/r04_bench_half2.err || { tail -5 gpurun_out/r04_bench_half2.err; exit 1; }
python - <<'P'
import json
d=json.load(open('gpurun_out/r04_bench_half2.json'))
print(d['value'], d['ms_per_step'], d['dtype'][:40], {k:d['roofline'][k] for k in ('achieved','peak','frac')})
print({k:(v['images_per_s'], v.get('gpu_vs_oracle_max_abs_err')) for k,v in d['other_modes'].items()})
print({k:d['cpu_baseline'].get(k) for k in ('gpu_vs_oracle_max_abs_err','gpu_vs_oracle_max_rel_err')}, d.get('max_abs_err_image0'), d.get('max_abs_err_mask_image0'))
P
