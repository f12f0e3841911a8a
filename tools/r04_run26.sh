#!/bin/bash
# round 4: after the 16 x 16 x 32 change: the rocprofv3 evidence of c2 in f32_half2 again + the default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 bash tools/profile_config.sh c2 f32_half2 r04 > gpurun_out/r04_profile_c2_f32_half2.log 2>&1 || { tail -5 gpurun_out/r04_profile_c2_f32_half2.log; exit 1; }
tail -1 gpurun_out/r04_profile_c2_f32_half2.log
timeout -k 10 800 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err || { tail -5 gpurun_out/r04_bench_default.err; exit 1; }
python -c "import json; d=json.load(open('gpurun_out/r04_bench_default.json')); print(json.dumps(d['summary']))"
