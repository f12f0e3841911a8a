#!/bin/bash
# round 4: rocprofv3 evidence of the final build (tools/profile_config.sh per configuration) + the default bench line
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for args in "c3 bf16 r04" "c3 bf16 r04 deformable" "c2 f32_half2 r04" "c2 f32_split3 r04" "c5 bf16 r04" "c5 bf16 r04 deformable"; do
  L=gpurun_out/r04_profile_$(echo $args | tr ' ' '_').log
  timeout -k 10 300 bash tools/profile_config.sh $args > $L 2>&1 || { echo "profile $args failed"; tail -5 $L; exit 1; }
  tail -1 $L
done
timeout -k 10 800 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err || { tail -5 gpurun_out/r04_bench_default.err; exit 1; }
python -c "import json; d=json.load(open('gpurun_out/r04_bench_default.json')); print(json.dumps(d['summary']))"
