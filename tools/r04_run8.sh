#!/bin/bash
# round 4, GPU call 8: the mixed compute mode — test, then the default bench line (which now carries its block)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_configs_gpu.py -x -q -s -k "mixed_mode" > gpurun_out/r04_t8.log 2>&1; RC=$?
grep -E "max abs err|passed|failed" gpurun_out/r04_t8.log | tail -12
if [ $RC -ne 0 ]; then tail -30 gpurun_out/r04_t8.log; exit 1; fi
timeout -k 10 400 python bench.py > gpurun_out/r04_bench_default2.json 2> gpurun_out/r04_bench_default2.err || { tail -5 gpurun_out/r04_bench_default2.err; exit 1; }
python -c "import json; d=json.load(open('gpurun_out/r04_bench_default2.json')); print(json.dumps(d['summary']))"
