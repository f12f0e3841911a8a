#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 800 python bench.py > gpurun_out/r04_bench_default.json 2> gpurun_out/r04_bench_default.err || { tail -5 gpurun_out/r04_bench_default.err; exit 1; }
python -c "import json; d=json.load(open('gpurun_out/r04_bench_default.json')); print(json.dumps(d['summary'])); print(json.dumps(d['other_modes']))"
