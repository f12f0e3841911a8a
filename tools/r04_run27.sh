#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=40 > gpurun_out/r04_durations.log 2>&1
grep -A45 "slowest 40 durations" gpurun_out/r04_durations.log | head -50; tail -2 gpurun_out/r04_durations.log
