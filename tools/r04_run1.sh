#!/bin/bash
# round 4, GPU call 1: the GPU suite, the bf16 error attribution, the CU-partition A/B, the power / clock poll
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 720 python -m pytest tests -m gpu -x -q > gpurun_out/r04_t1.log 2>&1; RC=$?
tail -5 gpurun_out/r04_t1.log
if [ $RC -eq 124 ] || [ $RC -eq 137 ]; then echo "pytest timed out: stopping"; exit 1; fi
timeout -k 10 240 python tools/bf16_error_attrib.py --size 1024 --out gpurun_out/r04_attrib_1024.json > gpurun_out/r04_attrib.log 2>&1 || { tail -20 gpurun_out/r04_attrib.log; }
tail -40 gpurun_out/r04_attrib.log
timeout -k 10 300 bash tools/ab_env.sh BRN_CU_PARTITION "0 1 2" "c3" > gpurun_out/r04_ab_cupart.log 2>&1; cat gpurun_out/r04_ab_cupart.log
timeout -k 10 120 bash tools/power_poll.sh c3 150 > gpurun_out/r04_power_c3.log 2>&1; tail -3 gpurun_out/r04_power_c3.log
