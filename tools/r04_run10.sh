#!/bin/bash
# round 4, GPU call 10: the one-transcendental GELU (bf16 outputs): op tests + goldens, then same-box A/B against the previous library
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_configs_gpu.py -x -q -k "bf16 or gelu or linear" > gpurun_out/r04_t10.log 2>&1; RC=$?
tail -4 gpurun_out/r04_t10.log
if [ $RC -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/r04_t10.log | head -20; exit 1; fi
timeout -k 10 400 bash tools/ab_lib.sh candle_birefnet_amd/libbirefnet_hip_ab0.so "c3 c5" 3 > gpurun_out/r04_ab_gelu.log 2>&1; cat gpurun_out/r04_ab_gelu.log
