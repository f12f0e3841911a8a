#!/bin/bash
# round 4, GPU call 11: the one-transcendental GELU in the fp32-class modes: tests, then same-box A/B at c2 against the previous library
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_ops_gpu.py tests/test_golden_gpu.py tests/test_model_gpu.py tests/test_swin_gpu.py -x -q > gpurun_out/r04_t11.log 2>&1; RC=$?
tail -4 gpurun_out/r04_t11.log
if [ $RC -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/r04_t11.log | head -20; exit 1; fi
timeout -k 10 300 bash tools/ab_lib.sh candle_birefnet_amd/libbirefnet_hip_ab0.so "c2" 4 > gpurun_out/r04_ab_gelu_c2.log 2>&1; cat gpurun_out/r04_ab_gelu_c2.log
timeout -k 10 100 python bench.py --config c2 --other-configs off --cpu-baseline on --steps 10 --warmup 3 --also "" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c2 err vs golden', d['max_abs_err_image0_vs_strided_golden'], 'vs oracle', d['cpu_baseline'].get('gpu_vs_oracle_max_abs_err'))"
