#!/bin/bash
# round 4, GPU call 4: op / swin / golden tests on the swizzled split-GEMM layout; same-box A/B of the two layouts at c2; p1 fp32 A/B
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests/test_ops_gpu.py tests/test_swin_gpu.py tests/test_golden_gpu.py tests/test_model_gpu.py -x -q > gpurun_out/r04_t4.log 2>&1; RC=$?
tail -6 gpurun_out/r04_t4.log
if [ $RC -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/r04_t4.log | head -20; echo "tests failed: stopping"; exit 1; fi
timeout -k 10 300 bash tools/ab_lib.sh candle_birefnet_amd/libbirefnet_hip_ab0.so "c2" 4 > gpurun_out/r04_ab_swz_c2.log 2>&1; cat gpurun_out/r04_ab_swz_c2.log
timeout -k 10 200 bash tools/ab_lib.sh candle_birefnet_amd/libbirefnet_hip_ab0.so "c2" 3 --compute f32_split2 > gpurun_out/r04_ab_swz_c2_split2.log 2>&1; cat gpurun_out/r04_ab_swz_c2_split2.log
timeout -k 10 200 bash tools/ab_env.sh BRN_P1_F32 "0 1" "c3" > gpurun_out/r04_ab_p1b.log 2>&1; cat gpurun_out/r04_ab_p1b.log
