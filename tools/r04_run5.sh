#!/bin/bash
# round 4, GPU call 5: full suite; A/Bs: offset-map padding, three sub-batch streams
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 840 python -m pytest tests -m gpu -x -q > gpurun_out/r04_t5.log 2>&1; RC=$?
tail -6 gpurun_out/r04_t5.log
if [ $RC -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/r04_t5.log | head -20; echo "tests failed: stopping"; exit 1; fi
timeout -k 10 200 bash tools/ab_env.sh BRN_OFFMOD_PAD8 "0 1" "c3" --deform-mode deformable > gpurun_out/r04_ab_offmod.log 2>&1; cat gpurun_out/r04_ab_offmod.log
timeout -k 10 300 bash tools/ab_env.sh BRN_SPLIT_STREAMS "2 3" "c3 c5" > gpurun_out/r04_ab_streams3.log 2>&1; cat gpurun_out/r04_ab_streams3.log
