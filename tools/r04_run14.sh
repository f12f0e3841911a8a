#!/bin/bash
# round 4: f32_half2 with the fp16-pair attention kernel: tests, A/B against the fp32-MFMA attention kernel, A/B of the activation scale
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s -k "half2 or window_attention_split" > gpurun_out/r04_half2_tests2.log 2>&1; rc=$?
grep -E "half2\]|passed|failed|Error|error" gpurun_out/r04_half2_tests2.log | tail -30
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 bash tools/ab_env.sh BRN_H2_ATT "0 1" "c2" --also "" 2>&1 | tee gpurun_out/r04_ab_h2att.log
