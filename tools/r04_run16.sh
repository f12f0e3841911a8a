#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 500 bash tools/ab_env.sh BRN_SK_T128 "200 250" "c2" --also "" 2>&1 | tee gpurun_out/r04_ab_skt128.log
timeout -k 10 300 bash tools/ab_env.sh BRN_H2_ASCALE "3 0 6" "c2" --also "" 2>&1 | tee gpurun_out/r04_ab_ascale.log
