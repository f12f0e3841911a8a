#!/bin/bash
# round 4: rocprofv3 evidence of c2 in mode f32_half2
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 bash tools/profile_config.sh c2 f32_half2 r04 > gpurun_out/r04_profile_c2_f32_half2.log 2>&1 || { tail -5 gpurun_out/r04_profile_c2_f32_half2.log; exit 1; }
tail -1 gpurun_out/r04_profile_c2_f32_half2.log
