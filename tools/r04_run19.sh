#!/bin/bash
# round 4: the fuzzers with the two new compute modes (f32_half2, f16), then the rocprofv3 evidence of c3 in f16
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 300 python tools/op_fuzz.py 80 11 > gpurun_out/r04_fuzz_op.log 2>&1; tail -3 gpurun_out/r04_fuzz_op.log
timeout -k 10 200 python tools/att_fuzz.py 40 12 > gpurun_out/r04_fuzz_att.log 2>&1; tail -3 gpurun_out/r04_fuzz_att.log
timeout -k 10 300 python tools/deform_fuzz.py 40 13 > gpurun_out/r04_fuzz_deform.log 2>&1; tail -3 gpurun_out/r04_fuzz_deform.log
timeout -k 10 400 python tools/model_fuzz.py 14 14 > gpurun_out/r04_fuzz_model.log 2>&1; tail -3 gpurun_out/r04_fuzz_model.log
timeout -k 10 300 python tools/model_fuzz.py 8 15 deformable > gpurun_out/r04_fuzz_model_def.log 2>&1; tail -3 gpurun_out/r04_fuzz_model_def.log
timeout -k 10 300 bash tools/profile_config.sh c3 f16 r04 > gpurun_out/r04_profile_c3_f16.log 2>&1 || { tail -5 gpurun_out/r04_profile_c3_f16.log; exit 1; }
tail -1 gpurun_out/r04_profile_c3_f16.log
