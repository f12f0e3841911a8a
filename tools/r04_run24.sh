#!/bin/bash
# round 4: 16 x 16 x 32 MFMAs in the 2-plane warp-specialised GEMM: the split-mode tests, then same-box A/B against the 32 x 32 x 16 build
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "split_modes or half2 or split_conv_exact or golden" > gpurun_out/r04_m16_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r04_m16_tests.log
[ $rc -eq 0 ] || { tail -40 gpurun_out/r04_m16_tests.log; exit $rc; }
timeout -k 10 600 bash tools/ab_lib.sh candle_birefnet_amd/libbirefnet_hip_ab0.so "c2" 4 2>&1 | tee gpurun_out/r04_ab_m16.log
timeout -k 10 300 bash tools/ab_lib.sh candle_birefnet_amd/libbirefnet_hip_ab0.so "c2" 2 --compute f32_split2 2>&1 | tee -a gpurun_out/r04_ab_m16.log
