#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "split_modes or half2 or split_conv_exact" > gpurun_out/r04_m16_tests2.log 2>&1; rc=$?
tail -2 gpurun_out/r04_m16_tests2.log
[ $rc -eq 0 ] || { tail -40 gpurun_out/r04_m16_tests2.log; exit $rc; }
timeout -k 10 600 bash tools/ab_lib.sh candle_birefnet_amd/libbirefnet_hip_ab0.so "c2" 4 2>&1 | tee gpurun_out/r04_ab_m16q.log
