#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 200 python -m pytest tests -m gpu -x -q -k "half2 and (linear or golden or 1024)" 2>&1 | tail -2
timeout -k 10 600 bash tools/ab_lib.sh candle_birefnet_amd/libbirefnet_hip_ab0.so "c2" 4 2>&1 | tee gpurun_out/r04_ab_aplpad.log
timeout -k 10 300 bash tools/ab_lib.sh candle_birefnet_amd/libbirefnet_hip_ab0.so "c2" 3 --compute f32_split2 2>&1 | tee -a gpurun_out/r04_ab_aplpad.log
