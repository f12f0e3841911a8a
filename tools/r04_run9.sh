#!/bin/bash
# round 4, GPU call 9: measured errors of the bf16 op-level tests (to set their tolerances at 3x measured)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -q -s -k "deform_conv2d_bf16_mode or window_attention_bf16_mode or linear_residual_layer_norm_bf16 or decblk_forward_bf16 or aspp_deformable_any_width_bf16" > gpurun_out/r04_t9.log 2>&1
grep -E "max abs err|passed|failed" gpurun_out/r04_t9.log | tail -60
