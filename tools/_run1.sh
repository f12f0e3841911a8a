set -e
python -m pytest tests/test_ops_gpu.py tests/test_golden_gpu.py tests/test_configs_gpu.py -m gpu -x -q -k "bf16" 2>&1 | tail -3
for i in 1 2; do BRN_GEMM_ACT=2 BRN_SWEEP_CFGS=2,3,-1 BRN_LIB_PATH=candle_birefnet_amd/libbirefnet_hip_diag.so python tools/gemm_bf16_sweep.py 40960x3072x768; done
python tools/bench_env_ab.py "BRN_LIB_PATH=candle_birefnet_amd/libbirefnet_hip_base.so" "X=1" -- --config c3
