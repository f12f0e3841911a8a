set -e
python -m pytest tests/test_ops_gpu.py tests/test_golden_gpu.py tests/test_configs_gpu.py tests/test_swin_gpu.py -m gpu -x -q -k "bf16 or attention" 2>&1 | tail -4
BRN_ATT_HPW=2 python -m pytest tests/test_ops_gpu.py tests/test_golden_gpu.py -m gpu -x -q -k "bf16 or attention" 2>&1 | tail -2
python tools/bench_env_ab.py "BRN_ATT_HPW=1" "BRN_ATT_HPW=2" "BRN_LIB_PATH=candle_birefnet_amd/libbirefnet_hip_base.so" -- --config c3 --profile-steps 1 2>&1 | tee gpurun_out/ab_att_hpw.txt
