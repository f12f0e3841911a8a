set -e
python -m pytest tests/test_ops_gpu.py tests/test_golden_gpu.py tests/test_configs_gpu.py tests/test_swin_gpu.py -m gpu -x -q -k "bf16 or attention" 2>&1 | tail -4
python tools/bench_env_ab.py "BRN_LIB_PATH=candle_birefnet_amd/libbirefnet_hip_base.so" "X=1" -- --config c3 2>&1 | tee gpurun_out/ab_base_new3.txt
python bench.py --config c3 --cpu-baseline off --also= > gpurun_out/bench_c3_att.json 2> gpurun_out/bench_c3_att.err
python -c "import json; d=json.load(open('gpurun_out/bench_c3_att.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline']['families']['window_attention'])"
