set -e
python -m pytest tests -m gpu -x -q 2>&1 | tail -5
echo "[final] tests done"
bash tools/profile_config.sh c2 f32_split3
bash tools/profile_config.sh c3 bf16
python bench.py --config c5 --cpu-baseline off --also= > gpurun_out/bench_c5.json 2> gpurun_out/bench_c5.err
python -c "import json; d=json.load(open('gpurun_out/bench_c5.json')); print('c5', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
