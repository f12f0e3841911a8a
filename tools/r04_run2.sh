#!/bin/bash
# round 4, GPU call 2: new kernels (deform v2, rowln) — tests, then same-box A/Bs, attribution with p1 fp32, stage timers
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -x -q -k "linear_residual or deform or decblk" > gpurun_out/r04_t2a.log 2>&1; RC=$?
tail -15 gpurun_out/r04_t2a.log
if [ $RC -ne 0 ]; then echo "op tests failed (rc $RC): stopping"; exit 1; fi
timeout -k 10 600 python -m pytest tests/test_configs_gpu.py tests/test_golden_gpu.py -x -q -s -k "bf16 or golden" > gpurun_out/r04_t2b.log 2>&1; RC=$?
grep -E "max abs err|passed|failed|Error" gpurun_out/r04_t2b.log | tail -30
if [ $RC -ne 0 ]; then echo "model tests failed (rc $RC): stopping"; tail -30 gpurun_out/r04_t2b.log; exit 1; fi
timeout -k 10 200 bash tools/ab_env.sh BRN_DEFORM_V "1 2" "c3" --deform-mode deformable > gpurun_out/r04_ab_deformv.log 2>&1; cat gpurun_out/r04_ab_deformv.log
timeout -k 10 300 bash tools/ab_env.sh BRN_ROWLN "0 1 3" "c3 c5" > gpurun_out/r04_ab_rowln.log 2>&1; cat gpurun_out/r04_ab_rowln.log
timeout -k 10 200 bash tools/ab_env.sh BRN_P1_F32 "0 1" "c3" > gpurun_out/r04_ab_p1.log 2>&1; cat gpurun_out/r04_ab_p1.log
timeout -k 10 240 python tools/bf16_error_attrib.py --size 1024 --out gpurun_out/r04_attrib_1024_p1f32.json > gpurun_out/r04_attrib2.log 2>&1; grep -E "^---|all|backbone  |decoder|sq\+dec" gpurun_out/r04_attrib2.log
for cm in bf16 f32_split2; do
  timeout -k 10 200 python bench.py --config c3 --compute $cm --other-configs off --cpu-baseline off --steps 5 --warmup 2 --also "" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cm', d['value'], d['ms_per_step'], d.get('stage_ms_profiled'))"
done > gpurun_out/r04_stage_ms.log 2>&1; cat gpurun_out/r04_stage_ms.log
