#!/bin/bash
# round 4: compute mode f16 (fp16 storage): first run — small goldens, c3 / c3 deformable / c5 deformable in both 16-bit modes, c3 timed in both
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_golden_gpu.py tests/test_configs_gpu.py -m gpu -x -q -s -k "bf16_modes_are_informational or c3_batch8 or c5_batch4_2048_bf16_deformable" > gpurun_out/r04_f16_tests.log 2>&1; rc=$?
grep -E "bf16|f16|passed|failed|Error|error" gpurun_out/r04_f16_tests.log | grep -v "^tests\|^E  \|def \|import" | tail -40
[ $rc -eq 0 ] || { tail -30 gpurun_out/r04_f16_tests.log; exit $rc; }
for i in 1 2; do for mode in bf16 f16; do
  python bench.py --config c3 --compute $mode --cpu-baseline off --profile-steps 0 --other-configs off --steps 20 --also "" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 $mode', d['value'], d['ms_per_step'])"
done; done | tee gpurun_out/r04_ab_f16.log
