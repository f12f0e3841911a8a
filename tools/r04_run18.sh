#!/bin/bash
# round 4: compute mode f16: the op / model / config tests that carry it + the same-box A/B against bf16
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -k "f16 or bf16_mode or swin_configs_in_every or pieces_in_every" > gpurun_out/r04_f16_tests2.log 2>&1; rc=$?
tail -5 gpurun_out/r04_f16_tests2.log
[ $rc -eq 0 ] || { tail -40 gpurun_out/r04_f16_tests2.log; exit $rc; }
for i in 1 2; do for cfg in c3 c5; do for mode in bf16 f16; do
  python bench.py --config $cfg --compute $mode --cpu-baseline off --profile-steps 0 --other-configs off --steps 20 --also "" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg $mode', d['value'], d['ms_per_step'])"
done; done; done | tee gpurun_out/r04_ab_f16b.log
