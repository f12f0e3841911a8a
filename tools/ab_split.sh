#!/bin/bash
# A/B on one box: 1 / 2 / 4 sub-batches on as many streams (BRN_SPLIT_STREAMS), configs c3 and c5, alternating
for i in 1 2; do
  for cfg in c3 c5; do
    for sp in 1 2 4; do
      BRN_SPLIT_STREAMS=$sp python bench.py --config $cfg --cpu-baseline off --profile-steps 0 --other-configs off --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg parts=$sp', d['value'], d['ms_per_step'], d['max_abs_err_image0_vs_strided_golden'])"
    done
  done
done
for sp in 1 2; do BRN_SPLIT_STREAMS=$sp python bench.py --config c3 --deform-mode deformable --cpu-baseline off --profile-steps 0 --other-configs off --steps 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 deformable parts=$sp', d['value'], d['ms_per_step'], d['max_abs_err_image0_vs_strided_golden'])"; done
