#!/bin/bash
# A/B of an environment switch on one box, alternating: tools/ab_env.sh VAR "0 1" "c3 c5" [extra bench args]
VAR=$1; VALS=$2; CFGS=$3; shift 3
for i in 1 2; do
  for cfg in $CFGS; do
    for v in $VALS; do
      env $VAR=$v python bench.py --config $cfg --cpu-baseline off --profile-steps 0 --other-configs off --steps 20 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$cfg $VAR=$v', d['value'], d['ms_per_step'], d['max_abs_err_image0_vs_strided_golden'])"
    done
  done
done
