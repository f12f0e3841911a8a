#!/bin/bash
# A/B of two library builds on one box, alternating: tools/ab_lib.sh <other.so> "c3 c5" [rounds] [extra bench args]
LIB=$1; CFGS=$2; N=${3:-3}; shift 3
for i in $(seq $N); do
  for cfg in $CFGS; do
    for lib in "" "$LIB"; do
      BRN_LIB_PATH=$lib python bench.py --config $cfg --steps 20 --warmup 5 --cpu-baseline off --also "" --profile-steps 0 --other-configs off "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', '${lib:-default}', d['value'], d['ms_per_step'])"
    done
  done
done
