#!/bin/bash
# A/B one build with / without an environment switch, alternating: tools/ab_env.sh VAR=value [rounds]
KV=$1; N=${2:-3}
for i in $(seq $N); do
  env $KV python bench.py --steps 20 --warmup 5 --cpu-baseline off --also "" --profile-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$KV', d['ms_per_step'])"
  python bench.py --steps 20 --warmup 5 --cpu-baseline off --also "" --profile-steps 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['ms_per_step'])"
done
