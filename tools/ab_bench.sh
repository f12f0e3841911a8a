#!/bin/bash
# A/B two builds of the library on the default bench, alternating: tools/ab_bench.sh <old.so> [rounds]
OLD=$1; N=${2:-4}
for i in $(seq $N); do
  for lib in "$OLD" ""; do
    BRN_LIB_PATH=$lib python bench.py --steps 20 --warmup 5 --cpu-baseline off --also "" --profile-steps 0 $BENCH_ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('${lib:-new}', d['ms_per_step'])"
  done
done
