#!/bin/bash
# per-launch table of one profiled forward: tools/dump_launches.sh <config> <out.csv> [ENV=VAL ...] [-- bench args]
CFG=$1; OUT=$2; shift 2
ENVS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do ENVS+=("$1"); shift; done
[ "$1" == "--" ] && shift
env "${ENVS[@]}" BRN_DUMP_LAUNCHES=$OUT python bench.py --config $CFG --cpu-baseline off --profile-steps 1 --other-configs off --steps 3 --warmup 2 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$CFG', d['value'], 'gemm ms', r['ms_per_step'], {k:v['ms'] for k,v in r['families'].items()})"
