#!/bin/bash
# round 4: how much of a batch-1 forward is idle time between dependent launches (one stream)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
D=gpurun_out/gaps_c2
rm -rf $D
BRN_SPLIT_STREAMS=1 BRN_BRANCH_STREAMS=0 rocprofv3 --kernel-trace --output-format csv -d $D -- python3 bench.py --config c2 --cpu-baseline off --also= --profile-steps 0 --other-configs off --mask-error off --steps 4 --warmup 2 > gpurun_out/gaps_c2.log 2>&1
python3 tools/launch_gaps.py $D 4
find $D -name "*kernel_trace.csv" -size +20M -delete
