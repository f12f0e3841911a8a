#!/bin/bash
# Collects the rocprofv3 evidence of one bench configuration into gpurun_out/prof_<cfg>/ and condenses it into profiles/ (run on the GPU box):
#   tools/profile_config.sh c2 f32_split3 r03      tools/profile_config.sh c3 bf16 r03      tools/profile_config.sh c3 bf16 r03 deformable
# Passes (MI355X_MICROARCH.md: counters in their own runs, --kernel-trace only beside --pmc; python3 directly after --):
#   1 --kernel-trace --stats (1 warm-up + 3 timed forwards)   2 --pmc FETCH_SIZE   3 --pmc WRITE_SIZE   4 --pmc SQ_* GRBM_GUI_ACTIVE (1 forward each)
# All passes run the batch on ONE stream (BRN_SPLIT_STREAMS=1): per-kernel durations and counters of kernels that run alone, the
# same condition as bench.py's per-launch HIP-event brackets (its profiled steps).  The timed region of bench.py runs batches >= 4 as
# two sub-batches on two streams; pass 1b records that mode's kernel trace too (<tag>_kernel_stats_2streams_*: durations of overlapping
# kernels, not comparable with the roofline block).
set -e
CFG=$1; MODE=$2; TAG=${3:-r03}; DEFORM=$4
SUF=${CFG}${DEFORM:+_deformable}_${MODE}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/prof_${SUF}
rm -rf $D && mkdir -p $D
B="python3 bench.py --config $CFG --compute $MODE --cpu-baseline off --also= --profile-steps 0 --other-configs off --mask-error off ${DEFORM:+--deform-mode deformable}"
export BRN_SPLIT_STREAMS=1 BRN_BRANCH_STREAMS=0   # one stream, no auxiliary branch streams: kernels run alone
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- $B --steps 3 --warmup 1 > $D/stats.log 2>&1
echo "[profile] stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/fetch -- $B --steps 1 --warmup 0 > $D/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/write -- $B --steps 1 --warmup 0 > $D/write.log 2>&1
echo "[profile] hbm counters done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $D/sq -- $B --steps 1 --warmup 0 > $D/sq.log 2>&1
echo "[profile] sq counters done"
BRN_DUMP_LAUNCHES=$D/launches.csv python3 bench.py --config $CFG --compute $MODE --cpu-baseline off --also= --profile-steps 1 --other-configs off --mask-error off ${DEFORM:+--deform-mode deformable} --steps 5 --warmup 2 > $D/bench_short.json 2> $D/bench_short.err
python3 tools/make_profiles.py --tag $TAG --suffix $SUF --stats $D/stats --fetch $D/fetch --write $D/write --launches $D/launches.csv --forwards 4
python3 tools/pmc_sq_summary.py $D/sq profiles/${TAG}_pmc_sq_${SUF}.csv
if [ "$CFG" != "c2" ]; then
  export BRN_SPLIT_STREAMS=2; unset BRN_BRANCH_STREAMS
  rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats2 -- $B --steps 3 --warmup 1 > $D/stats2.log 2>&1
  python3 tools/make_profiles.py --tag $TAG --suffix 2streams_$SUF --stats $D/stats2 --forwards 4
  rm -rf $D/stats2
fi
mkdir -p gpurun_out/profiles_out && cp profiles/${TAG}_*${SUF}* gpurun_out/profiles_out/
rm -rf $D/stats $D/fetch $D/write $D/sq
echo "[profile] $SUF condensed"
