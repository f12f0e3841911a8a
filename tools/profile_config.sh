#!/bin/bash
# Collects the rocprofv3 evidence of one bench configuration into gpurun_out/prof_<cfg>/ and condenses it into profiles/ (run on the GPU box):
#   tools/profile_config.sh c2 f32_split3      tools/profile_config.sh c3 bf16
# Passes (MI355X_MICROARCH.md: counters in their own runs, --kernel-trace only beside --pmc; python3 directly after --):
#   1 --kernel-trace --stats (1 warm-up + 3 timed forwards)   2 --pmc FETCH_SIZE   3 --pmc WRITE_SIZE   4 --pmc SQ_* GRBM_GUI_ACTIVE (1 forward each)
set -e
CFG=$1; MODE=$2; TAG=${3:-r02}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/prof_${CFG}
rm -rf $D && mkdir -p $D
B="python3 bench.py --config $CFG --cpu-baseline off --also= --profile-steps 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $D/stats -- $B --steps 3 --warmup 1 > $D/stats.log 2>&1
echo "[profile] stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/fetch -- $B --steps 1 --warmup 0 > $D/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/write -- $B --steps 1 --warmup 0 > $D/write.log 2>&1
echo "[profile] hbm counters done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $D/sq -- $B --steps 1 --warmup 0 > $D/sq.log 2>&1
echo "[profile] sq counters done"
BRN_DUMP_LAUNCHES=$D/launches.csv python3 bench.py --config $CFG --cpu-baseline off --also= --profile-steps 1 --steps 5 --warmup 2 > $D/bench_short.json 2> $D/bench_short.err
python3 tools/make_profiles.py --tag $TAG --suffix ${CFG}_${MODE} --stats $D/stats --fetch $D/fetch --write $D/write --launches $D/launches.csv --forwards 4
python3 tools/pmc_sq_summary.py $D/sq profiles/${TAG}_pmc_sq_${CFG}_${MODE}.csv
mkdir -p gpurun_out/profiles_out && cp profiles/${TAG}_*${CFG}_${MODE}* gpurun_out/profiles_out/
rm -rf $D/stats $D/fetch $D/write $D/sq
echo "[profile] $CFG $MODE condensed"
