#!/bin/bash
# round 4, GPU call 3: full GPU suite on the current build; per-launch tables for the rowln and deform-v2 A/Bs; SQ counters of c3-deformable
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 840 python -m pytest tests -m gpu -x -q > gpurun_out/r04_t3.log 2>&1; RC=$?
tail -6 gpurun_out/r04_t3.log
if [ $RC -eq 124 ] || [ $RC -eq 137 ]; then echo "pytest timed out: stopping"; exit 1; fi
if [ $RC -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/r04_t3.log | head -20; fi
for v in 0 1; do bash tools/dump_launches.sh c3 gpurun_out/r04_launches_c3_rowln$v.csv BRN_ROWLN=$v; done
for v in 1 2; do bash tools/dump_launches.sh c3 gpurun_out/r04_launches_c3def_v$v.csv BRN_DEFORM_V=$v -- --deform-mode deformable; done
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
D=gpurun_out/prof_sq_c3def
rm -rf $D && mkdir -p $D
export BRN_SPLIT_STREAMS=1 BRN_BRANCH_STREAMS=0
timeout -k 10 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
    --kernel-trace --output-format csv -d $D/sq -- python3 bench.py --config c3 --deform-mode deformable --cpu-baseline off --also= --profile-steps 0 --other-configs off --steps 1 --warmup 0 > $D/sq.log 2>&1
python3 tools/pmc_sq_summary.py $D/sq gpurun_out/r04_pmc_sq_c3def_v2.csv && rm -rf $D/sq
head -8 gpurun_out/r04_pmc_sq_c3def_v2.csv | cut -c1-260
