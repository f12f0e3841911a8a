#!/bin/bash
# round 4, GPU call 7: full GPU suite on the final build + smoke()
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_t7.log 2>&1; RC=$?
tail -6 gpurun_out/r04_t7.log
if [ $RC -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" gpurun_out/r04_t7.log | head -20; exit 1; fi
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
