#!/bin/bash
# round 4: the whole GPU suite + smoke() on the final build
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r04_full_suite.log 2>&1; rc=$?
tail -5 gpurun_out/r04_full_suite.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
