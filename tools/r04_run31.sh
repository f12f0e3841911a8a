#!/bin/bash
cd "$(dirname "$0")/.."
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "window_attention or half2" 2>&1 | tail -2
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
