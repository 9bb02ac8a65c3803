#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread 2>&1 | tee gpurun_out/gputests.txt | tail -3
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout -k 10 400 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-330
