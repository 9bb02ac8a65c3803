#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 1000 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread 2>&1 | tee gpurun_out/gputests.txt | tail -15
