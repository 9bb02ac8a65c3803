#!/bin/bash
set -e
mkdir -p gpurun_out
( while sleep 45; do date >> gpurun_out/heartbeat.txt; done ) &
HB=$!
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread 2>&1 | tee gpurun_out/gputests.txt | tail -6
timeout -k 10 600 python bench.py 2>gpurun_out/bench_err.txt | tee gpurun_out/bench_final.json
kill $HB
