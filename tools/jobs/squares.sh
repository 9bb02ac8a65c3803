#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread 2>&1 | tee gpurun_out/gputests.txt | tail -4
timeout -k 10 200 python bench.py --steps 5 --no-cpu --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), d['stage_ms_per_frame'])" | tee gpurun_out/squares_bench.txt
timeout -k 10 300 python -u tools/frame_kernel_check.py --cam 256x192 --aa 2 --scatter 1 --time 1080x960:8 --rebalance 0 --reps 3 2>&1 | grep "spp frame\|False\|rror" | tee -a gpurun_out/squares_bench.txt
timeout -k 10 300 python -u tools/frame_kernel_check.py --cam 256x192 --aa 2 --scatter 2 --time 1080x960:2 --rebalance 0 --reps 3 2>&1 | grep "spp frame\|False\|rror" | tee -a gpurun_out/squares_bench.txt
