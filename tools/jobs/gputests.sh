#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread 2>&1 | tee gpurun_out/gputests.txt | tail -4
timeout -k 10 300 python -u tools/frame_kernel_check.py --scene grid --cam 160x120 --aa 2 --scatter 2 --time 1080x960:2 --rebalance 0 --reps 2 2>&1 | grep "spp frame\|False\|rror"
