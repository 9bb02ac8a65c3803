#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 200 python -u tools/frame_kernel_check.py --cam 256x192 --aa 3 --scatter 2 --time "" 2>&1 | tee gpurun_out/fk1.txt
if grep -q "identical False" gpurun_out/fk1.txt; then exit 1; fi
timeout -k 10 200 python -u tools/frame_kernel_check.py --cam 200x150 --aa 1 --scatter 2 --time "1080x960:2" --rebalance 0,1,2 --reps 2 2>&1 | tee gpurun_out/fk3.txt
timeout -k 10 200 python -u tools/frame_kernel_check.py --cam 200x150 --aa 1 --scatter 2 --time "1080x960:4" --rebalance 0,2,3 --reps 2 2>&1 | tee gpurun_out/fk4.txt
