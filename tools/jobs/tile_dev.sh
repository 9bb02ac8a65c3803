#!/bin/bash
set -e
mkdir -p gpurun_out
{
for t in 0 40 0 40 120 24 8; do
echo "tile $t"
timeout -k 10 300 python -u tools/frame_kernel_check.py --cam 256x192 --aa 2 --scatter 1 --time 1080x960:8 --rebalance 0 --reps 4 --tile $t 2>&1 | grep "frame_kernel=0"
done
} | tee gpurun_out/tile_dev.txt
