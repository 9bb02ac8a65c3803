#!/bin/bash
set -e
mkdir -p gpurun_out
{
timeout -k 10 300 python -u tools/frame_kernel_check.py --cam 256x192 --aa 2 --scatter 2 --time 1080x960:2 --rebalance 0 --reps 3
timeout -k 10 300 python -u tools/frame_kernel_check.py --cam 256x192 --aa 2 --scatter 2 --time 1920x1080:2 --rebalance 0 --reps 2
timeout -k 10 300 python -u tools/frame_kernel_check.py --cam 256x192 --aa 2 --scatter 1 --time 1080x960:8 --rebalance 0 --reps 3
} 2>&1 | grep "spp frame\|False\|rror" | tee gpurun_out/fk_tree_check.txt
