#!/bin/bash
set -e
mkdir -p gpurun_out
for n in 1 2 4 8; do
echo "== rank 0 of $n"
timeout -k 10 300 python -u tools/frame_kernel_check.py --cam 100x80 --aa 1 --scatter 1 --time 1080x960:8 --rebalance 0 --tile 64 --ranks $n --reps 3 2>&1 | grep "spp frame\|False\|rror"
done | tee gpurun_out/fk_ranks.txt
