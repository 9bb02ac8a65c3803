#!/bin/bash
set -e
mkdir -p gpurun_out
O=gpurun_out/psplit.txt
rm -f $O
for sp in 1 1,1 1,1,1 1,2 2,1 1,1,1,1 1,2,1 1 1,1; do
echo "== RTAMD_PRIMARY_SPLIT=$sp" | tee -a $O
RTAMD_PRIMARY_SPLIT=$sp timeout -k 10 300 python -u tools/frame_kernel_check.py --cam 256x192 --aa 3 --scatter 1 --time 1080x960:8 --rebalance 0 --reps 3 2>&1 | grep "frame_kernel=0\|identical False\|rror" | tee -a $O
done
