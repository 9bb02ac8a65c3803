#!/bin/bash
set -e
mkdir -p gpurun_out
O=gpurun_out/fk_squares.txt
rm -f $O
run() { echo "== $*" | tee -a $O; timeout -k 10 300 python -u tools/frame_kernel_check.py "$@" 2>&1 | grep "spp frame\|identical False\|rror" | tee -a $O; }
export RTAMD_FRAME_SQUARES=1
run --cam 256x192 --aa 2 --scatter 1 --time 1080x960:8 --rebalance 0,1,2,3 --reps 3
run --cam 256x192 --aa 2 --scatter 2 --time 1080x960:2 --rebalance 0,1,2 --reps 3
unset RTAMD_FRAME_SQUARES
run --cam 256x192 --aa 2 --scatter 1 --time 1080x960:8 --rebalance 0,2 --reps 3
run --cam 256x192 --aa 2 --scatter 2 --time 1080x960:2 --rebalance 0,1,2 --reps 3
