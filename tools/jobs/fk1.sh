#!/bin/bash
set -e
mkdir -p gpurun_out
{
echo "== product (5 pixels per lane, 2 blocks per CU)"
timeout -k 10 300 python -u tools/frame_kernel_check.py --cam 200x150 --aa 2 --scatter 2 --time 1080x960:2 --rebalance 0 --reps 3 2>&1 | grep "rebalance 0\|spp frame\|False\|rror"
export RTAMD_LIB=rust-raytracer_amd/variants/librtamd_ppl9.so
for b in 2 1; do
echo "== 9 pixels per lane, $b blocks per CU"
timeout -k 10 300 python -u tools/frame_kernel_check.py --cam 200x150 --aa 2 --scatter 2 --time 1080x960:2 --rebalance 0 --blocks $b --reps 3 2>&1 | grep "rebalance 0\|spp frame\|False\|rror"
timeout -k 10 300 python -u tools/frame_kernel_check.py --cam 200x150 --aa 2 --scatter 1 --time 1080x960:8 --rebalance 0 --blocks $b --reps 3 2>&1 | grep "spp frame\|False\|rror"
done
} | tee gpurun_out/fk_ppl9.txt
