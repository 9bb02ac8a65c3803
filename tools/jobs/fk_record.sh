#!/bin/bash
set -e
mkdir -p gpurun_out
O=gpurun_out/ab_frame_kernel_runs.txt
rm -f $O
( while sleep 45; do date >> gpurun_out/heartbeat.txt; done ) &
HB=$!
run() { echo "== $*" | tee -a $O; timeout -k 10 300 python -u tools/frame_kernel_check.py "$@" 2>&1 | grep "spp frame\|identical False\|rror" | tee -a $O; }
run --cam 200x150 --aa 2 --scatter 1 --time 1080x960:8 --rebalance 0,2,3 --reps 3
run --cam 200x150 --aa 2 --scatter 2 --time 1080x960:2 --rebalance 0 --reps 3
run --cam 200x150 --aa 2 --scatter 2 --time 1080x960:4 --rebalance 0,2 --reps 2
run --cam 200x150 --aa 2 --scatter 3 --depth 4 --time 1080x960:2 --rebalance 0 --reps 2
run --cam 200x150 --aa 1 --scatter 2 --time 640x480:2 --rebalance 0 --reps 3
run --cam 200x150 --aa 1 --scatter 2 --time 1920x1080:2 --rebalance 0 --reps 2
run --cam 200x150 --aa 1 --scatter 2 --time 3840x2160:1 --rebalance 0 --reps 2
for sc in spheres grid textured; do run --scene $sc --cam 160x120 --aa 1 --scatter 2 --time 1080x960:2 --rebalance 0 --reps 2; done
for sc in spheres grid; do run --scene $sc --cam 160x120 --aa 1 --scatter 1 --time 1080x960:4 --rebalance 2 --reps 2; done
export RTAMD_LIB=rust-raytracer_amd/librtamd_diag.so
echo "== phase shares, frame kernel, scatter 1, 64 spp (diagnostic build)" | tee -a $O
timeout -k 10 200 python -u tools/phase_diag.py --aa 8 --depth 5 --frame --opt rebalance=2 2>&1 | grep -v "max_depth 1" | tail -10 | tee -a $O
echo "== phase shares, frame kernel, scatter 2, 4 spp (diagnostic build)" | tee -a $O
timeout -k 10 200 python -u tools/phase_diag.py --aa 2 --depth 5 --frame --scatter 2 --opt rebalance=0 2>&1 | grep -v "max_depth 1" | tail -10 | tee -a $O
unset RTAMD_LIB
kill $HB
