#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 200 python -u tools/frame_kernel_check.py --cam 256x192 --aa 3 --time "1080x960:8" --rebalance 2 --reps 2 2>&1 | tee gpurun_out/fk1.txt
export RTAMD_LIB=rust-raytracer_amd/librtamd_diag.so
timeout -k 10 200 python -u tools/phase_diag.py --aa 8 --depth 5 --frame --opt rebalance=2 2>&1 | grep -v "max_depth 1" | tail -10 | tee gpurun_out/fk_diag_b.txt
