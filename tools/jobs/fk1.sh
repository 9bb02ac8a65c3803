#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 200 python -u tools/frame_kernel_check.py --cam 256x192 --aa 2 --time "" 2>&1 | tee gpurun_out/fk1.txt
grep -q "identical True" gpurun_out/fk1.txt
timeout -k 10 300 python -u tools/frame_kernel_check.py --cam 200x150 --aa 3 --time "1080x960:8" --step-min 64,32,16 2>&1 | tee gpurun_out/fk2.txt
