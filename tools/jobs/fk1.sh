#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 240 python -u tools/frame_kernel_check.py --cam 256x192 --aa 2 --time "" 2>&1 | tee gpurun_out/fk1.txt
