#!/bin/bash
set -e
mkdir -p gpurun_out
RTAMD_HOST_TIMING=1 timeout -k 10 200 python -u tools/frame_kernel_once.py 0 8 4 2>&1 | tail -8 | tee gpurun_out/host_timing.txt
