#!/bin/bash
set -e
mkdir -p gpurun_out
{
timeout -k 10 200 python -u tools/strong_scaling_probe.py --cam 1080x960 --ranks 1,2,4,8
timeout -k 10 200 python -u tools/strong_scaling_probe.py --cam 1520x1360 --ranks 2
timeout -k 10 200 python -u tools/strong_scaling_probe.py --cam 2160x1920 --ranks 4
timeout -k 10 200 python -u tools/strong_scaling_probe.py --cam 3040x2720 --ranks 8
} 2>&1 | grep "rank 0 of" | tee gpurun_out/weak_probe.txt
