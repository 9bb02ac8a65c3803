#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 200 python -u tools/step_split_probe.py 2>&1 | grep "per step" | tee gpurun_out/step_split.txt
