#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 300 python -u tools/tmax_hint_probe.py 2>&1 | grep "primary rays\|rror" | tee gpurun_out/hint.txt
