#!/bin/bash
set -e
mkdir -p gpurun_out
( while sleep 45; do date >> gpurun_out/heartbeat.txt; done ) &
HB=$!
{
timeout -k 10 300 python -u tools/two_contexts_probe.py --scene main --cam 1080x960 --aa 8 --contexts 2
timeout -k 10 300 python -u tools/two_contexts_probe.py --scene main --cam 1520x1360 --aa 8 --contexts 2
timeout -k 10 300 python -u tools/two_contexts_probe.py --scene spheres --cam 3840x2160 --aa 4 --contexts 2
timeout -k 10 300 python -u tools/two_contexts_probe.py --scene grid --cam 7680x4320 --aa 2 --contexts 2 --reps 2
} 2>&1 | grep "spp,\|identical\|rror" | tee gpurun_out/twoctx.txt
kill $HB
