#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 200 python -u tools/trace_floor_probe.py 2>&1 | grep "rays:" | tee gpurun_out/trace_floor.txt
