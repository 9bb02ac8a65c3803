#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 240 python -u -m pytest tests/test_gpu_render.py -x -v -m gpu -k "tiles_equal" --timeout 120 --timeout-method thread 2>&1 | tail -6
