#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 240 python -u -m pytest tests/test_gpu_configs.py -x -v -m gpu -k "tiny_sphere or spheres_10000" --timeout 120 --timeout-method thread 2>&1 | tee gpurun_out/tiny_tree.txt | tail -30
