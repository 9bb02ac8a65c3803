#!/bin/bash
set -e
mkdir -p gpurun_out
export RTAMD_LIB=rust-raytracer_amd/librtamd_diag.so
timeout -k 10 200 python -u tools/phase_diag.py --aa 2 --depth 5 --frame --scatter 2 --opt rebalance=0 2>&1 | grep -v "max_depth 1" | tail -10 | tee gpurun_out/fk_diag_tree.txt
