#!/bin/bash
set -e
mkdir -p gpurun_out
export RTAMD_LIB=rust-raytracer_amd/librtamd_diag.so
timeout -k 10 200 python -u tools/phase_diag.py --aa 8 --depth 5 2>&1 | grep -v "max_depth 1" | tail -24 | tee gpurun_out/diag_d.txt
