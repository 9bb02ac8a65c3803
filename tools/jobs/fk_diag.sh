#!/bin/bash
set -e
mkdir -p gpurun_out
export RTAMD_LIB=rust-raytracer_amd/librtamd_diag.so
for r in 0 2; do
  echo "== rebalance=$r"
  timeout -k 10 200 python -u tools/phase_diag.py --aa 8 --depth 5 --frame --opt rebalance=$r 2>&1 | grep -v "max_depth 1" | tail -9
done | tee gpurun_out/fk_diag_reb.txt
