#!/bin/bash
set -e
mkdir -p gpurun_out
rm -f gpurun_out/fk_var.txt
for v in "" variants/librtamd_ahead1.so variants/librtamd_ahead3.so variants/librtamd_ahead64.so variants/librtamd_ppl5.so; do
echo "== ${v:-product}" | tee -a gpurun_out/fk_var.txt
if [ -n "$v" ]; then export RTAMD_LIB=rust-raytracer_amd/$v; fi
timeout -k 10 300 python -u tools/frame_kernel_check.py --cam 160x120 --aa 3 --time "1080x960:8" --rebalance 2 --step-min 64,40 --reps 2 2>&1 | grep "rebalance 2\|spp frame\|rror" | tee -a gpurun_out/fk_var.txt
done
