#!/bin/bash
# A/B of every library under rust-raytracer_amd/variants against the in-tree one on the headline bench (alternating)
set -e
mkdir -p gpurun_out
for rep in 1 2; do
for lib in rust-raytracer_amd/librtamd.so rust-raytracer_amd/variants/*.so; do
echo -n "$(basename $lib): "
RTAMD_LIB=$lib timeout -k 10 200 python bench.py --steps 6 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), d['stage_ms_per_frame'])"
done
done | tee gpurun_out/ab_variants.txt
