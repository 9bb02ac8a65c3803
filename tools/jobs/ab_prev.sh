#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread 2>&1 | tee gpurun_out/gputests.txt | tail -3
for lib in rust-raytracer_amd/variants/librtamd_prev.so rust-raytracer_amd/librtamd.so rust-raytracer_amd/variants/librtamd_prev.so rust-raytracer_amd/librtamd.so; do
echo -n "$(basename $lib): "
RTAMD_LIB=$lib timeout -k 10 200 python bench.py --steps 8 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), d['stage_ms_per_frame'])"
done | tee gpurun_out/ab_prev.txt
