#!/bin/bash
set -e
mkdir -p gpurun_out
( while sleep 45; do date >> gpurun_out/heartbeat.txt; done ) &
HB=$!
{
echo "== headline (bench.py), product = runs of skipped objects in one step, variant = one object per step"
bash tools/ab_bench.sh rust-raytracer_amd/librtamd.so rust-raytracer_amd/variants/librtamd_noskiprun.so 3
for wl in spheres_10k_3840x2160_64spp_depth5 bunny_grid_230x230_7680x4320_256spp_depth5; do
  for L in rust-raytracer_amd/librtamd.so rust-raytracer_amd/variants/librtamd_noskiprun.so rust-raytracer_amd/librtamd.so; do
    RTAMD_LIB=$PWD/$L timeout -k 10 200 python bench.py --workload $wl --aa 2 --steps 2 --no-cpu --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl', '$L', round(d['value'],1), round(d['ms_per_step'],2))"
  done
done
} | tee gpurun_out/ab_skip_run.txt
timeout -k 10 600 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread 2>&1 | tail -3
kill $HB
