#!/bin/bash
set -e
mkdir -p gpurun_out
O=gpurun_out/ab_squares_workloads.txt
rm -f $O
( while sleep 45; do date >> gpurun_out/heartbeat.txt; done ) &
HB=$!
run() { # name lib args...
  local name=$1 lib=$2; shift 2
  echo -n "$name $(basename $lib): " | tee -a $O
  RTAMD_LIB=$lib timeout -k 10 400 python bench.py --no-cpu-baseline --no-pmc "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],2), 'ms', d['stage_ms_per_frame'])" | tee -a $O
}
for lib in rust-raytracer_amd/variants/librtamd_rows.so rust-raytracer_amd/librtamd.so rust-raytracer_amd/variants/librtamd_rows.so rust-raytracer_amd/librtamd.so; do
  run primary_4k $lib --workload bunny_3840x2160_primary --steps 20
done
for lib in rust-raytracer_amd/variants/librtamd_rows.so rust-raytracer_amd/librtamd.so; do
  run spheres_4k_4spp $lib --workload spheres_10k_3840x2160_64spp_depth5 --aa 2 --steps 4
  run grid_8k_1spp $lib --workload bunny_grid_230x230_7680x4320_256spp_depth5 --aa 1 --steps 3
  run grid_flat_8k_1spp $lib --workload bunny_grid_flat_230x230_7680x4320_256spp_depth5 --aa 1 --steps 3
  run grid_merged_8k_1spp $lib --workload bunny_grid_merged_230x230_7680x4320_256spp_depth5 --aa 1 --steps 3
done
kill $HB
