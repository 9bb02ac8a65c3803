#!/bin/bash
set -e
mkdir -p gpurun_out
( while sleep 45; do date >> gpurun_out/heartbeat.txt; done ) &
HB=$!
run() { local name=$1; shift; echo -n "$name: "; timeout -k 10 400 python bench.py --no-cpu-baseline --no-pmc "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'Mrays/s', round(d['ms_per_step'],2), 'ms', d['stage_ms_per_frame'])"; }
{
run primary_4k --workload bunny_3840x2160_primary --steps 20
run spheres_4k_4spp --workload spheres_10k_3840x2160_64spp_depth5 --aa 2 --steps 4
run grid_8k_1spp --workload bunny_grid_230x230_7680x4320_256spp_depth5 --aa 1 --steps 3
run grid_flat_8k_1spp --workload bunny_grid_flat_230x230_7680x4320_256spp_depth5 --aa 1 --steps 3
run grid_merged_8k_1spp --workload bunny_grid_merged_230x230_7680x4320_256spp_depth5 --aa 1 --steps 3
} | tee gpurun_out/others.txt
kill $HB
