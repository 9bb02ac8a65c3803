set -x
( while sleep 45; do date >> gpurun_out/heartbeat.txt; done ) &
HB=$!
bash tools/collect_profiles.sh r3d_grid --workload bunny_grid_230x230_7680x4320_256spp_depth5 --steps 1 --warmup 0 > gpurun_out/cp_grid.log 2>&1 || { kill $HB; exit 1; }
bash tools/collect_profiles.sh r3d_grid_flat --workload bunny_grid_flat_230x230_7680x4320_256spp_depth5 --steps 1 --warmup 0 > gpurun_out/cp_flat.log 2>&1 || { kill $HB; exit 1; }
bash tools/collect_profiles.sh r3d_grid_merged --workload bunny_grid_merged_230x230_7680x4320_256spp_depth5 --steps 1 --warmup 0 > gpurun_out/cp_merged.log 2>&1 || { kill $HB; exit 1; }
kill $HB
