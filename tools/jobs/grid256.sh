( while sleep 45; do date >> gpurun_out/heartbeat.txt; done ) &
HB=$!
bash tools/collect_profiles.sh r3b_grid --workload bunny_grid_230x230_7680x4320_256spp_depth5 --steps 1 --warmup 0 > gpurun_out/cp_grid.log 2>&1
RC=$?
kill $HB
exit $RC
