set -x
( while sleep 45; do date >> gpurun_out/heartbeat.txt; done ) &
HB=$!
bash tools/collect_profiles.sh r3d_spheres --workload spheres_10k_3840x2160_64spp_depth5 --steps 3 > gpurun_out/cp_spheres.log 2>&1 || { kill $HB; exit 1; }
bash tools/collect_profiles.sh r3d_primary4k --workload bunny_3840x2160_primary --steps 20 > gpurun_out/cp_primary.log 2>&1 || { kill $HB; exit 1; }
kill $HB
