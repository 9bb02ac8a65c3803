set -x
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_final_tests.log 2>&1 || exit 1
bash tools/collect_profiles.sh r3b_bunny --steps 20 > gpurun_out/cp_bunny.log 2>&1 || exit 1
bash tools/collect_profiles.sh r3b_spheres --workload spheres_10k_3840x2160_64spp_depth5 --steps 3 > gpurun_out/cp_spheres.log 2>&1 || exit 1
bash tools/collect_profiles.sh r3b_primary4k --workload bunny_3840x2160_primary --steps 20 > gpurun_out/cp_primary.log 2>&1 || exit 1
