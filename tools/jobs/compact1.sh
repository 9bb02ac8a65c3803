set -x
L=rust-raytracer_amd/librtamd.so
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_compact_tests.log 2>&1 || exit 1
bash tools/sweep_opts.sh gpurun_out/r3_compact_grid.txt $L "ordered=0 ordered=1 ordered=1,ostack=3" --workload bunny_grid_230x230_7680x4320_256spp_depth5 --aa 2 --steps 3 || exit 1
bash tools/sweep_opts.sh gpurun_out/r3_compact_spheres_small.txt $L "ordered=0 ordered=1 ordered=1,ostack=7" --workload spheres_10k_3840x2160_64spp_depth5 --cam 1080x960 --aa 4 --steps 5 || exit 1
bash tools/sweep_opts.sh gpurun_out/r3_compact_spheres.txt $L "ordered=1 ordered=1,ostack=7" --workload spheres_10k_3840x2160_64spp_depth5 --aa 4 --steps 3 || exit 1
timeout -k 10 600 python -u tools/ordered_stress.py --rays 2000000 --rounds 2 --scenes grid,spheres > gpurun_out/r3_compact_stress.log 2>&1 || exit 1
