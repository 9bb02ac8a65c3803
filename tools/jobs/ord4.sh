set -x
L=rust-raytracer_amd/librtamd.so
timeout -k 10 600 python -u -m pytest tests/test_gpu_ordered.py -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_ord4_tests.log 2>&1 || exit 1
bash tools/sweep_opts.sh gpurun_out/r3_ord4_merged.txt $L "ordered=1 ordered=1,ostack=6 ordered=1,ostack=3 ordered=1,ostack=0" --workload bunny_grid_merged_230x230_7680x4320_256spp_depth5 --aa 2 --steps 3 || exit 1
bash tools/sweep_opts.sh gpurun_out/r3_ord4_spheres_small.txt $L "ordered=1 ordered=1,ostack=4 ordered=1,ostack=2" --workload spheres_10k_3840x2160_64spp_depth5 --cam 1080x960 --aa 4 --steps 5 || exit 1
