set -x
L=rust-raytracer_amd/librtamd.so
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/r3_lds1_tests.log 2>&1 || exit 1
bash tools/sweep_opts.sh gpurun_out/r3_lds1_bunny.txt $L "chunk=64 chunk=64 chunk=64" --steps 20 || exit 1
bash tools/sweep_opts.sh gpurun_out/r3_lds1_grid.txt $L "ordered=0 ordered=1 ordered=1,ostack=2 ordered=1,ostack=1" --workload bunny_grid_230x230_7680x4320_256spp_depth5 --aa 2 --steps 3 || exit 1
