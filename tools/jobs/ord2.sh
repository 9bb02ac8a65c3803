set -x
L=rust-raytracer_amd/librtamd.so
T=$1
timeout -k 10 500 python -u tools/ordered_stress.py --rays 1000000 --rounds 1 > gpurun_out/r3_${T}_stress.log 2>&1 || exit 1
bash tools/sweep_opts.sh gpurun_out/r3_${T}_merged.txt $L "ordered=0 ordered=1" --workload bunny_grid_merged_230x230_7680x4320_256spp_depth5 --aa 2 --steps 3 || exit 1
bash tools/sweep_opts.sh gpurun_out/r3_${T}_flat.txt $L "ordered=0 ordered=1" --workload bunny_grid_flat_230x230_7680x4320_256spp_depth5 --aa 2 --steps 3 || exit 1
bash tools/sweep_opts.sh gpurun_out/r3_${T}_spheres.txt $L "ordered=0 ordered=1" --workload spheres_10k_3840x2160_64spp_depth5 --aa 4 --steps 3 || exit 1
bash tools/sweep_opts.sh gpurun_out/r3_${T}_spheres_small.txt $L "ordered=0 ordered=1" --workload spheres_10k_3840x2160_64spp_depth5 --cam 1080x960 --aa 4 --steps 5 || exit 1
for o in ; do
  echo "== merged 100x100, 1920x1080, ordered=$o" >> gpurun_out/r3_${T}_diag.txt
  timeout -k 10 300 python tools/phase_diag.py --scene merged --grid 100 --cam 1920x1080 --aa 2 --depth 5 --opt ordered=$o 2>&1 | grep -A5 "max_depth 5" >> gpurun_out/r3_${T}_diag.txt || exit 1
done
