set -x
L=rust-raytracer_amd/librtamd.so
bash tools/sweep_opts.sh gpurun_out/r3_ord_merged.txt $L "ordered=0 ordered=1" --workload bunny_grid_merged_230x230_7680x4320_256spp_depth5 --aa 2 --steps 3 || exit 1
bash tools/sweep_opts.sh gpurun_out/r3_ord_flat.txt $L "ordered=0 ordered=1 ordered=1,deepen=0" --workload bunny_grid_flat_230x230_7680x4320_256spp_depth5 --aa 2 --steps 3 || exit 1
bash tools/sweep_opts.sh gpurun_out/r3_ord_spheres.txt $L "ordered=0 ordered=1" --workload spheres_10k_3840x2160_64spp_depth5 --aa 4 --steps 3 || exit 1
bash tools/sweep_opts.sh gpurun_out/r3_ord_spheres_small.txt $L "ordered=0 ordered=1" --workload spheres_10k_3840x2160_64spp_depth5 --cam 1080x960 --aa 4 --steps 5 || exit 1
for o in 0 1; do
  echo "== merged 100x100, 1920x1080, ordered=$o" >> gpurun_out/r3_ord_diag.txt
  timeout -k 10 300 python tools/phase_diag.py --scene merged --grid 100 --cam 1920x1080 --aa 2 --opt ordered=$o >> gpurun_out/r3_ord_diag.txt 2>&1 || exit 1
done
