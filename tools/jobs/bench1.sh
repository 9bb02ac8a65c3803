set -x
timeout -k 10 500 python bench.py > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err || exit 1
timeout -k 10 400 python bench.py --gpus 2 --no-cpu-baseline --no-pmc --steps 5 > gpurun_out/r3_bench_gloo2.json 2> gpurun_out/r3_bench_gloo2.err || exit 1
timeout -k 10 500 python bench.py --gpus 8 --no-cpu-baseline --no-pmc --steps 3 > gpurun_out/r3_bench_gloo8.json 2> gpurun_out/r3_bench_gloo8.err || exit 1
