#!/bin/bash
set -e
mkdir -p gpurun_out
( while sleep 45; do date >> gpurun_out/heartbeat.txt; done ) &
HB=$!
timeout -k 10 400 python bench.py --gpus 2 --steps 2 --no-cpu-baseline --no-pmc > gpurun_out/bunny_bench_gloo_2ranks_one_gpu_weak.json 2> gpurun_out/w2.err
timeout -k 10 600 python bench.py --gpus 8 --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > gpurun_out/bunny_bench_gloo_8ranks_one_gpu_weak.json 2> gpurun_out/w8.err
kill $HB
python -c "
import json
for n in (2,8):
    d=json.load(open(f'gpurun_out/bunny_bench_gloo_{n}ranks_one_gpu_weak.json')); print(n, d['scaling'], d['config']['width'], d['config']['height'], d['config']['backend'], d['multi_gpu']['owned_tiles'][:3], round(d['value'],1))"
