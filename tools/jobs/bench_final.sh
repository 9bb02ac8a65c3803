#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 400 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
cut -c1-400 gpurun_out/bench_final.json
timeout -k 10 400 python bench.py --gpus 2 --steps 3 --no-cpu-baseline --no-pmc > gpurun_out/bench_final_2ranks.json 2> gpurun_out/bench_final_2ranks.err
python -c "import json; d=json.load(open('gpurun_out/bench_final_2ranks.json')); print(d['value'], d['ms_per_step'], d['n_gpus'], d['config'].get('backend'), d.get('multi_gpu'))"
