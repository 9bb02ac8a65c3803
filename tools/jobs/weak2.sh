#!/bin/bash
set -e
mkdir -p gpurun_out
show() { python -c "import json,sys; d=json.load(open(sys.argv[1])); print(d['value'], d['ms_per_step'], d['n_gpus'], d['scaling'], d['config']['width'], d['config']['height'], d['config']['parallelism'], d['multi_gpu']['owned_tiles'], d['rays_per_step'])" $1; }
timeout -k 10 400 python bench.py --gpus 2 --steps 2 --no-cpu-baseline --no-pmc > gpurun_out/weak2.json 2> gpurun_out/weak2.err
show gpurun_out/weak2.json
timeout -k 10 400 python bench.py --gpus 2 --steps 2 --strong --no-cpu-baseline --no-pmc > gpurun_out/strong2.json 2> gpurun_out/strong2.err
show gpurun_out/strong2.json
timeout -k 10 600 python bench.py --gpus 8 --steps 1 --warmup 1 --no-cpu-baseline --no-pmc > gpurun_out/weak8.json 2> gpurun_out/weak8.err
show gpurun_out/weak8.json
