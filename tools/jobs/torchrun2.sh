#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline --no-pmc > gpurun_out/torchrun2.json 2> gpurun_out/torchrun2.err
python -c "import json; d=json.load(open('gpurun_out/torchrun2.json')); print(d['value'], d['ms_per_step'], d['n_gpus'], d['config'].get('backend'), d['config'].get('tile'), d.get('multi_gpu',{}).get('owned_tiles'))"
