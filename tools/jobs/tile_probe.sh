#!/bin/bash
set -e
mkdir -p gpurun_out
for t in 64 40 64 40 64 40 120 24; do
sed "s/^TILE = 64/TILE = $t/" bench.py > bench_tile_probe.py
echo -n "tile $t: "
timeout -k 10 200 python bench_tile_probe.py --steps 10 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), d['stage_ms_per_frame'])"
done | tee gpurun_out/tile_probe.txt
rm -f bench_tile_probe.py
