#!/bin/bash
set -e
mkdir -p gpurun_out
for o in static=6 static=5 static=4 chunk=128 chunk=256 static=6,min_batch=8 static=6,min_batch=2 static=6; do
echo -n "$o: "
timeout -k 10 200 python bench.py --steps 6 --no-cpu-baseline --no-pmc --opt $o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), d['stage_ms_per_frame'])"
done | tee gpurun_out/retune.txt
