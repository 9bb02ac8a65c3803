#!/bin/bash
set -e
mkdir -p gpurun_out
for o in static=6 static=8 static=7 static=8,chunk=64 static=6; do
echo -n "$o: "
timeout -k 10 200 python bench.py --steps 6 --no-cpu-baseline --no-pmc --opt $o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), d['stage_ms_per_frame'])"
done | tee gpurun_out/static8.txt
