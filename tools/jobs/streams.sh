#!/bin/bash
set -e
mkdir -p gpurun_out
for o in streams=1 streams=2 streams=3 streams=4 streams=2,max_paths=524288 streams=1; do
echo -n "$o: "
timeout -k 10 200 python bench.py --steps 5 --no-cpu-baseline --no-pmc --opt $o 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), d['stage_ms_per_frame'])"
done | tee gpurun_out/streams.txt
