#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread 2>&1 | tee gpurun_out/gputests.txt | tail -4
for m in 1 0 1 0; do
echo -n "thin_tail=$m: "
RTAMD_THIN_TAIL=$m timeout -k 10 200 python bench.py --steps 6 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), d['stage_ms_per_frame'])"
done | tee gpurun_out/thin_tail.txt
RTAMD_THIN_TAIL=1 timeout -k 10 200 python -u tools/trace_floor_probe.py 2>&1 | grep "rays:" | head -11 | tee -a gpurun_out/thin_tail.txt
