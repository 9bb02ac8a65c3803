#!/bin/bash
set -e
mkdir -p gpurun_out
( while sleep 45; do date >> gpurun_out/heartbeat.txt; done ) &
HB=$!
for m in 0 1; do
echo -n "spheres 4K 16spp merge=$m: "
RTAMD_MERGE_RESOLVE=$m timeout -k 10 300 python bench.py --workload spheres_10k_3840x2160_64spp_depth5 --aa 4 --steps 3 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), d['stage_ms_per_frame'])"
echo -n "grid 8K 4spp merge=$m: "
RTAMD_MERGE_RESOLVE=$m timeout -k 10 300 python bench.py --workload bunny_grid_230x230_7680x4320_256spp_depth5 --aa 2 --steps 2 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), d['stage_ms_per_frame'])"
echo -n "merged grid 8K 1spp merge=$m: "
RTAMD_MERGE_RESOLVE=$m timeout -k 10 300 python bench.py --workload bunny_grid_merged_230x230_7680x4320_256spp_depth5 --aa 1 --steps 2 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), round(d['ms_per_step'],2), d['stage_ms_per_frame'])"
done | tee gpurun_out/merge2.txt
kill $HB
