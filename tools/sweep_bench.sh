#!/bin/bash
# Runs bench.py (no CPU leg, no counter passes) once per library build in rust-raytracer_amd/variants/, twice round-robin.
# usage (through gpurun): bash tools/sweep_bench.sh <out file> [bench.py arguments]
OUT=$1
shift
mkdir -p "$(dirname $OUT)"
: > $OUT
for rep in 1 2; do
for lib in rust-raytracer_amd/variants/librtamd_*.so; do
    RTAMD_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-pmc "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('%-50s %9.1f Mrays/s  %8.2f ms/step  trace kernel %8.1f Mrays/s  stages %s' % ('$lib'.split('/')[-1], j['value'], j['ms_per_step'], j.get('trace_kernel_mrays_per_s', 0), j.get('stage_ms_per_frame')))" >> $OUT || { echo "FAILED $lib" >> $OUT; exit 1; }
done
done
sort $OUT
