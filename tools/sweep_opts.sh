#!/bin/bash
# Runs bench.py (no CPU leg, no counter passes) with one library build over a list of --opt strings.
# usage (through gpurun): bash tools/sweep_opts.sh <out file> <lib.so> "<opt1> <opt2> ..." [bench.py arguments]
OUT=$1; LIB=$2; OPTS=$3
shift 3
mkdir -p "$(dirname $OUT)"
for o in $OPTS; do
    RTAMD_LIB=$PWD/$LIB timeout -k 10 300 python bench.py --no-cpu-baseline --no-pmc --opt "$o" "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read())
print('%-40s %-24s %9.1f Mrays/s  %8.2f ms/step  trace kernel %8.1f Mrays/s' % ('$LIB'.split('/')[-1], '$o', j['value'], j['ms_per_step'], j.get('trace_kernel_mrays_per_s', 0)))" >> $OUT || { echo "FAILED $o" >> $OUT; exit 1; }
done
