#!/bin/bash
# A/B of two builds of the library inside one GPU call (different boxes differ by ~2 %): alternates the libraries.
# usage: bash tools/ab_bench.sh <libA.so> <libB.so> [rounds]
A=$1; B=$2; N=${3:-3}
for i in $(seq $N); do
  for L in $A $B; do
    RTAMD_LIB=$PWD/$L timeout -k 10 120 python bench.py --steps 4 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', round(d['value'],1), round(d['ms_per_step'],2))" || exit 1
  done
done
