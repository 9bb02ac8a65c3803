#!/bin/bash
# Runs tools/scene_rates.py once per library build in rust-raytracer_amd/variants/ (RTAMD_LIB selects the build).
# usage (through gpurun): bash tools/sweep_variants.sh <out file> [scene_rates.py arguments]
OUT=$1
shift
mkdir -p "$(dirname $OUT)"
: > $OUT
for lib in rust-raytracer_amd/variants/librtamd_*.so; do
    echo "== $lib" >> $OUT
    RTAMD_LIB=$PWD/$lib timeout -k 10 200 python tools/scene_rates.py "$@" >> $OUT 2>&1 || { echo "FAILED ($?)" >> $OUT; exit 1; }
done
cat $OUT
