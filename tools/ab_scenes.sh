#!/bin/bash
# A/B of two library builds on the synthetic scenes (tools/scene_rates.py), alternating inside one GPU call.
A=$1; B=$2
for L in $A $B $A $B; do echo "== $L"; RTAMD_LIB=$PWD/$L timeout -k 10 200 python tools/scene_rates.py 2>&1 | tail -4 | awk '{print $1, $(NF-1)}'; done
