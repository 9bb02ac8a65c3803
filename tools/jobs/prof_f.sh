#!/bin/bash
set -e
( while sleep 45; do date >> gpurun_out/heartbeat.txt; done ) &
HB=$!
bash tools/collect_profiles.sh r3f_bunny
kill $HB
cat gpurun_out/prof_r3f_bunny/bench.json | cut -c1-300
