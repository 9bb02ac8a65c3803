#!/bin/bash
# Collects the profile set of one build on the GPU box into gpurun_out/prof_<tag>/:
# bench line, rocprofv3 kernel stats + trace summary, HBM traffic and SQ / TCC counter passes (one --pmc pass each).
# usage: bash tools/collect_profiles.sh <tag>      (run from the repository root through gpurun)
set -e
TAG=${1:-vX}
R=$PWD
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- $B > /dev/null 2> $O/ks.err
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > /dev/null 2> $O/fetch.err
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > /dev/null 2> $O/write.err
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq -- $B > /dev/null 2> $O/sq.err
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc -- $B > /dev/null 2> $O/tcc.err
cd $R
f() { find $O/$1 -name "*$2" | head -1; }
python tools/ktrace_summary.py $O/ks > $O/kernel_trace_summary.txt
cp "$(f ks kernel_stats.csv)" $O/kernel_stats.csv
python tools/traffic_from_pmc.py "$(f fetch counter_collection.csv)" "$(f write counter_collection.csv)" bunny_1080x960_64spp_depth5 $O/hbm_traffic.json
python tools/pmc_summary.py "$(f sq counter_collection.csv)" > $O/pmc_sq_summary.txt
python tools/pmc_summary.py "$(f tcc counter_collection.csv)" > $O/pmc_tcc_summary.txt
rm -rf $O/ks $O/fetch $O/write $O/sq $O/tcc
ls -la $O
