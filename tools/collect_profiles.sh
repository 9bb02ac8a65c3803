#!/bin/bash
# Collects the profile set of one build on the GPU box into gpurun_out/prof_<tag>/:
# bench line (with its own counter passes and CPU baseline), rocprofv3 kernel stats + trace summary, SQ / TCC counter summaries.
# usage: bash tools/collect_profiles.sh <tag> [bench.py arguments, e.g. --workload NAME --aa 4 --steps 2]
#        (run from the repository root through gpurun)
set -e
TAG=${1:-vX}
shift || true
R=$PWD
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
timeout -k 10 500 python bench.py "$@" > $O/bench.json 2> $O/bench.err
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --inner --steps 2 --warmup 0 $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -- $B > /dev/null 2> $O/ks.err
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq -- $B > /dev/null 2> $O/sq.err
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $O/sq2 -- $B > /dev/null 2> $O/sq2.err
timeout -s KILL 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/tcc -- $B > /dev/null 2> $O/tcc.err
cd $R
f() { find $O/$1 -name "*$2" | head -1; }
python tools/ktrace_summary.py $O/ks > $O/kernel_trace_summary.txt
cp "$(f ks kernel_stats.csv)" $O/kernel_stats.csv
python tools/pmc_summary.py "$(f sq counter_collection.csv)" > $O/pmc_sq_summary.txt
python tools/pmc_summary.py "$(f sq2 counter_collection.csv)" > $O/pmc_wave_cycles_summary.txt
python tools/pmc_summary.py "$(f tcc counter_collection.csv)" > $O/pmc_tcc_summary.txt
rm -rf $O/ks $O/sq $O/sq2 $O/tcc
ls -la $O
