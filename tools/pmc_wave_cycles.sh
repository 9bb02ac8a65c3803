# Where the wave cycles of the kernels go: two rocprofv3 --pmc passes (issue / wait / instruction-class counters), summarised per kernel.
# usage (through gpurun, from the repository root): bash tools/pmc_wave_cycles.sh
R=$PWD; O=$R/gpurun_out/pmc2; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_IFETCH --output-format csv -d $O/a -- $B > /dev/null 2> $O/a.err
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INST_LEVEL_LDS --output-format csv -d $O/b -- $B > /dev/null 2> $O/b.err
cd $R
for d in a b; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tools/pmc_summary.py $f | grep -A12 "k_wave_trace" ; tail -2 $O/$d.err; done
rm -rf $O/a $O/b
