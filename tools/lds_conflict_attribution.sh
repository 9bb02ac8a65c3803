#!/bin/bash
# Which LDS accesses of the scheduler kernel the bank-conflict cycles belong to: one rocprofv3 counter pass per library build,
# the plain one and one per kind of access issued twice (make -C rust-raytracer_amd/csrc lds-dup; RTAMD_LDS_DUP in rt_sched.h).
# The growth of SQ_LDS_BANK_CONFLICT and SQ_LDS_IDX_ACTIVE over the plain build is that kind's share.
# usage (through gpurun, from the repository root): bash tools/lds_conflict_attribution.sh <out file> [bench.py arguments]
OUT=$PWD/$1
shift
R=$PWD
mkdir -p "$(dirname $OUT)"
: > $OUT
cd /tmp && export TMPDIR=/tmp
for lib in librtamd.so variants/librtamd_ldsdup1.so variants/librtamd_ldsdup2.so variants/librtamd_ldsdup4.so variants/librtamd_ldsdup8.so variants/librtamd_ldsdup16.so; do
    D=/tmp/ldsattr_$$
    rm -rf $D
    RTAMD_LIB=$R/rust-raytracer_amd/$lib timeout -s KILL 200 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_LDS_ADDR_CONFLICT --output-format csv -d $D -- python3 $R/bench.py --inner --steps 1 --warmup 0 "$@" > /dev/null 2> $D.err || { echo "FAILED $lib" >> $OUT; exit 1; }
    echo "== $lib" >> $OUT
    python3 $R/tools/pmc_summary.py "$(find $D -name '*counter_collection.csv' | head -1)" | grep -A8 "k_wave_trace" >> $OUT
    rm -rf $D $D.err
done
cat $OUT
