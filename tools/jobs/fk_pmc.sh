#!/bin/bash
set -e
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in 1; do
  rm -rf $R/gpurun_out/pmc_fk$mode
  timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_fk$mode -o fk -- python3 $R/tools/frame_kernel_once.py $mode > $R/gpurun_out/pmc_fk$mode.log 2>&1
  python3 $R/tools/pmc_by_kernel.py $R/gpurun_out/pmc_fk$mode | tee $R/gpurun_out/pmc_fk$mode.txt
  rm -rf $R/gpurun_out/pmc_fk$mode
done
