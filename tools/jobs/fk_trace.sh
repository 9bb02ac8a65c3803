#!/bin/bash
set -e
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in 0 1; do
  rm -rf $R/gpurun_out/kt_fk$mode
  timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_fk$mode -o fk -- python3 $R/tools/frame_kernel_once.py $mode 8 > $R/gpurun_out/kt_fk$mode.log 2>&1
  f=$(find $R/gpurun_out/kt_fk$mode -name "*kernel_stats.csv" | head -1)
  cp "$f" $R/gpurun_out/frame_kernel_mode${mode}_kernel_stats.csv
  rm -rf $R/gpurun_out/kt_fk$mode
done
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_fk1 -o fk -- python3 $R/tools/frame_kernel_once.py 1 8 > $R/gpurun_out/pmc_fk1.log 2>&1
python3 $R/tools/pmc_by_kernel.py $R/gpurun_out/pmc_fk1 | tee $R/gpurun_out/frame_kernel_mode1_pmc_sq.txt
rm -rf $R/gpurun_out/pmc_fk1
timeout -s KILL 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_fk0 -o fk -- python3 $R/tools/frame_kernel_once.py 0 8 > $R/gpurun_out/pmc_fk0.log 2>&1
python3 $R/tools/pmc_by_kernel.py $R/gpurun_out/pmc_fk0 | tee $R/gpurun_out/frame_kernel_mode0_pmc_sq.txt
rm -rf $R/gpurun_out/pmc_fk0
