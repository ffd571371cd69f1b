#!/bin/bash
# usage: tools/pmc_sq.sh <tag> <script and args>  -- shader-occupancy counters, one pass per group (kernel-trace only)
tag=$1; shift
export TMPDIR=/tmp
R=$PWD
cd /tmp
i=0
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmcsq_${tag}_$i -- python3 $R/"$@" > $R/gpurun_out/pmcsq_${tag}_$i.log 2>&1 || { echo "pass $i ($C) failed"; tail -3 $R/gpurun_out/pmcsq_${tag}_$i.log; }
done
cd $R
python tools/pmc_summary.py gpurun_out/pmcsq_${tag}_*/ 
