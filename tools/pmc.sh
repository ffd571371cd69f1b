#!/bin/bash
# usage: tools/pmc.sh <tag> <script and args>  -- two separate counter passes (FETCH_SIZE, WRITE_SIZE), kernel-trace only
tag=$1; shift
export TMPDIR=/tmp
R=$PWD
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_$C -- python3 $R/"$@" > $R/gpurun_out/pmc_${tag}_$C.log 2>&1 || exit 1
done
cd $R
python tools/pmc_summary.py gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE
