#!/bin/bash
# usage: tools/prof.sh <tag> <python script and args...>   (on the GPU box, from the repo root)
# kernel-trace statistics of a command, summary copied next to the log under gpurun_out/
tag=$1; shift
export TMPDIR=/tmp
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/"$@" > $R/gpurun_out/prof_$tag.log 2>&1
rc=$?
cd $R
cat gpurun_out/prof_$tag/*/*kernel_stats.csv | cut -d, -f1-7 | sed 's/(.*)"/"/' | head -12
exit $rc
