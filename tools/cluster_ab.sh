#!/bin/bash
# usage: tools/cluster_ab.sh <out> <log2n...>: rocprofv3 kernel time of the cluster cycle kernel in every A/B tree present
out=$1; shift
root=$(pwd)
export TMPDIR=/tmp
: > $root/$out
for n in "$@"; do
  for t in . .ab_v1 .ab_v2 .ab_v3; do
    [ -d $root/$t/gpu_se_amd/lib ] || continue
    d=$root/gpurun_out/prof_cab_${n}_$(echo $t | tr -d './')
    rm -rf $d
    (cd $root/$t && timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/cluster_prof.py $n > /dev/null 2>&1) || { echo "failed $t $n" >> $root/$out; exit 1; }
    echo "$t N=2^$n $(cat $d/*/*kernel_stats.csv | grep -E 'pf_cluster_kernel<true, true, true' | cut -d, -f1-7 | sed 's/(.*)"/"/')" >> $root/$out
  done
done
cat $root/$out
