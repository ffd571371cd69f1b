#!/bin/bash
# usage: tools/ab_gsf_prof.sh <N>: rocprofv3 kernel time of the GS-UKF cycle's kernels in every A/B tree present
root=$(pwd); export TMPDIR=/tmp
for t in . .ab_v1 .ab_v2 .ab_v3; do
  [ -d $root/$t/gpu_se_amd/lib ] || continue
  d=$root/gpurun_out/prof_gab_$(echo $t | tr -d './')x
  rm -rf $d
  (cd $root/$t && timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/gsf_bench.py $1 > /dev/null 2>&1) || exit 1
  python3 - $d $t <<'PY'
import sys,glob,csv
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if int(r['Calls'])>100: print(sys.argv[2], r['Name'][:50], r['Calls'], r['AverageNs'][:8], r['MinNs'], r['MaxNs'])
PY
done
