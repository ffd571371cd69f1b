#!/bin/bash
# usage: tools/ab_gsf.sh <out> <N...>: tools/gsf_bench.py from every A/B tree present, alternating, twice
out=$1; shift
root=$(pwd); : > $root/$out
for rep in 1 2; do
  for t in . .ab_v1 .ab_v2 .ab_v3 .ab_v4; do
    [ -d $root/$t/gpu_se_amd/lib ] || continue
    (cd $root/$t && timeout -k 10 200 python tools/gsf_bench.py "$@" 2>/dev/null | grep "GS-UKF" | sed "s|^|$t  |" >> $root/$out) || exit 1
  done
done
cat $root/$out
