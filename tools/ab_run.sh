#!/bin/bash
# usage: tools/ab_run.sh <out> <log2n...> -- runs tools/ab_cycle.py from every A/B tree present, alternating, twice
out=$1; shift
root=$(pwd)
: > $out
for rep in 1 2; do
  for n in "$@"; do
    for t in .ab_r03 . .ab_v1 .ab_v2 .ab_v3 .ab_v4; do
      [ -d $root/$t/gpu_se_amd/lib ] || continue
      (cd $root/$t && timeout -k 10 120 python tools/ab_cycle.py $n "$t" 2>/dev/null | grep "device us" >> $root/$out) || exit 1
    done
  done
done
cat $out
