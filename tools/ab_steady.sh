#!/bin/bash
# steady-state A/B: the cycle after 2000 cycles of tracking, round-3 tree against this one (with and without the weight scale)
root=$(pwd); out=$root/gpurun_out/ab_steady.txt; : > $out
for rep in 1 2; do
  (cd $root/.ab_r03 && GSE_AB_BURN=2000 timeout -k 10 200 python tools/ab_cycle.py 20 r03-steady 2>/dev/null | grep "device us" >> $out) || exit 1
  (cd $root && GSE_AB_BURN=2000 timeout -k 10 200 python tools/ab_cycle.py 20 r04-steady 2>/dev/null | grep "device us" >> $out) || exit 1
  (cd $root && GSE_AB_BURN=2000 GSE_BENCH_NO_SCALE=1 timeout -k 10 200 python tools/ab_cycle.py 20 r04-steady-noscale 2>/dev/null | grep "device us" >> $out) || exit 1
done
cat $out
