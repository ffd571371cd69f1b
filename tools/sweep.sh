#!/bin/bash
# particle-count sweep of bench.py (one JSON line each), N = 2^16 .. 2^24: cycles/s, the kernel time and roofline fraction
# on a filter drawn from its prior and after 2000 cycles of tracking
for L in 16 18 20 21 22 23 24; do
  python bench.py --log2-particles $L --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=d['steady_state']
f=lambda v: 'n/a' if v is None else '%.3f' % v
print('N=2^%-4s %9.1f cycles/s  %8.2f us/cycle  %6.2f Gparticles/s  from the prior: kernel %7.2f us, HBM frac %s, vs reference passes %.3f   steady state: %7.2f us, vs reference passes %.3f   with an estimate every cycle: %8.2f us' % ('$L', d['value'], d['ms_per_step']*1e3, d['particles_per_s']/1e9, r['device_us']['cycle'], f(r['frac']), r['vs_reference_passes']['frac'], s['device_us_per_cycle'], s['frac_vs_reference_passes'], d['with_estimate_every_cycle']['ms_per_step']*1e3))"
done
