#!/bin/bash
# particle-count sweep of bench.py (one JSON line each), N = 2^16 .. 2^24: cycles/s, the kernel time and roofline fraction
# on a filter drawn from its prior and after 2000 cycles of tracking
for L in 16 18 20 21 22 23 24; do
  python bench.py --log2-particles $L --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; s=d['steady_state']
print('N=2^%-4s %9.1f cycles/s  %8.2f us/cycle  %6.2f Gparticles/s  from the prior: kernel %7.2f us frac %.3f   steady state: %7.2f us frac %.3f' % ('$L', d['value'], d['ms_per_step']*1e3, d['particles_per_s']/1e9, r['device_us']['cycle'], r['frac'], s['device_us_per_cycle'], s['frac']))"
done
