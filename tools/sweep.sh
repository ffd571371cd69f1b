#!/bin/bash
# particle-count sweep of bench.py (one JSON line each), N = 2^16 .. 2^24
for L in 16 18 20 21 22 24; do
  python bench.py --log2-particles $L --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('N=2^%-4s %9.1f cycles/s  %8.2f us/cycle  %6.2f Gparticles/s  kernel %7.2f us  frac %.3f' % ('$L', d['value'], d['ms_per_step']*1e3, d['particles_per_s']/1e9, r['device_us']['cycle'], r['frac']))"
done
