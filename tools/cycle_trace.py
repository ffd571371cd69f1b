"""Per-cycle device time of the particle filter from its prior: one HIP event after every cycle's launch, the differences
are the launch-to-launch times (what bench.py averages).  The backing of bench.py's fresh-prior / steady-state figures and
of the bound on the slowest launch.  usage: cycle_trace.py [log2N] [cycles] [out.csv]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_se_amd, bench
from gpu_se_amd import sim_base

log2n = float(sys.argv[1]) if len(sys.argv) > 1 else 20
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 300
out = sys.argv[3] if len(sys.argv) > 3 else None
n = int(round(2 ** log2n))
io = bench.make_inputs(cycles, 1.0)
# (clocks up: a throw-away filter runs first, so that the trace shows the filter's transient and not the device's)
gpu_se_amd.seed(99)
_, _, _, warm = sim_base.get_parts(N_particles=n)
for u, z in io[:200]:
    warm.predict(u, 1.0); warm.update(u, z); warm.resample()
torch.cuda.synchronize()
gpu_se_amd.seed(1234)
_, _, _, pf = sim_base.get_parts(N_particles=n)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(cycles + 1)]
ev[0].record()
for c, (u, z) in enumerate(io):
    pf.predict(u, 1.0); pf.update(u, z); pf.resample()
    ev[c + 1].record()
torch.cuda.synchronize()
us = np.array([ev[c].elapsed_time(ev[c + 1]) * 1e3 for c in range(cycles)])
if out:
    with open(out, "w") as fh:
        fh.write("cycle,device_us\n")
        for c, v in enumerate(us):
            fh.write("%d,%.3f\n" % (c, v))
med = float(np.median(us))
print("N=2^%g  %d cycles from the prior: median %.2f us  mean %.2f  p99 %.2f  max %.2f (cycle %d)  max/median %.2f  launches > 2 x median: %d"
      % (log2n, cycles, med, us.mean(), np.percentile(us, 99), us.max(), int(us.argmax()), us.max() / med, int(np.sum(us > 2 * med))))
for lo in range(0, cycles, 50):
    seg = us[lo:lo + 50]
    print("  cycles %4d-%4d: mean %.2f us  median %.2f  max %.2f" % (lo, lo + len(seg) - 1, seg.mean(), np.median(seg), seg.max()))
