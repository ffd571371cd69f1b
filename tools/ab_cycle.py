"""A/B of the cycle kernel: device time per cycle at N = 2^log2n in THIS tree, printed; run it from two trees in one gpurun
call, alternating, to compare builds on the same box.  usage: ab_cycle.py <log2n> <label>"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import gpu_se_amd
from gpu_se_amd import sim_base
import bench
log2n, label = float(sys.argv[1]), sys.argv[2]
n = int(round(2 ** log2n))
io = bench.make_inputs(1500, 1.0)
bench.device_warm_up(torch, n, 1.0, cycles=500)
gpu_se_amd.seed(2)
_, _, _, pf = sim_base.get_parts(N_particles=n)
if os.environ.get("GSE_BENCH_NO_SCALE") == "1":
    pf.weight_scale = False
if os.environ.get("GSE_BENCH_NO_CLUSTER") == "1":
    pf.cluster = False
at = [0]
def cycle(_i):
    u, z = io[at[0] % len(io)]; at[0] += 1
    pf.predict(u, 1.0); pf.update(u, z); pf.resample()
for i in range(int(os.environ.get("GSE_AB_BURN", "10"))): cycle(i)
import time
out, host = [], []
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(300): cycle(i)
    host.append((time.perf_counter() - t0) / 300 * 1e6)      # enqueue time per cycle (the device runs behind)
    torch.cuda.synchronize()
    out.append(bench.time_method_events(torch, cycle, 300) * 1e6)
print("%-22s N=2^%g  device us/cycle: %s   host enqueue us/cycle: %s" % (label, log2n, " ".join("%.2f" % v for v in out), " ".join("%.2f" % v for v in host)), flush=True)
