"""Where the host's time per cycle goes (three methods from Python, mid-size filter): cProfile over many cycles."""
import os, sys, cProfile, pstats, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import gpu_se_amd, bench
from gpu_se_amd import sim_base
n = 1 << int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 14
io = bench.make_inputs(2000, 1.0)
gpu_se_amd.seed(2)
_, _, _, pf = sim_base.get_parts(N_particles=n)
def run(k):
    for i in range(k):
        u, z = io[i % len(io)]
        pf.predict(u, 1.0); pf.update(u, z); pf.resample()
run(500); torch.cuda.synchronize()
t0 = time.perf_counter(); run(5000); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue %.2f us/cycle, wall %.2f us/cycle" % ((t1 - t0) / 5000 * 1e6, (t2 - t0) / 5000 * 1e6))
pr = cProfile.Profile(); pr.enable(); run(5000); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
