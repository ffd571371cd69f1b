"""Cycles of mid-size filters for a rocprofv3 kernel trace: tile kernels then cluster kernels at each size (the kernel durations
are the profiler's; the host's enqueue rate, 11 us per cycle from Python, hides them from event timing).
usage: rocprofv3 --kernel-trace --stats -d out -- python3 tools/small_n_prof.py [log2N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_se_amd
from gpu_se_amd import sim_base
import bench
sizes = [int(v) for v in sys.argv[1:]] or [14, 16, 18]
io = bench.make_inputs(400, 1.0)
for log2n in sizes:
    for cluster in (False, True):
        gpu_se_amd.seed(2)
        _, _, _, pf = sim_base.get_parts(N_particles=1 << log2n)
        pf.cluster = cluster
        for i in range(300):
            u, z = io[i]
            pf.predict(u, 1.0); pf.update(u, z); pf.resample()
        torch.cuda.synchronize()
