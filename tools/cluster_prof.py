"""Cycles of ONE mid-size filter through the cluster kernels, for a rocprofv3 kernel trace (the host's enqueue rate hides the
kernel time from event timing).  usage: rocprofv3 --kernel-trace --stats -d out -- python3 tools/cluster_prof.py <log2N> [tile]"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import gpu_se_amd
from gpu_se_amd import sim_base
import bench
log2n = int(sys.argv[1])
io = bench.make_inputs(500, 1.0)
gpu_se_amd.seed(2)
_, _, _, pf = sim_base.get_parts(N_particles=1 << log2n)
pf.cluster = len(sys.argv) < 3
for i in range(500):
    u, z = io[i]
    pf.predict(u, 1.0); pf.update(u, z); pf.resample()
torch.cuda.synchronize()
