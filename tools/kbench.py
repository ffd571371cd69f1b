"""Per-kernel timing of the PF hot path at N = 2^20 (run under rocprofv3 --kernel-trace --stats for exact kernel
durations, or stand-alone for event timings).  Usage: python tools/kbench.py [log2N] [reps]"""
import sys
import time

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import gpu_se_amd
from gpu_se_amd import sim_base

log2n = float(sys.argv[1]) if len(sys.argv) > 1 else 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
n = int(round(2 ** log2n))
gpu_se_amd.seed(1)
_, _, _, pf = sim_base.get_parts(N_particles=n)
xss = np.array([1.55555556, 26.01626016, 5.2711538, 0.0, 15.18857143])
u = [0.06, 0.2]
z = np.array([280.2, 611.1])


def timed(name, fn):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t = time.perf_counter()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    host = (time.perf_counter() - t) / reps
    torch.cuda.synchronize()
    print("%-28s device %.2f us   host %.2f us" % (name, e0.elapsed_time(e1) / reps * 1e3, host * 1e6), flush=True)


import bench
io = bench.make_inputs(reps + 60, 1.0)
state = {"i": 0}


def cycle():
    uu, zz = io[state["i"] % len(io)]
    state["i"] += 1
    pf.predict(uu, 1.0)
    pf.update(uu, zz)
    pf.resample()


timed("predict", lambda: pf.predict(u, 0.1))
pf.particles = pf.particles * 0 + torch.as_tensor(xss, dtype=torch.float32, device="cuda")
pf.predict(u, 0.1)
timed("update", lambda: pf.update(u, z))
w = np.random.default_rng(0).random(n)
pf.weights = w / w.sum()
timed("resample(random w, +tile sums)", lambda: pf.resample())
gpu_se_amd.seed(2)
_, _, _, pf = sim_base.get_parts(N_particles=n)
timed("cycle (plant-tracking inputs)", cycle)
