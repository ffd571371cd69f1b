"""The driver's timed region (K = 20 cycles between synchronisations) several times over: wall, device (events), first-launch gap."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import gpu_se_amd, bench
from gpu_se_amd import sim_base
n = 1 << 20
io = bench.make_inputs(2000, 1.0)
bench.device_warm_up(torch, n, 1.0)
gpu_se_amd.seed(1234)
_, _, _, pf = sim_base.get_parts(N_particles=n)
at = [0]
def cycle():
    u, z = io[at[0] % len(io)]; at[0] += 1
    pf.predict(u, 1.0); pf.update(u, z); pf.resample()
for i in range(5): cycle()
for rep in range(8):
    torch.cuda.synchronize(); torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    t0 = time.perf_counter()
    e0.record()
    cycle()
    e1.record()
    th = time.perf_counter()
    for i in range(19): cycle()
    e2.record()
    tq = time.perf_counter()
    gpu_se_amd.synchronize()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print("region %d: wall %.1f us (%.2f per step)  host: first cycle %.1f, all enqueued after %.1f   device: e0->e1 %.1f, e0->e2 %.1f" % (
        rep, (t1 - t0) * 1e6, (t1 - t0) * 1e6 / 20, (th - t0) * 1e6, (tq - t0) * 1e6, e0.elapsed_time(e1) * 1e3, e0.elapsed_time(e2) * 1e3), flush=True)
