"""Where the driver-style run (K = 20, W = 5) loses time against K = 200: host enqueue time per cycle, the first launch after
an idle queue, the closing synchronisation.  usage: bracket_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import gpu_se_amd, bench
from gpu_se_amd.distributed import ShardedParticleFilter
n = 1 << 20
dt = 1.0
bench.device_warm_up(torch, n, dt)
for K, W in ((20, 5), (20, 5), (200, 20), (20, 50)):
    gpu_se_amd.seed(12345)
    pf = ShardedParticleFilter.from_get_parts(n_local=n, group=None)
    io = bench.make_inputs(K + W, dt)
    bench.device_warm_up(torch, n, dt, cycles=200)

    def cycle(i):
        u, z = io[i]
        pf.predict(u, dt); pf.update(u, z); pf.resample(exact=False)
    for i in range(W):
        cycle(i)
    pf.wait_collectives(); torch.cuda.synchronize(); torch.cuda.synchronize()
    stamps = [time.perf_counter()]
    for i in range(W, W + K):
        cycle(i)
        stamps.append(time.perf_counter())
    pf.wait_collectives()
    t_enq = time.perf_counter()
    gpu_se_amd.synchronize()
    t_spin = time.perf_counter()
    torch.cuda.synchronize(); torch.cuda.synchronize()
    t_end = time.perf_counter()
    per = np.diff(stamps) * 1e6
    print("K=%d W=%d: total %.1f us = %.2f us/cycle; host enqueue per cycle: first %.1f, median %.1f, max %.1f, sum %.1f; "
          "after the last enqueue: spin wait %.1f, torch syncs %.1f" % (
              K, W, (t_end - stamps[0]) * 1e6, (t_end - stamps[0]) * 1e6 / K, per[0], np.median(per), per.max(), per.sum(),
              (t_spin - t_enq) * 1e6, (t_end - t_spin) * 1e6))
