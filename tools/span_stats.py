"""Statistics of the resample structure under the benchmark's inputs: ESS, distinct ancestors and ancestor span per
tile of 1024 outputs.  (Design data for the gather stage.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_se_amd, bench
from gpu_se_amd import sim_base
n = 1 << 20
dt = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
gpu_se_amd.seed(1234)
_, _, _, pf = sim_base.get_parts(N_particles=n)
io = bench.make_inputs(60, dt)
for c, (u, z) in enumerate(io):
    pf.predict(u, dt)
    pf.update(u, z)
    w = pf.weights.get().astype(np.float64)
    ess = w.sum() ** 2 / (w * w).sum()
    idx = pf.resample(return_index=True).get().astype(np.int64)
    if c % 6 == 0:
        t = idx.reshape(-1, 1024)
        span = t[:, -1] - t[:, 0] + 1
        distinct = np.array([len(np.unique(r)) for r in t[::16]])
        w64 = idx.reshape(-1, 64)
        uniform = np.mean(w64[:, 0] == w64[:, -1])
        print("cycle %2d  ESS %9.0f (%.3f%%)  span median %8d  p90 %8d  max %8d   frac span<=2048: %.2f  <=8192: %.2f   distinct/tile median %4d   uniform 64-groups %.2f"
              % (c, ess, 100 * ess / n, np.median(span), np.percentile(span, 90), span.max(), np.mean(span <= 2048), np.mean(span <= 8192), np.median(distinct), uniform))
