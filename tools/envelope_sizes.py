"""ITSE of the resident particle-filter loop against the ensemble size (32 rollouts each), beside the reference's N = 1024."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpu_se_amd
from gpu_se_amd import sim_base
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "loop_envelope.npz"))
print("reference pf1024_c01", np.quantile(g["pf1024_c01_performance"], [0.1, 0.5, 0.9]))
for seed in (1234, 1, 77):
    for log2n in (10, 14, 16, 18):
        gpu_se_amd.seed(seed)
        sim = sim_base.BatchedSimulation(batch=32, N_particles=1 << log2n, dt_control=0.1, dt_predict=0.1, end_time=50, pf=True,
                                         resident=True, seed=1)
        sim.simulate()
        err = sim.ys_f - sim.ys[:, :, [0, 2]]
        print("seed", seed, "N=2^%d" % log2n, "ITSE q10/50/90", np.quantile(sim.performance, [0.1, 0.5, 0.9]).round(0),
              "rms err", np.sqrt(np.mean(err[100:] ** 2, axis=(0, 1))).round(3), flush=True)
