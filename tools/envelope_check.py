"""The Philox-driven resident closed loops against the reference's own Simulation (tests/golden/loop_envelope.npz, G11):
prints, per configuration, the ITSE distribution and windowed error / output statistics of both.  usage: envelope_check.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpu_se_amd
from gpu_se_amd import sim_base

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "loop_envelope.npz"))
for tag in g["names"]:
    n, dtc, dtp, end, is_pf = g[tag + "_cfg"]
    gpu_se_amd.seed(1234)
    sim = sim_base.BatchedSimulation(batch=32, N_particles=int(n), dt_control=float(dtc), dt_predict=float(dtp), end_time=int(end),
                                     pf=bool(is_pf), resident=True)
    sim.simulate()
    perf_d, perf_r = np.asarray(sim.performance), g[tag + "_performance"]
    err_d = sim.ys_f - sim.ys[:, :, [0, 2]]                      # (ticks, B, 2)
    err_r = np.transpose(g[tag + "_err"], (1, 0, 2))
    ys_d, ys_r = sim.ys[:, :, [0, 2]], np.transpose(g[tag + "_ys"], (1, 0, 2))
    us_d, us_r = sim.us, np.transpose(g[tag + "_us"], (1, 0, 2))
    q = lambda a: np.quantile(a, [0.1, 0.5, 0.9])
    print(tag, "ITSE device", q(perf_d), "reference", q(perf_r), "mpc", sim.mpc_frac)
    for lo in range(0, 500, 100):
        sl = slice(max(lo, 1), lo + 100)
        rms = lambda e: np.sqrt(np.mean(e[sl] ** 2, axis=(0, 1)))
        print("   ticks %3d-%3d  rms err dev %s ref %s | mean y dev %s ref %s (sd %s) | mean u dev %s ref %s" % (
            lo, lo + 100, rms(err_d), rms(err_r), ys_d[sl].mean(axis=(0, 1)), ys_r[sl].mean(axis=(0, 1)),
            ys_r[sl].mean(axis=0).std(axis=0), us_d[sl].mean(axis=(0, 1)), us_r[sl].mean(axis=(0, 1))))
