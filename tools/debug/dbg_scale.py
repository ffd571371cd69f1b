import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import gpu_se_amd
from gpu_se_amd import sim_base
from gpu_se_amd.filter import ParallelParticleFilter
from gpu_se_amd.model import Bioreactor
from oracle import gse_oracle as orc
for cluster in (True, False):
    gpu_se_amd.seed(61)
    sp, mp = sim_base.get_noise(); x0, _ = sim_base.get_noise()
    xss = orc.steady_state(); x0.means += xss[None, :].astype(np.float32)
    f = ParallelParticleFilter(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, 5003, x0, sp, mp)
    f.cluster = cluster
    rng = np.random.default_rng(5)
    for c in range(9):
        u = [float(rng.uniform(0.03, 0.09)), float(rng.uniform(0.1, 0.3))]
        z = np.array(orc.bioreactor_g(xss)) + rng.normal(0, 0.3, 2) + ([250.0, 0.0] if c == 7 else [0.0, 0.0])
        r = float(rng.random())
        f.predict(u, 0.1 if c % 3 else 1.0)
        f.update(u, z)
        w, k = f._scaled_weights()
        y0 = f.particles.get()[:, 0] * 180
        print(cluster, c, "k", k, "wmax", w.max(), "wsum", w.astype(np.float64).sum(), "y0 range", y0.min(), y0.max(), "z0", z[0], flush=True)
        f.resample(r=r)
