import sys; sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import gpu_se_amd
from gpu_se_amd import sim_base
from gpu_se_amd.filter import ParallelParticleFilter
from gpu_se_amd.model import Bioreactor
from oracle import gse_oracle as orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65553
fs = []
for cluster in (True, False):
    gpu_se_amd.seed(61)
    sp, mp = sim_base.get_noise(); x0, _ = sim_base.get_noise()
    xss = orc.steady_state(); x0.means += xss[None, :].astype(np.float32)
    f = ParallelParticleFilter(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, n, x0, sp, mp)
    f.cluster = cluster
    fs.append(f)
rng = np.random.default_rng(5)
for c in range(6):
    u = [float(rng.uniform(0.03, 0.09)), float(rng.uniform(0.1, 0.3))]
    z = np.array(orc.bioreactor_g(xss)) + rng.normal(0, 0.3, 2)
    r = float(rng.random())
    out = []
    for f in fs:
        f.predict(u, 0.1 if c % 3 else 1.0)
        f.update(u, z)
        w, k = f._scaled_weights()
        x = f.particles.get().copy()
        idx = f.resample(r=r, return_index=True).get()
        out.append((w[0], k, x, idx))
    (wa, ka, xa, ia), (wb, kb, xb, ib) = out
    bad_w = np.nonzero(wa != wb)[0]; bad_x = np.nonzero((xa != xb).any(axis=1))[0]; bad_i = np.nonzero(ia != ib)[0]
    print("cycle", c, "k", ka, kb, "bad w", len(bad_w), bad_w[:5], "bad x", len(bad_x), bad_x[:5], "bad idx", len(bad_i), bad_i[:5], flush=True)
    if len(bad_x):
        j = bad_x[0]; print("   x cluster", xa[j], "tile", xb[j], " quarters hit:", sorted(set((bad_x // 256).tolist()))[:20])
    if len(bad_i):
        j = bad_i[0]; print("   idx cluster", ia[j-2:j+3], "tile", ib[j-2:j+3], "quarters hit:", sorted(set((bad_i // 256).tolist()))[:20], "n", len(set((bad_i // 256).tolist())))
