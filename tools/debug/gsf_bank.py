import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import test_gpu_gsukf as T
from test_gpu_gsukf import *
n = 1 << 18
xss = orc.steady_state()
z = np.array(orc.bioreactor_g(xss)) + [0.2, -0.1]
rng = np.random.default_rng(18)
sd5 = np.sqrt([1e-4, 1e-7, 1e-3, 1e-3, 1e-7])
means = (xss[None, :] + rng.standard_normal((n, 5)) * sd5).astype(np.float32)
sp, mp = orc.bioreactor_noise()
o = orc.GSUKFOracle(n, None, sp, mp, means=means)
gf = make(n, means, deterministic=True)
def dev(gf, o, prior):
    sd = np.sqrt(np.abs(np.einsum("...ii->...i", prior.astype(np.float64))))
    em = np.abs(gf.means.get().astype(np.float64) - o.means) / sd
    ref = o.covariances.astype(np.float64); d = np.sqrt(np.abs(np.einsum("...ii->...i", ref)))
    ec = np.max(np.abs(gf.covariances.get().astype(np.float64) - ref) / (d[..., :, None] * d[..., None, :] + 1e-30))
    w = gf.weights.get().astype(np.float64); ow = o.weights.astype(np.float64)
    ew = np.abs(w - ow) / np.maximum(ow, 1e-300)
    print("mean err (sigma) max", em.max(), "cov rel max", ec, "weights rel max", ew.max(), "p99.9", np.quantile(ew, 0.999), "median", np.median(ew))
u = [0.05, 0.2]
for c in range(3):
    noise = (rng.standard_normal((n, 11, 5)) * sd5).astype(np.float32)
    DeterministicGaussianSum.preload(noise)
    gf.predict(u, 0.1); gf.update(u, z)
    o.predict(u, 0.1, noise=noise, loop=False); prior = o.covariances.copy(); o.update(u, z)
    dev(gf, o, prior)
    r = float(rng.random())
    w_dev = gf.weights.get()
    idx = gf.resample(r=r, return_index=True).get()
    own = orc.systematic_index(w_dev.astype(np.float64), r)
    own_o = orc.systematic_index(o.weights.astype(np.float64), r)
    print("  flips vs device weights", int(np.sum(idx != own)), "vs oracle weights", int(np.sum(idx != own_o)))
    o.resample(r=r, idx=idx.astype(np.int64))
