import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import gpu_se_amd
from gpu_se_amd import sim_base
from gpu_se_amd.gaussian_sum_dist import DeterministicGaussianSum
from oracle import gse_oracle as orc
from gpu_util import host
n = 1024
gpu_se_amd.seed(3)
_, _, _, pf = sim_base.get_parts(N_particles=n)
pf.state_pdf = sim_base.get_noise(deterministic=True)[0]
pf.record_ancestors = True
sp, mp = orc.bioreactor_noise(np.random.default_rng(1))
o = orc.ParticleFilterOracle(n, None, sp, mp, particles=host(pf.particles), weights_dtype=np.float32)
rng = np.random.default_rng(5)
xss = orc.steady_state()
mode = sys.argv[1] if len(sys.argv) > 1 else "every10"
for tick in range(40):
    noise = sp.draw((n,))
    u = [0.06, 0.2]
    DeterministicGaussianSum.preload(noise)
    pf.predict(u, 0.1)
    o.predict_vec(u, 0.1, noise=noise)
    if (tick % 10 == 9) if mode == "every10" else True:
        z = np.array(orc.bioreactor_g(xss)) + rng.normal(0, 0.3, 2)
        pf.update(u, z)
        o.update_vec(u, z)
        wd = host(pf._w)[0, :n]
        print(tick, "w max rel", np.max(np.abs(wd - o.weights) / np.maximum(o.weights, 1e-300)), "x equal", np.array_equal(host(pf._x)[0, :, :n].T, o.particles))
        r = float(rng.random())
        pf.resample(r=r)
        own = orc.systematic_index(o.weights, r)
        est = pf.point_estimate()
        idx = host(pf.ancestor_log[-1])[0, :n]
        print("   flips", int(np.sum(idx != own)))
        o.resample(r=r, idx=idx.astype(np.int64))
    e = pf.point_estimate()
    print(tick, "est diff", np.max(np.abs(e - o.point_estimate())), "x equal", np.array_equal(host(pf._x)[0, :, :n].T, o.particles))
