import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import gpu_se_amd
from gpu_se_amd import sim_base, _lib
from gpu_se_amd.filter import ParallelParticleFilter
from gpu_se_amd.model import Bioreactor
from oracle import gse_oracle as orc
n = 5003
fs = []
for cluster in (True, False):
    gpu_se_amd.seed(61)
    state_pdf, measurement_pdf = sim_base.get_noise()
    x0, _ = sim_base.get_noise()
    x0.means += orc.steady_state()[None, :].astype(np.float32)
    f = ParallelParticleFilter(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, n, x0, state_pdf, measurement_pdf)
    f.cluster = cluster
    seen = []
    def spy(ops, *args, _inner=f._fn_step, _seen=seen):
        _seen.append(int(ops)); return _inner(ops, *args)
    f._fn_step = spy
    f._seen = seen
    fs.append(f)
rng = np.random.default_rng(5)
xss = orc.steady_state()
for c in range(9):
    u = [float(rng.uniform(0.03, 0.09)), float(rng.uniform(0.1, 0.3))]
    z = np.array(orc.bioreactor_g(xss)) + rng.normal(0, 0.3, 2) + ([-500.0, 0.0] if c == 7 else [0.0, 0.0])
    r = float(rng.random())
    out = []
    for f in fs:
        f.predict(u, 0.1 if c % 3 else 1.0)
        f.update(u, z)
        w, k = f._scaled_weights()
        out.append((w.copy(), k.copy(), f._w_uniform))
        f.resample(r=r, return_index=True)
    print(c, "k", out[0][1], out[1][1], "equal w", np.array_equal(out[0][0], out[1][0]), "ops", [hex(o) for o in fs[0]._seen[-3:]], [hex(o) for o in fs[1]._seen[-3:]])
