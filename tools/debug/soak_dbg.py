import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import gpu_se_amd
from gpu_se_amd import sim_base
from gpu_se_amd.filter import ParallelParticleFilter
from gpu_se_amd.model import Bioreactor
from oracle import gse_oracle as orc
xss = orc.steady_state()
rng = np.random.default_rng(1)
n = 5003
fs = []
for cluster in (True, False):
    gpu_se_amd.seed(61)
    state_pdf, measurement_pdf = sim_base.get_noise()
    x0, _ = sim_base.get_noise()
    x0.means += xss[None, :].astype(np.float32)
    f = ParallelParticleFilter(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, n, x0, state_pdf, measurement_pdf)
    f.cluster = cluster
    fs.append(f)
for c in range(16):
    u = [float(rng.uniform(0.03, 0.09)), float(rng.uniform(0.1, 0.3))]
    jump = rng.random() < 0.02
    z = np.array(orc.bioreactor_g(xss)) + rng.normal(0, 0.3, 2) + ([float(rng.normal(0, 300)), 0.0] if jump else [0.0, 0.0])
    r = float(rng.random())
    dt = 1.0 if rng.random() < 0.2 else 0.1
    out = []
    mode = os.environ.get("MODE", "lazy")
    for f in fs:
        f.predict(u, dt); f.update(u, z)
        if mode == "lazy":
            if c == 13:
                w, k = f._scaled_weights()
                print("   c13 stored max", w.max(), "k", k, "nonzero", int((w > 0).sum()), "sum", float(w.astype(np.float64).sum()))
            f.resample(r=r)
            if c >= 13:
                f.record_ancestors = False
                f._want_index = True
                f._flush()
                idx = f._last_index[0, :n].cpu().numpy()
            else:
                idx = np.zeros(1)
            if c != 13:
                w, k = np.zeros((1, 1), np.float32), np.zeros(1)
        else:
            w, k = f._scaled_weights()
            idx = f.resample(r=r, return_index=True).get()
        out.append((w[0].copy(), k.copy(), idx.copy(), f._x.clone()))
    wa, wb = out[0][0].astype(np.float64) * 2.0 ** out[0][1][0], out[1][0].astype(np.float64) * 2.0 ** out[1][1][0]
    d = np.nonzero(out[0][2] != out[1][2])[0]
    print(c, "jump" if jump else "", "k", out[0][1], out[1][1], "true w equal", np.array_equal(wa, wb), "sum", wa.sum(), "idx diff", len(d), d[:5], out[0][2][d[:5]], out[1][2][d[:5]], "x equal", torch.equal(out[0][3], out[1][3]), flush=True)
