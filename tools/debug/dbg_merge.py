import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
import gpu_se_amd, bench
from gpu_se_amd import sim_base
from oracle import gse_oracle as orc
n = 1 << 20
gpu_se_amd.seed(1234)
_, _, _, f = sim_base.get_parts(N_particles=n)
io = bench.make_inputs(60, 1.0)
_, omp = orc.bioreactor_noise()
for i in range(10):
    u, z = io[i]
    f.predict(u, 1.0); f.update(u, z)
    w, k = f._scaled_weights()
    x = f.particles.get()
    y = np.stack([x[:, 0] * 180.0, x[:, 2] * 116.0], axis=1).astype(np.float64)
    lik = omp.pdf(np.asarray(z)[None, :] - y)
    with np.errstate(divide="ignore"):
        l2 = np.log2(lik)
    samp = (np.arange(64) * n) // 64
    print(i, "k", k, "stored max", w.max(), "inf", int(np.isinf(w).sum()), "nan", int(np.isnan(w).sum()),
          "log2 lik: max %.1f median %.1f min %.1f; samples max %.1f" % (l2.max(), np.median(l2), l2[np.isfinite(l2)].min(), l2[samp].max()),
          "y0 range %.1f %.1f z0 %.1f" % (y[:, 0].min(), y[:, 0].max(), z[0]), flush=True)
    f.resample()
