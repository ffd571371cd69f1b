"""Device time of gse_pf_moments_resampled alone (the estimate of a closed-loop tick) on weights a running filter produces.
usage: momres_bench.py [B] [log2N] [cycles]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_se_amd
from gpu_se_amd import sim_base
from gpu_se_amd.filter import BatchedParticleFilter
from gpu_se_amd.model import Bioreactor
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 18
cycles = int(sys.argv[3]) if len(sys.argv) > 3 else 6
gpu_se_amd.seed(3)
sp, mp = sim_base.get_noise()
x0, _ = sim_base.get_noise()
xss = Bioreactor.find_SS(np.array([0.06, 0.2]), np.array([260 / 180, 640 / 24.6, 1000 / 116, 0, 0]))
x0.means += xss[None, :].astype(np.float32)
f = BatchedParticleFilter(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, 1 << log2n, x0, sp, mp, batch=B)
rng = np.random.default_rng(0)
z0 = np.array(Bioreactor.static_outputs(xss, None))
u = np.repeat([[0.06, 0.2]], B, axis=0)
for c in range(cycles):
    f.predict(u, 0.1)
    f.update(u, z0[None, :] + rng.normal(0, 0.25, size=(B, 2)))
    f.resample()
out = torch.zeros((B, 26), dtype=torch.float64, device="cuda")
assert f._resampled_moments_into(out)
torch.cuda.synchronize()
w = f._w[:, :f.N_particles].double()
ess = (w.sum(1) ** 2 / (w * w).sum(1)).cpu().numpy()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 50
e0.record()
for _ in range(reps):
    f._resampled_moments_into(out)
e1.record()
torch.cuda.synchronize()
print("B=%d N=2^%d: resampled moments %.1f us per call (incl. finish); ESS/N median %.4f" % (B, log2n, e0.elapsed_time(e1) * 1e3 / reps, np.median(ess) / f.N_particles))
if os.environ.get("GSE_MOMRES_STATS"):
    w0 = f._w[0, :f.N_particles].double().cpu().numpy()
    n = len(w0)
    c = np.cumsum(w0) / w0.sum()
    idx = np.minimum(np.searchsorted(c, (np.arange(n) + 0.37) / n, side="left"), n - 1)
    cnt = np.bincount(idx, minlength=n)
    heavy = cnt > 0
    print("particles with offspring: %.4f of N; max count %d" % (heavy.mean(), cnt.max()))
    h = heavy.reshape(-1, 4, 64, 4)                   # tile, quarter, lane, p
    per = h.sum(axis=2)                               # (tile, quarter, p): heavy lanes per ballot
    most = per.max(axis=2)
    print("steps: %d; most == 0: %.3f, 1..16: %.3f, > 16: %.3f; mean heavy per step %.2f; list flushes ~%d, in-place steps %d"
          % (most.size, (most == 0).mean(), ((most > 0) & (most <= 16)).mean(), (most > 16).mean(), per.sum(axis=2).mean(),
             per.sum(axis=2)[most <= 16].sum() // 64, (most > 16).sum()))
