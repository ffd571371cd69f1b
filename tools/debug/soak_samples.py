"""Soak: the weight scale's carried samples (GSE_OP_SAMPLES) against samples fetched from x_in / w_in, tile kernels, thousands of
cycles with jumps: the same particles; the same true weights wherever they are compared (between update and resample)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import gpu_se_amd
from gpu_se_amd import sim_base
from gpu_se_amd.filter import ParallelParticleFilter
from gpu_se_amd.model import Bioreactor
from oracle import gse_oracle as orc
xss = orc.steady_state()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
cycles = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
for n in (1 << 16, 3 << 16, 1 << 20):
    fs = []
    for _ in range(2):
        gpu_se_amd.seed(61)
        state_pdf, measurement_pdf = sim_base.get_noise()
        x0, _ = sim_base.get_noise()
        x0.means += xss[None, :].astype(np.float32)
        f = ParallelParticleFilter(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, n, x0, state_pdf, measurement_pdf)
        f.cluster = False
        fs.append(f)
    carried, fetched = fs
    bad = engaged = 0
    for c in range(cycles):
        u = [float(rng.uniform(0.03, 0.09)), float(rng.uniform(0.1, 0.3))]
        jump = rng.random() < 0.03
        z = np.array(orc.bioreactor_g(xss)) + rng.normal(0, 0.3, 2) + ([float(rng.normal(0, 300)), 0.0] if jump else [0.0, 0.0])
        r = float(rng.random())
        dt = 1.0 if rng.random() < 0.2 else 0.1
        for f in fs:
            f.predict(u, dt)
            if f is fetched:
                f._samples_ok = False
            f.update(u, z)
        if jump or c % 50 == 49:
            ka, kb = carried._wscale.cpu().numpy(), fetched._wscale.cpu().numpy()
            engaged += int(ka[0] < -100)
            wa = np.ldexp(carried._w[0, :n].cpu().numpy().astype(np.float64), int(ka[0]))
            wb = np.ldexp(fetched._w[0, :n].cpu().numpy().astype(np.float64), int(kb[0]))
            heavy = wb > wb.max() * 2.0 ** -100                      # (what both scales can hold)
            if not (np.array_equal(wa[heavy], wb[heavy]) and torch.equal(carried._x[:, :, :n], fetched._x[:, :, :n])) or not np.isfinite(wa.sum()):
                print("N=%d cycle %d: differ (k %s %s, sum %g)" % (n, c, ka, kb, wa.sum()), flush=True); bad += 1; break
        for f in fs:
            f.resample(r=r)
    print("N=%d: %d cycles, scale engaged %d times, %s" % (n, cycles, engaged, "ok" if not bad else "FAILED"), flush=True)
