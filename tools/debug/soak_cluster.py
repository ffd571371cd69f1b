"""Soak: cluster kernels against tile kernels, and the GS-UKF's one-launch cycle against its two launches, over thousands of
cycles with wandering inputs, measurement jumps and random r -- bit-equality of the ensembles every few cycles."""
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
for n in [int(v) for v in os.environ.get("SIZES", "5003,16384,65600,131072,262144").split(",")]:
    fs = []
    for cluster in (True, False):
        gpu_se_amd.seed(61)
        state_pdf, measurement_pdf = sim_base.get_noise()
        x0, _ = sim_base.get_noise()
        x0.means += xss[None, :].astype(np.float32)
        f = ParallelParticleFilter(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, n, x0, state_pdf, measurement_pdf)
        f.cluster = cluster
        fs.append(f)
    bad = 0
    for c in range(cycles):
        u = [float(rng.uniform(0.03, 0.09)), float(rng.uniform(0.1, 0.3))]
        jump = rng.random() < 0.02
        z = np.array(orc.bioreactor_g(xss)) + rng.normal(0, 0.3, 2) + ([float(rng.normal(0, 300)), 0.0] if jump else [0.0, 0.0])
        r = float(rng.random())
        dt = 1.0 if rng.random() < 0.2 else 0.1
        for f in fs:
            f.predict(u, dt); f.update(u, z); f.resample(r=r)
        if c % 25 == 24 or jump:
            for f in fs:
                f._flush()
            same = torch.equal(fs[0]._x[:, :, :n], fs[1]._x[:, :, :n]) and torch.equal(fs[0]._w[:, :n], fs[1]._w[:, :n])   # (not the padding)
            if not same:
                bad += 1
                dx = (fs[0]._x[:, :, :n] != fs[1]._x[:, :, :n]).nonzero()
                dw = (fs[0]._w[:, :n] != fs[1]._w[:, :n]).nonzero()
                print("N=%d cycle %d: ensembles differ: x at %d places (columns %s), w at %d places (columns %s); ld %d" % (
                    n, c, len(dx), dx[:4, 2].tolist(), len(dw), dw[:4, 1].tolist(), fs[0]._ld), flush=True)
                break
            est = fs[0].point_estimate()
            if not np.all(np.isfinite(est)):
                print("N=%d cycle %d: estimate not finite" % (n, c), est, flush=True); bad += 1; break
    print("PF N=%d: %d cycles, %s" % (n, cycles, "ok" if not bad else "FAILED"), flush=True)
# GS-UKF: one launch against two
for n in (5000, 12 * 1024 + 4, 1 << 16):
    gs = []
    for one in (True, False):
        gpu_se_amd.seed(3)
        _, _, _, gf = sim_base.get_parts(N_particles=n, pf=False)
        gf._cycle_lanes = one
        gs.append(gf)
    gs[1]._seed = gs[0]._seed
    bad = 0
    for c in range(cycles // 4):
        u = [float(rng.uniform(0.03, 0.09)), float(rng.uniform(0.1, 0.3))]
        jump = rng.random() < 0.03
        z = np.array(orc.bioreactor_g(xss)) + rng.normal(0, 0.3, 2) + ([float(rng.normal(0, 40)), 0.0] if jump else [0.0, 0.0])
        r = float(rng.random())
        for gf in gs:
            gf.predict(u, 0.1); gf.update(u, z); gf.resample(r=r)
        if c % 25 == 24 or jump:
            same = np.array_equal(gs[0].means.get(), gs[1].means.get()) and np.array_equal(gs[0].weights.get(), gs[1].weights.get())
            if not same:
                print("GS-UKF N=%d cycle %d: banks differ" % (n, c), flush=True); bad += 1; break
    print("GS-UKF N=%d: %d cycles, %s" % (n, cycles // 4, "ok" if not bad else "FAILED"), flush=True)
