"""The launches the tail is made of, for `rocprofv3 --kernel-trace`: (a) the first 300 cycles of a 2^20-particle filter from
its prior, (b) every adversarial weight pattern of tests/gpu_util.py at 2^20 through the single-launch cycle.  A marker
kernel (gse_model_g on 1 row) separates the two parts in the trace; tools/tail_summary.py turns the trace into the table
under profiles/."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import gpu_se_amd, bench
from gpu_se_amd import sim_base
from gpu_util import adversarial_weights
n = 1 << 20
cycles = int(sys.argv[1]) if len(sys.argv) > 1 else 300
io = bench.make_inputs(cycles, 1.0)
gpu_se_amd.seed(99)
_, _, _, warm = sim_base.get_parts(N_particles=n)
for u, z in io[:100]:
    warm.predict(u, 1.0); warm.update(u, z); warm.resample()
torch.cuda.synchronize()


def marker():
    warm.g_vectorize(np.zeros((1, 5), dtype=np.float32), [0.0, 0.0])
    torch.cuda.synchronize()


marker()
gpu_se_amd.seed(1234)
_, _, _, pf = sim_base.get_parts(N_particles=n)
for u, z in io:
    pf.predict(u, 1.0); pf.update(u, z); pf.resample()
torch.cuda.synchronize()
marker()
xss = np.array([1.55555556, 26.01626016, 5.2711538, 0.0, 15.18857143], dtype=np.float32)
u, z = [0.06, 0.2], np.array([280.0, 611.4538406938659])
rng = np.random.default_rng(n)
for name, w in adversarial_weights(n, rng).items():
    for r in (0.0, 0.37, 1.0 - 2.0 ** -53):
        pf.particles = np.repeat(xss[None, :], n, axis=0)
        pf.weights = (w / np.nansum(w) if np.nansum(w) > 0 else w).astype(np.float32)
        pf.update(u, z)
        pf.resample(r=r)
        pf.predict(u, 0.1)
        pf.update(u, z)                       # the single launch: gather (inverse map) + predict + update
        torch.cuda.synchronize()
        marker()
        print(name, r, flush=True)
