"""GS-UKF cycle time (BASELINE config 3: 64 Gaussians; plus a sweep) through the drop-in class."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_se_amd
from gpu_se_amd import sim_base
from gpu_se_amd.model import Bioreactor
xss = Bioreactor.find_SS([0.06, 0.2], [260 / 180, 640 / 24.6, 1000 / 116, 0, 0])
z = np.array(Bioreactor.static_outputs(xss, None))
for n in [int(a) for a in sys.argv[1:]] or [64, 1024, 16384, 262144]:
    gpu_se_amd.seed(3)
    _, _, _, gf = sim_base.get_parts(N_particles=n, pf=False)
    if os.environ.get("GSE_GSF_TWO_LAUNCHES") == "1":
        gf._cycle_lanes = False           # (A/B: ancestors by their own kernel, as until round 4)
    u = [0.06, 0.2]
    def cycle():
        gf.predict(u, 0.1); gf.update(u, z); gf.resample()
    for _ in range(200): cycle()          # (clocks the device up after the set-up: 9 ms at 2^18)
    torch.cuda.synchronize()
    reps = 300
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t = time.perf_counter(); e0.record()
    for _ in range(reps): cycle()
    e1.record(); torch.cuda.synchronize()
    wall = (time.perf_counter() - t) / reps
    dev = e0.elapsed_time(e1) / reps * 1e-3
    est = gf.point_estimate()
    print("GS-UKF N=%7d  wall %.1f us/cycle (%.0f cycles/s)  device %.1f us   760 B/Gaussian -> %.1f GB/s   est Cg=%.4f" %
          (n, wall * 1e6, 1 / wall, dev * 1e6, 760.0 * n / dev / 1e9, est[0]), flush=True)
