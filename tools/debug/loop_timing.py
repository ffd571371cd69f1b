import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from gpu_se_amd import sim_base
for pf, log2n in ((True, 18), (False, 6)):
    sim = sim_base.BatchedSimulation(batch=32, N_particles=1 << log2n, dt_control=0.1, dt_predict=0.1, end_time=50, seed=1, resident=True, pf=pf)
    sim.K._device_law()
    warm = sim_base.BatchedSimulation(batch=32, N_particles=(1 << 10) if pf else (1 << log2n), dt_control=0.1, dt_predict=0.1, end_time=2, seed=2, resident=True, pf=pf)
    warm.K = sim.K; warm.simulate(); torch.cuda.synchronize()
    t = time.perf_counter(); sim.simulate(); torch.cuda.synchronize(); wall = time.perf_counter() - t
    print("pf" if pf else "gsf", "wall %.4f" % wall, {k: round(float(v), 4) for k, v in sim.timing.items()})
