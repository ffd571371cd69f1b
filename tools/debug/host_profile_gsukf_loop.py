import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, gpu_se_amd
from gpu_se_amd import sim_base
def make(seed):
    return sim_base.BatchedSimulation(batch=32, N_particles=64, dt_control=0.1, dt_predict=0.1, end_time=50, seed=seed, resident=True, pf=False)
w = make(1); w.K._device_law(); w.simulate(); torch.cuda.synchronize()
s = make(2); s.K = w.K
t = time.perf_counter(); s.simulate(); torch.cuda.synchronize(); print("wall ms", (time.perf_counter() - t) * 1e3, {k: round(float(v) * 1e3, 3) for k, v in s.timing.items()})
s = make(3); s.K = w.K
pr = cProfile.Profile(); pr.enable(); s.simulate(); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
