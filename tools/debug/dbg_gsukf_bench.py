import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
from gpu_se_amd import sim_base
bench.device_warm_up(torch, 1 << 20, 1.0, cycles=300)
print(json.dumps(bench.gsukf_block(torch), indent=1))
for _ in range(2):
    r = bench.closed_loop_block(torch, log2n=6, pf=False)
    print({k: r[k] for k in ("device_s", "wall_s", "host_enqueue_s", "wall_over_device", "median_ITSE")})
sim = sim_base.BatchedSimulation(batch=32, N_particles=64, dt_control=0.1, dt_predict=0.1, end_time=50, seed=1, resident=True, pf=False)
sim.simulate()
print({k: round(float(v) * 1e3, 3) for k, v in sim.timing.items()}, "ms")
