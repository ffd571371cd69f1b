"""Closed-loop throughput (BASELINE config 5 per GPU: 32 rollouts x 2^18 particles): ticks/s of BatchedSimulation."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_se_amd
from gpu_se_amd import sim_base
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 18
end_time = float(sys.argv[3]) if len(sys.argv) > 3 else 5
gpu_se_amd.seed(1)
sim = sim_base.BatchedSimulation(batch=B, N_particles=1 << log2n, dt_control=0.1, dt_predict=0.1, end_time=end_time, seed=1,
                                 controller=sim_base.hold_controller if os.environ.get("GSE_HOLD") else None)
torch.cuda.synchronize()
t = time.perf_counter()
sim.simulate()
torch.cuda.synchronize()
el = time.perf_counter() - t
ticks = len(sim.ts) - 1
print("closed loop: B=%d rollouts x 2^%d particles, %d ticks in %.3f s -> %.1f ticks/s, %.3g particle-cycles/s, "
      "%.1f ms per tick (estimator cycle + 2 estimates + host plant); median ITSE %.3g"
      % (B, log2n, ticks, el, ticks / el, ticks * B * (1 << log2n) / el, 1e3 * el / ticks, np.median(sim.performance)))
