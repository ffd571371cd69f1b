"""Closed-loop throughput (BASELINE config 5 per GPU: 32 rollouts x 2^18 particles): ticks/s of BatchedSimulation.
usage: closedloop_bench.py [B] [log2N] [end_time] [dt_control];  GSE_HOLD=1: hold controller instead of the MPC;
GSE_RESIDENT=1: plant, controller and histories on the device (no host round trip per tick).
Several GPUs (BASELINE config 5: 256 rollouts = 32 per GPU on 8 GPUs): launch with torch.distributed.run, one process per
GPU -- the rollouts are independent replicas (own seeds), the only exchange is the final gather of the ITSE scalars;
GSE_BENCH_REHEARSAL=1 uses gloo and lets the ranks share one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_se_amd
from gpu_se_amd import sim_base
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
log2n = int(sys.argv[2]) if len(sys.argv) > 2 else 18
end_time = float(sys.argv[3]) if len(sys.argv) > 3 else 5
dt_control = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1
world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
rehearsal = bool(os.environ.get("GSE_BENCH_REHEARSAL"))
if world > 1:
    import torch.distributed as dist
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(0 if rehearsal else local)
    dist.init_process_group(backend="gloo" if rehearsal else "nccl", rank=rank, world_size=world)
gpu_se_amd.seed(1 + rank)
resident = bool(os.environ.get("GSE_RESIDENT"))
use_pf = not os.environ.get("GSE_GSF")       # GSE_GSF=1: banks of 2^log2N Gaussians (GS-UKF) as the estimator
sim = sim_base.BatchedSimulation(batch=B, N_particles=1 << log2n, dt_control=dt_control, dt_predict=0.1, end_time=end_time,
                                 seed=1 + rank, resident=resident, pf=use_pf,
                                 controller=sim_base.hold_controller if os.environ.get("GSE_HOLD") else None)
t = time.perf_counter()
if not os.environ.get("GSE_HOLD"):
    sim.K._device_law()                     # condense the QP and upload the law before the clock starts
print("controller set-up %.2f s" % (time.perf_counter() - t))
warm = sim_base.BatchedSimulation(batch=B, N_particles=(1 << 10) if use_pf else (1 << log2n), dt_control=dt_control, dt_predict=0.1,
                                  end_time=2, seed=2, resident=resident, pf=use_pf, controller=sim.controller)
warm.K = sim.K
warm.simulate()                             # loads every kernel of the loop before the clock starts
torch.cuda.synchronize()
if world > 1:
    dist.barrier()
t = time.perf_counter()
sim.simulate()
torch.cuda.synchronize()
el = time.perf_counter() - t
ticks = len(sim.ts) - 1
if world > 1:
    dev = "cpu" if rehearsal else "cuda"
    worst = torch.tensor([el], dtype=torch.float64, device=dev)
    dist.all_reduce(worst, op=dist.ReduceOp.MAX)
    scores = [torch.zeros(B, dtype=torch.float64, device=dev) for _ in range(world)]
    dist.all_gather(scores, torch.as_tensor(sim.performance, dtype=torch.float64, device=dev))
    if rank == 0:
        allp = torch.cat(scores).cpu().numpy()
        print("%d ranks x %d rollouts x 2^%d particles: %d ticks in %.3f s (slowest rank) -> %.1f rollout-ticks/s, "
              "%.3g particle-cycles/s; median ITSE over %d rollouts %.3g"
              % (world, B, log2n, ticks, worst.item(), world * B * ticks / worst.item(),
                 world * B * ticks * (1 << log2n) / worst.item(), len(allp), np.median(allp)))
    dist.barrier()
    dist.destroy_process_group()
    if rank != 0:
        sys.exit(0)
print("closed loop: B=%d rollouts x 2^%d particles, %d ticks in %.3f s -> %.1f ticks/s, %.3g particle-cycles/s, "
      "%.2f ms per tick (%s); median ITSE %.3g"
      % (B, log2n, ticks, el, ticks / el, ticks * B * (1 << log2n) / el, 1e3 * el / ticks,
         "resident loop" if resident else "host plant", np.median(sim.performance)))
if resident:
    print("  resident loop seconds:", {k: round(float(v), 4) for k, v in sim.timing.items()},
          "(device_loop: HIP events around the ticks)")
