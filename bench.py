#!/usr/bin/env python3
"""Benchmark of the hot path: particle-filter cycles per second (predict + update + systematic resample) on
the bioreactor model at N = 2^20 particles per MI355X (BASELINE.json config 2), through the drop-in class
API -> ctypes C ABI -> HIP kernels.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         bench.py --gpus N --steps K --warmup W

N > 1: one process per GPU, each rank owns a contiguous shard of 2^20 particles of one big ensemble (weak
scaling, independent Philox sub-streams, island resampling); the only collective is an RCCL all-reduce of
the global weight sum per cycle, issued asynchronously.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES = {"predict": 40, "update": 28, "resample": 72}     # SURVEY 8d, per particle
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--log2-particles", type=float, default=20.0, help="particles per GPU = 2^this")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exact-resample", action="store_true",
                    help="N > 1: one systematic resample of the whole ensemble per cycle (all-gather of tile sums + "
                         "all-to-all of particles over xGMI) instead of the default island resampling")
    ap.add_argument("--cpu-log2-particles", type=float, default=17.0)
    return ap.parse_args()


def make_inputs(steps, dt, seed=0):
    """Per-cycle inputs: u ~ [U(0,.1), U(0,.2)] as in the reference protocol (sim_base.py:196-199).  The
    measurement is realistic -- a host simulation of the plant under the same u (Euler step + state noise,
    sim_base.py:280-285) observed through g plus measurement noise -- instead of get_random_io's y, which is
    in the wrong units and underflows every likelihood."""
    from gpu_se_amd import sim_base
    from gpu_se_amd.model import Bioreactor
    rng = np.random.default_rng(seed)
    plant = Bioreactor(Bioreactor.find_SS([0.06, 0.2], [260 / 180, 640 / 24.6, 1000 / 116, 0, 0]))
    sd = np.sqrt([1e-4, 1e-7, 1e-3, 1e-3, 1e-7])
    out = []
    for _ in range(steps):
        u, _ = sim_base.get_random_io(rng)
        plant.step(dt, u)
        plant.X += rng.standard_normal(5) * sd * (1.0 if rng.uniform() < 0.75 else np.sqrt(10.0))
        _, z = sim_base.get_realistic_io(rng, plant.X)
        out.append((u, z))
    return out


def pmc_traffic(n):
    """HBM bytes per launch of the cycle kernel from the committed PMC profile (rocprofv3 cannot run inside this
    process): profiles/pmc_traffic.json, valid for N = 2^20; None otherwise."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if n != 1 << 20 or not os.path.exists(path):
        return None
    with open(path) as fh:
        return json.load(fh).get("traffic_bytes_per_launch")


def time_method_events(torch, fn, reps):
    """Average device time of one call of fn, HIP events on the stream the kernels are launched on."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def cpu_baseline(log2n, dt=1.0):
    """The loop-faithful oracle (one Python call per particle, like filter/particle.py:54-103) on one host core,
    one cycle at 2^log2n particles; plus the vectorised oracle at the same size."""
    from oracle import gse_oracle as orc
    n = int(round(2 ** log2n))
    rng = np.random.default_rng(0)
    sp, mp = orc.bioreactor_noise(rng)
    x0, _ = orc.bioreactor_noise(rng)
    xss = orc.steady_state()
    x0.means = x0.means + xss[None, :].astype(np.float32)
    u = np.array([0.05, 0.1])
    z = np.array(orc.bioreactor_g(xss)) + [0.1, -0.2]
    pf = orc.ParticleFilterOracle(n, x0, sp, mp, rng=rng)
    t0 = time.perf_counter()
    pf.predict(u, dt)
    t1 = time.perf_counter()
    pf.update(u, z)
    t2 = time.perf_counter()
    pf.resample()
    t3 = time.perf_counter()
    loop = dict(predict=t1 - t0, update=t2 - t1, resample=t3 - t2)
    pv = orc.ParticleFilterOracle(n, x0, sp, mp, rng=rng)
    t0 = time.perf_counter()
    for _ in range(3):
        pv.predict_vec(u, dt)
        pv.update_vec(u, z)
        pv.resample(loop=False)
    vec = (time.perf_counter() - t0) / 3
    return n, loop, vec


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import gpu_se_amd
    from gpu_se_amd import sim_base

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with torch.distributed.run (one process per GPU)")
    # rehearsal on a one-GPU box: GSE_BENCH_REHEARSAL=1 maps every rank to device 0 and uses gloo for the collectives
    rehearsal = os.environ.get("GSE_BENCH_REHEARSAL") == "1"
    backend = "gloo" if rehearsal else "nccl"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("GSE_FORCE_DIST") == "1"      # the latter: exercise RCCL with one rank
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if rehearsal:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))

    n = int(round(2 ** args.log2_particles))
    gpu_se_amd.seed(1234)
    from gpu_se_amd.distributed import ShardedParticleFilter
    pf = ShardedParticleFilter.from_get_parts(n_local=n, group=dist.group.WORLD if use_dist else None)
    steps, warmup = args.steps, args.warmup
    dt = 1.0                                                    # pf_run_seq.py:48
    io = make_inputs(steps + warmup, dt)

    def cycle(i):
        u, z = io[i]
        pf.predict(u, dt)
        pf.update(u, z)
        pf.resample(exact=args.exact_resample)

    for i in range(warmup):
        cycle(i)
    pf.wait_collectives()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        cycle(i)
    pf.wait_collectives()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    est = pf.point_estimate()                                   # merges the shards (all-reduce of the moments)
    assert np.all(np.isfinite(est)), est

    result = None
    if rank == 0:
        # ---- device time of the cycle's kernel (HIP events on the launch stream), rank 0 only, local filter.  In steady
        #      state one cycle is ONE launch, pf_step_kernel<gather, predict, update, scan in+out>: resample() and
        #      predict() only record their arguments, update() launches the fused pass.
        lp = pf.local

        def local_cycle(i):                 # rank 0 only: the local filter, no collective (other ranks are at the barrier)
            u, z = io[i % len(io)]
            lp.predict(u, dt)
            lp.update(u, z)
            lp.resample()
        reps = max(50, min(steps, 400))
        per = {"cycle": time_method_events(torch, local_cycle, reps)}
        dom = "cycle"
        achieved = 140.0 * n / per[dom] / 1e9
        roofline = {"bound": "hbm", "kernel": "pf_step_kernel<gather,predict,update,scan>",
                    "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(n),
                    "algorithmic_bytes_per_particle": 140,
                    "device_us": {k: round(v * 1e6, 2) for k, v in per.items()},
                    "note": "one launch per cycle; achieved = 140 B x N / average launch-to-launch device time"}
        value = world * steps / elapsed
        result = {
            "metric": "PF steps/sec (predict+update+resample) @ N particles, bioreactor model",
            "value": round(value, 2), "unit": "cycles/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64 model/likelihood/cumsum on f32 state",
            "data": "synthetic",
            "config": {"workload": "ParallelParticleFilter HIP, N=2^%g particles per GPU, systematic resample "
                                   "(BASELINE config 2)" % args.log2_particles,
                       "particles_per_gpu": n, "total_particles": n * world, "dt": dt,
                       "parallelism": ("particle shards x%d, EXACT global resampling (all-gather of tile sums + all-to-all "
                                       "of particles per cycle), posterior merge at estimates" % world)
                       if (world > 1 and args.exact_resample) else
                       ("particle shards x%d, island resampling, async RCCL all-reduce of the weight sum every "
                        "%d cycles, posterior merge (all-reduce of 22 doubles per rank) at estimates"
                        % (world, pf.reduce_every)) if world > 1 else "single GPU"},
            "particles_per_s": round(value * n, 1),
            "roofline": roofline,
        }
        if not args.no_cpu_baseline:
            cn, loop, vec = cpu_baseline(args.cpu_log2_particles, dt)
            scale = n / cn
            result["cpu_baseline"] = {
                "value": round(1.0 / (sum(loop.values()) * scale), 5), "unit": "cycles/s", "cores": 1, "kind": "port",
                "sample": "1 cycle of the loop-faithful oracle (oracle/gse_oracle.py ParticleFilterOracle) at N=2^%g, "
                          "scaled linearly to N=2^%g" % (args.cpu_log2_particles, args.log2_particles),
                "seconds_on_sample": {k: round(v, 3) for k, v in loop.items()},
                "vectorised_numpy_cycles_per_s": round(1.0 / (vec * scale), 4),
                "host": "%d logical cpus" % (os.cpu_count() or 0)}
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
