#!/usr/bin/env python3
"""Benchmark of the hot path: particle-filter cycles per second (predict + update + systematic resample) on
the bioreactor model at N = 2^20 particles per MI355X (BASELINE.json config 2), through the drop-in class
API -> ctypes C ABI -> HIP kernels.

  python bench.py --gpus 1 --steps 200 --warmup 20
  python bench.py --gpus N ...          starts its own N rank processes (before anything touches a GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         bench.py --gpus N --steps K --warmup W          (the driver's form: RANK / WORLD_SIZE come from the env)

N > 1: one process per GPU, each rank owns a contiguous shard of ONE big ensemble (weak scaling: 2^20 particles per rank for
EVERY N, so that the driver's 1 / 2 / 4 / 8 series is one workload), independent Philox sub-streams, island resampling; the only
collective is a small asynchronous RCCL all-reduce of the shards' log-weights every 32nd cycle and the posterior merge at
estimates.  Beside `value` the line of an N > 1 run carries `config4_shape` (2^21 particles per rank: BASELINE
config 4 at N = 8) and `config5` (32 closed-loop rollouts x 2^18 particles per rank: BASELINE config 5 at N = 8).  Rank 0 prints
ONE JSON line.

`value` is the whole job's throughput in the metric's unit: particle-cycles per second over all ranks divided by 2^20,
i.e. "cycles per second of 2^20-particle filters" -- at N = 1 with 2^20 particles exactly the cycles per second of
BASELINE config 2; `ensemble_cycles_per_s` and `particles_per_s` are reported beside it.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES = {"predict": 40, "update": 28, "resample": 72}     # SURVEY 8d, per particle
GSUKF_ALG_BYTES = 760                                          # SURVEY 8d, per Gaussian per cycle
HBM_PEAK_GBS = 8000.0
UNIT_PARTICLES = 1 << 20


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--log2-particles", type=float, default=None,
                    help="particles per GPU = 2^this (default 20 for every --gpus N)")
    ap.add_argument("--burn-in", type=int, default=0,
                    help="extra untimed cycles before the W warm-up steps (default 0: `value` is measured on a filter drawn "
                         "from its prior, SURVEY 8d; the running-estimator figure is reported beside it as `steady_state`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary measurements (per-method protocol, GS-UKF, closed loop)")
    ap.add_argument("--exact-resample", action="store_true",
                    help="N > 1: one systematic resample of the whole ensemble per cycle (all-gather of tile sums + "
                         "all-to-all of particles over xGMI) instead of the default island resampling")
    args = ap.parse_args()
    if args.log2_particles is None:
        args.log2_particles = 20.0
    return args


# ---------------------------------------------------------------------------------------------------------------------
# N > 1 without a launcher: this process starts the ranks itself.  It has not imported torch and never touches a GPU;
# the ranks are fresh child processes (no exec of a process that initialised the GPU).
# ---------------------------------------------------------------------------------------------------------------------
def self_launch(args):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0:
                    rc = rc or code
                    for q in pending:           # a rank died: the others would wait at a collective for ever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


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


def device_warm_up(torch, n, dt, cycles=500):
    """`cycles` cycles of a throw-away filter of n particles (nothing of it survives): raises the device's clocks without
    touching the state of the filter that is measured."""
    import gpu_se_amd
    from gpu_se_amd import sim_base
    keep = dict(gpu_se_amd._seed_state)           # (the throw-away filter must not move the streams of what is measured)
    gpu_se_amd.seed(99)
    _, _, _, tmp = sim_base.get_parts(N_particles=n)
    gpu_se_amd._seed_state.update(keep)
    io = make_inputs(64, dt, seed=7)
    for c in range(cycles):
        u, z = io[c % len(io)]
        tmp.predict(u, dt)
        tmp.update(u, z)
        tmp.resample()
    torch.cuda.synchronize()
    del tmp
    return cycles


CYCLE_KERNEL_SOURCES = ("gse_pf_kernels.h", "gse_device.h", "gse_host.h", "gse_merge_device.h", "gse_pf.hip")


def csrc_hash():
    """sha256 over the sources the cycle kernel is compiled from and the ABI header: ties a committed PMC profile to the build
    it was taken on (the other translation units do not touch that kernel)."""
    h = hashlib.sha256()
    for name in sorted(CYCLE_KERNEL_SOURCES):
        with open(os.path.join(ROOT, "gpu_se_amd", "csrc", name), "rb") as fh:
            h.update(name.encode() + b"\0" + fh.read())
    with open(os.path.join(ROOT, "include", "gse.h"), "rb") as fh:
        h.update(b"gse.h\0" + fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(n):
    """HBM bytes per launch of the cycle kernel from the committed PMC profile (rocprofv3 cannot run inside this
    process): profiles/pmc_traffic.json, per size (N = 2^20, 2^21), with the hash of the sources it was taken on."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None, None
    with open(path) as fh:
        d = json.load(fh)
    by_n = d.get("by_n") or ({str(1 << 20): d} if "traffic_bytes_per_launch" in d else {})
    e = by_n.get(str(n))
    if e is None:
        return None, None
    return e.get("traffic_bytes_per_launch"), e.get("csrc_sha256_16")


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


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(n_full, dt=1.0):
    """The loop-faithful oracle (one Python call per particle, like filter/particle.py:54-103) on ONE host core:
    one whole cycle at the benchmark's N (no extrapolation), 100 cycles of BASELINE config 1 (N = 1024), and the
    vectorised-NumPy oracle at the benchmark's N as the stronger CPU line."""
    from oracle import gse_oracle as orc
    rng = np.random.default_rng(0)
    sp, mp = orc.bioreactor_noise(rng)
    x0, _ = orc.bioreactor_noise(rng)
    xss = orc.steady_state()
    x0.means = x0.means + xss[None, :].astype(np.float32)
    u = np.array([0.05, 0.1])
    z = np.array(orc.bioreactor_g(xss)) + [0.1, -0.2]

    def one_cycle(pf):
        t0 = time.perf_counter()
        pf.predict(u, dt)
        t1 = time.perf_counter()
        pf.update(u, z)
        t2 = time.perf_counter()
        pf.resample()
        t3 = time.perf_counter()
        return t1 - t0, t2 - t1, t3 - t2

    full = one_cycle(orc.ParticleFilterOracle(n_full, x0, sp, mp, rng=rng))
    small = orc.ParticleFilterOracle(1024, x0, sp, mp, rng=rng)
    per = np.array([one_cycle(small) for _ in range(100)])
    pv = orc.ParticleFilterOracle(n_full, x0, sp, mp, rng=rng)
    t0 = time.perf_counter()
    for _ in range(3):
        pv.predict_vec(u, dt)
        pv.update_vec(u, z)
        pv.resample(loop=False)
    vec = (time.perf_counter() - t0) / 3
    return full, np.median(per, axis=0), vec


# ---------------------------------------------------------------------------------------------------------------------
# secondary measurements (rank 0 of a one-GPU run): every number DESIGN.md section 5 quotes comes out of this line
# ---------------------------------------------------------------------------------------------------------------------
def protocol_sum(torch, n, io, dt, runs=100):
    """SURVEY 8d / results/pf_openloop/pf_run_seq.py:15-130: predict, update and resample timed one call at a time, each
    followed by a device synchronisation (one kernel per method, filter.lazy = False), medians of `runs`, summed."""
    import gpu_se_amd
    from gpu_se_amd import sim_base
    gpu_se_amd.seed(3)
    _, _, _, f = sim_base.get_parts(N_particles=n)
    f.lazy = False

    def timed(fn):
        gpu_se_amd.synchronize()
        t = time.perf_counter()
        fn()
        gpu_se_amd.synchronize()                  # (polls the stream: gse_stream_synchronize)
        return time.perf_counter() - t
    for u, z in io[:5]:
        f.predict(u, dt); f.update(u, z); f.resample()
    t = {"predict": [], "update": [], "resample": []}
    for i in range(runs):
        u, z = io[(5 + i) % len(io)]
        t["predict"].append(timed(lambda: f.predict(u, dt)))
        t["update"].append(timed(lambda: f.update(u, z)))
        t["resample"].append(timed(lambda: f.resample()))
    med = {k: round(1e6 * float(np.median(v)), 2) for k, v in t.items()}
    f.lazy = True
    cyc = []
    for i in range(runs):
        u, z = io[(5 + i) % len(io)]
        cyc.append(timed(lambda: (f.predict(u, dt), f.update(u, z), f.resample())))
    return {"predict_us": med["predict"], "update_us": med["update"], "resample_us": med["resample"],
            "sum_us": round(sum(med.values()), 2), "fused_cycle_synchronised_us": round(1e6 * float(np.median(cyc)), 2),
            "runs": runs, "note": "each call followed by a device synchronisation, one kernel per method (lazy=False); "
                                  "`value` instead times back-to-back cycles without synchronisation"}


def gsukf_block(torch, sizes=(64, 4096, 1 << 16, 1 << 18)):
    """BASELINE config 3 and the large-bank end of results/gsf_openloop/gsf_run_seq.py:483-484: device time per GS-UKF
    cycle (HIP events), 760 algorithmic bytes per Gaussian."""
    import gpu_se_amd
    from gpu_se_amd import sim_base
    from gpu_se_amd.model import Bioreactor
    xss = Bioreactor.find_SS([0.06, 0.2], [260 / 180, 640 / 24.6, 1000 / 116, 0, 0])
    z = np.array(Bioreactor.static_outputs(xss, None))
    u = [0.06, 0.2]
    out = {}
    for n in sizes:
        gpu_se_amd.seed(3)
        _, _, _, gf = sim_base.get_parts(N_particles=n, pf=False)

        def cycle(_i):
            gf.predict(u, 0.1); gf.update(u, z); gf.resample()
        # (setting a bank up leaves the device idle for a while: 200 untimed cycles -- 9 ms at 2^18 -- clock it up again, as
        #  device_warm_up does for the particle filter's figure; tools/gsf_bench.py has always measured this way)
        for i in range(200 if n > 4096 else 2000):          # (small banks: 2000 x 11 us, or the clocks are still on their way up)
            cycle(i)
        reps = 300
        t0 = time.perf_counter()
        dev_methods = time_method_events(torch, cycle, reps)
        wall_methods = (time.perf_counter() - t0) / reps
        # ... and the same cycles enqueued by ONE call (run_cycles -> gse_gsukf_cycles, round 5: the class's launches from C++)
        us_k, zs_k = np.tile(np.asarray(u, dtype=np.float64), (reps, 1)), np.tile(np.asarray(z, dtype=np.float64), (reps, 1))
        gf.run_cycles(us_k[:50], zs_k[:50], 0.1)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        gf.run_cycles(us_k, zs_k, 0.1)
        e1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps
        dev = e0.elapsed_time(e1) * 1e-3 / reps
        est = gf.point_estimate()
        assert np.all(np.isfinite(est)), est
        entry = {"device_us_per_cycle": round(dev * 1e6, 2), "wall_us_per_cycle": round(wall * 1e6, 2),
                 "wall_over_device": round(wall / dev, 3),
                 "by_three_method_calls": {"device_us_per_cycle": round(dev_methods * 1e6, 2),
                                           "wall_us_per_cycle": round(wall_methods * 1e6, 2)}}
        if n <= 64:
            entry["bound"] = "launch latency (one launch, one workgroup): no roofline fraction"
        elif n <= 4096:
            entry["bound"] = "launch latency (%s per cycle): no roofline fraction" % (
                "one launch" if getattr(gf, "_cycle_lanes", False) else "two launches")
        else:
            gbs = GSUKF_ALG_BYTES * n / dev / 1e9
            entry.update({"achieved_GBps": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
                          "algorithmic_bytes_per_gaussian": GSUKF_ALG_BYTES,
                          "launches_per_cycle": 1 if getattr(gf, "_cycle_lanes", False) else 2})
        out["N=%d" % n] = entry
    return out


def reference_itse(pf, n):
    """Median (and 10 / 90 % quantiles) of the ITSE ``performance`` of the reference's OWN Simulation (control every tick) for the
    recorded ensemble size nearest to n: tests/golden/loop_envelope.npz (G11: 32 seeds; particle filter N = 256, 1024, GS-UKF
    N = 64) and loop_envelope_large.npz (G11b: 16 seeds at N = 16384) -- the reference's NumPy classes run by
    tests/golden/make_golden.py.  The reference's ITSE GROWS with N (4.37e5, 4.53e5, 5.11e5 at 256, 1024, 16384: its estimate
    lags the plant by one tick, sim_base.py:250-262, which a sharper posterior does not forgive); it cannot be run at 2^18."""
    if not pf and n == 64:
        # (G11c + G11: 288 runs of the reference's GS-UKF loop, 64 Gaussians, control every tick; round 5 -- the 32 runs of G11
        #  alone have a median of 1.416e6, a low draw of a distribution whose median is 1.52e6: tests/test_gpu_mpc.py)
        paths = [os.path.join(ROOT, "tests", "golden", nm) for nm in ("loop_envelope_gsf.npz", "loop_envelope.npz")]
        if all(os.path.exists(q) for q in paths):
            perf = np.concatenate([np.load(q)["gsf64_c01_performance"] for q in paths])
            q = np.quantile(perf, [0.1, 0.5, 0.9])
            return {"config": "gsf64_c01 (G11c + G11)", "N": 64, "seeds": int(len(perf)), "median_ITSE": float(q[1]),
                    "q10": float(q[0]), "q90": float(q[2]),
                    "standard_error_of_a_32_rollout_median": round(1.2533 * float(perf.std()) / np.sqrt(32.0), 1)}
    best = None
    for name in ("loop_envelope.npz", "loop_envelope_large.npz"):
        path = os.path.join(ROOT, "tests", "golden", name)
        if not os.path.exists(path):
            continue
        g = np.load(path)
        for tag in g["names"]:
            tag = str(tag)
            cfg = g[tag + "_cfg"]
            if bool(cfg[4]) != bool(pf) or float(cfg[1]) != 0.1:
                continue
            dist = abs(np.log2(cfg[0]) - np.log2(n))
            if best is None or dist < best[0]:
                best = (dist, tag, g[tag + "_performance"], int(cfg[0]))
    if best is None:
        return None
    q = np.quantile(best[2], [0.1, 0.5, 0.9])
    return {"config": best[1], "N": best[3], "seeds": int(len(best[2])), "median_ITSE": float(q[1]), "q10": float(q[0]),
            "q90": float(q[2])}


def run_cycles_block(torch, dt, sizes=(14, 16, 18, 20)):
    """Open-loop cycles enqueued by ONE call (ParallelParticleFilter.run_cycles -> gse_pf_cycles: the loop over predict /
    update / resample restated in C++, the same launches): device microseconds per cycle by HIP events and cycles per second
    of wall time.  From Python the three method calls cost the host 11 us per cycle -- more than the device needs below 2^18
    particles, so only this way does a mid-size filter run at the speed of its kernels (four workgroups per tile there)."""
    import gpu_se_amd
    from gpu_se_amd import sim_base
    out = {}
    for log2n in sizes:
        n = 1 << log2n
        io = make_inputs(700, dt, seed=5)
        us, zs = np.array([u for u, _ in io]), np.array([z for _, z in io])
        gpu_se_amd.seed(6)
        _, _, _, f = sim_base.get_parts(N_particles=n)
        f.run_cycles(us[:100], zs[:100], dt)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        f.run_cycles(us[100:], zs[100:], dt)
        e1.record()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        dev = e0.elapsed_time(e1) * 1e-3 / 600
        assert np.all(np.isfinite(f.point_estimate()))
        out["N=2^%d" % log2n] = {"device_us_per_cycle": round(dev * 1e6, 2), "cycles_per_s_wall": round(600 / wall, 1),
                                 "kernels": "four workgroups per tile (GSE_OP_CLUSTER)" if f.cluster else "one workgroup per tile"}
    return out


def closed_loop_block(torch, B=32, log2n=18, end_time=50, pf=True):
    """BASELINE config 5 per GPU: 32 closed-loop rollouts x 2^18 particles, 499 ticks, MPC every tick, everything
    resident on the device (sim_base.BatchedSimulation(resident=True)).  pf=False: the same loop with banks of 2^log2n
    Gaussians (GS-UKF) as the estimator (results/gsf_closedloop/bioreactor_performance_gsf.py)."""
    import gpu_se_amd
    from gpu_se_amd import sim_base
    # seed-stable (round 5): every filter and mixture draws the next stream key of the global seed, so a block's rollouts used to
    # change whenever a block was added in front of it -- and with them its 32-rollout median ITSE (sd 5e4 between seeds for the
    # GS-UKF loop: tests/test_gpu_mpc.py).  Each block now runs on its own fixed seed and puts the stream state back.
    keep = dict(gpu_se_amd._seed_state)
    gpu_se_amd.seed(0xC105ED00 + 64 * B + log2n + (0 if pf else 1 << 20))
    sim = sim_base.BatchedSimulation(batch=B, N_particles=1 << log2n, dt_control=0.1, dt_predict=0.1, end_time=end_time,
                                     seed=1, resident=True, pf=pf)
    sim.K._device_law()
    warm = sim_base.BatchedSimulation(batch=B, N_particles=(1 << 10) if pf else (1 << log2n), dt_control=0.1, dt_predict=0.1,
                                      end_time=2, seed=2, resident=True, pf=pf)
    warm.K = sim.K
    warm.simulate()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.simulate()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    gpu_se_amd._seed_state.update(keep)
    ticks = len(sim.ts) - 1
    dev_s = float(sim.timing["device_loop"])          # HIP events around the ticks' launches (excludes the host's final statistics)
    return {"rollouts": B, "particles_per_rollout" if pf else "gaussians_per_rollout": 1 << log2n, "ticks": ticks,
            "device_s": round(dev_s, 4),
            "wall_s": round(wall, 4), "host_enqueue_s": round(float(sim.timing["enqueue"]), 4),
            "ticks_per_s_device": round(ticks / dev_s, 1),
            "particle_cycles_per_s_device": round(ticks * B * (1 << log2n) / dev_s, 1),
            "median_ITSE": float(np.median(sim.performance)), "mpc_solved_fraction": round(float(sim.mpc_frac), 4),
            "wall_over_device": round(wall / dev_s, 3),
            "reference_ITSE": reference_itse(pf, 1 << log2n)}


def config5_block(torch, dist, world, rank, rehearsal, B=32, log2n=18, end_time=50, base_seed=1000):
    """BASELINE config 5 (256 stochastic closed-loop rollouts x 2^18 particles on 8 GPUs) at its per-rank shape: every rank runs
    BatchedSimulation(batch=32, N=2^18, resident=True) with seed = base + rank -- independent replicas, no collective until the
    gather of the ITSE scalars -- between a barrier + synchronize on both sides; wall = max over the ranks.  Protocol:
    results/pf_closedloop/bioreactor_performance_pf.py:19-58, :104-118 (Monte-Carlo repetitions of the closed loop)."""
    import gpu_se_amd
    from gpu_se_amd import sim_base
    gpu_se_amd.seed(base_seed + rank)
    sim = sim_base.BatchedSimulation(batch=B, N_particles=1 << log2n, dt_control=0.1, dt_predict=0.1, end_time=end_time,
                                     seed=base_seed + rank, resident=True, pf=True)
    sim.K._device_law()
    warm = sim_base.BatchedSimulation(batch=B, N_particles=1 << 10, dt_control=0.1, dt_predict=0.1, end_time=2,
                                      seed=base_seed + 500 + rank, resident=True, pf=True)
    warm.K = sim.K
    warm.simulate()                                 # (loads every kernel of the loop before the clock starts)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sim.simulate()
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev = "cpu" if rehearsal else "cuda"
    worst = torch.tensor([wall, float(sim.timing["device_loop"])], dtype=torch.float64, device=dev)
    dist.all_reduce(worst, op=dist.ReduceOp.MAX)
    scores = [torch.zeros(B, dtype=torch.float64, device=dev) for _ in range(world)]
    dist.all_gather(scores, torch.as_tensor(np.asarray(sim.performance, dtype=np.float64), device=dev))
    solved = torch.tensor([float(sim.mpc_frac)], dtype=torch.float64, device=dev)
    dist.all_reduce(solved, op=dist.ReduceOp.MIN)
    allp = torch.cat(scores).cpu().numpy()
    ticks = len(sim.ts) - 1
    wall, dev_s = float(worst[0].item()), float(worst[1].item())
    return {"workload": "closed-loop MPC + PF: %d rollouts x 2^%d particles on %d ranks (%d per rank%s), %d ticks, MPC every tick, "
                        "resident on the device" % (world * B, log2n, world, B, ": BASELINE config 5" if world == 8 else "", ticks),
            "rollouts": world * B, "rollouts_per_rank": B, "particles_per_rollout": 1 << log2n, "ticks": ticks,
            "wall_s": round(wall, 4), "device_s_slowest_rank": round(dev_s, 4),
            "rollout_ticks_per_s": round(world * B * ticks / wall, 1),
            "particle_cycles_per_s": round(world * B * ticks * float(1 << log2n) / wall, 1),
            "seeds": "base %d + rank (gpu_se_amd.seed and the simulation's own)" % base_seed,
            "median_ITSE": float(np.median(allp)), "ITSE_q10_q90": [float(q) for q in np.quantile(allp, [0.1, 0.9])],
            "distinct_ITSE_values": int(len(np.unique(allp))),
            "mpc_solved_fraction_min_over_ranks": round(float(solved.item()), 4),
            "collectives": "none in the loops; one all-gather of %d ITSE scalars per rank at the end" % B,
            "reference_ITSE": reference_itse(True, 1 << log2n)}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist
    import gpu_se_amd

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d does not match WORLD_SIZE=%d" % (args.gpus, world))
    # rehearsal on a one-GPU box: GSE_BENCH_REHEARSAL=1 maps every rank to device 0 and uses gloo for the collectives
    rehearsal = os.environ.get("GSE_BENCH_REHEARSAL") == "1"
    backend = "gloo" if rehearsal else "nccl"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get("GSE_FORCE_DIST") == "1"      # the latter: exercise RCCL with one rank
    if use_dist:
        # stdout carries ONE JSON line: keep RCCL's version banner (printed at NCCL_DEBUG=VERSION, which some images set)
        # out of it; GSE_NCCL_DEBUG passes another level through
        os.environ["NCCL_DEBUG"] = os.environ.get("GSE_NCCL_DEBUG", "WARN")
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
    steps, warmup, burn_in = args.steps, args.warmup, max(args.burn_in, 0)
    # the island bookkeeping's all-reduce fires every reduce_every-th resample: at the driver's K = 20 the default of 32 would
    # leave the timed region without a single collective -- at least four of them now, whatever K is
    pf.reduce_every = max(1, min(32, steps // 4))
    dt = 1.0                                                    # pf_run_seq.py:48
    io = make_inputs(burn_in + steps + warmup, dt)

    def bracket(fn, first, count, spf):
        """`count` calls fn(first), fn(first + 1), ... bracketed by barrier + synchronize on both sides; seconds, max over ranks."""
        spf.wait_collectives()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        before = spf.collectives_issued
        t0 = time.perf_counter()
        for i in range(first, first + count):
            fn(i)
        spf.wait_collectives()
        # (the stream is polled until the last cycle has ended, then the bracket's synchronisations find nothing left to wait for:
        #  a blocking wait alone adds its wake-up latency -- 20-40 us, a tenth of the driver's K = 20 cycles -- to the timed region)
        gpu_se_amd.synchronize()
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el, spf.collectives_issued - before

    def cycle(i):
        u, z = io[i]
        pf.predict(u, dt)
        pf.update(u, z)
        pf.resample(exact=args.exact_resample)

    # `value` is measured as SURVEY 8d prescribes: particles ~ prior, W warm-up cycles, K timed cycles.  (Round 2 put 2000
    # untimed cycles in front; the figure of a filter that has been running that long is now `steady_state` in the same line.)
    # The DEVICE is brought up to its clocks first, by a throw-away filter of the same size: setting the workload up (a second of
    # host work: the plant simulation behind the inputs) leaves it idle, and an idle MI355X needs a few milliseconds of work to
    # clock up -- that ramp, not the filter's state, was most of what separated round 2's from-the-prior figure from its
    # steady-state one (tools/cycle_trace.py, which has always warmed the device this way, sees 5 % between them).
    device_warmup = device_warm_up(torch, n, dt)
    for i in range(burn_in):
        cycle(i)
    io = io[burn_in:]
    for i in range(warmup):
        cycle(i)
    if world > 1:
        # (one untimed exchange of the shards' log-weights: the first collective on a buffer sets its channels up -- warm-up,
        #  as the W cycles are for the kernels; the timed region issues its own, counted below)
        pf._exchange_shares()
    elapsed, collectives_timed = bracket(cycle, warmup, steps, pf)
    est = pf.point_estimate()                                   # merges the shards (all-reduce of the moments)
    assert np.all(np.isfinite(est)), est

    # ---- the same cycles, each ENDING IN THE POSTERIOR MERGE (the reference's loop asks for an estimate every tick,
    #      sim_base.py:262, :287-295).  Round 5: nothing is flushed -- the cycle's ONE launch, then ONE more (gse_pf_merge_begin:
    #      the moments of the ensemble the pending resample will produce, its last workgroup finishes them and writes the
    #      shard's row), the all-reduce of [P][22] doubles (asynchronous on RCCL), and with more than one rank gse_merge_end
    #      (combination + estimate); a single shard is combined by the first launch.  The merged moments of every cycle go to
    #      a record row; the covariance sizes of all rows come from one gse_moments_estimates call at the end of the region, as
    #      in the resident closed loops.  Stream-ordered, nothing comes to the host.  Same bracket as above.
    # (its own count of timed cycles, at least 200: at the driver's K = 20 the region's two ends and the one call that turns the
    #  recorded rows into covariance sizes would be a quarter of what is measured -- `fresh_prior` does the same for the kernel)
    est_steps = max(steps, 200)
    io2 = make_inputs(est_steps + 5, dt, seed=11)
    record = torch.zeros((est_steps + 5, 26), dtype=torch.float64, device="cuda")
    sizes = torch.zeros((est_steps + 5, 6), dtype=torch.float64, device="cuda")
    from gpu_se_amd import _lib as _gl

    def cycle_with_estimate(i):
        u, z = io2[i]
        pf.predict(u, dt)
        pf.update(u, z)
        pf.resample()
        pf.merge_begin(into=record[i])
        pf.merge_end()
        if i == est_steps + 4:
            _gl.call("gse_moments_estimates", record.data_ptr(), est_steps + 5, sizes.data_ptr(), torch.cuda.current_stream().cuda_stream)
    for i in range(5):
        cycle_with_estimate(i)
    elapsed_est, collectives_est = bracket(cycle_with_estimate, 5, est_steps, pf)
    est_rows = sizes.cpu().numpy()
    assert np.all(np.isfinite(est_rows)) and np.all(est_rows[:, 5] > 0), "merged estimates are not finite"
    assert np.allclose(est_rows[-1, :5], pf.estimate_device.cpu().numpy()[:5], rtol=1e-12)

    # ---- N > 1 only: the two multi-GPU configurations of BASELINE.json at their per-rank shapes, as blocks of the same line
    config4 = config5 = None
    if world > 1:
        # config 4: 2^24 particles over 8 GPUs = 2^21 per rank (here: 2^21 per rank on however many ranks there are)
        n4 = 1 << 21
        gpu_se_amd.seed(4321)
        pf4 = ShardedParticleFilter.from_get_parts(n_local=n4, group=dist.group.WORLD)
        pf4.reduce_every = pf.reduce_every
        io4 = make_inputs(steps + warmup, dt, seed=21)

        def cycle4(i):
            u, z = io4[i]
            pf4.predict(u, dt)
            pf4.update(u, z)
            pf4.resample()
        for i in range(warmup):
            cycle4(i)
        pf4._exchange_shares()
        el4, coll4 = bracket(cycle4, warmup, steps, pf4)
        est4 = pf4.point_estimate()
        assert np.all(np.isfinite(est4)), est4
        # ... and the same shape with the posterior merge at the end of every cycle (what a loop around config 4 would run)
        rec4 = torch.zeros((est_steps + 5, 26), dtype=torch.float64, device="cuda")
        io4e = make_inputs(est_steps + 5, dt, seed=22)

        def cycle4_est(i):
            u, z = io4e[i]
            pf4.predict(u, dt)
            pf4.update(u, z)
            pf4.resample()
            pf4.merge_begin(into=rec4[i])
            pf4.merge_end()
        for i in range(5):
            cycle4_est(i)
        el4e, coll4e = bracket(cycle4_est, 5, est_steps, pf4)
        assert np.all(np.isfinite(rec4[-1].cpu().numpy()))
        config4 = {"workload": "PF N=2^%d particles sharded over %d ranks (2^21 per rank%s), island resampling, posterior merge at "
                               "the end" % (21 + int(np.log2(world)) if (world & (world - 1)) == 0 else 0, world,
                                            ": BASELINE config 4" if world == 8 else ""),
                   "particles_per_rank": n4, "total_particles": n4 * world, "steps": steps, "warmup": warmup,
                   "ms_per_step": round(el4 / steps * 1e3, 5), "ensemble_cycles_per_s": round(steps / el4, 2),
                   "value": round(float(world) * n4 * steps / el4 / UNIT_PARTICLES, 2), "unit": "cycles/s per 2^20 particles",
                   "collectives_in_timed_region": int(coll4), "exact_resamples": pf4.exact_resamples,
                   "frac_vs_reference_passes": round(140.0 * n4 * steps / el4 / 1e9 / HBM_PEAK_GBS, 4),
                   "with_estimate_every_cycle": {"ms_per_step": round(el4e / est_steps * 1e3, 5), "timed_cycles": est_steps,
                                                 "collectives_in_timed_region": int(coll4e),
                                                 "frac_of_164B_per_particle": round(164.0 * n4 * est_steps / el4e / 1e9 / HBM_PEAK_GBS, 4)}}
        del pf4
        # config 5: 256 closed-loop rollouts x 2^18 particles on 8 GPUs = 32 rollouts per rank, independent replicas (seed =
        # base + rank), no collective until the final gather of the ITSE scalars
        # (results/pf_closedloop/bioreactor_performance_pf.py:19-58, :104-118)
        config5 = config5_block(torch, dist, world, rank, rehearsal)

    # what torch.distributed / RCCL saw (so that a multi-GPU line can be checked against the launcher's own record)
    rccl_info = {"initialised": bool(use_dist), "world_size": world, "backend": None, "devices": [local_rank]}
    if use_dist:
        ids = [None] * world
        dist.all_gather_object(ids, {"rank": rank, "device": torch.cuda.current_device(),
                                     "name": torch.cuda.get_device_name(torch.cuda.current_device()),
                                     "pci": getattr(torch.cuda.get_device_properties(torch.cuda.current_device()), "pci_bus_id", None)})
        probe = torch.ones(1, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(probe)                                   # one collective through the backend itself
        rccl_info = {"initialised": True, "world_size": dist.get_world_size(), "backend": dist.get_backend(),
                     "devices": ids, "all_reduce_of_ones": float(probe.item()),
                     "nccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None}

    result = None
    if rank == 0:
        # ---- device time of the cycle's kernel (HIP events on the launch stream), rank 0 only, local filter.  In steady
        #      state one cycle is ONE launch, pf_step_kernel<gather, predict, update, scan in+out>: resample() and
        #      predict() only record their arguments, update() launches the fused pass.
        # The roofline figure belongs to the same workload as `value`: a filter drawn from its prior, 10 warm-up cycles, then
        # >= 100 timed ones (reference protocol results/pf_openloop/pf_run_seq.py:341-348).  The driver's K may be
        # as small as 20, so this is a run of its own: a fresh local filter (rank 0 only, no collective: the other ranks are at
        # the barrier), HIP events on the launch stream around `reps` cycles.  The same filter then runs on to 2000 cycles and
        # is timed again: `steady_state`.
        from gpu_se_amd import sim_base as _sb
        lp_io = make_inputs(2000 + 2 * 400 + 10, dt, seed=1)
        device_warm_up(torch, n, dt)                # (the host work above left the device idle)
        gpu_se_amd.seed(4321)
        _, _, _, lp = _sb.get_parts(N_particles=n)
        lp_at = [0]

        def local_cycle(_i):
            u, z = lp_io[lp_at[0]]
            lp_at[0] += 1
            lp.predict(u, dt)
            lp.update(u, z)
            lp.resample()
        for i in range(10):
            local_cycle(i)
        reps = max(100, min(steps, 400))
        torch.cuda.synchronize()
        t_w = time.perf_counter()
        per = {"cycle": time_method_events(torch, local_cycle, reps)}
        fresh_wall = (time.perf_counter() - t_w) / reps
        achieved = 140.0 * n / per["cycle"] / 1e9
        while lp_at[0] < 2000:
            local_cycle(0)
        torch.cuda.synchronize()
        t_w = time.perf_counter()
        steady = time_method_events(torch, local_cycle, reps)
        steady_wall = (time.perf_counter() - t_w) / reps
        steady_gbs = 140.0 * n / steady / 1e9
        traffic, traffic_hash = pmc_traffic(n)
        sources = csrc_hash()
        single_launch = not lp._prefix_global
        current = bool(traffic) and traffic_hash == sources
        measured = (traffic / per["cycle"] / 1e9) if traffic else None
        # Round 5: `frac` is what the launch really moves through HBM (PMC bytes of the committed profile of
        # THIS kernel at THIS size) / this run's device time per launch / peak -- a bandwidth, which cannot pass 1.  The figure
        # against the reference's five passes over a particle (SURVEY 8d's 140 bytes), which CAN pass 1 because the fused launch
        # moves 0.38 of those bytes, is `vs_reference_passes`.
        roofline = {"bound": "valu-issue",
                    "roofline_of": "hbm (1.8 flop/B is far below the vector ridge, SURVEY 8d): `peak`; the fused launch is limited "
                                   "by vector-instruction issue (437 wave-instructions per particle, profiles/r04_pmc_sq.txt) and "
                                   "its fixed launch cost, which is why it reaches `frac` of that peak with no wasted traffic "
                                   "(traffic / algorithmic bytes = %s)" % (round(traffic / (140.0 * n), 3) if traffic else "n/a"),
                    "kernel": "pf_step_kernel<gather,predict,update,scan>" if single_launch
                    else "tile_scan_kernel (one workgroup) + pf_step_kernel<gather,predict,update,scan from the scanned tile "
                         "prefix> (two launches per cycle beyond 2^21 particles)",
                    "achieved": None if measured is None else round(measured, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": None if measured is None else round(measured / HBM_PEAK_GBS, 4),
                    "traffic": traffic,
                    "traffic_is_of_these_sources": current,
                    "traffic_profiled_on_csrc": traffic_hash, "csrc_sha256_16": sources,
                    "vs_reference_passes": {"algorithmic_bytes_per_particle": 140, "GBps": round(achieved, 1),
                                            "frac": round(achieved / HBM_PEAK_GBS, 4),
                                            "note": "140 B x N / device time per cycle / 8 TB/s: the reference's passes (predict 40 + "
                                                    "update 28 + resample 72 B) priced at HBM peak against ONE fused launch -- a "
                                                    "speed-up over a pass-per-method design, not a bandwidth; may pass 1"},
                    "device_us": {k: round(v * 1e6, 2) for k, v in per.items()},
                    "measured_on": "a filter drawn from its prior, 10 warm-up cycles, %d timed cycles" % reps,
                    "note": "achieved = HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE x2-corrected + WRITE_SIZE, separate passes: "
                            "profiles/pmc_traffic.json) / average launch-to-launch device time of one cycle (HIP events on the launch "
                            "stream, this run); null when no profile of this size is committed; traffic_is_of_these_sources says "
                            "whether the kernel's sources have changed since the profile was taken"}
        fresh_prior = {"cycles_per_s": round(1.0 / fresh_wall, 1), "device_us_per_cycle": round(per["cycle"] * 1e6, 2),
                       "frac_vs_reference_passes": round(achieved / HBM_PEAK_GBS, 4), "warmup": 10, "timed_cycles": reps,
                       "note": "SURVEY 8d protocol: particles ~ prior, 10 warm-ups, then the timed cycles back to back (one "
                               "synchronisation at the end); `value` above is the same workload at the driver's K and W"}
        steady_state = {"cycles_per_s": round(1.0 / steady_wall, 1), "device_us_per_cycle": round(steady * 1e6, 2),
                        "frac_vs_reference_passes": round(steady_gbs / HBM_PEAK_GBS, 4),
                        "hbm_frac": None if not traffic else round(traffic / steady / 1e9 / HBM_PEAK_GBS, 4),
                        "after_cycles": 2000, "timed_cycles": reps,
                        "note": "the same filter after 2000 cycles of tracking the plant (the posterior is narrow, ancestor "
                                "ranges are 1-2 tiles): what round 2 reported as `value`"}
        particle_cycles = float(world) * n * steps / elapsed
        value = particle_cycles / UNIT_PARTICLES
        result = {
            "metric": "PF steps/sec (predict+update+resample) @ N particles, bioreactor model",
            "value": round(value, 2), "unit": "cycles/s per 2^20 particles (particle-cycles/s / 2^20)",
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64 model/likelihood/cumsum on f32 state",
            "data": "synthetic",
            "config": {"workload": "ParallelParticleFilter HIP, N=2^%g particles per GPU, systematic resample (%s)"
                                   % (args.log2_particles,
                                      "BASELINE config 2" if (world == 1 and n == 1 << 20)
                                      else "weak scaling of BASELINE config 2: the same 2^20 particles per GPU at every N"),
                       "particles_per_gpu": n, "particles_per_rank": [cnt for _, cnt in pf._spans],
                       "total_particles": n * world, "dt": dt, "burn_in_cycles": burn_in, "reduce_every": pf.reduce_every,
                       "device_warmup_cycles": device_warmup,
                       "parallelism": ("particle shards x%d, EXACT global resampling (all-gather of tile sums + all-to-all "
                                       "of particles per cycle), posterior merge at estimates" % world)
                       if (world > 1 and args.exact_resample) else
                       ("particle shards x%d, island resampling, async RCCL all-reduce of the shards' log-weights every "
                        "%d cycles (exact resample when the effective number of islands drops below %.2g x %d: %d in "
                        "this run), posterior merge (all-reduce of 22 doubles per rank) at estimates"
                        % (world, pf.reduce_every, pf.ess_fraction, world, pf.exact_resamples))
                       if world > 1 else "single GPU"},
            "ensemble_cycles_per_s": round(steps / elapsed, 2),
            "particles_per_s": round(particle_cycles, 1),
            "collectives_in_timed_region": int(collectives_timed),
            "with_estimate_every_cycle": {
                "value": round(float(world) * n * est_steps / elapsed_est / UNIT_PARTICLES, 2),
                "unit": "cycles/s per 2^20 particles", "ms_per_step": round(elapsed_est / est_steps * 1e3, 5),
                "timed_cycles": est_steps,
                "collectives_in_timed_region": int(collectives_est),
                "launches_per_cycle": 3 if use_dist else 2,      # (+ the all-reduce's own kernel with a process group)
                "frac_of_164B_per_particle": round(164.0 * n * est_steps / elapsed_est / 1e9 / HBM_PEAK_GBS, 4),
                "note": "every cycle ends in the posterior merge over all ranks, nothing flushed: the cycle's one launch, then "
                        "gse_pf_merge_begin (moments of the ensemble the pending resample will produce; the launch's last workgroup "
                        "finishes them and writes the shard's row), all-reduce of [P][22] doubles (RCCL, async), gse_merge_end "
                        "(combination + mean; a single shard: inside the first launch); merged moments recorded per cycle, the "
                        "covariance sizes of all rows from one gse_moments_estimates call at the end of the timed region; "
                        "stream-ordered, no host copy.  164 B = 140 + the 24 B of a moments pass (SURVEY 8d)"},
            "roofline": roofline,
            "fresh_prior": fresh_prior,
            "steady_state": steady_state,
        }
        result["config"]["rccl"] = rccl_info
        if config4 is not None:
            result["config4_shape"] = config4
        if config5 is not None:
            result["config5"] = config5
        if world == 1 and not args.no_extras:
            result["protocol_sum_us"] = protocol_sum(torch, n, io, dt)
            result["run_cycles"] = run_cycles_block(torch, dt)
            result["gsukf"] = gsukf_block(torch)
            result["closed_loop"] = closed_loop_block(torch)
            # one real-time loop, and a few: the ticks are enqueued from C++ (gse_pf_closed_loop), so the wall time of a small
            # loop is its device time, not the host's
            result["closed_loop_small"] = {"B=%d,N=2^%d" % (b_, l_): closed_loop_block(torch, B=b_, log2n=l_)
                                           for b_, l_ in ((1, 16), (1, 18), (4, 16), (8, 18))}
            result["closed_loop_gsukf"] = closed_loop_block(torch, log2n=6, pf=False)     # 32 rollouts x 64 Gaussians (config 3's bank)
        if world == 1 and not args.no_cpu_baseline:
            full, small, vec = cpu_baseline(n, dt)
            result["cpu_baseline"] = {
                "value": round(1.0 / sum(full), 5), "unit": "cycles/s", "cores": 1, "kind": "port",
                "sample": "1 whole cycle of the loop-faithful oracle (oracle/gse_oracle.py ParticleFilterOracle, one "
                          "Python call per particle as filter/particle.py:54-103) at N=%d, not extrapolated" % n,
                "seconds_on_sample": {k: round(v, 3) for k, v in zip(("predict", "update", "resample"), full)},
                "config1_N1024": {"cycles_per_s": round(1.0 / float(np.sum(small)), 2),
                                  "median_ms": {k: round(1e3 * float(v), 3)
                                                for k, v in zip(("predict", "update", "resample"), small)},
                                  "runs": 100},
                "vectorised_numpy_cycles_per_s": round(1.0 / vec, 4),
                "host": "%s, %d logical cpus" % (cpu_model_name(), os.cpu_count() or 0)}
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return result


if __name__ == "__main__":
    main()
