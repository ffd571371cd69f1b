"""Closed-loop driver and the batch of filters behind it, -m gpu."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

import gpu_se_amd                                            # noqa: E402
from gpu_se_amd import sim_base                              # noqa: E402
from gpu_se_amd.filter import BatchedParticleFilter, ParallelParticleFilter   # noqa: E402
from gpu_se_amd.model import Bioreactor                      # noqa: E402
from oracle import gse_oracle as orc                         # noqa: E402


def _parts(seed):
    gpu_se_amd.seed(seed)
    sp, mp = sim_base.get_noise()
    x0, _ = sim_base.get_noise()
    x0.means += orc.steady_state()[None, :].astype(np.float32)
    return x0, sp, mp


def test_batched_filter_with_one_filter_equals_the_plain_class():
    n = 3000
    a = ParallelParticleFilter(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, n, *_parts(5))
    b = BatchedParticleFilter(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, n, *_parts(5), batch=1)
    z = np.array(orc.bioreactor_g(orc.steady_state())) + [0.1, -0.2]
    for c in range(4):
        for f in (a, b):
            f.predict([0.05, 0.2], 0.1)
            f.update([0.05, 0.2], z)
            f.resample(r=0.25 + 0.1 * c)
    np.testing.assert_array_equal(a.particles.get(), b.particles.get())
    np.testing.assert_array_equal(a.point_estimate(), b.point_estimate()[0])


def test_batched_filter_tracks_per_filter_measurements():
    """Three filters in one batch, each fed its own measurement: every estimate follows ITS measurement."""
    n, B = 1 << 14, 3
    f = BatchedParticleFilter(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, n, *_parts(9), batch=B)
    assert tuple(f.particles.shape) == (B, n, 5) and tuple(f.weights.shape) == (B, n)
    xss = orc.steady_state()
    shift = np.array([-0.004, 0.0, 0.004])
    z = np.stack([(xss[0] + shift) * 180, np.full(B, xss[2] * 116)], axis=1)
    u = np.repeat([[0.06, 0.2]], B, axis=0)
    for c in range(15):
        f.predict(u, 0.1)
        f.update(u, z)
        f.resample()
    est, cov = f.estimates()
    assert est.shape == (B, 5) and cov.shape == (B,)
    np.testing.assert_allclose(est[:, 0], xss[0] + shift, atol=1.2e-3)
    assert est[0, 0] < est[1, 0] < est[2, 0]


def test_closed_loop_simulation_tracks_the_plant():
    gpu_se_amd.seed(1)
    sim = sim_base.Simulation(N_particles=1 << 14, dt_control=0.1, dt_predict=0.1, end_time=6, seed=3,
                              controller=sim_base.hold_controller)
    sim.simulate()
    steps = len(sim.ts)
    assert sim.xs.shape == (steps, 5) and sim.ys_f.shape == (steps, 2) and sim.us.shape == (steps, 2)
    assert sim.predict_count == steps - 1 and sim.update_count == steps - 1
    # the estimate at tick t has seen the measurement of tick t-1 (then the plant moves on by a step plus state noise
    # of 1.8-5.7 mg/L): compare it with the plant output it was told about
    err = np.abs(sim.ys_f[5:] - sim.ys[4:-1, [0, 2]])
    assert np.median(err[:, 0]) < 0.6 and np.median(err[:, 1]) < 0.8          # mg/L; measurement sigma is 0.25 / 0.28
    assert np.isfinite(sim.performance) and np.all(np.isfinite(sim.covariance_point_size))


def test_batched_closed_loops():
    gpu_se_amd.seed(2)
    sim = sim_base.BatchedSimulation(batch=4, N_particles=1 << 13, dt_control=0.1, dt_predict=0.1, end_time=4, seed=7,
                                     controller=sim_base.hold_controller)
    sim.simulate()
    steps = len(sim.ts)
    assert sim.xs.shape == (steps, 4, 5) and sim.ys_f.shape == (steps, 4, 2)
    assert sim.performance.shape == (4,) and np.all(np.isfinite(sim.performance))
    err = np.abs(sim.ys_f[5:] - sim.ys[4:-1][:, :, [0, 2]])
    assert np.median(err[..., 0]) < 0.6
    assert len(np.unique(np.round(sim.performance, 6))) == 4                      # independent rollouts


def test_closed_loops_with_the_gaussian_sum_filter():
    """pf=False: the reference's closed loop with the GS-UKF as the estimator (sim_base.py:215-219;
    results/gsf_closedloop/bioreactor_performance_gsf.py:20-56), one rollout and a batch of rollouts."""
    gpu_se_amd.seed(4)
    one = sim_base.Simulation(N_particles=64, dt_control=0.1, dt_predict=0.1, end_time=4, pf=False, seed=3,
                              controller=sim_base.hold_controller)
    one.simulate()
    steps = len(one.ts)
    assert one.xs_f.shape[1] == 5 and one.ys_f.shape == (steps, 2)
    err = np.abs(one.ys_f[5:] - one.ys[4:-1, [0, 2]])
    assert np.median(err[:, 0]) < 0.6 and np.median(err[:, 1]) < 0.8
    sim = sim_base.BatchedSimulation(batch=4, N_particles=64, dt_control=0.1, dt_predict=0.1, end_time=4, pf=False, seed=7,
                                     controller=sim_base.hold_controller)
    sim.simulate()
    assert sim.xs.shape == (steps, 4, 5) and sim.ys_f.shape == (steps, 4, 2)
    assert sim.performance.shape == (4,) and np.all(np.isfinite(sim.performance))
    err = np.abs(sim.ys_f[5:] - sim.ys[4:-1][:, :, [0, 2]])
    assert np.median(err[..., 0]) < 0.6 and np.median(err[..., 1]) < 0.8
    assert len(np.unique(np.round(sim.performance, 6))) == 4                      # independent rollouts


def test_resident_loop_with_large_banks_of_gaussians():
    """B x N >= 2^16 Gaussians: predict + update of a tick are ONE launch reading u and z from device memory
    (gse_gsukf_predict_update_dev); the loop stays on the device and the estimate follows the plant."""
    gpu_se_amd.seed(8)
    sim = sim_base.BatchedSimulation(batch=2, N_particles=1 << 15, dt_control=0.1, dt_predict=0.1, end_time=3, pf=False, seed=5,
                                     resident=True)
    assert sim.f._fuse_pu
    sim.simulate()
    err = np.abs(sim.ys_f[5:] - sim.ys[4:-1][:, :, [0, 2]])
    assert np.median(err[..., 0]) < 0.6 and np.median(err[..., 1]) < 0.8
    assert np.all(np.isfinite(sim.performance)) and sim.mpc_frac > 0.9


@pytest.mark.parametrize("n", [64, 100, 200])
def test_resident_loop_with_the_gaussian_sum_filter(n):
    """pf=False, resident=True: the whole loop on the device with the batch of GS-UKFs as the estimator -- inputs and
    measurements read from device memory (gse_gsukf_cycle_dev for 64 Gaussians; predict + update in one launch with the
    deferred row copy beyond), the moments in the particle filter's layout for the controller kernel.  Against the host
    loop: same statistics."""
    gpu_se_amd.seed(6)
    dev = sim_base.BatchedSimulation(batch=8, N_particles=n, dt_control=0.1, dt_predict=0.1, end_time=5, pf=False, seed=3,
                                     resident=True)
    dev.simulate()
    gpu_se_amd.seed(6)
    host = sim_base.BatchedSimulation(batch=8, N_particles=n, dt_control=0.1, dt_predict=0.1, end_time=5, pf=False, seed=3)
    host.simulate()
    steps = len(dev.ts)
    assert dev.xs.shape == (steps, 8, 5) and dev.ys_f.shape == (steps, 8, 2) and dev.us.shape == (steps, 8, 2)
    assert np.all(np.isfinite(dev.performance)) and dev.mpc_frac > 0.9
    for sim in (dev, host):
        err = np.abs(sim.ys_f[5:] - sim.ys[4:-1][:, :, [0, 2]])
        assert np.median(err[..., 0]) < 0.6 and np.median(err[..., 1]) < 0.8
    # the estimate recorded on the device is the filter's own (host formulas on the moments the kernels left)
    assert np.all(np.isfinite(dev.covariance_point_size)) and np.all(dev.covariance_point_size > 0)
    assert abs(np.median(dev.performance) / np.median(host.performance) - 1) < 0.5


# ------------------------------------------------------------------------------ sharded filter across two ranks
def _sharded_worker(rank, world, port, out, n_local=1 << 15):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)     # both ranks share the one GPU of the test box
    torch.cuda.set_device(0)
    import gpu_se_amd as g
    from gpu_se_amd.distributed import ShardedParticleFilter
    g.seed(77)
    spf = ShardedParticleFilter.from_get_parts(n_local=n_local, group=dist.group.WORLD)
    spf.reduce_every = 2
    xss = orc.steady_state()
    z = np.array(orc.bioreactor_g(xss)) + [0.3, -0.2]
    for c in range(6):
        spf.predict([0.06, 0.2], 0.1)
        spf.update([0.06, 0.2], z)
        spf.resample()
    spf.wait_collectives()
    # the posterior merge on the device (gse_merge_row -> all-reduce -> gse_merge_rows; nothing but the final 26 doubles comes
    # to the host) against the host merge of rounds 1-3 (rows packed and combined with NumPy): the same formulas; exp / log of
    # the shard shares come from two libraries, and the covariance undoes the pivot shift in another order
    # (round 5) first the merge WITHOUT a flush: a resample is pending on both ranks, the moments are those of the ensembles the
    # resamples will produce, the rows' log-weights what the copies' launches will make of the shards' -- nothing is copied
    from gpu_se_amd.filter.particle import moments_to_mean_cov
    assert spf.local._pend_gather
    spf.merge_begin()
    assert spf.local._pend_gather
    pend = torch.cat([spf.merge_end(), spf.estimate_device]).cpu().numpy()
    _, pend_mean, pend_cov = moments_to_mean_cov(pend[0], pend[1:6], pend[6:21], pend[21:26])
    host_mean, host_cov = spf._merged_host()
    dev_mean, dev_cov = spf._merged()
    np.testing.assert_allclose(dev_mean, host_mean, rtol=1e-14, atol=0)
    np.testing.assert_allclose(dev_cov, host_cov, rtol=1e-9, atol=1e-22)
    np.testing.assert_allclose(pend_mean, host_mean, rtol=1e-13, atol=0)
    np.testing.assert_allclose(pend_cov, host_cov, rtol=1e-8, atol=1e-22)
    np.testing.assert_allclose(pend[26:31], host_mean, rtol=1e-13, atol=0)          # gse_merge_end's estimate
    assert spf.collectives_issued >= 3
    est = spf.point_estimate()
    cov = spf.point_covariance()
    lo = spf.local._element_offset
    out[rank] = (est, float(cov), lo, float(spf.local._island[0, 0].item()))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_filter_two_ranks_share_one_posterior():
    """Two processes (gloo group, both on the test box's GPU), each owning half of the particles with its own Philox
    sub-stream: after the merge both report the SAME posterior, and it sits where a single filter puts it."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    manager = mp.Manager()
    out = manager.dict()
    mp.spawn(_sharded_worker, args=(2, port, out), nprocs=2, join=True)
    (e0, c0, lo0, l0), (e1, c1, lo1, l1) = out[0], out[1]
    np.testing.assert_array_equal(e0, e1)
    assert c0 == c1 and lo0 == 0 and lo1 == 1 << 15 and l0 != l1
    xss = orc.steady_state()
    assert abs(e0[0] * 180 - (xss[0] * 180 + 0.3)) < 0.4          # follows the measurement (sigma 0.245)


def test_sharded_filter_two_ranks_at_config4_shard_size():
    """BASELINE config 4 shards the 2^24-particle ensemble into 2^21 particles per rank: two such shards (the test box
    has one GPU; both ranks share it) run the single-launch cycle and merge to one posterior."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    manager = mp.Manager()
    out = manager.dict()
    mp.spawn(_sharded_worker, args=(2, port, out, 1 << 21), nprocs=2, join=True)
    (e0, c0, lo0, l0), (e1, c1, lo1, l1) = out[0], out[1]
    np.testing.assert_array_equal(e0, e1)
    assert c0 == c1 and lo0 == 0 and lo1 == 1 << 21 and l0 != l1
    xss = orc.steady_state()
    assert abs(e0[0] * 180 - (xss[0] * 180 + 0.3)) < 0.4


# ------------------------------------------------------------------------------ island policy
def _policy_worker(rank, world, port, out):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import gpu_se_amd as g
    from gpu_se_amd import distributed as gd
    g.seed(5)
    spf = gd.ShardedParticleFilter.from_get_parts(n_local=1 << 14, group=dist.group.WORLD)
    spf.reduce_every = 4
    spf.ess_fraction = 0.75                      # two islands: act when their effective number falls below 1.5
    xss = orc.steady_state()
    rng = np.random.default_rng(3)                                   # the same inputs on both ranks
    ess_log, trace = [], []
    for c in range(500):
        u = [rng.uniform(0.03, 0.09), rng.uniform(0.1, 0.3)]
        z = np.array(orc.bioreactor_g(xss)) + rng.normal(0, 0.3, 2)
        if c < 60:
            z = z + [0.5 * rank, 0.0]            # the shards are told different things for a while: their weights drift apart
        spf.predict(u, 0.1)
        spf.update(u, z)
        spf.resample()
        ess_log.append(spf.last_island_ess)
        trace.append(spf.exact_resamples)
    spf.wait_collectives()
    spf.local._flush()
    logs = torch.zeros(world, dtype=torch.float64)
    logs[rank] = float(spf.local._island[0, 0].item())
    dist.all_reduce(logs)
    shares, _ = gd.shard_shares(logs.numpy(), np.ones(world))
    est = spf.point_estimate()
    out[rank] = (shares, spf.exact_resamples, min(ess_log), trace, est)
    dist.barrier()
    dist.destroy_process_group()


def test_island_policy_keeps_the_shards_alive():
    """500 cycles on two shards whose likelihoods are pushed apart for the first 60: the shards' log-weights are
    exchanged every 4th cycle, the effective number of islands falls below 0.75 P, an exact resample of the whole
    ensemble levels them (every rank takes the decision at the same cycle), and at the end no shard is dead weight."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    manager = mp.Manager()
    out = manager.dict()
    mp.spawn(_policy_worker, args=(2, port, out), nprocs=2, join=True)
    (sh0, k0, ess0, tr0, e0), (sh1, k1, ess1, tr1, e1) = out[0], out[1]
    assert k0 == k1 and k0 >= 1                       # the policy fired, on both ranks alike ...
    assert tr0 == tr1                                 # ... at the same cycles
    assert ess0 < 1.5 and ess0 == ess1                # it saw the islands apart
    np.testing.assert_array_equal(sh0, sh1)
    assert sh0.min() >= 1e-3, sh0                     # no shard has become dead weight
    np.testing.assert_array_equal(e0, e1)
    assert np.all(np.isfinite(e0))


# ------------------------------------------------------------------------------ exact resampling across two ranks
def _mixed_worker(rank, world, port, out):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    import gpu_se_amd as g
    from gpu_se_amd.distributed import ShardedParticleFilter
    g.seed(32)
    n_local = 8 * 1024
    spf = ShardedParticleFilter.from_get_parts(n_local=n_local, group=dist.group.WORLD)
    spf.ess_fraction = 0.0                                           # the exact resample below is asked for explicitly
    xss = orc.steady_state()
    z = np.array(orc.bioreactor_g(xss)) + [0.3 + 0.3 * rank, -0.2]   # the ranks see different likelihood mass
    for c in range(2):                                               # island cycles: the shards' A_p drift apart
        spf.predict([0.06, 0.2], 0.1)
        spf.update([0.06, 0.2], z)
        spf.resample(r=0.37)
    spf.predict([0.06, 0.2], 0.1)
    spf.update([0.06, 0.2], z)
    est_before = spf.point_estimate()                                # between update and resample: live S0 differ
    x, w = spf.local.particles.get().copy(), spf.local.weights.get().copy()
    log_a = float(spf.local._island[0, 0].item())
    spf.resample(r=0.61, exact=True)
    out[rank] = (x, w, log_a, spf.local.particles.get().copy(), est_before, float(spf.local._island[0, 0].item()))
    dist.barrier()
    dist.destroy_process_group()


def test_exact_resample_after_island_resamples_uses_the_island_weights():
    """resample() followed by resample(exact=True): particle i of shard p weighs A_p w_i in the whole ensemble, so the
    exact resample must scale the shards' weights by their island weights (fp32, exp(log A_p - max log A)) -- the
    offspring are bit for bit those of ONE filter holding the concatenation with the scaled weights; and the estimate
    taken between update() and resample() is the weighted mean of that same concatenation."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    manager = mp.Manager()
    out = manager.dict()
    mp.spawn(_mixed_worker, args=(2, port, out), nprocs=2, join=True)
    (x_a, w_a, la, new_a, est_a, ia), (x_b, w_b, lb, new_b, est_b, ib) = out[0], out[1]
    assert la != lb                                                   # the island weights did drift apart
    n_local = len(w_a)
    top = max(la, lb)
    scaled = np.concatenate([w_a * np.float32(np.exp(la - top)), w_b * np.float32(np.exp(lb - top))])
    state_pdf, meas_pdf = sim_base.get_noise()
    x0, _ = sim_base.get_noise()
    one = ParallelParticleFilter(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, 2 * n_local, x0, state_pdf, meas_pdf)
    one.particles = np.concatenate([x_a, x_b])
    one.weights = scaled
    wn = scaled.astype(np.float64) / scaled.astype(np.float64).sum()
    np.testing.assert_allclose(est_a, wn @ np.concatenate([x_a, x_b]).astype(np.float64), rtol=1e-6)
    np.testing.assert_array_equal(est_a, est_b)
    one.resample(r=0.61)
    np.testing.assert_array_equal(np.concatenate([new_a, new_b]), one.particles.get())
    assert ia == ib == np.log(0.5)                                    # the shards are level again


def _exact_worker(rank, world, port, out):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)     # collectives on the host; both ranks on one GPU
    torch.cuda.set_device(0)
    import gpu_se_amd as g
    from gpu_se_amd.distributed import ShardedParticleFilter
    g.seed(31)
    n_local = 12 * 1024
    spf = ShardedParticleFilter.from_get_parts(n_local=n_local, group=dist.group.WORLD)
    xss = orc.steady_state()
    z = np.array(orc.bioreactor_g(xss)) + [0.3 + 0.2 * rank, -0.2]   # the ranks see different likelihood mass
    records = []
    for c in range(3):
        spf.predict([0.06, 0.2], 0.1)
        spf.update([0.06, 0.2], z)
        before = (spf.local.particles.get().copy(), spf.local.weights.get().copy())
        spf.resample(r=0.25 + 0.2 * c, exact=True)
        records.append((before, spf.local.particles.get().copy(), spf.local.weights.get().copy(), dict(spf.last_exact)))
    est = spf.point_estimate()
    out[rank] = (records, est)
    dist.barrier()
    dist.destroy_process_group()


def test_exact_resample_two_ranks_equals_one_filter_on_the_concatenation():
    """two shards resampled as ONE ensemble give, bit for bit, the particles a single filter holding
    the concatenation produces with the same r."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    manager = mp.Manager()
    out = manager.dict()
    mp.spawn(_exact_worker, args=(2, port, out), nprocs=2, join=True)
    n_local = 12 * 1024
    state_pdf, meas_pdf = sim_base.get_noise()
    x0, _ = sim_base.get_noise()
    one = ParallelParticleFilter(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, 2 * n_local, x0, state_pdf, meas_pdf)
    crossed = 0
    for c in range(3):
        (x_a, w_a), new_a, wn_a, info_a = out[0][0][c]
        (x_b, w_b), new_b, wn_b, info_b = out[1][0][c]
        assert info_a["J"] == info_b["J"] and info_a["r"] == info_b["r"]
        # island weights enter the concatenation un-normalised exactly as the shards held them
        one.particles = np.concatenate([x_a, x_b])
        one.weights = np.concatenate([w_a, w_b])
        idx = one.resample(r=0.25 + 0.2 * c, return_index=True).get()
        want = one.particles.get()
        np.testing.assert_array_equal(np.concatenate([new_a, new_b]), want)
        np.testing.assert_array_equal(wn_a, np.full(n_local, np.float32(1.0 / n_local)))
        crossed += int(np.sum(idx[:n_local] >= n_local) + np.sum(idx[n_local:] < n_local))
        assert info_a["send"][1] == int(np.sum(idx[n_local:] < n_local))
    assert crossed > 0
    np.testing.assert_array_equal(out[0][1], out[1][1])               # and the merged estimate is shared


# ------------------------------------------------------------------------------ the collectives through RCCL itself
def _rccl_worker(rank, world, port, out):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    import gpu_se_amd as g
    from gpu_se_amd.distributed import ShardedParticleFilter
    g.seed(31)
    n_local = 12 * 1024 + 100                                        # (a ragged last tile)
    spf = ShardedParticleFilter.from_get_parts(n_local=n_local, group=dist.group.WORLD)
    assert spf._merge_device.type == "cuda" and dist.get_backend() == "nccl"
    xss = orc.steady_state()
    z = np.array(orc.bioreactor_g(xss)) + [0.3, -0.2]
    log = []
    for c in range(4):
        spf.predict([0.06, 0.2], 0.1)
        spf.update([0.06, 0.2], z)
        if c == 1:
            spf.resample()                                           # an island resample in between: the log-weights move
            spf._exchange_shares()                                   # the policy's all-reduce, asynchronous on RCCL
            spf.wait_collectives()
            log.append(("shares", spf._shares.cpu().numpy().copy(), float(spf.local._island[0, 0].item())))
            # ... and the policy's look at them: a side stream brought them to pinned memory behind the collective, the policy
            # reads that copy shares_lag resamples later without waiting for the stream the cycles run on
            assert spf._shares_event is not None and spf._shares_cycle is not None and spf.shares_lag == 4
            spf.last_island_ess = -1.0
            for extra in range(spf.shares_lag):
                assert spf._shares_cycle is not None and spf.last_island_ess == -1.0
                spf.predict([0.06, 0.2], 0.1)
                spf.update([0.06, 0.2], z)                           # (the launch that copies counts the resample)
                spf.resample()
                spf._consider_shares()                               # (resample() calls it when there is more than one rank)
            assert spf._shares_cycle is None and spf.last_island_ess == 1.0
            np.testing.assert_array_equal(spf._shares_host.numpy(), log[-1][1])
            continue
        before = (spf.local.particles.get().copy(), spf.local.weights.get().copy())
        spf._exact_resample(0.25 + 0.2 * c)                          # all-gather of tile sums + all-to-all of rows, on RCCL
        log.append(("exact", before, spf.local.particles.get().copy(), spf.local.weights.get().copy(), dict(spf.last_exact)))
    # the posterior merge, on the device from end to end: the shard's row written by gse_merge_row, an ASYNCHRONOUS all-reduce
    # on RCCL, gse_merge_rows -- against the host merge of rounds 1-3 (one shard: its share is exp(0) = 1, so the mean is
    # the host's bit for bit; the covariance undoes the pivot shift in another order)
    # (round 5) ... first without a flush: one more cycle leaves a resample pending, the merge takes the moments of the ensemble
    # it will produce and the copy keeps riding the next launch; then the flushed merge of the same state
    from gpu_se_amd.filter.particle import moments_to_mean_cov
    spf.predict([0.06, 0.2], 0.1)
    spf.update([0.06, 0.2], z)
    spf.resample()
    spf.merge_begin()
    assert spf._merge_work is not None and spf.local._pend_gather
    pend = spf.merge_end().cpu().numpy()
    _, pend_mean, _ = moments_to_mean_cov(pend[0], pend[1:6], pend[6:21], pend[21:26])
    host_mean, host_cov = spf._merged_host()
    np.testing.assert_allclose(pend_mean, host_mean, rtol=1e-13, atol=0)
    spf.merge_begin()
    assert spf._merge_work is not None                               # (issued with async_op=True on the nccl backend)
    moments = spf.merge_end()
    assert moments.is_cuda and tuple(moments.shape) == (26,)
    dev_mean, dev_cov = spf._merged()
    np.testing.assert_array_equal(dev_mean, host_mean)
    np.testing.assert_allclose(dev_cov, host_cov, rtol=1e-9, atol=1e-22)
    est = spf.point_estimate()                                       # the posterior merge: all-reduce on RCCL
    out[rank] = (log, est, spf.local.point_estimate())
    dist.barrier()
    dist.destroy_process_group()


def test_collectives_run_through_rccl_with_one_rank():
    """No second GPU on this box, so RCCL carries ONE rank -- but it is RCCL that carries it: the asynchronous all-reduce of
    the island log-weights, the all-gather + all-to-all of the exact resample and the all-reduce of the posterior merge are
    issued on the nccl backend with device tensors (the 8-GPU form is the driver's to run; here the entry points are loaded,
    called and give the single-filter answers bit for bit)."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    manager = mp.Manager()
    out = manager.dict()
    mp.spawn(_rccl_worker, args=(1, port, out), nprocs=1, join=True)
    log, est, local_est = out[0]
    n = 12 * 1024 + 100
    state_pdf, meas_pdf = sim_base.get_noise()
    x0, _ = sim_base.get_noise()
    one = ParallelParticleFilter(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, n, x0, state_pdf, meas_pdf)
    seen = 0
    for entry in log:
        if entry[0] == "shares":
            np.testing.assert_array_equal(entry[1], [entry[2]])
            continue
        _, (x, w), new_x, new_w, info = entry
        one.particles, one.weights = x, w
        one.resample(r=info["r"])
        np.testing.assert_array_equal(new_x, one.particles.get())
        np.testing.assert_array_equal(new_w, np.full(n, np.float32(1.0 / n)))
        assert info["send"] == [n] and info["recv"] == [n] and info["J"] == [0, n]
        seen += 1
    assert seen == 3
    # (the merged estimate is normalised, the local class's is the raw weights @ particles with n x fp32(1 / n) != 1)
    np.testing.assert_allclose(est, local_est, rtol=1e-7)


def test_config5_size_cycle_properties():
    """BASELINE config 5 per GPU (32 filters x 2^18 particles, 8.4 M particles in one launch): size-independent
    properties of one fused cycle -- ancestors sorted and in range per filter, offspring counts within one of N w,
    offspring equal to their ancestors, weights reset, every filter following ITS measurement."""
    gpu_se_amd.seed(8)
    n, B = 1 << 18, 32
    f = BatchedParticleFilter(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, n, *_parts(9), batch=B)
    xss = orc.steady_state()
    shift = np.linspace(-0.004, 0.004, B)
    z = np.stack([(xss[0] + shift) * 180, np.full(B, xss[2] * 116)], axis=1)
    u = np.repeat([[0.06, 0.2]], B, axis=0)
    for c in range(3):
        f.predict(u, 0.1)
        f.update(u, z)
        f.resample()
    f.predict(u, 0.1)
    f.update(u, z)
    x = f.particles.get().copy()                                   # (B, N, 5) before the resample
    w = f.weights.get().astype(np.float64)
    r = np.linspace(0.05, 0.95, B)
    idx = f.resample(r=r, return_index=True).get()
    xn = f.particles.get()
    for b in range(0, B, 5):
        assert np.all(np.diff(idx[b]) >= 0) and idx[b].min() >= 0 and idx[b].max() < n
        counts = np.bincount(idx[b], minlength=n)
        assert np.all(np.abs(counts - w[b] / w[b].sum() * n) < 1.0 + 1e-6)
        np.testing.assert_array_equal(xn[b], x[b][idx[b]])
    np.testing.assert_array_equal(f.weights.get(), np.full((B, n), np.float32(1.0 / n)))
    est = f.point_estimate()
    assert np.all(np.diff(est[:, 0]) > 0)                          # ordered like the measurements
    np.testing.assert_allclose(est[:, 0], xss[0] + shift, atol=1.5e-3)


@pytest.mark.parametrize("n,dt_control,dt_predict,end_time", [(64, 0.1, 0.1, 6), (64, 1.0, 0.1, 6), (17, 0.1, 0.1, 3), (64, 1.0, 1.0, 6)])
def test_persistent_loop_equals_the_launch_per_piece_loop(n, dt_control, dt_predict, end_time):
    """gse_gsukf_closed_loop -- every tick of B rollouts inside ONE launch, the banks in LDS throughout -- against the resident
    loop that launches the filter cycle, the moments, the controller and the plant one after the other: every record bit for
    bit (same device functions, same Philox elements and steps, same uniform numbers of the resamples), the filters' final
    banks included.  Control every tick and every tenth; predicts between control ticks; a bank that is not a multiple of
    anything."""
    B = 5
    sims = []
    for persistent in (True, False):
        gpu_se_amd.seed(77)
        sim = sim_base.BatchedSimulation(batch=B, N_particles=n, dt_control=dt_control, dt_predict=dt_predict, end_time=end_time,
                                         pf=False, resident=True, seed=3, persistent=persistent)
        assert sim._persistent_capable() == persistent
        sim.simulate()
        sims.append(sim)
    one, many = sims
    for name in ("us", "xs", "ys", "ys_meas", "moments", "xs_f", "ys_f", "covariance_point_size", "performance"):
        np.testing.assert_array_equal(getattr(one, name), getattr(many, name), err_msg=name)
    assert one.mpc_frac == many.mpc_frac == 1.0
    np.testing.assert_array_equal(one.f.means.get(), many.f.means.get())
    np.testing.assert_array_equal(one.f.covariances.get(), many.f.covariances.get())
    np.testing.assert_array_equal(one.f.weights.get(), many.f.weights.get())
    assert one.f._step == many.f._step


@pytest.mark.parametrize("n,dt_control,dt_predict,end_time", [(4096, 0.1, 0.1, 5), (5000, 1.0, 0.1, 6), (1 << 16, 0.1, 0.1, 3)])
def test_fused_loop_tail_equals_the_separate_launches(n, dt_control, dt_predict, end_time):
    """gse_pf_moments_resampled_tail -- the reduction of the deferred moments, the controller call and the plant in ONE launch
    behind the partial sums -- against the resident particle-filter loop that launches gse_pf_moments_resampled,
    gse_loop_control and gse_plant_step one after the other: every record and the filters' particles bit for bit, with control
    every tick and every tenth (plant-only ticks in between), a ragged ensemble, and the injected-noise mode the parity
    tests use."""
    B = 3
    sims = []
    for fused in (True, False):
        gpu_se_amd.seed(78)
        sim = sim_base.BatchedSimulation(batch=B, N_particles=n, dt_control=dt_control, dt_predict=dt_predict, end_time=end_time,
                                         pf=True, resident=True, seed=4, fused_tail=fused)
        sim.simulate()
        sims.append(sim)
    one, many = sims
    for name in ("us", "xs", "ys", "ys_meas", "moments", "xs_f", "ys_f", "covariance_point_size", "performance"):
        np.testing.assert_array_equal(getattr(one, name), getattr(many, name), err_msg=name)
    assert one.mpc_frac == many.mpc_frac
    np.testing.assert_array_equal(one.f.particles.get(), many.f.particles.get())
    np.testing.assert_array_equal(one.f.weights.get(), many.f.weights.get())


@pytest.mark.parametrize("B,n,dt_control,dt_predict,end_time", [
    (1, 1 << 16, 0.1, 0.1, 4), (4, 5003, 1.0, 0.1, 6), (3, 1 << 14, 1.0, 1.0, 6), (8, 1 << 12, 0.1, 0.1, 3)])
def test_native_loop_equals_the_python_loop(B, n, dt_control, dt_predict, end_time):
    """gse_pf_closed_loop -- the ticks of the resident particle-filter loop enqueued from C++ -- against the same loop enqueued
    from Python (native_loop = False): the same launches with the same arguments in the same order, so every record, the
    particles and the weights bit for bit; and the filter object carries on identically afterwards (the loop hands back where
    the ensemble is and what is still pending: a resample whose copy rides the next launch)."""
    sims = []
    for native in (True, False):
        gpu_se_amd.seed(91)
        sim = sim_base.BatchedSimulation(batch=B, N_particles=n, dt_control=dt_control, dt_predict=dt_predict, end_time=end_time,
                                         pf=True, resident=True, seed=4, native_loop=native)
        assert sim._native_capable() == native
        sim.simulate()
        sims.append(sim)
    one, many = sims
    for name in ("us", "xs", "ys", "ys_meas", "moments", "xs_f", "ys_f", "covariance_point_size", "performance"):
        np.testing.assert_array_equal(getattr(one, name), getattr(many, name), err_msg=name)
    assert one.mpc_frac == many.mpc_frac
    # one more cycle through the class API on both (before anything flushes the pending stages)
    u = np.tile([0.05, 0.15], (B, 1))
    z = np.tile([280.0, 611.0], (B, 1))
    for sim in sims:
        sim.f.predict(u, 0.1)
        sim.f.update(u, z)
        sim.f.resample()
    np.testing.assert_array_equal(one.f.particles.get(), many.f.particles.get())
    np.testing.assert_array_equal(one.f.weights.get(), many.f.weights.get())
    np.testing.assert_array_equal(one.f.point_estimate(), many.f.point_estimate())


@pytest.mark.parametrize("B,n,dt_control", [(1, 1 << 16, 0.1), (2, 50000, 1.0)])
def test_resident_loop_with_cluster_launches_equals_the_tile_launches(B, n, dt_control):
    """The resident particle-filter loop of a mid-size ensemble runs four workgroups per tile (GSE_OP_CLUSTER; the deferred
    moments read the quarter-format slot): every record bit for bit those of the same loop on the one-workgroup-per-tile
    kernels, enqueued from C++ and from Python."""
    sims = []
    for cluster, native in ((True, True), (False, True), (True, False)):
        gpu_se_amd.seed(92)
        sim = sim_base.BatchedSimulation(batch=B, N_particles=n, dt_control=dt_control, dt_predict=0.1, end_time=4, pf=True,
                                         resident=True, seed=4, native_loop=native)
        assert sim.f.cluster
        sim.f.cluster = cluster
        sim.simulate()
        sims.append(sim)
    for other in sims[1:]:
        for name in ("us", "xs", "ys", "ys_meas", "moments", "xs_f", "ys_f", "covariance_point_size", "performance"):
            np.testing.assert_array_equal(getattr(sims[0], name), getattr(other, name), err_msg=name)
    np.testing.assert_array_equal(sims[0].f.particles.get(), sims[1].f.particles.get())
    np.testing.assert_array_equal(sims[0].f.particles.get(), sims[2].f.particles.get())
