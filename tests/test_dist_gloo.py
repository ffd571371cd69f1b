"""The multi-rank posterior merge, exercised with world_size 2 on the gloo backend (CPU).  The device part of a
shard (its moments and log-weight) is produced here by the NumPy oracle; what is under test is the host logic of
gpu_se_amd.distributed: shard bounds, the merge row, the all-reduce-as-all-gather, and the combination."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gpu_se_amd import distributed as gd
from oracle import gse_oracle as orc


def test_shard_bounds():
    for n, world in ((10, 3), (1 << 20, 8), (7, 8), (1000, 1)):
        spans = [gd.shard_bounds(n, world, r) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == n
        for (lo, c), (lo2, _) in zip(spans, spans[1:]):
            assert lo + c == lo2
        assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_shard_bounds_on_tile_boundaries():
    """What ShardedParticleFilter splits by: whole tiles per rank, the ragged end on the last -- any ensemble with a tile per
    rank can be resampled exactly (the island policy may call for that in the middle of a run); smaller ones split evenly."""
    for n, world in ((10000, 4), (1 << 20, 8), (8 * 1024 + 1, 8), (4096, 4), (123457, 3)):
        spans = [gd.shard_bounds(n, world, r, 1024) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == n
        for (lo, c), (lo2, _) in zip(spans, spans[1:]):
            assert lo + c == lo2 and c % 1024 == 0 and c > 0
        assert spans[-1][1] > 0
    assert [gd.shard_bounds(3000, 4, r, 1024) for r in range(4)] == [gd.shard_bounds(3000, 4, r) for r in range(4)]
    assert [c for _, c in (gd.shard_bounds(10000, 4, r, 1024) for r in range(4))] == [3072, 3072, 2048, 1808]


def _shard_row(x, w, log_island, common_pivot):
    kp = x[0].astype(np.float64)
    d = x.astype(np.float64) - kp
    w = w.astype(np.float64)
    return gd.pack_merge_row(log_island, w.sum(), w @ d, d.T @ (d * w[:, None]), kp, common_pivot)


def _problem():
    rng = np.random.default_rng(5)
    xss = orc.steady_state()
    x = (xss[None, :] + rng.standard_normal((3000, 5)) * np.sqrt([1e-3, 1e-6, 1e-2, 1e-2, 1e-6])).astype(np.float32)
    w = rng.random(3000) ** 3
    return x, w, xss


def test_merge_equals_single_filter():
    """Two shards with island log-weights log(sum of their weights) and locally NORMALISED weights reproduce the
    posterior mean / covariance of the undivided weighted ensemble."""
    x, w, xss = _problem()
    rows = []
    for lo, hi in ((0, 1200), (1200, 3000)):
        ws = w[lo:hi]
        rows.append(_shard_row(x[lo:hi], ws / ws.sum() * 7.0, np.log(ws.sum()), xss))   # any local scale
    mean, cov = gd.merge_rows(np.array(rows), xss)
    wn = w / w.sum()
    want_mean = wn @ x.astype(np.float64)
    dd = x.astype(np.float64) - want_mean
    np.testing.assert_allclose(mean, want_mean, rtol=1e-12)
    np.testing.assert_allclose(cov, dd.T @ (dd * wn[:, None]), rtol=1e-8, atol=1e-16)
    # a shard with zero weight is ignored
    rows.append(np.zeros(22))
    mean2, _ = gd.merge_rows(np.array(rows), xss)
    np.testing.assert_allclose(mean2, mean, rtol=1e-14)


def test_merge_between_update_and_resample_uses_the_live_weight_sums():
    """Particle i of shard p weighs A_p w_i in the whole ensemble.  Between update() and resample() the local weights
    are NOT normalised and their sums S0_p differ from shard to shard: the merge must weigh shard p by A_p S0_p."""
    x, w, xss = _problem()
    rows = []
    for (lo, hi), local_scale in (((0, 1200), 0.37), ((1200, 3000), 12.5)):
        # w_i = A_p * w_local_i with an arbitrary split of the scale between the island weight and the local weights
        rows.append(_shard_row(x[lo:hi], w[lo:hi] / local_scale, np.log(local_scale), xss))
    assert abs(rows[0][1] - rows[1][1]) > 1.0                         # the S0 differ
    mean, cov = gd.merge_rows(np.array(rows), xss)
    wn = w / w.sum()
    want_mean = wn @ x.astype(np.float64)
    dd = x.astype(np.float64) - want_mean
    np.testing.assert_allclose(mean, want_mean, rtol=1e-12)
    np.testing.assert_allclose(cov, dd.T @ (dd * wn[:, None]), rtol=1e-8, atol=1e-16)


def test_shard_shares_and_island_ess():
    a, live = gd.shard_shares([np.log(3.0), np.log(1.0), -np.inf, 0.0], [1.0, 1.0, 1.0, 0.0])
    np.testing.assert_allclose(a, [0.75, 0.25, 0.0, 0.0], rtol=1e-15)
    assert list(live) == [True, True, False, False]
    assert gd.island_ess(np.full(8, -700.0)) == pytest.approx(8.0)
    assert gd.island_ess([0.0, -800.0, -900.0]) == pytest.approx(1.0)
    assert gd.island_ess([0.0, 0.0, np.log(2.0)]) == pytest.approx(16.0 / 6.0)
    with pytest.raises(FloatingPointError):
        gd.shard_shares([-np.inf, np.nan], [1.0, 1.0])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, w, xss = _problem()
    lo, cnt = gd.shard_bounds(len(w), world, rank)
    ws = w[lo:lo + cnt]
    row = _shard_row(x[lo:lo + cnt], ws / ws.sum(), np.log(ws.sum()), xss)
    # exactly what ShardedParticleFilter._merged does with its row
    buf = torch.zeros((world, gd.MERGE_WIDTH), dtype=torch.float64)
    buf[rank] = torch.as_tensor(row)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    mean, cov = gd.merge_rows(buf.numpy(), xss)
    # and the per-cycle weight-sum all-reduce, asynchronous
    wsum = torch.tensor([ws.sum()], dtype=torch.float64)
    work = dist.all_reduce(wsum, op=dist.ReduceOp.SUM, async_op=True)
    work.wait()
    out[rank] = (mean, cov, float(wsum.item()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_merge_over_gloo():
    world = 2
    manager = mp.Manager()
    out = manager.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    x, w, xss = _problem()
    wn = w / w.sum()
    want = wn @ x.astype(np.float64)
    for rank in range(world):
        mean, cov, wsum = out[rank]
        np.testing.assert_allclose(mean, want, rtol=1e-12)
        np.testing.assert_allclose(wsum, w.sum(), rtol=1e-12)
    np.testing.assert_array_equal(out[0][0], out[1][0])


# ------------------------------------------------------------------------------ exact resampling across ranks
def test_output_boundaries_split_the_outputs_exactly():
    """The outputs that descend from a shard, derived from the cumulative weight at the shard ends alone, against a
    brute-force search over the whole ensemble (incl. weightless shards, r at the ends of [0, 1), tiny ensembles)."""
    rng = np.random.default_rng(1)
    for n in (8, 1000, 4096, 12345):
        for trial in range(40):
            w = rng.random(n) ** 4
            cuts = sorted(rng.choice(np.arange(1, n), size=3, replace=False))
            bounds = [0] + list(cuts) + [n]
            if trial % 5 == 0:
                w[bounds[1]:bounds[2]] = 0.0                       # a shard without weight gets no offspring
            c = np.cumsum(w) / w.sum()
            c[-1] = 1.0
            r = (0.0, 1.0 - 2.0 ** -53, rng.random())[trial % 3]
            anc = np.searchsorted(c, (np.arange(n) + r) / n, side="left")
            J = gd.output_boundaries([c[k - 1] for k in bounds[1:]], n, r)
            assert J[0] == 0 and J[-1] == n
            for q in range(4):
                mine = np.flatnonzero((anc >= bounds[q]) & (anc < bounds[q + 1]))
                assert len(mine) == J[q + 1] - J[q]
                if len(mine):
                    assert mine[0] == J[q] and mine[-1] == J[q + 1] - 1
    send, recv = gd.exact_resample_plan([0, 5, 5, 30, 40], [(0, 10), (10, 10), (20, 10), (30, 10)], 2)
    assert send == [5, 10, 10, 0] and recv == [0, 0, 10, 0]


def _exact_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x, w, _ = _problem_big()
    n_total = len(w)
    spans = [gd.shard_bounds(n_total, world, q) for q in range(world)]
    lo, cnt = spans[rank]
    xs, ws = x[lo:lo + cnt], w[lo:lo + cnt].astype(np.float64)
    # the host logic of ShardedParticleFilter._exact_resample, with NumPy standing in for the four kernels
    shard = object.__new__(gd.ShardedParticleFilter)
    shard.group, shard.world, shard.rank, shard._merge_device = dist.group.WORLD, world, rank, torch.device("cpu")
    shard._comm = gd.TorchComm(dist.group.WORLD)
    tiles = [-(-c // 1024) for _, c in spans]
    mine = np.add.reduceat(ws, np.arange(0, cnt, 1024))
    sums = shard._all_gather_uneven(torch.as_tensor(mine), tiles).numpy()
    prefix = np.concatenate([[0.0], np.cumsum(sums)])
    t0 = sum(tiles[:rank])
    c_local = (prefix[t0] + np.cumsum(ws)) / prefix[-1]
    ends = prefix[np.cumsum(tiles)] / prefix[-1]
    c_local[-1] = ends[rank]
    r = 0.6180339887
    J = gd.output_boundaries(ends, n_total, r)
    send, recv = gd.exact_resample_plan(J, spans, rank)
    j = np.arange(J[rank], J[rank + 1])
    anc = np.searchsorted(c_local, (j + r) / n_total, side="left")
    got = shard._all_to_all_rows(torch.as_tensor(xs[anc]), send, recv).numpy()
    out[rank] = (got, lo, cnt, J)
    dist.barrier()
    dist.destroy_process_group()


def _problem_big():
    rng = np.random.default_rng(9)
    n = 3 * 4096
    x = rng.standard_normal((n, 5)).astype(np.float32)
    w = rng.random(n) ** 6
    w[4096:6000] = 0.0
    return x, w, None


def test_exact_resample_exchange_over_gloo():
    """Three ranks: every rank ends up with exactly its slice of the systematic resample of the whole ensemble."""
    world = 3
    manager = mp.Manager()
    out = manager.dict()
    mp.spawn(_exact_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    x, w, _ = _problem_big()
    n = len(w)
    c = np.cumsum(w.astype(np.float64)) / w.sum()
    anc = np.minimum(np.searchsorted(c, (np.arange(n) + 0.6180339887) / n, side="left"), n - 1)
    moved = 0
    for rank in range(world):
        got, lo, cnt, J = out[rank]
        assert got.shape == (cnt, 5)
        # cumulative sums associate differently here (tile prefix + in-tile sums): allow the neighbour on exact ties
        same = np.all(got == x[anc[lo:lo + cnt]], axis=1)
        assert same.mean() > 0.999
        moved += int(np.sum((anc[lo:lo + cnt] < lo) | (anc[lo:lo + cnt] >= lo + cnt)))
    assert moved > 0                                                # particles did cross ranks


def test_island_policy_reads_the_shares_at_a_fixed_lag():
    """On device groups the exchanged log-weights come back through pinned memory and the policy looks at them
    ``shares_lag`` resamples after the exchange -- a function of the resample COUNT alone, so every rank decides at the same
    resample whatever its side stream has or has not finished (host logic only: the event is a stand-in)."""
    class Event:
        waited = 0

        def synchronize(self):
            Event.waited += 1

    shard = object.__new__(gd.ShardedParticleFilter)
    shard.world, shard.reduce_every, shard.ess_fraction = 4, 8, 0.5
    shard._shares_event, shard._shares_host = Event(), torch.tensor([0.0, -50.0, -50.0, -50.0], dtype=torch.float64)
    shard._force_exact, shard.last_island_ess = False, 4.0
    assert shard.shares_lag == 4
    shard._shares_cycle, shard._resamples = 16, 16
    for count in range(16, 20):                              # nothing is read before the lag has passed
        shard._resamples = count
        shard._consider_shares()
        assert shard._shares_cycle == 16 and Event.waited == 0 and not shard._force_exact
    shard._resamples = 20
    shard._consider_shares()
    assert shard._shares_cycle is None and Event.waited == 1
    assert shard.last_island_ess < 0.5 * 4 and shard._force_exact      # one island carries everything: the next resample is exact
    for every, lag in ((1, 1), (2, 1), (5, 4), (32, 4)):
        shard.reduce_every = every
        assert shard.shares_lag == lag
    shard._shares_event = None                                # host groups (gloo): the old one-cycle lag
    assert shard.shares_lag == 1


# ------------------------------------------------------------------------------ the comm interface (round 5)
def _comm_ops(comm, rank, world):
    """The five collectives of gpu_se_amd.distributed.TorchComm's interface on small host tensors: what a rank ends up with."""
    t = torch.zeros(world, dtype=torch.float64)
    t[rank] = rank + 1.5
    work = comm.all_reduce(t, async_op=True)
    work.wait()
    s = torch.full((3,), float(rank))
    comm.all_reduce(s)
    parts = [torch.empty(2) for _ in range(world)]
    comm.all_gather(parts, torch.tensor([float(rank), 10.0 * rank]))
    send = [(rank + q) % 3 + 1 for q in range(world)]
    recv = [(p + rank) % 3 + 1 for p in range(world)]
    src = torch.arange(sum(send), dtype=torch.float32).reshape(-1, 1).repeat(1, 5) + 100 * rank
    dst = torch.empty((sum(recv), 5), dtype=torch.float32)
    comm.all_to_all_single(dst, src, recv, send)
    return t.tolist(), s.tolist(), [p.tolist() for p in parts], dst[:, 0].tolist(), (comm.rank, comm.world, comm.backend)


def _comm_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out[rank] = _comm_ops(gd.TorchComm(dist.group.WORLD), rank, world)
    dist.barrier()
    dist.destroy_process_group()


def test_comm_interface_over_gloo_and_between_threads():
    """ShardedParticleFilter / ShardedGaussianSumFilter talk to their peers through five methods (gpu_se_amd.distributed.TorchComm
    on a torch.distributed group; round 5).  The -m gpu tests drive EIGHT shards from eight threads of one process through a
    stand-in with the same interface (tests/gpu_util.ThreadComm: a GPU box admits six processes on its card) -- here the two are
    held to each other on the CPU: world size 2 over gloo (processes) and world size 2 and 8 between threads give every rank the
    same sums, the same gathered parts and the same all-to-all pieces."""
    import threading
    from tests.gpu_util import ThreadComm, ThreadWorld
    manager = mp.Manager()
    out = manager.dict()
    mp.spawn(_comm_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    for world in (2, 8):
        shared = ThreadWorld(world)
        got = {}
        threads = [threading.Thread(target=lambda r=r: got.__setitem__(r, _comm_ops(ThreadComm(shared, r), r, world))) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert sorted(got) == list(range(world))
        for r in range(world):
            red, s, parts, a2a, ident = got[r]
            assert red == [p + 1.5 for p in range(world)] and s == [float(sum(range(world)))] * 3
            assert parts == [[float(p), 10.0 * p] for p in range(world)] and ident == (r, world, "thread")
            want = []
            for p in range(world):                                  # rank p sends rows [lo, lo + cnt) of its source to rank r
                sendp = [(p + q) % 3 + 1 for q in range(world)]
                lo = sum(sendp[:r])
                want += [float(100 * p + k) for k in range(lo, lo + sendp[r])]
            assert a2a == want
            if world == 2:
                assert out[r][:4] == got[r][:4] and out[r][4] == (r, 2, "gloo")
