"""The multi-rank posterior merge, exercised with world_size 2 on the gloo backend (CPU).  The device part of a
shard (its moments and log-weight) is produced here by the NumPy oracle; what is under test is the host logic of
gpu_se_amd.distributed: shard bounds, the merge row, the all-reduce-as-all-gather, and the combination."""
import os
import socket

import numpy as np
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
