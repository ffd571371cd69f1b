"""Helpers shared by the -m gpu tests."""
import ctypes

import numpy as np
import torch

from gpu_se_amd import _lib
from gpu_se_amd.gaussian_sum_dist import MultivariateGaussianSum


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a), device="cuda")
    return t.to(dtype) if dtype is not None else t


def host(t):
    return t.detach().as_subclass(torch.Tensor).cpu().numpy()


def stream():
    return torch.cuda.current_stream().cuda_stream


def to_state_major(x_rows, ld=None):
    """(N, 5) host rows -> device [5][ld] fp32 buffer (padding filled with NaN to catch any use of it)."""
    x_rows = np.asarray(x_rows, dtype=np.float32)
    n = x_rows.shape[0]
    ld = ld or -(-n // 4) * 4
    buf = np.full((5, ld), np.nan, dtype=np.float32)
    buf[:, :n] = x_rows.T
    return dev(buf), ld


def padded(v, ld, fill=np.nan, dtype=np.float32):
    out = np.full(ld, fill, dtype=dtype)
    out[:len(v)] = v
    return dev(out)


def product_mixtures():
    from gpu_se_amd import sim_base
    return sim_base.get_noise()


def workspace(B, N):
    return torch.zeros(_lib.load().gse_workspace_bytes(B, N), dtype=torch.uint8, device="cuda")


def adversarial_weights(n, rng):
    """Weight patterns chosen to hurt a systematic resample (test_inverse_map_resample_on_adversarial_weights; also what
    tools/tail_profile.py times): exact ties, one heavy particle at tile edges, a bump over the index, a long weightless
    stretch over a floor, sums that are zero / NaN -- and, round 3, contiguous lineages a few hundred times lighter than the
    rest (every output of some workgroup then descends from another tile: the widest work lists the gather stage can meet)."""
    idx_of = np.arange(n)
    light_half = np.where(idx_of < n // 2, 1.0, 1.0 / 512.0) * (0.5 + rng.random(n))
    steps = 10.0 ** -(4.0 * (idx_of * 8 // max(n, 1)) / 7.0) * (0.5 + rng.random(n))
    return {
        "uniform": np.ones(n),
        "one_hot_first": (idx_of == 0).astype(float),
        "one_hot_last": (idx_of == n - 1).astype(float),
        "one_hot_tile_edge": ((idx_of == min(1023, n - 1)) | (idx_of == min(1024, n - 1))).astype(float),
        "zeros_interleaved": rng.random(n) * (rng.random(n) > 0.7),
        "bump": np.exp(-0.5 * ((idx_of - 0.3 * n) / (0.01 * n + 1)) ** 2) + 1e-30,
        "floor_and_spikes": np.where(rng.random(n) < 8.0 / n, 1.0, 1e-7),
        "weightless_stretch": np.where((idx_of > 0.2 * n) & (idx_of < 0.9 * n), 0.0, rng.random(n) ** 8),
        "light_half": light_half,
        "eight_steps_down": steps,
        "zero_sum": np.zeros(n),
        "nan": np.full(n, np.nan),
    }


class ThreadWorld:
    """Shared state of the ranks of a ThreadComm world (one per world)."""

    def __init__(self, world):
        import threading
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.result = None
        self.lock = threading.Lock()


class _Done:
    def wait(self):
        pass


class ThreadComm:
    """The collectives of gpu_se_amd.distributed.TorchComm between THREADS of one process (rank = thread): sums in rank order
    (the same bits on every rank), device or host tensors.  Lets one GPU box, which admits six processes on its card, carry
    the eight shards of BASELINE config 4.  Every rank must call the same collectives in the same order."""
    backend = "thread"

    def __init__(self, shared, rank):
        self.shared, self.rank, self.world = shared, rank, shared.world

    def all_reduce(self, tensor, async_op=False):
        s = self.shared
        s.slots[self.rank] = tensor
        s.barrier.wait()
        if self.rank == 0:
            acc = s.slots[0].clone()
            for p in range(1, self.world):
                acc += s.slots[p].to(acc.device)
            s.result = acc
        s.barrier.wait()
        tensor.copy_(s.result)
        s.barrier.wait()
        return _Done() if async_op else None

    def all_gather(self, parts, tensor):
        s = self.shared
        s.slots[self.rank] = tensor
        s.barrier.wait()
        for p in range(self.world):
            parts[p].copy_(s.slots[p])
        s.barrier.wait()

    def all_to_all_single(self, dst, src, recv, send):
        s = self.shared
        s.slots[self.rank] = (src, list(send))
        s.barrier.wait()
        at = 0
        for p in range(self.world):
            psrc, psend = s.slots[p]
            lo, cnt = sum(psend[:self.rank]), psend[self.rank]
            assert cnt == recv[p], (self.rank, p, cnt, recv[p])
            dst[at:at + cnt].copy_(psrc[lo:lo + cnt])
            at += cnt
        s.barrier.wait()
