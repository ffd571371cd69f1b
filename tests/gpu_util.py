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
    return torch.empty(_lib.load().gse_workspace_bytes(B, N), dtype=torch.uint8, device="cuda")
