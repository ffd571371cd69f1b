"""gpu_se_amd -- MI355X-native particle-filter / Gaussian-sum-UKF engine for the bioreactor MPC loop.

Drop-in mirror of the GPU half of AlgorithmicAmoeba/gpu_se's ``filter``, ``gaussian_sum_dist`` and
``sim_base`` modules: Python host code -> ctypes -> libgse_hip.so (hand-written gfx950 HIP kernels);
torch-ROCm tensors are used only as device buffers.
"""
from gpu_se_amd import _lib
from gpu_se_amd._lib import GseError

_seed_state = {"seed": 0x5EED5EED, "next_stream": 0}


def seed(value):
    """Seed every RNG stream created afterwards (Philox keys of the device draws and the host draw of r)."""
    _seed_state["seed"] = int(value) & 0xFFFFFFFFFFFFFFFF
    _seed_state["next_stream"] = 0


def _new_stream_seed():
    """A distinct 64-bit Philox key per consumer, derived from the global seed (SplitMix64 step)."""
    _seed_state["next_stream"] += 1
    z = (_seed_state["seed"] + 0x9E3779B97F4A7C15 * _seed_state["next_stream"]) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def version():
    return _lib.load().gse_version()


def synchronize(spin=True):
    """Wait for the current stream's work (polling it by default: a microsecond after the last kernel ends instead of the
    wake-up latency of torch.cuda.synchronize -- what a caller timing single method calls wants)."""
    from gpu_se_amd.devarray import stream_ptr
    _lib.check(_lib.load().gse_stream_synchronize(stream_ptr(), 1 if spin else 0))


__all__ = ["seed", "version", "synchronize", "GseError"]
