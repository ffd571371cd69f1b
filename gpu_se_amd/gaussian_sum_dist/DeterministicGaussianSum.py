"""Noise that replays: the test double of the reference (gaussian_sum_dist/DeterministicGaussianSum.py).

draw() hands out the same cached values on every call, so two filters fed by DeterministicGaussianSum
objects see identical "noise".  Filters recognise this class and take the noise-injection kernels
(gse_pf_predict_injected / gse_gsukf_predict with a noise buffer) -- the parity mode.
"""
import numpy
import torch

from gpu_se_amd.devarray import DevArray, device
from gpu_se_amd.gaussian_sum_dist.MultivariateGaussianSum import MultivariateGaussianSum


class DeterministicGaussianSum(MultivariateGaussianSum):
    """MultivariateGaussianSum whose draws are served from one class-level buffer
    (DeterministicGaussianSum.py:27, :50-61)."""

    _values = numpy.array([], dtype=numpy.float32)

    def __init__(self, means, covariances, weights, library=None):
        super().__init__(means, covariances, weights, library)

    @classmethod
    def preload(cls, values):
        """Install the replay buffer (e.g. noise recorded from the reference)."""
        cls._values = numpy.asarray(values, dtype=numpy.float32).ravel().copy()

    def draw(self, shape=(1,)):
        if not isinstance(shape, tuple):
            shape = (shape,)
        size = int(numpy.prod(shape)) * self._Nx
        have = DeterministicGaussianSum._values.size
        if have < size:
            need = -(-(size - have) // self._Nx)
            fresh = super().draw(need).get()
            DeterministicGaussianSum._values = numpy.hstack([DeterministicGaussianSum._values, fresh.ravel()])
        out = DeterministicGaussianSum._values[:size].reshape(shape + (self._Nx,))
        return DevArray.wrap(torch.as_tensor(numpy.ascontiguousarray(out), device=device()))
