"""Gaussian-sum distribution whose pdf() and draw() run as HIP kernels.

Mirror of the reference's gaussian_sum_dist.MultivariateGaussianSum with library=cupy
(gaussian_sum_dist/MultivariateGaussianSum.py:7-97): same constructor, attributes and methods.
"""
import ctypes

import numpy
import torch

import gpu_se_amd
from gpu_se_amd import _lib
from gpu_se_amd.devarray import DevArray, as_device, device, stream_ptr


class MultivariateGaussianSum:
    """Multivariate Gaussian-sum distribution evaluated and sampled on the MI355X.

    Parameters
    ----------
    means : array (N_distributions x Nx)
    covariances : array (N_distributions x Nx x Nx)
    weights : array (N_distributions)
    library : ignored
        The reference switches between numpy and cupy here (MultivariateGaussianSum.py:27-31);
        this class always computes on the HIP device.  The small parameter arrays themselves stay on
        the host as NumPy arrays (they travel to the kernels as kernel arguments), so callers that
        mutate them in place -- sim_base.get_parts does ``x0.means += X`` (sim_base.py:104) -- work.
    """

    def __init__(self, means, covariances, weights, library=None):
        self.lib = library
        means = numpy.asarray(means)
        self.means = numpy.array(means, dtype=numpy.float32)
        self.weights = numpy.array(weights, dtype=numpy.float32)
        self.covariances = numpy.array(covariances, dtype=numpy.float32)
        # fp64 inverse of the covariances as given, fp32 constants from the fp32 covariances (:33-37)
        self._inverse_covariances = numpy.linalg.inv(numpy.asarray(covariances, dtype=numpy.float64))
        self._Nd, self._Nx = means.shape
        if not (1 <= self._Nd <= _lib.GSE_MAX_ND and 1 <= self._Nx <= _lib.GSE_N_MAX_NX):
            raise ValueError("MultivariateGaussianSum on the HIP device supports Nd <= %d, Nx <= %d (got %d, %d)"
                             % (_lib.GSE_MAX_ND, _lib.GSE_N_MAX_NX, self._Nd, self._Nx))
        # (more than GSE_NX = 5 dimensions -- the noise of a user model of 6 .. 8 states, round 5 --: draw and pdf go through
        #  gse_mixture_draw_n / gse_mixture_pdf_n, the dimension-generic kernels' mixture; _pod(), the fast kernels' struct, is
        #  not available for those)
        self._wide = self._Nx > _lib.GSE_NX
        self._constants = ((2 * numpy.pi) ** (-self._Nx / 2)
                           / numpy.sqrt(numpy.linalg.det(self.covariances))).astype(numpy.float32)
        self._seed = gpu_se_amd._new_stream_seed()
        self._draws = 0

    # ------------------------------------------------------------------ C-ABI view
    def _pod(self):
        """struct gse_mixture for the CURRENT attribute values.  The attributes are plain arrays that callers may
        change in place at any time, so the struct is cached against their bytes (three tiny arrays)."""
        key = (numpy.asarray(self.means).tobytes(), numpy.asarray(self.weights).tobytes(),
               numpy.asarray(self.covariances).tobytes())
        cached = self.__dict__.get("_pod_cache")
        if cached is not None and cached[0] == key:
            return cached[1]
        pod = self._build_pod()
        self._pod_cache = (key, pod)
        return pod

    def _pod_n(self):
        """struct gse_mixture_n for the CURRENT attribute values (any dimension up to 8)."""
        m = _lib.MixtureN()
        n = self._Nx
        m.nd, m.n = self._Nd, n
        w = numpy.asarray(self.weights, dtype=numpy.float32)
        cum = numpy.cumsum(w, dtype=numpy.float32) / numpy.float32(w.sum(dtype=numpy.float32))
        means = numpy.asarray(self.means, dtype=numpy.float32).reshape(self._Nd, n)
        chol = numpy.linalg.cholesky(numpy.asarray(self.covariances, dtype=numpy.float64)).astype(numpy.float32)
        for d in range(self._Nd):
            m.cumw[d] = float(cum[d])
            m.cw[d] = float(numpy.float64(numpy.float32(self._constants[d])) * numpy.float64(w[d]))
            for i in range(n):
                m.mean[d][i] = float(means[d, i])
                for j in range(n):
                    m.chol[d][i][j] = float(chol[d, i, j])
                    m.inv[d][i][j] = float(self._inverse_covariances[d, i, j])
        return m

    def _build_pod(self):
        if self._wide:
            raise ValueError("a mixture of %d dimensions has no fast-kernel struct (gse_mixture: Nx <= %d)" % (self._Nx, _lib.GSE_NX))
        m = _lib.Mixture()
        m.nd, m.nx = self._Nd, self._Nx
        means = numpy.asarray(self.means, dtype=numpy.float32).reshape(self._Nd, self._Nx)
        chol = numpy.linalg.cholesky(numpy.asarray(self.covariances, dtype=numpy.float64)).astype(numpy.float32)
        for d in range(self._Nd):
            m.weights[d] = float(self.weights[d])
            m.consts[d] = float(self._constants[d])
            for i in range(self._Nx):
                m.means[d][i] = float(means[d, i])
                for j in range(self._Nx):
                    m.chol[d][i][j] = float(chol[d, i, j])
                    m.inv_cov[d][i][j] = float(self._inverse_covariances[d, i, j])
        return m

    def _rng(self, element_offset=0, filter_offset=0):
        r = _lib.Rng(seed=self._seed, step=self._draws, element_offset=element_offset, filter_offset=filter_offset,
                     reserved=0)
        self._draws += 1
        return r

    # ------------------------------------------------------------------ API
    def pdf(self, x):
        """Value of the probability density function at the rows of x (m x Nx) -> device array (m,) fp64
        (MultivariateGaussianSum.py:39-63).  fp32 input takes the fp32 route of the reference's GPU class
        (differences x - mean in fp32), anything else is evaluated from fp64 points."""
        if isinstance(x, torch.Tensor):
            t = x.as_subclass(torch.Tensor)
            t = t.to(device()) if t.dtype in (torch.float32, torch.float64) else t.to(device(), torch.float64)
        else:
            a = numpy.atleast_2d(numpy.asarray(x))
            a = a if a.dtype == numpy.float32 else a.astype(numpy.float64)
            t = torch.as_tensor(numpy.ascontiguousarray(a), device=device())
        if t.dim() == 1:
            t = t[None, :]
        t = t.contiguous()
        if t.shape[-1] != self._Nx:
            raise ValueError("pdf: points have %d columns, the mixture has Nx = %d" % (t.shape[-1], self._Nx))
        m = t.shape[0]
        out = torch.empty(m, dtype=torch.float64, device=t.device)
        if self._wide:
            t = t.to(torch.float64).contiguous()
            _lib.call("gse_mixture_pdf_n", t.data_ptr(), m, ctypes.byref(self._pod_n()), out.data_ptr(), stream_ptr())
            return DevArray.wrap(out)
        pod = self._pod()
        fn = "gse_mixture_pdf_f32" if t.dtype == torch.float32 else "gse_mixture_pdf_f64"
        _lib.call(fn, t.data_ptr(), m, ctypes.byref(pod), out.data_ptr(), stream_ptr())
        return DevArray.wrap(out)

    def draw(self, shape=(1,), _element_offset=0):
        """Samples, device array (*shape x Nx) fp32 (MultivariateGaussianSum.py:65-97).  The component of
        every sample is drawn independently (the reference lays components out in contiguous blocks --
        each row has the same distribution); Philox4x32-10 + Box-Muller + the Cholesky factor."""
        if not isinstance(shape, tuple):
            shape = (shape,)
        size = int(numpy.prod(shape))
        out = torch.empty((size, self._Nx), dtype=torch.float32, device=device())
        # (_element_offset: global index of the first sample -- a shard of a larger draw takes ITS rows of it)
        if self._wide:
            rng = self._rng(element_offset=int(_element_offset))
            _lib.call("gse_mixture_draw_n", out.data_ptr(), size, ctypes.byref(self._pod_n()), ctypes.byref(rng), stream_ptr())
            return DevArray.wrap(out.reshape(shape + (self._Nx,)))
        pod, rng = self._pod(), self._rng(element_offset=int(_element_offset))
        _lib.call("gse_mixture_draw", out.data_ptr(), size, 0, ctypes.byref(pod), ctypes.byref(rng), stream_ptr())
        return DevArray.wrap(out.reshape(shape + (self._Nx,)))

    def _draw_state_major(self, out, n, ld, element_offset=0, filter_offset=0):
        """Draw n samples straight into a state-major [Nx][ld] buffer (how the filters draw their prior)."""
        pod, rng = self._pod(), self._rng(element_offset, filter_offset)
        _lib.call("gse_mixture_draw", out.data_ptr(), n, ld, ctypes.byref(pod), ctypes.byref(rng), stream_ptr())
