"""The two filters for user models beyond the fast kernels' bounds (round 5; VERDICT r04 missing #3).

The reference compiles ANY Python ``f`` / ``g`` of any dimension with Numba (filter/particle.py:176-208, filter/gs_ukf.py:243-275).
Models of up to 5 states / 2 inputs / exactly 2 outputs run this build's fast kernels compiled for the model (usermodel.py);
models of up to 8 states, 4 inputs, 1 .. 4 outputs come here: the dimension-generic kernels of csrc/gse_generic_kernels.h, compiled
in the process by hiprtc with the model's dimensions as constants and launched by name (gse_model_launch) -- from Python, the way
the reference launches its Numba kernels.  Same class API (filter/particle.py:117-327, filter/gs_ukf.py:186-449); particles are rows
of Nx floats; the resample's cumulative weights and ancestors come from the model-independent entry points
gse_cumsum_normalised / gse_systematic_index.  Correct and parallel, not tuned.

``ParallelParticleFilter(f, g, ...)`` and ``ParallelGaussianSumUnscentedKalmanFilter(f, g, ...)`` return these classes by
themselves when the model needs them.
"""
import ctypes

import numpy
import torch

import gpu_se_amd
from gpu_se_amd import _lib, usermodel
from gpu_se_amd.devarray import DevArray, as_device, device, stream_ptr
from gpu_se_amd.gaussian_sum_dist import DeterministicGaussianSum

MAX_NX, MAX_NU, MAX_NY, MAX_ND = 8, 4, 4, 4


class _IO(ctypes.Structure):
    """struct gse::GenericIO."""
    _fields_ = [("u", ctypes.c_double * MAX_NU), ("z", ctypes.c_double * MAX_NY), ("dt", ctypes.c_double)]


class _RngKey(ctypes.Structure):
    """struct gse::RngKey (csrc/gse_device.h); make_rng_key's packing."""
    _fields_ = [("k0", ctypes.c_uint32), ("k1", ctypes.c_uint32), ("step_lo", ctypes.c_uint32), ("hi", ctypes.c_uint32),
                ("elem0", ctypes.c_uint64)]


_PURPOSE_PREDICT, _PURPOSE_GSUKF = 1, 3


def _rng_key(seed, step, purpose, element_offset=0):
    return _RngKey(k0=seed & 0xFFFFFFFF, k1=(seed >> 32) & 0xFFFFFFFF, step_lo=step & 0xFFFFFFFF,
                   hi=(((step >> 32) & 0xFF) << 24) | ((purpose & 0xF) << 20), elem0=element_offset)


def _device_mixture(pdf, n):
    """A MultivariateGaussianSum of dimension n as a gse_mixture_n in device memory (rebuilt on every call: callers mutate the
    host attributes, sim_base.py:104)."""
    if pdf._Nx != n:
        raise ValueError("a %d-dimensional mixture where the model needs %d" % (pdf._Nx, n))
    raw = numpy.frombuffer(bytes(pdf._pod_n()), dtype=numpy.uint8).copy()
    return torch.as_tensor(raw, device=device())


class _GenericBase:
    def _setup(self, f, g, N_particles, x0, state_pdf, measurement_pdf):
        self._model = usermodel.compile_generic_model(f, g)
        self._Nx, self._Nu, self._Ny = self._model.nx, self._model.nu, self._model.ny
        for name, pdf, want in (("x0", x0, self._Nx), ("state_pdf", state_pdf, self._Nx), ("measurement_pdf", measurement_pdf, self._Ny)):
            if pdf._Nx != want:
                raise ValueError("%s is %d-dimensional, the model needs %d" % (name, pdf._Nx, want))
        self.f, self.g = f, g
        self.N_particles = int(N_particles)
        self.state_pdf, self.measurement_pdf = state_pdf, measurement_pdf
        self._seed = gpu_se_amd._new_stream_seed()
        self._step = 0
        self._host_rng = numpy.random.default_rng(self._seed & 0xFFFFFFFF)
        n = self.N_particles
        self._ws = torch.zeros(_lib.load().gse_workspace_bytes(1, n), dtype=torch.uint8, device=device())
        self._cumsum = torch.empty(n, dtype=torch.float64, device=device())
        self._idx = torch.empty(n, dtype=torch.int64, device=device())
        self._last_index = None

    def _io(self, u, z=None, dt=0.0):
        io = _IO()
        uu = numpy.asarray(u, dtype=numpy.float64).reshape(-1)
        if len(uu) != self._Nu:
            raise ValueError("the model takes %d inputs, got %d" % (self._Nu, len(uu)))
        for k in range(self._Nu):
            io.u[k] = float(uu[k])
        if z is not None:
            zz = numpy.asarray(z, dtype=numpy.float64).reshape(-1)
            if len(zz) != self._Ny:
                raise ValueError("the model has %d outputs, got a measurement of %d" % (self._Ny, len(zz)))
            for k in range(self._Ny):
                io.z[k] = float(zz[k])
        io.dt = float(dt)
        return io

    def _launch(self, name, n_threads, block, *args):
        self._model.launch(name, n_threads, block, args, stream_ptr())

    @staticmethod
    def _p(t):
        return ctypes.c_void_p(t.data_ptr() if t is not None else None)

    def _ancestors(self, w, r):
        """Systematic resample's ancestors of the weights w (filter/particle.py:296-310): device int64 [N]."""
        n = self.N_particles
        r = self._host_rng.random() if r is None else float(numpy.asarray(r, dtype=numpy.float64).reshape(-1)[0])
        _lib.call("gse_cumsum_normalised", w.data_ptr(), self._cumsum.data_ptr(), n, self._ws.data_ptr(), stream_ptr())
        _lib.call("gse_systematic_index", self._cumsum.data_ptr(), self._idx.data_ptr(), n, r, stream_ptr())
        return self._idx

    def _partials(self, kernel, *tensors):
        n = self.N_particles
        width = 1 + self._Nx + self._Nx * (self._Nx + 1) // 2
        nblk = max(1, min(256, -(-n // 256)))
        out = torch.empty((nblk, width), dtype=torch.float64, device=device())
        self._model.launch(kernel, nblk * 256, 256, [self._p(t) for t in tensors] + [ctypes.c_int64(n), self._p(out)], stream_ptr())
        rows = out.cpu().numpy()
        tot = rows[0].copy()
        for k in range(1, nblk):
            tot += rows[k]
        nx = self._Nx
        s0, s1 = tot[0], tot[1:1 + nx]
        s2 = numpy.zeros((nx, nx))
        s2[numpy.triu_indices(nx)] = tot[1 + nx:]
        s2 = s2 + s2.T - numpy.diag(numpy.diag(s2))
        return s0, s1, s2

    def f_vectorize(self, particles, u, dt):
        """out[i] = f(particles[i], u, dt), device array fp32 (filter/particle.py:176-191)."""
        p = as_device(particles, torch.float32)
        lead = p.shape[:-1]
        p2 = p.reshape(-1, self._Nx).contiguous()
        out = torch.empty_like(p2)
        self._launch("generic_f_rows_kernel", p2.shape[0], 256, self._p(p2), self._p(out), ctypes.c_int64(p2.shape[0]), self._io(u, dt=dt))
        return DevArray.wrap(out.reshape(lead + (self._Nx,)))

    def g_vectorize(self, particles, u, _y_dummy=None):
        """out[i] = g(particles[i], u), device array fp32 (filter/particle.py:193-208)."""
        p = as_device(particles, torch.float32)
        lead = p.shape[:-1]
        p2 = p.reshape(-1, self._Nx).contiguous()
        out = torch.empty((p2.shape[0], self._Ny), dtype=torch.float32, device=p2.device)
        self._launch("generic_g_rows_kernel", p2.shape[0], 256, self._p(p2), self._p(out), ctypes.c_int64(p2.shape[0]), self._io(u))
        return DevArray.wrap(out.reshape(lead + (self._Ny,)))


class GenericParticleFilter(_GenericBase):
    """ParallelParticleFilter (filter/particle.py:117-327) for a model of up to 8 states, 4 inputs, 4 outputs."""

    def __init__(self, f, g, N_particles, x0, state_pdf, measurement_pdf):
        self._setup(f, g, N_particles, x0, state_pdf, measurement_pdf)
        n = self.N_particles
        self._x = x0.draw(n).as_subclass(torch.Tensor).reshape(n, self._Nx).contiguous()            # (particle.py:49)
        self._x_alt = torch.empty_like(self._x)
        self._w = torch.full((n,), 1.0 / n, dtype=torch.float32, device=self._x.device)             # (particle.py:50)
        self._w_alt = torch.empty_like(self._w)

    @property
    def particles(self):
        return DevArray.wrap(self._x)

    @particles.setter
    def particles(self, value):
        self._x.copy_(as_device(value, torch.float32).reshape(self.N_particles, self._Nx))

    @property
    def weights(self):
        return DevArray.wrap(self._w)

    @weights.setter
    def weights(self, value):
        self._w.copy_(as_device(value, torch.float32).reshape(self.N_particles))

    def predict(self, u, dt):
        """x_i += f(x_i, u, dt), then x += state noise (filter/particle.py:265-277)."""
        n = self.N_particles
        io = self._io(u, dt=dt)
        if isinstance(self.state_pdf, DeterministicGaussianSum):
            inj = self.state_pdf.draw(n).as_subclass(torch.Tensor).reshape(n, self._Nx).contiguous()
            self._launch("generic_pf_predict_kernel<true>", n, 256, self._p(self._x), ctypes.c_int64(n), io, self._p(None),
                         _rng_key(0, 0, 0), self._p(inj))
            return
        pdf = _device_mixture(self.state_pdf, self._Nx)
        key = _rng_key(self._seed, self._step, _PURPOSE_PREDICT)
        self._step += 1
        self._launch("generic_pf_predict_kernel<false>", n, 256, self._p(self._x), ctypes.c_int64(n), io, self._p(pdf), key,
                     self._p(None))

    def update(self, u, z):
        """w_i *= measurement_pdf(z - g(x_i, u)) (filter/particle.py:279-294)."""
        n = self.N_particles
        pdf = _device_mixture(self.measurement_pdf, self._Ny)
        self._launch("generic_pf_update_kernel", n, 256, self._p(self._x), self._p(self._w), ctypes.c_int64(n), self._io(u, z),
                     self._p(pdf))

    def resample(self, r=None, return_index=False):
        """Systematic resample (filter/particle.py:296-316)."""
        n = self.N_particles
        idx = self._ancestors(self._w, r)
        self._launch("generic_gather_kernel", n, 256, self._p(self._x), self._p(idx), self._p(self._x_alt), self._p(self._w_alt),
                     ctypes.c_int64(n), ctypes.c_int(self._Nx))
        self._x, self._x_alt = self._x_alt, self._x
        self._w, self._w_alt = self._w_alt, self._w
        if return_index:
            return DevArray.wrap(idx.clone())

    def _sums(self):
        s0, s1, s2 = self._partials("generic_moments_kernel", self._x, self._w)
        pivot = self._x[0].cpu().numpy().astype(numpy.float64)
        from gpu_se_amd.filter.particle import moments_to_mean_cov
        return moments_to_mean_cov(s0, s1, s2[numpy.triu_indices(self._Nx)], pivot)

    def point_estimate(self):
        """weights @ particles (filter/particle.py:318-320)."""
        return self._sums()[1]

    def point_covariance(self):
        """Largest singular value of the weighted covariance (filter/particle.py:322-327)."""
        return numpy.linalg.svd(self._sums()[2], compute_uv=False)[0]


class GenericGaussianSumFilter(_GenericBase):
    """ParallelGaussianSumUnscentedKalmanFilter (filter/gs_ukf.py:186-449) for a model of up to 8 states, 4 inputs, 4 outputs."""

    def __init__(self, f, g, N_particles, x0, state_pdf, measurement_pdf):
        self._setup(f, g, N_particles, x0, state_pdf, measurement_pdf)
        n, nx = self.N_particles, self._Nx
        dev = device()
        self._means = x0.draw(n).as_subclass(torch.Tensor).reshape(n, nx).contiguous()                     # (gs_ukf.py:50)
        cov0 = torch.as_tensor(numpy.asarray(state_pdf.covariances[0], dtype=numpy.float32), device=dev)
        self._covs = cov0[None, :, :].repeat(n, 1, 1).contiguous()                                          # (:52)
        self._w = torch.full((n,), 1.0 / n, dtype=torch.float32, device=dev)                                # (:54)
        self._means_alt, self._covs_alt, self._w_alt = (torch.empty_like(t) for t in (self._means, self._covs, self._w))
        self._N_sigmas = 2 * nx + 1
        w_sigma = numpy.full(self._N_sigmas, 1 / (2 * nx + 8 / 5), dtype=numpy.float32)                    # (gs_ukf.py:62-67)
        w_sigma[0] = 1 / (1 + 5 / 4 * nx)
        self._w_sigma_host = w_sigma
        self._w_sigma = DevArray.wrap(torch.as_tensor(w_sigma, device=dev))
        self._status = torch.zeros(1, dtype=torch.int32, device=dev)

    means = property(lambda self: DevArray.wrap(self._means))
    covariances = property(lambda self: DevArray.wrap(self._covs))
    weights = property(lambda self: DevArray.wrap(self._w))

    @means.setter
    def means(self, value):
        self._means.copy_(as_device(value, torch.float32).reshape(self.N_particles, self._Nx))

    @covariances.setter
    def covariances(self, value):
        self._covs.copy_(as_device(value, torch.float32).reshape(self.N_particles, self._Nx, self._Nx))

    @weights.setter
    def weights(self, value):
        self._w.copy_(as_device(value, torch.float32).reshape(self.N_particles))

    def _w01(self):
        return ctypes.c_float(float(self._w_sigma_host[0])), ctypes.c_float(float(self._w_sigma_host[1]))

    def _check_status(self):
        bad = int(self._status.item())
        if bad:
            self._status.zero_()
            raise numpy.linalg.LinAlgError("%d covariance matrices are not positive definite" % bad)

    def _get_sigma_points(self):
        """Sigma points of the current Gaussians, device array (N x (2 Nx + 1) x Nx) (filter/gs_ukf.py:332-346)."""
        n = self.N_particles
        out = torch.empty((n, self._N_sigmas, self._Nx), dtype=torch.float32, device=self._means.device)
        self._launch("generic_gsukf_sigma_kernel", n, 64, self._p(self._means), self._p(self._covs), self._p(out), ctypes.c_int64(n),
                     self._p(self._status))
        return DevArray.wrap(out)

    def predict(self, u, dt):
        """(filter/gs_ukf.py:348-367)"""
        n = self.N_particles
        io = self._io(u, dt=dt)
        w0, w1 = self._w01()
        if isinstance(self.state_pdf, DeterministicGaussianSum):
            inj = self.state_pdf.draw((n, self._N_sigmas)).as_subclass(torch.Tensor).reshape(n, self._N_sigmas, self._Nx).contiguous()
            self._launch("generic_gsukf_predict_kernel<true>", n, 64, self._p(self._means), self._p(self._covs), ctypes.c_int64(n), io,
                         w0, w1, self._p(None), _rng_key(0, 0, 0), self._p(inj), self._p(self._status))
            return
        pdf = _device_mixture(self.state_pdf, self._Nx)
        key = _rng_key(self._seed, self._step, _PURPOSE_GSUKF)
        self._step += 1
        self._launch("generic_gsukf_predict_kernel<false>", n, 64, self._p(self._means), self._p(self._covs), ctypes.c_int64(n), io,
                     w0, w1, self._p(pdf), key, self._p(None), self._p(self._status))

    def update(self, u, z):
        """(filter/gs_ukf.py:369-407)"""
        n = self.N_particles
        w0, w1 = self._w01()
        pdf = _device_mixture(self.measurement_pdf, self._Ny)
        self._launch("generic_gsukf_update_kernel", n, 64, self._p(self._means), self._p(self._covs), self._p(self._w),
                     ctypes.c_int64(n), self._io(u, z), w0, w1, self._p(pdf), self._p(self._status))

    def resample(self, r=None, return_index=False):
        """(filter/gs_ukf.py:409-436)"""
        n, nx = self.N_particles, self._Nx
        idx = self._ancestors(self._w, r)
        self._launch("generic_gather_kernel", n, 256, self._p(self._means), self._p(idx), self._p(self._means_alt),
                     self._p(self._w_alt), ctypes.c_int64(n), ctypes.c_int(nx))
        self._launch("generic_gather_kernel", n, 256, self._p(self._covs), self._p(idx), self._p(self._covs_alt), self._p(None),
                     ctypes.c_int64(n), ctypes.c_int(nx * nx))
        self._means, self._means_alt = self._means_alt, self._means
        self._covs, self._covs_alt = self._covs_alt, self._covs
        self._w, self._w_alt = self._w_alt, self._w
        if return_index:
            return DevArray.wrap(idx.clone())

    def _sums(self):
        s0, s1, s2 = self._partials("generic_gsukf_moments_kernel", self._means, self._covs, self._w)
        self._check_status()
        pivot = self._means[0].cpu().numpy().astype(numpy.float64)
        # S2 already holds sum w (P + d d^T): undo the pivot shift of the d d^T part only (P does not move)
        mean = s1 + pivot * s0
        e = mean - pivot
        cov = s2 - numpy.outer(s1, e) - numpy.outer(e, s1) + s0 * numpy.outer(e, e)
        return s0, mean, cov

    def point_estimate(self):
        """weights @ means (filter/gs_ukf.py:438-440)."""
        return self._sums()[1]

    def point_covariance(self):
        """Largest singular value of sum w P + sum w (mu - m)(mu - m)^T (filter/gs_ukf.py:442-449)."""
        return numpy.linalg.svd(self._sums()[2], compute_uv=False)[0]
