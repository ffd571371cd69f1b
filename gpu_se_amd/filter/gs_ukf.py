"""Gaussian-sum unscented Kalman filter on the MI355X: drop-in for the reference's
filter.ParallelGaussianSumUnscentedKalmanFilter (filter/gs_ukf.py:186-449)."""
import ctypes

import numpy
import torch

import gpu_se_amd
from gpu_se_amd import _lib
from gpu_se_amd.devarray import DevArray, as_device, device, stream_ptr
from gpu_se_amd.filter.particle import _KernelLauncher, _parallel_resample, _require_builtin_model, rows_f, rows_g
from gpu_se_amd.gaussian_sum_dist import DeterministicGaussianSum


class ParallelGaussianSumUnscentedKalmanFilter:
    """Gaussian Sum Unscented Kalman Filter class implemented to run on the GPU (MI355X).

    Parameters as for ParallelParticleFilter; N_particles is the number of Gaussians.

    Attributes
    ----------
    means : device array (N x Nx) fp32
    covariances : device array (N x Nx x Nx) fp32
    weights : device array (N) fp32
    """

    _parallel_resample = _KernelLauncher(_parallel_resample)

    def __init__(self, f, g, N_particles, x0, state_pdf, measurement_pdf):
        _require_builtin_model(f, g)
        _lib.load()
        self.f = f
        self.g = g
        self.N_particles = int(N_particles)
        self.state_pdf = state_pdf
        self.measurement_pdf = measurement_pdf
        n = self.N_particles
        dev = device()

        self._means = x0.draw(n).as_subclass(torch.Tensor).reshape(n, _lib.GSE_NX).contiguous()          # (gs_ukf.py:50)
        cov0 = torch.as_tensor(numpy.asarray(state_pdf.covariances[0], dtype=numpy.float32), device=dev)
        self._covs = cov0[None, :, :].repeat(n, 1, 1).contiguous()                                        # (:52)
        self._w = torch.full((n,), 1.0 / n, dtype=torch.float32, device=dev)                              # (:54)
        self._means_alt = torch.empty_like(self._means)
        self._covs_alt = torch.empty_like(self._covs)
        self._w_alt = torch.empty_like(self._w)

        self._Nx = _lib.GSE_NX
        self._Ny = _lib.GSE_NY
        self._N_sigmas = 2 * self._Nx + 1
        # sigma weights: w_mu + 2 n w_sigma = 1 and w_mu / w_sigma = 0.4 / 0.25  (gs_ukf.py:62-67)
        w_sigma = numpy.full(self._N_sigmas, 1 / (2 * self._Nx + 8 / 5), dtype=numpy.float32)
        w_sigma[0] = 1 / (1 + 5 / 4 * self._Nx)
        self._w_sigma_host = w_sigma
        self._w_sigma_c = (_lib.c_f * self._N_sigmas)(*[float(v) for v in w_sigma])
        self._w_sigma = DevArray.wrap(torch.as_tensor(w_sigma, device=dev))

        self._ws = torch.empty(_lib.load().gse_workspace_bytes(1, n), dtype=torch.uint8, device=dev)
        self._moments = torch.empty(46, dtype=torch.float64, device=dev)
        self._status = torch.zeros(1, dtype=torch.int32, device=dev)
        self._seed = gpu_se_amd._new_stream_seed()
        self._step = 0
        self._host_rng = numpy.random.default_rng(self._seed & 0xFFFFFFFF)

        # deferred stages (small banks): predict() and update() are recorded and resample() runs all three in ONE launch
        self.lazy = True
        self._pend_predict = None
        self._pend_update = None
        self._fusable = n <= _lib.GSE_GSUKF_CYCLE_MAX
        self._fuse_pu = n >= _lib.GSE_GSUKF_LANES_MIN      # large banks: predict() is deferred and rides update()'s launch
        self._c_u = (_lib.c_d * _lib.GSE_NU)()
        self._c_z = (_lib.c_d * _lib.GSE_NY)()
        self._c_r = (_lib.c_d * 1)()

        self._threads_per_block = self._tpb = 1024
        self._blocks_per_grid = self._bpg = (n - 1) // 1024 + 1
        self._y_dummy = DevArray.wrap(torch.zeros(self._Ny, dtype=torch.float32, device=dev))

    # ------------------------------------------------------------------ attributes
    def _flush(self):
        """Run whatever predict / update is pending with the one-kernel-per-method calls."""
        if self._pend_predict is not None:
            u, dt = self._pend_predict
            self._pend_predict = None
            self._predict_now(u, dt)
        if self._pend_update is not None:
            u, z = self._pend_update
            self._pend_update = None
            self._update_now(u, z)

    @property
    def means(self):
        self._flush()
        return DevArray.wrap(self._means)

    @means.setter
    def means(self, value):
        self._flush()
        self._means.copy_(as_device(value, torch.float32).reshape(self.N_particles, self._Nx))

    @property
    def covariances(self):
        self._flush()
        return DevArray.wrap(self._covs)

    @covariances.setter
    def covariances(self, value):
        self._flush()
        self._covs.copy_(as_device(value, torch.float32).reshape(self.N_particles, self._Nx, self._Nx))

    @property
    def weights(self):
        self._flush()
        return DevArray.wrap(self._w)

    @weights.setter
    def weights(self, value):
        self._flush()
        self._w.copy_(as_device(value, torch.float32).reshape(self.N_particles))

    def _check_status(self):
        """Raise like the reference does when a covariance is not positive definite even after the
        1e-10 retry (gs_ukf.py:336-339 lets torch's RuntimeError escape).  Synchronises."""
        bad = int(self._status.item())
        if bad:
            self._status.zero_()
            raise numpy.linalg.LinAlgError("%d covariance matrices are not positive definite" % bad)

    # ------------------------------------------------------------------ the cycle
    def _get_sigma_points(self):
        """Return the sigma points for the current Gaussians, device array (N x 11 x Nx) (gs_ukf.py:332-346)."""
        self._flush()
        n = self.N_particles
        out = torch.empty((n, self._N_sigmas, self._Nx), dtype=torch.float32, device=self._means.device)
        _lib.call("gse_gsukf_sigma_points", self._means.data_ptr(), self._covs.data_ptr(), out.data_ptr(), 1, n,
                  self._status.data_ptr(), stream_ptr())
        return DevArray.wrap(out)

    def predict(self, u, dt):
        """Prediction step (gs_ukf.py:348-367).  With at most 64 Gaussians the step is deferred so that resample() can
        run predict + update + resample as one launch (the bank is launch-latency-bound)."""
        if self.lazy and (self._fusable or self._fuse_pu) and self._pend_predict is None and self._pend_update is None:
            self._pend_predict = ([float(v) for v in u], float(dt))
            return
        self._flush()
        self._predict_now(u, dt)

    def _predict_now(self, u, dt):
        uu = _lib.doubles(u, _lib.GSE_NU)
        n = self.N_particles
        if isinstance(self.state_pdf, DeterministicGaussianSum):
            noise = self.state_pdf.draw((n, self._N_sigmas)).as_subclass(torch.Tensor).contiguous()
            _lib.call("gse_gsukf_predict", self._means.data_ptr(), self._covs.data_ptr(), 1, n, uu, float(dt),
                      self._w_sigma_c, None, None, noise.data_ptr(), self._status.data_ptr(), stream_ptr())
        else:
            pod = self.state_pdf._pod()
            rng = _lib.Rng(seed=self._seed, step=self._step)
            self._step += 1
            _lib.call("gse_gsukf_predict", self._means.data_ptr(), self._covs.data_ptr(), 1, n, uu, float(dt),
                      self._w_sigma_c, ctypes.byref(pod), ctypes.byref(rng), None, self._status.data_ptr(),
                      stream_ptr())

    def update(self, u, z):
        """Update step: local UKF update of every Gaussian, then the global weight update (gs_ukf.py:369-407)."""
        if self.lazy and self._fusable and self._pend_predict is not None and self._pend_update is None:
            self._pend_update = ([float(v) for v in u], [float(v) for v in z])
            return
        if self.lazy and self._fuse_pu and self._pend_predict is not None and self._pend_update is None:
            (up, dt), self._pend_predict = self._pend_predict, None
            self._predict_update_now(up, dt, z)
            return
        self._flush()
        self._update_now(u, z)

    def _predict_update_now(self, u, dt, z):
        """predict(u, dt) and update(., z) of a large bank in one launch (gse_gsukf_predict_update): same bits as the two."""
        uu, zz = _lib.doubles(u, _lib.GSE_NU), _lib.doubles(z, _lib.GSE_NY)
        n = self.N_particles
        pod_m = self.measurement_pdf._pod()
        if isinstance(self.state_pdf, DeterministicGaussianSum):
            noise = self.state_pdf.draw((n, self._N_sigmas)).as_subclass(torch.Tensor).contiguous()
            _lib.call("gse_gsukf_predict_update", self._means.data_ptr(), self._covs.data_ptr(), self._w.data_ptr(), 1, n, uu,
                      float(dt), zz, self._w_sigma_c, None, None, noise.data_ptr(), ctypes.byref(pod_m),
                      self._status.data_ptr(), stream_ptr())
        else:
            pod_s = self.state_pdf._pod()
            rng = _lib.Rng(seed=self._seed, step=self._step)
            self._step += 1
            _lib.call("gse_gsukf_predict_update", self._means.data_ptr(), self._covs.data_ptr(), self._w.data_ptr(), 1, n, uu,
                      float(dt), zz, self._w_sigma_c, ctypes.byref(pod_s), ctypes.byref(rng), None, ctypes.byref(pod_m),
                      self._status.data_ptr(), stream_ptr())

    def _update_now(self, u, z):
        uu = _lib.doubles(u, _lib.GSE_NU)
        zz = _lib.doubles(z, _lib.GSE_NY)
        pod = self.measurement_pdf._pod()
        _lib.call("gse_gsukf_update", self._means.data_ptr(), self._covs.data_ptr(), self._w.data_ptr(), 1,
                  self.N_particles, uu, zz, self._w_sigma_c, ctypes.byref(pod), self._status.data_ptr(),
                  self._ws.data_ptr(), stream_ptr())

    def resample(self, r=None, return_index=False):
        """Systematic resample of the Gaussians based on their weights (gs_ukf.py:409-436)."""
        rr = _lib.doubles([self._host_rng.random() if r is None else r], 1)
        n = self.N_particles
        idx = torch.empty(n, dtype=torch.int32, device=self._w.device) if return_index else None
        if self._pend_predict is not None and self._pend_update is not None:
            # the whole cycle in one launch of one workgroup (gse_gsukf_cycle); in place
            (u, dt), (_, z) = self._pend_predict, self._pend_update
            self._pend_predict = self._pend_update = None
            uu, zz = _lib.doubles(u, _lib.GSE_NU), _lib.doubles(z, _lib.GSE_NY)
            pod_m = self.measurement_pdf._pod()
            if isinstance(self.state_pdf, DeterministicGaussianSum):
                noise = self.state_pdf.draw((n, self._N_sigmas)).as_subclass(torch.Tensor).contiguous()
                _lib.call("gse_gsukf_cycle", self._means.data_ptr(), self._covs.data_ptr(), self._w.data_ptr(),
                          idx.data_ptr() if return_index else None, 1, n, uu, dt, zz, rr, self._w_sigma_c, None, None,
                          noise.data_ptr(), ctypes.byref(pod_m), self._status.data_ptr(), stream_ptr())
            else:
                pod_s = self.state_pdf._pod()
                rng = _lib.Rng(seed=self._seed, step=self._step)
                self._step += 1
                _lib.call("gse_gsukf_cycle", self._means.data_ptr(), self._covs.data_ptr(), self._w.data_ptr(),
                          idx.data_ptr() if return_index else None, 1, n, uu, dt, zz, rr, self._w_sigma_c,
                          ctypes.byref(pod_s), ctypes.byref(rng), None, ctypes.byref(pod_m), self._status.data_ptr(),
                          stream_ptr())
            if return_index:
                return DevArray.wrap(idx)
            return
        self._flush()
        _lib.call("gse_gsukf_resample", self._means.data_ptr(), self._covs.data_ptr(), self._w.data_ptr(),
                  self._means_alt.data_ptr(), self._covs_alt.data_ptr(), self._w_alt.data_ptr(),
                  idx.data_ptr() if return_index else None, 1, n, rr, 0, self._ws.data_ptr(), stream_ptr())
        self._means, self._means_alt = self._means_alt, self._means
        self._covs, self._covs_alt = self._covs_alt, self._covs
        self._w, self._w_alt = self._w_alt, self._w
        if return_index:
            return DevArray.wrap(idx)

    # ------------------------------------------------------------------ estimates
    def _moment_sums(self):
        self._flush()
        _lib.call("gse_gsukf_moments", self._means.data_ptr(), self._covs.data_ptr(), self._w.data_ptr(), 1,
                  self.N_particles, self._moments.data_ptr(), self._ws.data_ptr(), stream_ptr())
        m = self._moments.cpu().numpy()
        self._check_status()
        from gpu_se_amd.filter.particle import moments_to_mean_cov
        s0, mean, cov_mean = moments_to_mean_cov(m[0], m[1:6], m[6:21], m[36:41])
        cc = numpy.zeros((self._Nx, self._Nx))
        cc[numpy.triu_indices(self._Nx)] = m[21:36]
        cov_cov = cc + cc.T - numpy.diag(numpy.diag(cc))
        return s0, mean, cov_cov + cov_mean

    def point_estimate(self):
        """Returns the point estimate of the filter: weights @ means (gs_ukf.py:438-440)."""
        return self._moment_sums()[1]

    def point_covariance(self):
        """Returns the maximum singular value of the filter's covariance (gs_ukf.py:442-449)."""
        return numpy.linalg.svd(self._moment_sums()[2], compute_uv=False)[0]

    # ------------------------------------------------------------------ pieces the benchmark scripts call
    def f_vectorize(self, particles, u, dt):
        """out[..., :] = f(particles[..., :], u, dt), device array fp32 (gs_ukf.py:243-258; gsf_run_seq.py:220)."""
        return rows_f(_lib.call, self._Nx, particles, [float(v) for v in numpy.asarray(u, dtype=numpy.float64).reshape(-1)], dt)

    def g_vectorize(self, particles, u, _y_dummy=None):
        """out[..., :] = g(particles[..., :], u), device array (.. x Ny) fp32 (gs_ukf.py:260-275; gsf_run_seq.py:285)."""
        return rows_g(_lib.call, self._Nx, particles, [float(v) for v in numpy.asarray(u, dtype=numpy.float64).reshape(-1)])
