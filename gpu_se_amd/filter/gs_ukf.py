"""Gaussian-sum unscented Kalman filter on the MI355X: drop-in for the reference's
filter.ParallelGaussianSumUnscentedKalmanFilter (filter/gs_ukf.py:186-449)."""
import ctypes

import numpy
import torch

import gpu_se_amd
from gpu_se_amd import _lib
from gpu_se_amd.devarray import DevArray, as_device, device, stream_ptr
from gpu_se_amd.filter.particle import _KernelLauncher, _model_library, _parallel_resample, rows_f, rows_g
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

    def __new__(cls, f, g, *args, **kwargs):
        # (as ParallelParticleFilter.__new__: models beyond the fast kernels' bounds get the dimension-generic class)
        from gpu_se_amd.model import Bioreactor
        if cls is ParallelGaussianSumUnscentedKalmanFilter and not (f is Bioreactor.homeostatic_DEs and g is Bioreactor.static_outputs):
            from gpu_se_amd import usermodel
            if usermodel.needs_generic_kernels(f, g):
                from gpu_se_amd.filter.generic import GenericGaussianSumFilter
                if kwargs.get("element_offset") or kwargs.get("_batch", 1) != 1:
                    raise NotImplementedError("shards and batches run the fast kernels' models only")
                return GenericGaussianSumFilter(f, g, *args)
        return object.__new__(cls)

    def __init__(self, f, g, N_particles, x0, state_pdf, measurement_pdf, _batch=1, element_offset=0):
        # the bioreactor functions run the kernels shipped in the library; any other (f, g) is translated and its kernels
        # are compiled for the device at run time (gs_ukf.py:243-275 hands them to Numba)
        self._lib, self._model = _model_library(f, g)
        self._Nx = self._model.nx if self._model else _lib.GSE_NX
        for name, pdf, want in (("x0", x0, self._Nx), ("state_pdf", state_pdf, self._Nx),
                                ("measurement_pdf", measurement_pdf, _lib.GSE_NY)):
            if pdf._Nx != want:
                raise ValueError("%s is %d-dimensional, the model needs %d" % (name, pdf._Nx, want))
        self.f = f
        self.g = g
        self.N_particles = int(N_particles)
        self.state_pdf = state_pdf
        self.measurement_pdf = measurement_pdf
        n = self.N_particles
        B = self._B = int(_batch)                     # independent banks sharing every launch (1 for this class)
        if not 1 <= B <= _lib.GSE_MAX_B:
            raise ValueError("batch size must be in 1..%d" % _lib.GSE_MAX_B)
        dev = device()

        # (element_offset: global index of this bank's first Gaussian when it is a SHARD of a larger bank -- its prior rows and
        #  its sigma-point noise are then those the whole bank draws for these Gaussians: gpu_se_amd/distributed.py)
        self._element_offset = int(element_offset)
        if self._element_offset and B != 1:
            raise ValueError("a shard is one bank")
        self._means = x0.draw((B, n), _element_offset=self._element_offset).as_subclass(torch.Tensor) \
            .reshape(B, n, self._Nx).contiguous()                                                         # (gs_ukf.py:50)
        cov0 = torch.as_tensor(numpy.asarray(state_pdf.covariances[0], dtype=numpy.float32), device=dev)
        self._covs = cov0[None, None, :, :].repeat(B, n, 1, 1).contiguous()                               # (:52)
        self._w = torch.full((B, n), 1.0 / n, dtype=torch.float32, device=dev)                            # (:54)
        self._means_alt = torch.empty_like(self._means)
        self._covs_alt = torch.empty_like(self._covs)
        self._w_alt = torch.empty_like(self._w)

        self._Ny = _lib.GSE_NY
        self._N_sigmas = 2 * self._Nx + 1
        # sigma weights: w_mu + 2 n w_sigma = 1 and w_mu / w_sigma = 0.4 / 0.25  (gs_ukf.py:62-67)
        w_sigma = numpy.full(self._N_sigmas, 1 / (2 * self._Nx + 8 / 5), dtype=numpy.float32)
        w_sigma[0] = 1 / (1 + 5 / 4 * self._Nx)
        self._w_sigma_host = w_sigma
        self._w_sigma_c = (_lib.c_f * self._N_sigmas)(*[float(v) for v in w_sigma])
        self._w_sigma = DevArray.wrap(torch.as_tensor(w_sigma, device=dev))

        self._ws = torch.zeros(_lib.load().gse_workspace_bytes(B, n), dtype=torch.uint8, device=dev)
        self._moments = torch.empty((B, _lib.GSE_GSUKF_MOMENTS), dtype=torch.float64, device=dev)
        self._status = torch.zeros(1, dtype=torch.int32, device=dev)
        self._seed = gpu_se_amd._new_stream_seed()
        self._step = 0
        self._filter_offset = 0                       # stream number of bank 0 (bank b of a batch draws from stream b)
        self._host_rng = numpy.random.default_rng(self._seed & 0xFFFFFFFF)

        # deferred stages (small banks): predict() and update() are recorded and resample() runs all three in ONE launch
        self.lazy = True
        self._pend_predict = None
        self._pend_update = None
        self._pend_gather = None                      # ancestors [B, N] int32 of a resample whose row copy is still to come
        self._replay = None                           # noise a DeterministicGaussianSum handed out at predict() time
        self._sums_valid = False                      # the workspace holds the quarter sums of the current weights
        self._sums_slot = 0                           # ... in this slot (gse_gsukf_cycle_lanes alternates between the two)
        # a resample of a large bank whose ANCESTORS are still to be found: the next predict + update launch finds them itself
        # (gse_gsukf_cycle_lanes: one launch per cycle); holds the resample's uniform numbers [B]
        self._pend_resample = None
        self._c_r_pend = (_lib.c_d * B)()
        # (policy: while the launch leaves the chip room -- at most 512 workgroups of 256 Gaussians, two per CU; beyond, the
        #  kernel is bound by its instruction count and the two launches are faster: gse_gsukf_cycle_lanes.h)
        tiles = (n + _lib.GSE_TILE - 1) // _lib.GSE_TILE
        self._cycle_lanes = (self._model is None and B * n >= _lib.GSE_GSUKF_LANES_MIN and n > _lib.GSE_GSUKF_CYCLE_MAX and
                             n % 4 == 0 and tiles <= _lib.GSE_CYCLE_LANES_MAX_TILES and
                             4 * tiles * B <= _lib.GSE_CYCLE_LANES_MAX_WORKGROUPS)
        self._sums_ok = True                          # ... may be used at all: not once a writable view of the weights is out
        self.record_ancestors = False                 # True: every resample's ancestors [B, N] int32 are kept in
        self.ancestor_log = []                        # ancestor_log (device tensors; parity tests of whole loops)
        self.ancestor_flips = []                      # resample(ancestors=..., r=...): own ancestors that differ, per resample
        self._idx_buf = None
        self._fusable = n <= _lib.GSE_GSUKF_CYCLE_MAX
        self._fuse_pu = B * n >= _lib.GSE_GSUKF_LANES_MIN  # large banks: predict() is deferred and rides update()'s launch
        # persistent argument buffers: a call only rewrites a few doubles (keeps the host side of a launch short)
        self._c_u = (_lib.c_d * (_lib.GSE_NU * B))()
        self._c_z = (_lib.c_d * (_lib.GSE_NY * B))()
        self._c_r = (_lib.c_d * B)()

        self._threads_per_block = self._tpb = 1024
        self._blocks_per_grid = self._bpg = (n - 1) // 1024 + 1
        self._y_dummy = DevArray.wrap(torch.zeros(self._Ny, dtype=torch.float32, device=dev))

    # calls that leave the weights (and the quarter sums a gse_gsukf_predict_update_sums left of them) as they are
    _KEEPS_WEIGHTS = ("gse_gsukf_predict", "gse_gsukf_predict_dev", "gse_gsukf_gather_rows", "gse_gsukf_moments",
                      "gse_gsukf_moments26", "gse_gsukf_sigma_points", "gse_model_f", "gse_model_g_u")

    def _call(self, name, *args):
        if name not in self._KEEPS_WEIGHTS:
            self._sums_valid = False
        rc = getattr(self._lib, name)(*args)
        if rc:
            (self._model.check if self._model else _lib.check)(rc)

    # ------------------------------------------------------------------ attributes
    def _flush(self):
        """Run whatever is pending with the one-kernel-per-method calls: the row copy of a deferred resample, predict, update."""
        if self._pend_resample is not None:
            # the ancestors nobody has found yet: by the resample's own kernel, from the sums the update left
            self._pend_resample = None
            if self._idx_buf is None:
                self._idx_buf = torch.empty((self._B, self.N_particles), dtype=torch.int32, device=self._w.device)
            self._call("gse_gsukf_resample_begin_slot", self._w.data_ptr(), self._w_alt.data_ptr(), self._idx_buf.data_ptr(),
                       self._B, self.N_particles, self._c_r_pend, self._ws.data_ptr(), self._sums_slot, stream_ptr())
            self._w, self._w_alt = self._w_alt, self._w
            self._pend_gather = self._idx_buf
        if self._pend_gather is not None:
            idx, self._pend_gather = self._pend_gather, None
            self._call("gse_gsukf_gather_rows", self._means.data_ptr(), self._covs.data_ptr(), idx.data_ptr(),
                       self._means_alt.data_ptr(), self._covs_alt.data_ptr(), self._B, self.N_particles, stream_ptr())
            self._means, self._means_alt = self._means_alt, self._means
            self._covs, self._covs_alt = self._covs_alt, self._covs
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
        return DevArray.wrap(self._means[0] if self._B == 1 else self._means)

    @means.setter
    def means(self, value):
        self._flush()
        self._means.copy_(as_device(value, torch.float32).reshape(self._B, self.N_particles, self._Nx))

    @property
    def covariances(self):
        self._flush()
        return DevArray.wrap(self._covs[0] if self._B == 1 else self._covs)

    @covariances.setter
    def covariances(self, value):
        self._flush()
        self._covs.copy_(as_device(value, torch.float32).reshape(self._B, self.N_particles, self._Nx, self._Nx))

    @property
    def weights(self):
        self._flush()
        # the view is writable and may be kept: from here on the quarter sums an update launch leaves behind cannot be trusted
        # to describe the weights a later resample finds (sticky: that resample forms its tile sums itself, one launch more)
        self._sums_valid = False
        self._sums_ok = False
        return DevArray.wrap(self._w[0] if self._B == 1 else self._w)

    @weights.setter
    def weights(self, value):
        self._flush()
        self._sums_valid = False
        self._w.copy_(as_device(value, torch.float32).reshape(self._B, self.N_particles))

    def _inputs(self, u):
        """Inputs as the kernels take them: two per bank (a user model with one input gets a zero for the other)."""
        nu = self._model.nu if self._model is not None else _lib.GSE_NU
        if nu == _lib.GSE_NU:
            return self._per_bank(u, _lib.GSE_NU, self._c_u)
        a = numpy.asarray(u, dtype=numpy.float64).reshape(-1, nu)
        a = numpy.concatenate([a, numpy.zeros((a.shape[0], _lib.GSE_NU - nu))], axis=1)
        return self._per_bank(a, _lib.GSE_NU, self._c_u)

    def _io(self, u, z=None):
        """(u argument, z argument, `where` flags) of a launch: host values packed for the C ABI, or -- torch tensors
        (B, 2) float64 on the device -- their pointers, read when the kernel runs (a closed loop on the device)."""
        where = 0
        if torch.is_tensor(u):
            u_arg, where = self._device_io(u, _lib.GSE_NU).data_ptr(), where | _lib.GSE_IO_U_DEVICE
        else:
            u_arg = self._inputs(u)
        if z is None:
            z_arg = None
        elif torch.is_tensor(z):
            z_arg, where = self._device_io(z, _lib.GSE_NY).data_ptr(), where | _lib.GSE_IO_Z_DEVICE
        else:
            z_arg = self._per_bank(z, _lib.GSE_NY, self._c_z)
        if where and self._model is not None:
            raise NotImplementedError("inputs in device memory are read by the built-in model's kernels only")
        return u_arg, z_arg, where

    def _device_io(self, t, width):
        if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.numel() == self._B * width):
            raise ValueError("device inputs must be contiguous float64 CUDA tensors of shape (%d, %d)" % (self._B, width))
        return t

    def _keep(self, v):
        """What a deferred stage remembers of an argument: the tensor itself (read when the kernel runs) or a host copy."""
        if torch.is_tensor(v):
            return v
        if self._B == 1:                                  # ((2,) or (1, 2): a batch of one bank)
            return [float(x) for x in numpy.asarray(v, dtype=numpy.float64).reshape(-1)]
        return numpy.array(v, dtype=numpy.float64)

    def _per_bank(self, values, width, buf=None):
        """Host values for every bank as the C ABI takes them ([B][width] doubles; one row is broadcast to all banks), written
        into the persistent buffer ``buf`` (a launch copies them into its kernel arguments before it returns)."""
        B = self._B
        if buf is None:
            buf = (_lib.c_d * (B * width))()
        if B == 1 and len(values) == width:               # the plain class: two scalars, no array round trip
            for i in range(width):
                buf[i] = values[i]
            return buf
        a = numpy.broadcast_to(numpy.asarray(values, dtype=numpy.float64).reshape(-1, width), (B, width)).ravel()
        for i in range(B * width):
            buf[i] = a[i]
        return buf

    def _noise_rows(self):
        """Replayed noise of a DeterministicGaussianSum for every sigma point of every bank."""
        return self.state_pdf.draw((self._B, self.N_particles, self._N_sigmas)).as_subclass(torch.Tensor).contiguous()

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
        n, B = self.N_particles, self._B
        out = torch.empty((B, n, self._N_sigmas, self._Nx), dtype=torch.float32, device=self._means.device)
        self._call("gse_gsukf_sigma_points", self._means.data_ptr(), self._covs.data_ptr(), out.data_ptr(), B, n,
                  self._status.data_ptr(), stream_ptr())
        return DevArray.wrap(out[0] if B == 1 else out)

    def predict(self, u, dt):
        """Prediction step (gs_ukf.py:348-367).  With at most 64 Gaussians the step is deferred so that resample() can
        run predict + update + resample as one launch (the bank is launch-latency-bound)."""
        deferred = self.lazy and (self._fusable or self._fuse_pu) and self._pend_predict is None and self._pend_update is None
        if not deferred:
            self._flush()
        # a DeterministicGaussianSum (the parity mode) is drawn from NOW, as the reference draws inside predict(): the values
        # stay with the stage, whatever the replay buffer holds by the time a deferred launch runs
        self._replay = self._noise_rows() if isinstance(self.state_pdf, DeterministicGaussianSum) else None
        if deferred:
            self._pend_predict = (self._keep(u), float(dt))
            return
        self._predict_now(u, dt)

    def _draw_args(self):
        """(state_pdf pod, rng, noise pointer) of a predict: replayed noise for a DeterministicGaussianSum, else Philox."""
        if self._replay is not None:
            self._noise, self._replay = self._replay, None
            return None, None, self._noise.data_ptr()
        rng = _lib.Rng(seed=self._seed, step=self._step, filter_offset=self._filter_offset,
                       element_offset=self._element_offset * self._N_sigmas)
        self._step += 1
        return ctypes.byref(self.state_pdf._pod()), ctypes.byref(rng), None

    def _predict_now(self, u, dt):
        pod, rng, noise = self._draw_args()
        u_arg, _, where = self._io(u)
        if where:
            return self._call("gse_gsukf_predict_dev", self._means.data_ptr(), self._covs.data_ptr(), self._B, self.N_particles,
                              u_arg, float(dt), self._w_sigma_c, pod, rng, noise, self._status.data_ptr(), where, stream_ptr())
        self._call("gse_gsukf_predict", self._means.data_ptr(), self._covs.data_ptr(), self._B, self.N_particles,
                   u_arg, float(dt), self._w_sigma_c, pod, rng, noise, self._status.data_ptr(), stream_ptr())

    def update(self, u, z):
        """Update step: local UKF update of every Gaussian, then the global weight update (gs_ukf.py:369-407)."""
        if self._model is not None and self._pend_predict is not None and not self._same_inputs(self._pend_predict[0], u):
            self._flush()                             # a fused launch carries ONE u and a user model's g may read it
                                                      # (gs_ukf.py:369-407 evaluates g with update's own u): predict first
        if self.lazy and self._fusable and self._pend_predict is not None and self._pend_update is None:
            self._pend_update = (self._keep(u), self._keep(z))
            return
        if self.lazy and self._fuse_pu and self._pend_predict is not None and self._pend_update is None:
            (up, dt), self._pend_predict = self._pend_predict, None
            self._predict_update_now(up, dt, z)
            return
        self._flush()
        self._update_now(u, z)

    def _same_inputs(self, a, b):
        if torch.is_tensor(a) or torch.is_tensor(b):
            return torch.is_tensor(a) and torch.is_tensor(b) and a.data_ptr() == b.data_ptr()
        return numpy.array_equal(numpy.asarray(a, dtype=numpy.float64).reshape(-1), numpy.asarray(b, dtype=numpy.float64).reshape(-1))

    def _predict_update_now(self, u, dt, z):
        """predict(u, dt) and update(., z) of a large bank in one launch (gse_gsukf_predict_update): same bits as the two."""
        pod, rng, noise = self._draw_args()
        u_arg, z_arg, where = self._io(u, z)
        if self._pend_resample is not None:
            # ... and the whole resample in front of them, ancestors included (gse_gsukf_cycle_lanes): from the sums of one
            # slot of the workspace, leaving those of the new weights in the other
            self._pend_resample = None
            slot = self._sums_slot
            self._call("gse_gsukf_cycle_lanes", self._means.data_ptr(), self._covs.data_ptr(), self._w.data_ptr(),
                       self._means_alt.data_ptr(), self._covs_alt.data_ptr(), self._w_alt.data_ptr(), None, self._B,
                       self.N_particles, u_arg, float(dt), z_arg, self._c_r_pend, self._w_sigma_c, pod, rng, noise,
                       ctypes.byref(self.measurement_pdf._pod()), self._status.data_ptr(), self._ws.data_ptr(),
                       where | (_lib.GSE_IO_SUMS_SLOT1 if slot else 0), stream_ptr())
            self._means, self._means_alt = self._means_alt, self._means
            self._covs, self._covs_alt = self._covs_alt, self._covs
            self._w, self._w_alt = self._w_alt, self._w
            self._sums_valid, self._sums_slot = True, 1 - slot
            return
        # ... and, if a resample left it to this launch, its row copy: every Gaussian is read from its ancestor's row, the
        # results go to the other pair of arrays.
        idx, self._pend_gather = self._pend_gather, None
        out_m, out_c = (self._means_alt, self._covs_alt) if idx is not None else (self._means, self._covs)
        args = (self._means.data_ptr(), self._covs.data_ptr(), idx.data_ptr() if idx is not None else None, out_m.data_ptr(),
                out_c.data_ptr(), self._w.data_ptr(), self._B, self.N_particles, u_arg, float(dt), z_arg, self._w_sigma_c, pod, rng,
                noise, ctypes.byref(self.measurement_pdf._pod()), self._status.data_ptr())
        if self._model is None and (self.N_particles > _lib.GSE_SMALL_RESAMPLE_MAX or self._cycle_lanes) and self._sums_ok:
            # ... and leaves the quarter sums of the new weights in the workspace: the resample that follows needs no
            # tile-sum launch (valid until anything else touches the weights or the workspace)
            self._call("gse_gsukf_predict_update_sums", *args, self._ws.data_ptr(), where, stream_ptr())
            self._sums_valid, self._sums_slot = True, 0
        else:
            self._call("gse_gsukf_predict_update_from", *args, where, stream_ptr())
        if idx is not None:
            self._means, self._means_alt = self._means_alt, self._means
            self._covs, self._covs_alt = self._covs_alt, self._covs

    def _update_now(self, u, z):
        u_arg, z_arg, where = self._io(u, z)
        args = (self._means.data_ptr(), self._covs.data_ptr(), self._w.data_ptr(), self._B, self.N_particles, u_arg, z_arg,
                self._w_sigma_c, ctypes.byref(self.measurement_pdf._pod()), self._status.data_ptr(), self._ws.data_ptr())
        if where:
            return self._call("gse_gsukf_update_dev", *args, where, stream_ptr())
        self._call("gse_gsukf_update", *args, stream_ptr())

    def _resample_given(self, ancestors, r=None):
        """resample() with the ancestors GIVEN ((N,) or (B, N) integers): the replay of a recorded run -- Gaussian j copies the
        row of Gaussian ancestors[j] (gse_gsukf_gather_rows), weights 1 / N.  Runs at once.  With ``r`` the filter's OWN
        resample runs first and the number of ancestors in which it differs from the given ones is appended to
        ``ancestor_flips`` (per bank, device tensors) before the given ones are copied."""
        self._flush()
        n, B = self.N_particles, self._B
        idx = as_device(ancestors, torch.int32).reshape(B, n).clamp_(0, n - 1).contiguous()
        src_m, src_c, dst_m, dst_c = self._means, self._covs, self._means_alt, self._covs_alt
        swap = True
        if r is not None:
            keep, self.record_ancestors = self.record_ancestors, False
            own = self.resample(r=r, return_index=True).as_subclass(torch.Tensor).reshape(B, n)
            self.record_ancestors = keep
            self.ancestor_flips.append((own != idx).sum(dim=1))
            if self._pend_gather is not None:          # (large banks: the rows have not moved yet -- the given ancestors instead)
                self._pend_gather = None
            else:                                      # the rows as they were are in the other pair of arrays now
                src_m, src_c, dst_m, dst_c = self._means_alt, self._covs_alt, self._means, self._covs
                swap = False
        self._call("gse_gsukf_gather_rows", src_m.data_ptr(), src_c.data_ptr(), idx.data_ptr(), dst_m.data_ptr(), dst_c.data_ptr(),
                   B, n, stream_ptr())
        if swap:
            self._means, self._means_alt = self._means_alt, self._means
            self._covs, self._covs_alt = self._covs_alt, self._covs
        self._w.fill_(1.0 / n)
        self._sums_valid = False
        if self.record_ancestors:
            self.ancestor_log.append(idx)

    def resample(self, r=None, return_index=False, ancestors=None):
        """Systematic resample of the Gaussians based on their weights (gs_ukf.py:409-436).  ``ancestors``: copy these instead of
        computing them (replay of a recorded run)."""
        if ancestors is not None:
            return self._resample_given(ancestors, r)
        wanted = return_index
        n, B = self.N_particles, self._B
        if B == 1:
            self._c_r[0] = self._host_rng.random() if r is None else float(numpy.asarray(r, dtype=numpy.float64).reshape(-1)[0])
            rr = self._c_r
        else:
            rr = self._per_bank(self._host_rng.random(B) if r is None else r, 1, self._c_r)
        return_index = return_index or self.record_ancestors
        idx = torch.empty((B, n), dtype=torch.int32, device=self._w.device) if return_index else None
        if self.record_ancestors:
            self.ancestor_log.append(idx)
        if self._pend_predict is not None and self._pend_update is not None:
            # the whole cycle in one launch, one workgroup per bank (gse_gsukf_cycle); in place
            (u, dt), (_, z) = self._pend_predict, self._pend_update
            self._pend_predict = self._pend_update = None
            pod, rng, noise = self._draw_args()
            u_arg, z_arg, where = self._io(u, z)
            args = (self._means.data_ptr(), self._covs.data_ptr(), self._w.data_ptr(), idx.data_ptr() if return_index else None,
                    B, n, u_arg, dt, z_arg, rr, self._w_sigma_c, pod, rng, noise, ctypes.byref(self.measurement_pdf._pod()),
                    self._status.data_ptr())
            if where:
                self._call("gse_gsukf_cycle_dev", *args, where, stream_ptr())
            else:
                self._call("gse_gsukf_cycle", *args, stream_ptr())
        elif self.lazy and self._fuse_pu and self._pend_predict is None and self._pend_update is None:
            # large banks: ancestors and weights now, the rows with the next predict + update (one launch, _predict_update_now)
            self._flush()
            if self._cycle_lanes and self._sums_valid and idx is None:
                # ... or nothing now: the update left the sums the resample needs, and the next predict + update launch
                # finds the ancestors of its own Gaussians (any accessor in between: _flush runs the resample's kernels)
                for b in range(B):
                    self._c_r_pend[b] = rr[b]
                self._pend_resample = True
                return
            if idx is None:                          # (a buffer of the filter's own unless the caller keeps the ancestors)
                if self._idx_buf is None:
                    self._idx_buf = torch.empty((B, n), dtype=torch.int32, device=self._w.device)
                idx = self._idx_buf
            if self._sums_valid:                     # (the resample replaces the weights: _call clears the flag)
                self._call("gse_gsukf_resample_begin_slot", self._w.data_ptr(), self._w_alt.data_ptr(), idx.data_ptr(), B, n, rr,
                           self._ws.data_ptr(), self._sums_slot, stream_ptr())
            else:
                self._call("gse_gsukf_resample_begin", self._w.data_ptr(), self._w_alt.data_ptr(), idx.data_ptr(), B, n, rr,
                           self._ws.data_ptr(), stream_ptr())
            self._w, self._w_alt = self._w_alt, self._w
            self._pend_gather = idx
        else:
            self._flush()
            self._call("gse_gsukf_resample", self._means.data_ptr(), self._covs.data_ptr(), self._w.data_ptr(),
                      self._means_alt.data_ptr(), self._covs_alt.data_ptr(), self._w_alt.data_ptr(),
                      idx.data_ptr() if return_index else None, B, n, rr, 0, self._ws.data_ptr(), stream_ptr())
            self._means, self._means_alt = self._means_alt, self._means
            self._covs, self._covs_alt = self._covs_alt, self._covs
            self._w, self._w_alt = self._w_alt, self._w
        if wanted:
            return DevArray.wrap(idx[0] if B == 1 else idx)

    def run_cycles(self, us, zs, dt, rs=None):
        """K cycles predict(us[k], dt) -> update(us[k], zs[k]) -> resample() enqueued by ONE call (gse_gsukf_cycles: the loop over
        the three methods that results/gsf_openloop/gsf_run_seq.py:16-134 runs from Python, restated in C++ for the class's
        three regimes): the same launches, hence the same banks, weights and ancestors as K rounds of the methods, without the
        interpreter's 20-50 us per cycle between them -- which is more than the device needs up to 2^17 Gaussians.  us, zs:
        (K, 2) host arrays ((K, B, 2) for a batch); rs: optional (K,) / (K, B) uniform numbers (default: the filter's own host
        stream, in the order resample() would draw them).  Philox noise, the built-in model."""
        n, B, K = self.N_particles, self._B, len(us)
        native = (self._model is None and not isinstance(self.state_pdf, DeterministicGaussianSum) and self.lazy and
                  not self.record_ancestors and self._sums_ok and K > 0 and
                  (self._fusable or self._cycle_lanes or (self._fuse_pu and n > _lib.GSE_SMALL_RESAMPLE_MAX)))
        if not native:
            for k in range(K):                              # (whatever gse_gsukf_cycles does not take: the methods themselves)
                self.predict(us[k], dt)
                self.update(us[k], zs[k])
                self.resample(r=None if rs is None else rs[k])
            return
        if self._pend_predict is not None or self._pend_update is not None:
            self._flush()
        regime = (_lib.GSE_GSUKF_REGIME_CYCLE if self._fusable else
                  _lib.GSE_GSUKF_REGIME_LANES if self._cycle_lanes else _lib.GSE_GSUKF_REGIME_TWO)
        if regime != _lib.GSE_GSUKF_REGIME_LANES and self._pend_resample is not None:
            self._flush()
        if regime == _lib.GSE_GSUKF_REGIME_LANES and self._pend_gather is not None:
            self._flush()                                   # (ancestors somebody asked for: copy the rows first)
        u = numpy.ascontiguousarray(numpy.broadcast_to(numpy.asarray(us, dtype=numpy.float64).reshape(K, -1, 2), (K, B, 2)))
        z = numpy.ascontiguousarray(numpy.broadcast_to(numpy.asarray(zs, dtype=numpy.float64).reshape(K, -1, 2), (K, B, 2)))
        if rs is None:
            r = numpy.array([self._host_rng.random(B) if B > 1 else [self._host_rng.random()] for _ in range(K)])
        else:
            r = numpy.broadcast_to(numpy.asarray(rs, dtype=numpy.float64).reshape(K, -1), (K, B))
        r = numpy.ascontiguousarray(r, dtype=numpy.float64)
        addr = lambda obj: ctypes.cast(ctypes.pointer(obj), ctypes.c_void_p).value
        if regime == _lib.GSE_GSUKF_REGIME_TWO and self._idx_buf is None:
            self._idx_buf = torch.empty((B, n), dtype=torch.int32, device=self._w.device)
        if self._pend_gather is not None and self._pend_gather is not self._idx_buf:
            self._flush()                                   # (the caller keeps that index array: do not write over it)
        spec = _lib.GsukfCyclesSpec()
        spec.B, spec.regime, spec.N, spec.dt = B, regime, n, float(dt)
        spec.mu[0], spec.mu[1] = self._means.data_ptr(), self._means_alt.data_ptr()
        spec.P[0], spec.P[1] = self._covs.data_ptr(), self._covs_alt.data_ptr()
        spec.w[0], spec.w[1] = self._w.data_ptr(), self._w_alt.data_ptr()
        spec.idx = self._idx_buf.data_ptr() if self._idx_buf is not None else None
        spec.workspace, spec.status = self._ws.data_ptr(), self._status.data_ptr()
        spec.w_sigma = ctypes.cast(self._w_sigma_c, ctypes.c_void_p).value
        spec.state_pdf, spec.measurement_pdf = addr(self.state_pdf._pod()), addr(self.measurement_pdf._pod())
        spec.rng = _lib.Rng(seed=self._seed, step=self._step, filter_offset=self._filter_offset,
                       element_offset=self._element_offset * self._N_sigmas)
        spec.r = r.ctypes.data
        spec.pend_resample = 1 if self._pend_resample is not None else 0
        spec.pend_gather = 1 if self._pend_gather is not None else 0
        spec.sums_valid, spec.sums_slot = (1 if self._sums_valid else 0), self._sums_slot
        for b in range(B):
            spec.r_pending[b] = self._c_r_pend[b]
        rc = self._lib.gse_gsukf_cycles(ctypes.byref(spec), K, u.ctypes.data, z.ctypes.data, stream_ptr())
        if rc:
            _lib.check(rc)
        if spec.bank_current:
            self._means, self._means_alt = self._means_alt, self._means
            self._covs, self._covs_alt = self._covs_alt, self._covs
        if spec.w_current:
            self._w, self._w_alt = self._w_alt, self._w
        self._step = int(spec.rng.step)
        self._pend_resample = True if spec.pend_resample else None
        self._pend_gather = self._idx_buf if spec.pend_gather else None
        self._sums_valid, self._sums_slot = bool(spec.sums_valid), int(spec.sums_slot)
        for b in range(B):
            self._c_r_pend[b] = spec.r_pending[b]

    # ------------------------------------------------------------------ estimates
    def _moment_sums_all(self):
        """Per bank (sum w, w @ mu, sum w P + sum w (mu - mean)(mu - mean)^T) from one device pass (synchronises)."""
        self._flush()
        self._call("gse_gsukf_moments", self._means.data_ptr(), self._covs.data_ptr(), self._w.data_ptr(), self._B,
                  self.N_particles, self._moments.data_ptr(), self._ws.data_ptr(), stream_ptr())
        mm = self._moments.cpu().numpy()
        self._check_status()
        from gpu_se_amd.filter.particle import moments_to_mean_cov
        out = []
        nx = self._Nx
        tri = nx * (nx + 1) // 2
        at = numpy.cumsum([0, 1, nx, tri, tri, nx])                     # S0 | S1 | S2 | C | pivot
        for m in mm:
            s0, mean, cov_mean = moments_to_mean_cov(m[0], m[at[1]:at[2]], m[at[2]:at[3]], m[at[4]:at[5]])
            cc = numpy.zeros((nx, nx))
            cc[numpy.triu_indices(nx)] = m[at[3]:at[4]]
            out.append((s0, mean, cc + cc.T - numpy.diag(numpy.diag(cc)) + cov_mean))
        return out

    def _moment_sums(self):
        return self._moment_sums_all()[0]

    def _moments26_into(self, out):
        """The moments in the particle filter's layout ([B, 26] device tensor; sum w P folded into the second moments): what the
        closed loop on the device records per tick and hands to its controller kernel.  No synchronisation."""
        if self._model is not None:
            raise NotImplementedError("the 26-wide moment rows are laid out for the 5-state built-in model")
        self._flush()
        self._call("gse_gsukf_moments26", self._means.data_ptr(), self._covs.data_ptr(), self._w.data_ptr(), self._B,
                   self.N_particles, out.data_ptr(), self._ws.data_ptr(), stream_ptr())

    def point_estimate(self):
        """Returns the point estimate of the filter: weights @ means (gs_ukf.py:438-440)."""
        return self._moment_sums()[1]

    def point_covariance(self):
        """Returns the maximum singular value of the filter's covariance (gs_ukf.py:442-449)."""
        return numpy.linalg.svd(self._moment_sums()[2], compute_uv=False)[0]

    # ------------------------------------------------------------------ pieces the benchmark scripts call
    def f_vectorize(self, particles, u, dt):
        """out[..., :] = f(particles[..., :], u, dt), device array fp32 (gs_ukf.py:243-258; gsf_run_seq.py:220)."""
        return rows_f(self._call, self._Nx, particles, list(self._inputs(u))[:_lib.GSE_NU], dt)

    def g_vectorize(self, particles, u, _y_dummy=None):
        """out[..., :] = g(particles[..., :], u), device array (.. x Ny) fp32 (gs_ukf.py:260-275; gsf_run_seq.py:285)."""
        return rows_g(self._call, self._Nx, particles, list(self._inputs(u))[:_lib.GSE_NU])


class BatchedGaussianSumUnscentedKalmanFilter(ParallelGaussianSumUnscentedKalmanFilter):
    """B independent Gaussian-sum filters (same model and noise; separate Gaussians, inputs, measurements and random
    streams) advanced by the same kernel launches -- the batch dimension of the C ABI.  The reference repeats its GS-UKF
    closed loops as a sequential Python loop (results/gsf_closedloop/bioreactor_performance_gsf.py:104-118); here they share
    a GPU.  Methods as the plain class with a leading batch axis: u (B, 2) or (2,), z (B, 2); means (B, N, 5); covariances
    (B, N, 5, 5); weights (B, N); point_estimate() -> (B, 5); point_covariance() -> (B,)."""

    def __init__(self, f, g, N_particles, x0, state_pdf, measurement_pdf, batch):
        super().__init__(f, g, N_particles, x0, state_pdf, measurement_pdf, _batch=batch)

    @property
    def batch(self):
        return self._B

    def estimates(self):
        """(point estimates (B, 5), covariance sizes (B,)) from ONE device pass and one synchronisation."""
        sums = self._moment_sums_all()
        mean = numpy.stack([s[1] for s in sums])
        cov = numpy.stack([s[2] for s in sums])
        return mean, numpy.linalg.svd(cov, compute_uv=False)[:, 0]

    def point_estimate(self):
        return self.estimates()[0]

    def point_covariance(self):
        return self.estimates()[1]
