"""Bootstrap particle filter on the MI355X: drop-in for the reference's filter.ParallelParticleFilter
(filter/particle.py:117-327).  Same constructor, methods and attributes; the device work goes through
the C ABI of libgse_hip.so (include/gse.h)."""
import ctypes

import numpy
import torch

import gpu_se_amd
from gpu_se_amd import _lib
from gpu_se_amd.devarray import DevArray, as_device, device, stream_ptr
from gpu_se_amd.gaussian_sum_dist import DeterministicGaussianSum
from gpu_se_amd.model import Bioreactor


def _model_library(f, g):
    """The library whose kernels carry the model (f, g): the shipped one for the bioreactor functions, otherwise the
    one gpu_se_amd.usermodel translates and compiles for the pair (the reference lets Numba JIT arbitrary Python,
    particle.py:176-208).  Returns (ctypes library, compiled model or None)."""
    if f is Bioreactor.homeostatic_DEs and g is Bioreactor.static_outputs:
        return _lib.load(), None
    from gpu_se_amd import usermodel
    model = usermodel.compile_model(f, g)
    return model.lib, model


class _KernelLauncher:
    """Callable that also accepts ``[blocks, threads]`` -- the shape of a Numba kernel launch, which the
    reference's benchmark scripts use (pf_run_seq.py:282-287).  The launch geometry is chosen by the
    library, so the subscript is accepted and ignored."""

    def __init__(self, fn):
        self._fn = fn

    def __getitem__(self, _launch_config):
        return self._fn

    def __call__(self, *args):
        return self._fn(*args)


def _parallel_resample(cumsum, sample_index, random_number, N_particles):
    """sample_index[i] = first k with cumsum[k] >= (i + random_number)/N  (particle.py:223-263)."""
    c = as_device(cumsum, torch.float64)
    if not (isinstance(sample_index, torch.Tensor) and sample_index.dtype == torch.int64 and sample_index.is_cuda):
        raise TypeError("sample_index must be an int64 device array")
    _lib.call("gse_systematic_index", c.data_ptr(), sample_index.data_ptr(), int(N_particles), float(random_number),
              stream_ptr())


class ParallelParticleFilter:
    """Particle filter class implemented to run on the GPU (MI355X).

    Parameters
    ----------
    f : callable
        The state transition function x_{k+1} += f(x_k, u_k)  -- Bioreactor.homeostatic_DEs
    g : callable
        The state observation function y_k = g(x_k, u_k)      -- Bioreactor.static_outputs
    N_particles : int
    x0 : MultivariateGaussianSum         the initial distribution
    state_pdf, measurement_pdf : MultivariateGaussianSum   state and measurement noise

    Attributes
    ----------
    particles : device array (N_particles x Nx) fp32 -- a strided view of the state-major storage
    weights : device array (N_particles) fp32
    """

    _parallel_resample = _KernelLauncher(_parallel_resample)

    def __new__(cls, f, g, *args, **kwargs):
        # a user model beyond the fast kernels' bounds (more than 5 states / 2 inputs, other than 2 outputs; round 5): the
        # dimension-generic class takes over (filter/generic.py) -- the reference's constructor takes any model Numba compiles
        if cls is ParallelParticleFilter and not (f is Bioreactor.homeostatic_DEs and g is Bioreactor.static_outputs):
            from gpu_se_amd import usermodel
            if usermodel.needs_generic_kernels(f, g):
                from gpu_se_amd.filter.generic import GenericParticleFilter
                if kwargs.get("element_offset") or kwargs.get("_batch", 1) != 1:
                    raise NotImplementedError("shards and batches run the fast kernels' models only")
                return GenericParticleFilter(f, g, *args)
        return object.__new__(cls)

    def __init__(self, f, g, N_particles, x0, state_pdf, measurement_pdf, element_offset=0, _batch=1):
        lib, self._model = _model_library(f, g)
        self._lib = lib
        self._nx = self._model.nx if self._model else _lib.GSE_NX      # states of the model (storage always has 5 rows)
        for name, pdf, want in (("x0", x0, self._nx), ("state_pdf", state_pdf, self._nx),
                                ("measurement_pdf", measurement_pdf, _lib.GSE_NY)):
            if pdf._Nx != want:
                raise ValueError("%s is %d-dimensional, the model needs %d" % (name, pdf._Nx, want))
        self.f = f
        self.g = g
        self.N_particles = int(N_particles)
        self.state_pdf = state_pdf
        self.measurement_pdf = measurement_pdf
        n = self.N_particles
        B = self._B = int(_batch)                           # independent filters sharing every launch (1 for this class)
        if not 1 <= B <= _lib.GSE_MAX_B:
            raise ValueError("batch size must be in 1..%d" % _lib.GSE_MAX_B)
        dev = device()
        self._ld = -(-n // 4) * 4
        self._x = torch.zeros((B, _lib.GSE_NX, self._ld), dtype=torch.float32, device=dev)
        self._x_alt = torch.zeros_like(self._x)
        self._w = torch.zeros((B, self._ld), dtype=torch.float32, device=dev)
        self._w_alt = torch.zeros_like(self._w)
        self._ws = torch.zeros(lib.gse_workspace_bytes(B, n), dtype=torch.uint8, device=dev)
        # The reference's weights are fp64 from the first resample on (particle.py:103, :316), so ``weights *= pdf`` survives
        # likelihoods down to 1e-308.  The fp32 weights here keep that RANGE through a launch-uniform power of two
        # (GSE_OP_WEIGHT_SCALE, include/gse.h): true weight = stored weight * 2^_wscale[b], _wscale a view of the workspace
        # the kernels maintain (0 unless a likelihood fell below 2^-40).  ``weight_scale = False``: the plain fp32 product.
        at = lib.gse_workspace_weight_scale_offset(B, n)
        self._wscale = self._ws[at:at + 8 * B].view(torch.float64)
        self.weight_scale = True
        # sticky status word per filter in the workspace (include/gse.h, GSE_WSTATUS_*): the kernels raise it when a resample
        # meets a weight sum that is +inf or NaN; read where the host synchronises anyway (_check_status)
        at = lib.gse_workspace_status_offset(B, n)
        self._wstatus = self._ws[at:at + 4 * B].view(torch.int32)
        self._moments = torch.empty((B, 26), dtype=torch.float64, device=dev)
        self._tiles_valid = False
        self._island = None                                 # [2][B] fp64 shard bookkeeping (ShardedParticleFilter)
        self._element_offset = int(element_offset)        # global index of particle 0 (sharded ensembles)
        self._seed = gpu_se_amd._new_stream_seed()
        self._step = 0
        self._host_rng = numpy.random.default_rng(self._seed & 0xFFFFFFFF)
        # persistent argument buffers: a call only rewrites a few doubles (keeps the host side of a launch short)
        self._c_u = (_lib.c_d * (_lib.GSE_NU * B))()
        self._c_z = (_lib.c_d * (_lib.GSE_NY * B))()
        self._c_r = (_lib.c_d * B)()
        self._c_rng = _lib.Rng(seed=self._seed, step=0, element_offset=self._element_offset)
        self._c_rng_ref = ctypes.byref(self._c_rng)
        self._fn_step, self._fn_resample_begin = lib.gse_pf_step, lib.gse_pf_resample_begin
        self._pend_gather = False
        self._gather_slot = None                            # slot of the scans a pending gather reads (None: cumsum array)
        self._scan_slot = None                              # slot holding scans of the CURRENT weights, if any
        self._samples_ok = False                            # ... and the weight scale's samples of the CURRENT particles beside it
        self._w_uniform = True                              # every weight is 1 / N (the constructor's, or a resample's since)
        self._scan_capable = True                           # the cycle kernel resamples from tile sums + per-thread prefixes
        self._prefix_global = (-(-n // _lib.GSE_TILE)) > _lib.GSE_SCAN_MAX_TILES    # ... beyond 2^21 from a scanned tile prefix
        # Mid-size filters (B x tiles <= 256, i.e. less than one generation of workgroups): four workgroups per tile, one
        # particle per thread (GSE_OP_CLUSTER) -- the same results, a shorter chain of dependent latencies.  The scan slots such
        # launches write are in quarter format; _slot_fmt remembers which kind of launch wrote a slot ('q' / 't').
        self.cluster = self._model is None and B * (-(-n // _lib.GSE_TILE)) <= _lib.GSE_CLUSTER_MAX_TILES
        self._slot_fmt = [None, None]
        self._pend_predict = None
        self._want_index = False
        self._last_index = None
        self.record_ancestors = False                       # True: every resample's ancestors [B, ld] int32 are kept in
        self.ancestor_log = []                              # ancestor_log (device tensors; parity tests of whole loops)
        self.ancestor_flips = []                            # resample(ancestors=..., r=...): own ancestors that differ, per resample
        self.lazy = True                                    # False: every method launches its own kernel at once
        self._moments_host = None                           # cached device moments (cleared by anything that changes x, w)
        self._dev_index = dev.index if dev.index is not None else torch.cuda.current_device()

        # particles ~ x0 (particle.py:49), drawn on the device straight into the state-major buffer
        for b in range(B):
            x0._draw_state_major(self._x[b], n, self._ld, self._element_offset, filter_offset=b)
        self._w[:, :n] = 1.0 / n                           # (particle.py:50)

        # launch geometry the reference computes and its benchmark scripts read (particle.py:160-174)
        if n >= 1024:
            self._tpb, self._bpg = 1024, (n - 1) // 1024 + 1
        else:
            self._tpb, self._bpg = 32 * ((n - 1) // 32 + 1), 1
        self._y_dummy = DevArray.wrap(torch.zeros(_lib.GSE_NY, dtype=torch.float32, device=dev))

    def _call(self, name, *args):
        rc = getattr(self._lib, name)(*args)
        if rc:
            self._check(rc)

    def _check(self, rc):
        (self._model.check if self._model else _lib.check)(rc)

    # -- per-filter host inputs -> the persistent ctypes buffers ([B][2]; one row is broadcast to all filters)
    def _fill(self, buf, values, width):
        B = self._B
        if B == 1:
            if type(values) in (list, tuple) and len(values) == width:     # (the common call: a short list of Python floats)
                for i in range(width):
                    buf[i] = values[i]
                return
            flat = numpy.asarray(values, dtype=numpy.float64).reshape(-1)
            for i in range(width):
                buf[i] = flat[i]
            return
        a = numpy.broadcast_to(numpy.asarray(values, dtype=numpy.float64).reshape(-1, width) if width > 1
                               else numpy.asarray(values, dtype=numpy.float64).reshape(-1, 1), (B, width))
        flat = a.ravel()
        for i in range(B * width):
            buf[i] = flat[i]

    def _host_u(self, u):
        """Inputs as the kernels take them: two per filter (a user model with one input gets a zero for the other)."""
        if self._B == 1 and self._model is None and type(u) in (list, tuple) and len(u) == 2:
            return (float(u[0]), float(u[1]))
        a = numpy.asarray(u, dtype=numpy.float64)
        if self._model is not None and self._model.nu < _lib.GSE_NU:
            a = a.reshape(-1, self._model.nu) if self._B > 1 else a.reshape(1, -1)
            a = numpy.concatenate([a, numpy.zeros((a.shape[0], _lib.GSE_NU - a.shape[1]))], axis=1)
        if self._B == 1:
            u0, u1 = a.reshape(-1)
            return (float(u0), float(u1))
        return numpy.array(a, dtype=numpy.float64)

    def _device_io(self, t, width):
        """A per-filter input given as a device tensor ([B][width] fp64, contiguous): passed by pointer."""
        if not (t.is_cuda and t.dtype == torch.float64 and t.is_contiguous() and t.numel() == self._B * width):
            raise ValueError("device inputs must be contiguous float64 CUDA tensors of shape (%d, %d)" % (self._B, width))
        return t

    # ------------------------------------------------------------------ deferred stages
    # resample() and predict() do not launch their particle pass at once: resample() computes the cumulative
    # weights (one small kernel) and leaves the gather pending, predict() leaves the model step pending; the next
    # update() then runs gather + predict + update as ONE fused pass (gse_pf_step).  Anything that looks at the
    # particles or weights in between first flushes what is pending with the same kernel, so every result is what
    # the eager sequence of calls would have produced.
    def _flush(self):
        if self._pend_gather or self._pend_predict is not None:
            self._launch_step(update=None)

    def _launch_step(self, update):
        """Run the pending gather / predict stages, plus an update if ``update`` = (u, z) is given."""
        injected = self._pend_predict is not None and self._pend_predict[2] is not None
        scan_ok = self._scan_capable and self.measurement_pdf._Nd == 2
        # which family of kernels: a pending gather reads its slot with the kind of launch that wrote it; otherwise the policy
        gather_q = self._pend_gather and self._gather_slot is not None and self._slot_fmt[self._gather_slot] == "q"
        if gather_q and injected:
            pend, self._pend_predict = self._pend_predict, None         # (the cluster kernels draw their noise: copy first)
            self._launch_step(update=None)
            self._pend_predict, gather_q = pend, False
        use_cluster = bool(gather_q or (self.cluster and self._model is None and scan_ok and not injected and
                                        not self._pend_gather and not self._prefix_global))
        if use_cluster:
            return self._launch_cluster(update)
        whole = lambda: (self._pend_gather and self._gather_slot is not None and self._pend_predict is not None and
                         update is not None and scan_ok and     # gather + predict + update in ONE launch, scans in and out
                         not (injected and self._prefix_global))  # (no injected-noise variant of the scanned-prefix mode)
        if injected and self._pend_gather and self._gather_slot is not None and not whole():
            # injected noise has the scan variants of the whole cycle only: otherwise run the gather on its own first
            pend, self._pend_predict = self._pend_predict, None
            self._launch_step(update=None)
            self._pend_predict = pend
        if self._model is not None and update is not None and self._pend_predict is not None and \
                not torch.is_tensor(self._pend_predict[0]) and not torch.is_tensor(update[0]) and \
                not numpy.array_equal(numpy.asarray(self._pend_predict[0], dtype=numpy.float64).reshape(-1),
                                      numpy.asarray(self._host_u(update[0]), dtype=numpy.float64).reshape(-1)):
            self._launch_step(update=None)                      # one launch carries one u: predict with its own first
        device_io = (self._pend_predict is not None and torch.is_tensor(self._pend_predict[0])) or \
            (update is not None and torch.is_tensor(update[1]))
        whole_cycle = whole()
        if device_io and self._pend_gather and (not whole_cycle or self._prefix_global):
            # inputs that live on the device are read by the single-launch cycle or by the variants without gather: in any
            # other combination copy the ancestors first
            pend, self._pend_predict = self._pend_predict, None
            self._launch_step(update=None)
            self._pend_predict = pend
        ops = _lib.GSE_OP_WEIGHT_SCALE if (self.weight_scale and (update is not None or self._pend_gather)) else 0
        n = self.N_particles
        gather = self._pend_gather
        if ops and update is not None and not gather and self._w_uniform:
            ops |= _lib.GSE_OP_UNIFORM_WEIGHTS              # (the scale's samples say enough: no pass over all particles first)
        # (predict + update without a gather is out of place too when the weights are scaled: the launch reads the scale's
        #  sample particles from x_in while other workgroups write their new states)
        swap_x = gather or (bool(ops) and update is not None and self._pend_predict is not None)
        x_in = self._x
        x_out = self._x_alt if swap_x else self._x
        w_in = self._w
        # (... and an update writes its weights to the other buffer: every workgroup scores the OLD weights of the samples)
        swap_w = gather or (bool(ops) and update is not None)
        w_out = self._w_alt if swap_w else self._w
        idx_ptr = None
        if gather:
            ops |= _lib.GSE_OP_GATHER
            if self._gather_slot is not None:
                ops |= _lib.GSE_OP_SCAN_IN | (_lib.GSE_OP_SCAN_IN_SLOT1 if self._gather_slot else 0)
                if self._samples_ok:
                    ops |= _lib.GSE_OP_SAMPLES
                if self._prefix_global:
                    ops |= _lib.GSE_OP_PREFIX_GLOBAL
            if self._want_index or self.record_ancestors:
                self._last_index = torch.empty((self._B, self._ld), dtype=torch.int32, device=self._x.device)
                idx_ptr = self._last_index.data_ptr()
                if self.record_ancestors:
                    self.ancestor_log.append(self._last_index)
        pod_state = rng_ref = noise_ptr = None
        dt = 0.0
        cu = self._c_u
        if self._pend_predict is not None:
            ops |= _lib.GSE_OP_PREDICT
            u, dt, replay = self._pend_predict
            if torch.is_tensor(u):                          # inputs that live on the device (closed loop on the device)
                ops |= _lib.GSE_OP_U_DEVICE
                cu = ctypes.cast(ctypes.c_void_p(u.data_ptr()), _lib.P_D)
            else:
                self._fill(cu, u, _lib.GSE_NU)
            if injected:
                buf = torch.zeros_like(self._x)
                buf[:, :self._nx, :n] = replay.reshape(self._B, n, self._nx).transpose(1, 2)   # (B, n, Nx) replayed values
                self._noise_buf = buf
                noise_ptr = buf.data_ptr()
            else:
                pod_state = ctypes.byref(self.state_pdf._pod())
                self._c_rng.step = self._step
                self._step += 1
                rng_ref = self._c_rng_ref
        pod_meas = None
        cz = self._c_z
        scan_out = None
        if update is not None:
            ops |= _lib.GSE_OP_UPDATE
            if self._model is not None and not torch.is_tensor(update[0]) and not (ops & _lib.GSE_OP_U_DEVICE):
                self._fill(cu, self._host_u(update[0]), _lib.GSE_NU)       # a user model's g may read u
            if torch.is_tensor(update[1]):
                ops |= _lib.GSE_OP_Z_DEVICE
                cz = ctypes.cast(ctypes.c_void_p(self._device_io(update[1], _lib.GSE_NY).data_ptr()), _lib.P_D)
            else:
                self._fill(cz, update[1], _lib.GSE_NY)
            pod_meas = ctypes.byref(self.measurement_pdf._pod())
            if scan_ok and (not injected or not gather or whole_cycle):
                # leave the tile-local scans of the new weights behind: the next resample then needs no kernel
                scan_out = 1 if (gather and self._gather_slot == 0) else 0
                ops |= _lib.GSE_OP_SCAN_OUT | (_lib.GSE_OP_SCAN_OUT_SLOT1 if scan_out else 0)
        rc = self._fn_step(ops, x_in.data_ptr(), x_out.data_ptr(), w_in.data_ptr(), w_out.data_ptr(), idx_ptr, self._ld,
                           self._B, n, cu, dt, cz, self._c_r, pod_state, rng_ref, noise_ptr, pod_meas,
                           self._island.data_ptr() if self._island is not None else None, self._ws.data_ptr(),
                           stream_ptr(self._dev_index))
        if rc:
            self._check(rc)
        if swap_x:
            self._x, self._x_alt = self._x_alt, self._x
        if swap_w:
            self._w, self._w_alt = self._w_alt, self._w
        self._pend_gather = False
        self._gather_slot = None
        self._pend_predict = None
        self._moments_host = None
        # GSE_OP_SAMPLES: this launch left the weight scale's sample particles beside the slot it filled; they describe the
        # particles until somebody else touches them (the accessors, a launch of another kind)
        self._samples_ok = bool(update is not None and scan_out is not None and (ops & _lib.GSE_OP_WEIGHT_SCALE))
        if update is not None:
            self._w_uniform = False
        elif gather:
            self._w_uniform = True
        if update is not None:
            self._tiles_valid = True
            self._scan_slot = scan_out
            if scan_out is not None:
                self._slot_fmt[scan_out] = "t"

    def _launch_cluster(self, update):
        """_launch_step with four workgroups per tile (GSE_OP_CLUSTER): any combination of the pending gather (from a
        quarter-format slot), the pending predict (Philox) and an update, host or device-resident inputs, one launch."""
        n, gather = self.N_particles, self._pend_gather
        ops = _lib.GSE_OP_CLUSTER | (_lib.GSE_OP_WEIGHT_SCALE if (self.weight_scale and (update is not None or gather)) else 0)
        # (predict + update without a gather: out of place when the weights are scaled -- the scale's anchor is read from x_in)
        swap_x = gather or (bool(ops & _lib.GSE_OP_WEIGHT_SCALE) and update is not None and self._pend_predict is not None)
        x_out = self._x_alt if swap_x else self._x
        swap_w = gather or (bool(ops & _lib.GSE_OP_WEIGHT_SCALE) and update is not None)      # (see _launch_step)
        w_out = self._w_alt if swap_w else self._w
        if (ops & _lib.GSE_OP_WEIGHT_SCALE) and update is not None and not gather and self._w_uniform:
            ops |= _lib.GSE_OP_UNIFORM_WEIGHTS
        idx_ptr = None
        if gather:
            ops |= _lib.GSE_OP_GATHER | _lib.GSE_OP_SCAN_IN | (_lib.GSE_OP_SCAN_IN_SLOT1 if self._gather_slot else 0)
            if self._want_index or self.record_ancestors:
                self._last_index = torch.empty((self._B, self._ld), dtype=torch.int32, device=self._x.device)
                idx_ptr = self._last_index.data_ptr()
                if self.record_ancestors:
                    self.ancestor_log.append(self._last_index)
        pod_state = rng_ref = pod_meas = None
        dt, cu, cz = 0.0, self._c_u, self._c_z
        if self._pend_predict is not None:
            ops |= _lib.GSE_OP_PREDICT
            u, dt, _ = self._pend_predict
            if torch.is_tensor(u):
                ops |= _lib.GSE_OP_U_DEVICE
                cu = ctypes.cast(ctypes.c_void_p(u.data_ptr()), _lib.P_D)
            else:
                self._fill(cu, u, _lib.GSE_NU)
            pod_state = ctypes.byref(self.state_pdf._pod())
            self._c_rng.step = self._step
            self._step += 1
            rng_ref = self._c_rng_ref
        scan_out = None
        if update is not None:
            ops |= _lib.GSE_OP_UPDATE
            if torch.is_tensor(update[1]):
                ops |= _lib.GSE_OP_Z_DEVICE
                cz = ctypes.cast(ctypes.c_void_p(self._device_io(update[1], _lib.GSE_NY).data_ptr()), _lib.P_D)
            else:
                self._fill(cz, update[1], _lib.GSE_NY)
            if self._pend_predict is None and torch.is_tensor(update[0]):        # (nothing reads u here; keep the flags consistent)
                pass
            pod_meas = ctypes.byref(self.measurement_pdf._pod())
            scan_out = 1 if (gather and self._gather_slot == 0) else 0
            ops |= _lib.GSE_OP_SCAN_OUT | (_lib.GSE_OP_SCAN_OUT_SLOT1 if scan_out else 0)
        rc = self._fn_step(ops, self._x.data_ptr(), x_out.data_ptr(), self._w.data_ptr(), w_out.data_ptr(), idx_ptr, self._ld,
                           self._B, n, cu, dt, cz, self._c_r, pod_state, rng_ref, None, pod_meas,
                           self._island.data_ptr() if self._island is not None else None, self._ws.data_ptr(),
                           stream_ptr(self._dev_index))
        if rc:
            self._check(rc)
        if swap_x:
            self._x, self._x_alt = self._x_alt, self._x
        if swap_w:
            self._w, self._w_alt = self._w_alt, self._w
        self._pend_gather = False
        self._gather_slot = None
        self._pend_predict = None
        self._moments_host = None
        self._samples_ok = False
        if update is not None:
            self._w_uniform = False
        elif gather:
            self._w_uniform = True
        if update is not None:
            self._tiles_valid = False                           # (the workspace's tile sums are not written by these launches)
            self._scan_slot = scan_out
            self._slot_fmt[scan_out] = "q"

    def _slot_arg(self, slot):
        """A scan slot as gse_pf_moments_resampled takes it: | GSE_SLOT_QUARTERS when a cluster launch wrote it."""
        return int(slot) | (_lib.GSE_SLOT_QUARTERS if self._slot_fmt[slot] == "q" else 0)

    # ------------------------------------------------------------------ attributes
    def _handed_out(self):
        """``particles`` / ``weights`` return writable views of the internal buffers (the reference's attributes are plain
        arrays that callers mutate in place: ``pf.weights[:] = ...``).  Whatever was derived from the old contents -- the
        cached moments, the tile sums and tile prefixes a deferred resample would use -- is therefore dropped the moment a
        view leaves: the next estimate / resample recomputes it (one small kernel) instead of using stale values."""
        self._moments_host = None
        self._samples_ok = False
        self._tiles_valid = False
        self._scan_slot = None
        self._w_uniform = False

    @property
    def particles(self):
        self._flush()
        self._moments_host = None
        self._samples_ok = False
        v = self._x[:, :self._nx, :self.N_particles].transpose(1, 2)   # (B, N, Nx) strided view
        return DevArray.wrap(v[0] if self._B == 1 else v)

    @particles.setter
    def particles(self, value):
        v = as_device(value, torch.float32).reshape(self._B, self.N_particles, self._nx)
        self._flush()
        self._x[:, :self._nx, :self.N_particles] = v.transpose(1, 2)
        self._moments_host = None
        self._samples_ok = False

    def _check_status(self, status=None):
        """Turn a raised status word into an exception (and clear it).  ``status``: the words as a host array if the caller has
        copied them already (with its own results: one synchronisation), else they are fetched here."""
        if status is None:
            status = self._wstatus.cpu().numpy()
        status = numpy.asarray(status).astype(numpy.int64).reshape(-1)
        if numpy.any(status & _lib.GSE_WSTATUS_SUM_NOT_FINITE):
            self._wstatus.zero_()
            bad = numpy.nonzero(status & _lib.GSE_WSTATUS_SUM_NOT_FINITE)[0].tolist()
            raise FloatingPointError(
                "a resample met weights whose sum is not finite (filter%s %s)%s: the ensemble has collapsed onto particle 0" % (
                    "s" if len(bad) > 1 else "", bad,
                    " -- the stored fp32 weights overflowed the launch's power-of-two scale: a particle weighed more than 2^127 "
                    "times the 64 sample particles the scale is anchored on" if self.weight_scale else ""))

    @property
    def weight_log2_scale(self):
        """Binary exponent the stored weights are to be multiplied by, per filter (host array (B,); synchronises)."""
        self._flush()
        return self._wscale.cpu().numpy().copy() if self.weight_scale else numpy.zeros(self._B)

    def _scaled_weights(self):
        """(stored fp32 weights (B, N) as a host array, their binary exponents (B,)) without folding the two."""
        self._flush()
        return self._w[:, :self.N_particles].cpu().numpy(), self.weight_log2_scale

    @property
    def weights(self):
        """The weights as a device array (N,) / (B, N).  While the stored fp32 numbers ARE the weights (binary exponent 0:
        always, unless a likelihood fell below 2^-40 since the last resample) this is a writable view of them, like the
        reference's attribute.  With an exponent engaged the true weights are below fp32's range -- what the reference holds in
        fp64 (filter/particle.py:103, :316) -- and an fp64 COPY of the true weights is handed out instead; the filter's state is
        not touched (round 4 folded the exponent into the fp32 numbers here, which flushed every weight to zero and sent
        the next resample to particle 0).  Synchronises."""
        self._flush()
        n = self.N_particles
        if self.weight_scale and bool((self._wscale != 0).any().item()):
            v = self._w[:, :n].double() * torch.exp2(self._wscale)[:, None]
            return DevArray.wrap(v[0] if self._B == 1 else v)
        self._handed_out()
        v = self._w[:, :n]
        return DevArray.wrap(v[0] if self._B == 1 else v)

    @weights.setter
    def weights(self, value):
        # the benchmark scripts assign a host array here (pf_run_seq.py:124-125)
        v = as_device(value, torch.float32).reshape(self._B, self.N_particles)
        self._flush()
        self._w[:, :self.N_particles] = v
        if self.weight_scale:
            self._wscale.zero_()
        self._tiles_valid = False
        self._scan_slot = None
        self._moments_host = None
        self._samples_ok = False
        self._w_uniform = False

    # ------------------------------------------------------------------ the cycle
    def predict(self, u, dt):
        """Performs a prediction step on the particles: x_i += f(x_i, u, dt) then x += state noise
        (particle.py:265-277).  u: (N_inputs) host array-like; dt: float.  Deferred -- see above."""
        if self._pend_predict is not None:
            self._flush()                                   # two predicts in a row: run the first one now
        if torch.is_tensor(u):
            u = self._device_io(u, _lib.GSE_NU)             # read by the kernel when it runs: do not overwrite before
        else:
            u = self._host_u(u)
        # a DeterministicGaussianSum (the parity mode) is drawn from NOW, as the reference draws inside predict(): the values
        # stay with the deferred stage, whatever the replay buffer holds by the time it runs
        replay = (self.state_pdf.draw((self._B, self.N_particles)).as_subclass(torch.Tensor)
                  if isinstance(self.state_pdf, DeterministicGaussianSum) else None)
        self._pend_predict = (u, float(dt), replay)
        if not self.lazy:
            self._flush()

    def update(self, u, z):
        """Performs an update step on the particles: w_i *= measurement_pdf(z - g(x_i, u))
        (particle.py:279-294).  u, z: host array-likes.  Runs the pending resample-gather and predict in the same
        pass."""
        self._launch_step(update=(u, z))

    def _resample_given(self, ancestors, r=None):
        """resample() with the ancestors GIVEN ((N,) or (B, N) integers): the replay of a recorded run (the reference's own
        choices, tests/golden/closed_loop.npz) -- offspring j copies particle ancestors[j], weights 1 / N.  Runs at once.
        With ``r`` the filter's OWN resample runs first (cumulative weights, ancestors: everything a normal resample does)
        and the number of ancestors in which it differs from the given ones is appended to ``ancestor_flips`` (per filter,
        device tensors) -- the replay then checks the weights too -- before the given ones are copied."""
        self._flush()
        n = self.N_particles
        idx = torch.zeros((self._B, self._ld), dtype=torch.int32, device=self._x.device)
        idx[:, :n] = as_device(ancestors, torch.int32).reshape(self._B, n)
        if r is not None:
            keep, self.record_ancestors = self.record_ancestors, False
            own = self.resample(r=r, return_index=True).as_subclass(torch.Tensor).reshape(self._B, n)
            self.record_ancestors = keep
            self.ancestor_flips.append((own != idx[:, :n]).sum(dim=1))
            src, dst, wdst = self._x_alt, self._x, self._w          # the ensemble as it was; over the filter's own offspring
        else:
            src, dst, wdst = self._x, self._x_alt, self._w_alt
        self._call("gse_pf_gather_index", src.data_ptr(), idx.data_ptr(), dst.data_ptr(), wdst.data_ptr(),
                   self._ld, self._B, n, stream_ptr(self._dev_index))
        if r is None:
            self._x, self._x_alt = self._x_alt, self._x
            self._w, self._w_alt = self._w_alt, self._w
        if self.weight_scale:
            self._wscale.zero_()
        self._tiles_valid, self._scan_slot, self._moments_host, self._samples_ok = False, None, None, False
        self._w_uniform = True                              # (index_gather_kernel wrote 1 / N everywhere)
        self._last_index = idx
        if self.record_ancestors:
            self.ancestor_log.append(idx)

    def resample(self, r=None, return_index=False, ancestors=None):
        """Systematic resample of the particles based on their weights (particle.py:296-316).  The one
        uniform number comes from a host generator (no device round trip); ``r`` overrides it (tests).  The
        cumulative weights are computed now; the copy of the ancestors is deferred -- see above.  ``ancestors``: copy these
        instead of computing them (replay of a recorded run)."""
        if ancestors is not None:
            return self._resample_given(ancestors, r)
        self._flush()
        if self._B == 1:
            self._c_r[0] = self._host_rng.random() if r is None else float(numpy.asarray(r, dtype=numpy.float64).reshape(-1)[0])
        else:
            self._fill(self._c_r, self._host_rng.random(self._B) if r is None else r, 1)
        if self._scan_slot is not None:
            # the preceding update left the tile sums and per-thread tile prefixes of these weights in the workspace: nothing
            # to launch now (lazy = False: the gather that follows at once resamples from them, one launch instead of two)
            self._gather_slot = self._scan_slot
            if self._prefix_global:
                # (beyond 2^21 particles: the tile prefix once, by one workgroup per filter, instead of in every workgroup)
                flags = _lib.GSE_RESAMPLE_PREFIX_FROM_SLOT | (_lib.GSE_RESAMPLE_SLOT1 if self._scan_slot else 0)
                rc = self._fn_resample_begin(self._w.data_ptr(), self._ld, self._B, self.N_particles, flags, None,
                                             self._ws.data_ptr(), stream_ptr(self._dev_index))
                if rc:
                    self._check(rc)
        else:
            flags = (_lib.GSE_RESAMPLE_TILES_VALID if self._tiles_valid else 0) | \
                (_lib.GSE_RESAMPLE_WEIGHT_SCALE if self.weight_scale else 0)
            rc = self._fn_resample_begin(self._w.data_ptr(), self._ld, self._B, self.N_particles, flags,
                                         self._island.data_ptr() if self._island is not None else None,
                                         self._ws.data_ptr(), stream_ptr(self._dev_index))
            if rc:
                self._check(rc)
            self._gather_slot = None
        self._tiles_valid = False
        self._scan_slot = None
        self._pend_gather = True
        self._want_index = bool(return_index)
        if return_index or not self.lazy:
            self._flush()
        if return_index:
            v = self._last_index[:, :self.N_particles]
            return DevArray.wrap(v[0] if self._B == 1 else v)

    def run_cycles(self, us, zs, dt, rs=None):
        """K cycles predict(us[k], dt) -> update(us[k], zs[k]) -> resample() enqueued by ONE call (gse_pf_cycles: the loop over
        the three methods that results/pf_openloop/pf_run_seq.py:15-130 runs from Python, restated in C++): the same launches,
        hence the same particles, weights and ancestors as K rounds of the methods, without the interpreter's 11 us per cycle
        between them -- which is more than the device needs for a filter of up to 2^18 particles.  us, zs: (K, 2) host arrays
        ((K, B, 2) for a batch); rs: optional (K,) / (K, B) uniform numbers (default: the filter's own host stream, in the order
        resample() would draw them).  Philox noise, the built-in model, a 2-component measurement mixture, N <= 2^21."""
        if self._model is not None or isinstance(self.state_pdf, DeterministicGaussianSum) or self.measurement_pdf._Nd != 2 \
                or self.N_particles > (1 << 21) or not self.lazy or self.record_ancestors:
            for k in range(len(us)):                        # (whatever gse_pf_cycles does not take: the methods themselves)
                self.predict(us[k], dt)
                self.update(us[k], zs[k])
                self.resample(r=None if rs is None else rs[k])
            return
        B, K = self._B, len(us)
        if self._pend_predict is not None or (self._pend_gather and self._gather_slot is None):
            self._flush()
        u = numpy.ascontiguousarray(numpy.broadcast_to(numpy.asarray(us, dtype=numpy.float64).reshape(K, -1, 2), (K, B, 2)))
        z = numpy.ascontiguousarray(numpy.broadcast_to(numpy.asarray(zs, dtype=numpy.float64).reshape(K, -1, 2), (K, B, 2)))
        if rs is None:
            r = numpy.array([self._host_rng.random(B) if B > 1 else [self._host_rng.random()] for _ in range(K)])
        else:
            r = numpy.broadcast_to(numpy.asarray(rs, dtype=numpy.float64).reshape(K, -1), (K, B))
        r = numpy.ascontiguousarray(r, dtype=numpy.float64)
        # which kernel family: a pending copy reads its slot with the kind of launch that wrote it, otherwise the policy
        if self._pend_gather:
            clustered = self._slot_fmt[self._gather_slot] == "q"
        else:
            clustered = bool(self.cluster and not self._prefix_global)
        addr = lambda obj: ctypes.cast(ctypes.pointer(obj), ctypes.c_void_p).value
        spec = _lib.PfLoopSpec()
        spec.B, spec.ticks, spec.N, spec.ld, spec.dt = B, K, self.N_particles, self._ld, float(dt)
        spec.flags = (_lib.GSE_PF_LOOP_WEIGHT_SCALE if self.weight_scale else 0) | (_lib.GSE_PF_LOOP_CLUSTER if clustered else 0)
        spec.x[0], spec.x[1], spec.w[0], spec.w[1] = (t.data_ptr() for t in (self._x, self._x_alt, self._w, self._w_alt))
        spec.workspace = self._ws.data_ptr()
        spec.state_pdf, spec.measurement_pdf = addr(self.state_pdf._pod()), addr(self.measurement_pdf._pod())
        spec.rng = _lib.Rng(seed=self._seed, step=self._step, element_offset=self._element_offset)
        spec.r = r.ctypes.data
        spec.pend_gather = 1 if self._pend_gather else 0
        spec.gather_slot = self._gather_slot if self._pend_gather else -1
        spec.samples_ok = 1 if (self._pend_gather and self._samples_ok) else 0
        spec.w_uniform = 1 if self._w_uniform else 0
        spec.island = self._island.data_ptr() if self._island is not None else None     # (a shard: distributed.py)
        for b in range(B):
            spec.r_pending[b] = self._c_r[b]
        rc = self._lib.gse_pf_cycles(ctypes.byref(spec), K, u.ctypes.data, z.ctypes.data, stream_ptr(self._dev_index))
        if rc:
            self._check(rc)
        if spec.x_current:
            self._x, self._x_alt = self._x_alt, self._x
        if spec.w_current:
            self._w, self._w_alt = self._w_alt, self._w
        self._step = int(spec.rng.step)
        self._pend_gather, self._gather_slot, self._scan_slot = True, int(spec.gather_slot), None
        self._slot_fmt[self._gather_slot] = "q" if clustered else "t"
        self._tiles_valid, self._moments_host, self._want_index = False, None, False
        self._samples_ok = bool(spec.samples_ok)
        self._w_uniform = bool(spec.w_uniform)
        for b in range(B):
            self._c_r[b] = spec.r_pending[b]

    # ------------------------------------------------------------------ estimates
    def _raw_moments_arrays(self):
        """(S0 (B,), S1 (B, 5), S2 (B, 5, 5), pivot (B, 5)) with d = x - pivot, from the device (synchronises)."""
        self._flush()
        if self._moments_host is None:                      # nothing changed since the last estimate: reuse it
            self._call("gse_pf_moments", self._x.data_ptr(), self._w.data_ptr(), self._ld, self._B, self.N_particles,
                       self._moments.data_ptr(), self._ws.data_ptr(), stream_ptr())
            host = self._moments.cpu().numpy()
            self._check_status()
            self._moments_host = host
            if self.weight_scale:
                # the kernel sums the STORED weights; the reference's quantities are un-normalised sums of the true ones
                k = self._wscale.cpu().numpy()
                if numpy.any(numpy.isposinf(self._moments_host[:, 0])):
                    # (not expected: after a resample the scale's samples are within 2 log2 N binades of the heaviest particle,
                    #  and updates without one take their anchor from all particles -- but never silently)
                    raise FloatingPointError("the stored weights overflowed the launch's power-of-two scale (weight sum = inf)")
                if numpy.any(k != 0):
                    self._moments_host[:, :21] = numpy.ldexp(self._moments_host[:, :21], k.astype(numpy.int64)[:, None])
        m = self._moments_host
        iu = numpy.triu_indices(_lib.GSE_NX)
        s2 = numpy.zeros((self._B, _lib.GSE_NX, _lib.GSE_NX))
        s2[:, iu[0], iu[1]] = m[:, 6:21]
        s2[:, iu[1], iu[0]] = m[:, 6:21]
        return m[:, 0].copy(), m[:, 1:6].copy(), s2, m[:, 21:26].copy()

    def _resampled_moments_into(self, out):
        """Moments of the ensemble the pending resample WILL produce, written to the device tensor ``out`` [B, 26] without
        copying a particle (gse_pf_moments_resampled) -- for loops on the device that need the estimate between
        resample() and the next predict() while the resample's copy rides the next cycle's single launch.  Returns False
        (nothing done) unless a resample is pending on stored tile prefixes and nothing else is."""
        if not (self._pend_gather and self._gather_slot is not None and self._pend_predict is None) or \
                self.N_particles > (1 << 21):
            return False
        self._call("gse_pf_moments_resampled", self._x.data_ptr(), self._w.data_ptr(), self._ld, self._B, self.N_particles,
                   self._c_r, self._slot_arg(self._gather_slot), out.data_ptr(), self._ws.data_ptr(), stream_ptr(self._dev_index))
        return True

    def _resampled_moments_tail(self, out, tail):
        """_resampled_moments_into followed, in the same launch as the reduction that finishes the moments, by the controller
        call and the plant step ``tail`` (a _lib.LoopTail) describes: gse_pf_moments_resampled_tail, two launches for what
        were five.  Returns False (nothing done) where _resampled_moments_into would."""
        import ctypes
        if not (self._pend_gather and self._gather_slot is not None and self._pend_predict is None) or \
                self.N_particles > (1 << 21):
            return False
        self._call("gse_pf_moments_resampled_tail", self._x.data_ptr(), self._w.data_ptr(), self._ld, self._B, self.N_particles,
                   self._c_r, self._slot_arg(self._gather_slot), out.data_ptr(), self._ws.data_ptr(), ctypes.byref(tail),
                   stream_ptr(self._dev_index))
        return True

    def _moments26_into(self, out):
        """gse_pf_moments of the current ensemble into the device tensor ``out`` [B, 26]; no synchronisation (closed loop
        on the device)."""
        self._flush()
        self._call("gse_pf_moments", self._x.data_ptr(), self._w.data_ptr(), self._ld, self._B, self.N_particles,
                   out.data_ptr(), self._ws.data_ptr(), stream_ptr(self._dev_index))

    def _raw_moments_all(self):
        s0, s1, s2, pv = self._raw_moments_arrays()
        return [(float(s0[b]), s1[b], s2[b], pv[b]) for b in range(self._B)]

    def _raw_moments(self):
        return self._raw_moments_all()[0]

    def _moment_sums_all(self):
        """Per filter (S0, w@x, sum w (x - w@x)(x - w@x)^T) -- the reference's un-normalised quantities."""
        k = self._nx
        return [moments_to_mean_cov(s0, s1[:k], s2[:k, :k][numpy.triu_indices(k)], pivot[:k])
                for s0, s1, s2, pivot in self._raw_moments_all()]

    def _moment_sums(self):
        return self._moment_sums_all()[0]

    def point_estimate(self):
        """Returns the point estimate of the filter: weights @ particles, host array (5,) (particle.py:318-320)."""
        return self._moment_sums()[1]

    def point_covariance(self):
        """Returns the maximum singular value of the filter's covariance (particle.py:322-327)."""
        cov = self._moment_sums()[2]
        return numpy.linalg.svd(cov, compute_uv=False)[0]

    # ------------------------------------------------------------------ pieces the benchmark scripts call
    def _row_u(self, u):
        return self._host_u(u) if self._B == 1 else numpy.asarray(u, dtype=numpy.float64).reshape(-1)[:2]

    def f_vectorize(self, particles, u, dt):
        """out[i] = f(particles[i], u, dt) as a device array (M x Nx) fp32 (particle.py:176-191)."""
        return rows_f(self._call, self._nx, particles, self._row_u(u), dt)

    def g_vectorize(self, particles, u, _y_dummy=None):
        """out[i] = g(particles[i], u) as a device array (M x Ny) fp32 (particle.py:193-208)."""
        return rows_g(self._call, self._nx, particles, self._row_u(u))


class BatchedParticleFilter(ParallelParticleFilter):
    """B independent particle filters (same model and noise, separate particles, inputs, measurements and random
    streams) advanced by the same kernel launches -- the batch dimension of the C ABI.  This is how independent
    closed-loop rollouts (the Monte-Carlo repetitions of results/pf_closedloop/bioreactor_performance_pf.py:104-118)
    share a GPU.  Methods as ParallelParticleFilter with a leading batch axis: u (B, 2) or (2,), z (B, 2);
    particles (B, N, 5); weights (B, N); point_estimate() -> (B, 5); point_covariance() -> (B,)."""

    def __init__(self, f, g, N_particles, x0, state_pdf, measurement_pdf, batch):
        super().__init__(f, g, N_particles, x0, state_pdf, measurement_pdf, _batch=batch)

    @property
    def batch(self):
        return self._B

    def estimates(self):
        """(point estimates (B, 5), covariance sizes (B,)) from ONE device pass and one synchronisation; the pivot
        shift of moments_to_mean_cov, vectorised over the batch."""
        s0, s1, s2, pv = self._raw_moments_arrays()
        mean = s1 + pv * s0[:, None]
        e = mean - pv
        cov = (s2 - s1[:, :, None] * e[:, None, :] - e[:, :, None] * s1[:, None, :]
               + s0[:, None, None] * e[:, :, None] * e[:, None, :])
        return mean, numpy.linalg.svd(cov, compute_uv=False)[:, 0]

    def point_estimate(self):
        return self.estimates()[0]

    def point_covariance(self):
        return self.estimates()[1]


def _rows5(particles, nx):
    """(..., nx) array-like -> contiguous (M, 5) fp32 device rows (unused states zero) and the leading shape."""
    p = as_device(particles, torch.float32)
    lead = p.shape[:-1]
    p2 = p.reshape(-1, nx)
    if nx < _lib.GSE_NX:
        p2 = torch.cat([p2, torch.zeros((p2.shape[0], _lib.GSE_NX - nx), dtype=torch.float32, device=p2.device)], dim=1)
    return p2.contiguous(), lead


def rows_f(call, nx, particles, uu, dt):
    """f applied to every row through gse_model_f of the library behind ``call`` (shared by both filter classes)."""
    p2, lead = _rows5(particles, nx)
    out = torch.empty_like(p2)
    call("gse_model_f", p2.data_ptr(), out.data_ptr(), p2.shape[0], _lib.doubles(uu, _lib.GSE_NU), float(dt), stream_ptr())
    return DevArray.wrap(out[:, :nx].reshape(lead + (nx,)))


def rows_g(call, nx, particles, uu):
    """g applied to every row through gse_model_g_u."""
    p2, lead = _rows5(particles, nx)
    out = torch.empty((p2.shape[0], _lib.GSE_NY), dtype=torch.float32, device=p2.device)
    call("gse_model_g_u", p2.data_ptr(), out.data_ptr(), p2.shape[0], _lib.doubles(uu, _lib.GSE_NU), stream_ptr())
    return DevArray.wrap(out.reshape(lead + (_lib.GSE_NY,)))


def moments_to_mean_cov(s0, s1, s2_upper, pivot):
    """Undo the pivot shift of gse_pf_moments: with d = x - p the kernel returns S0 = sum w, S1 = sum w d,
    S2 = sum w d d^T.  The reference's quantities (weights are NOT normalised there, particle.py:318-327):
        mean = w @ x                      = S1 + p S0
        cov  = sum w (x - mean)(x - mean)^T = S2 - S1 e^T - e S1^T + S0 e e^T,   e = mean - p."""
    s1 = numpy.asarray(s1, dtype=numpy.float64)
    pivot = numpy.asarray(pivot, dtype=numpy.float64)
    nx = len(s1)
    s2 = numpy.zeros((nx, nx))
    s2[numpy.triu_indices(nx)] = s2_upper
    s2 = s2 + s2.T - numpy.diag(numpy.diag(s2))
    mean = s1 + pivot * s0
    e = mean - pivot
    cov = s2 - numpy.outer(s1, e) - numpy.outer(e, s1) + s0 * numpy.outer(e, e)
    return float(s0), mean, cov
