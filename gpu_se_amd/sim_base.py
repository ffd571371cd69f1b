"""Entry points of the reference's sim_base.py: get_parts, get_noise, get_random_io, performance and the closed-loop
drivers: plant + estimator + MPC, one rollout (Simulation) or B rollouts (BatchedSimulation).
"""
import numpy

from gpu_se_amd import _lib, controller
from gpu_se_amd import filter as _filter
from gpu_se_amd import gaussian_sum_dist, model


def get_parts(dt_control=1, N_particles=2 * 15, gpu=True, pf=True):
    """Returns the parts needed for a closed-loop simulation (sim_base.py:10-114).

    Returns
    -------
    bioreactor : model.Bioreactor   the nonlinear plant at its steady state
    lin_model : model.LinearModel   linearisation about u = [0.04, 0.1], states / outputs (Cg, Cfa), both inputs
    K : controller.MPC              P = 300 // dt_control, M = 200 // dt_control, Q = diag(0.1, 1), R = I, u >= 0
    pf : filter.ParallelParticleFilter or filter.ParallelGaussianSumUnscentedKalmanFilter
    """
    if not gpu:
        raise NotImplementedError("gpu_se_amd provides the device classes only (gpu=True); the NumPy classes of "
                                  "the reference are restated under oracle/ as test infrastructure")
    shape = numpy.array([260 / 180, 640 / 24.6, 1000 / 116, 0, 0])        # Ng, Nx, Nfa, Ne, Nh
    bioreactor = model.Bioreactor(X0=model.Bioreactor.find_SS(numpy.array([0.06, 0.2]), shape), high_N=False)
    lin_model, K = get_controller(bioreactor, dt_control)

    my_filter = _filter.ParallelParticleFilter if pf else _filter.ParallelGaussianSumUnscentedKalmanFilter
    state_pdf, measurement_pdf = get_noise()
    x0, _ = get_noise()
    x0.means += numpy.asarray(bioreactor.X[numpy.newaxis, :], dtype=numpy.float32)     # prior centred on X_ss (:104)
    estimator = my_filter(
        f=bioreactor.homeostatic_DEs,
        g=bioreactor.static_outputs,
        N_particles=N_particles,
        x0=x0,
        state_pdf=state_pdf,
        measurement_pdf=measurement_pdf)
    return bioreactor, lin_model, K, estimator


def get_controller(bioreactor, dt_control=1):
    """The linear model and the MPC of get_parts (sim_base.py:55-86)."""
    shape = numpy.array([260 / 180, 640 / 24.6, 1000 / 116, 0, 0])
    lin_model = model.LinearModel.create_LinearModel(
        bioreactor, x_bar=model.Bioreactor.find_SS(numpy.array([0.04, 0.1]), shape), u_bar=numpy.array([0.04, 0.1]),
        T=dt_control)
    lin_model.select_subset(states=[0, 2], inputs=[0, 1], outputs=[0, 2])           # Cg, Cfa; Fg_in, Fm_in
    K = controller.MPC(
        P=int(300 // dt_control), M=max(int(200 // dt_control), 1), Q=numpy.diag([0.1, 1]), R=numpy.diag([1, 1]),
        lin_model=lin_model, ysp=lin_model.yn2d(numpy.array([280, 850]), subselect=False),
        u_bounds=[numpy.array([0, numpy.inf]) - lin_model.u_bar[0], numpy.array([0, numpy.inf]) - lin_model.u_bar[1]])
    return lin_model, K


def get_noise(lib=None, deterministic=False):
    """Returns state and measurement noise (sim_base.py:117-161).  ``lib`` is accepted for signature
    compatibility; the distributions always evaluate on the HIP device."""
    distribution = (gaussian_sum_dist.DeterministicGaussianSum if deterministic
                    else gaussian_sum_dist.MultivariateGaussianSum)
    state_pdf = distribution(
        means=numpy.zeros(shape=(2, 5)),
        covariances=numpy.array([numpy.diag([1e-4, 1e-7, 1e-3, 1e-3, 1e-7]),
                                 numpy.diag([1e-3, 1e-6, 1e-2, 1e-2, 1e-6])]),
        weights=numpy.array([0.75, 0.25]),
        library=lib)
    measurement_pdf = distribution(
        means=numpy.array([[1e-1, 0], [0, -1e-1]]),
        covariances=numpy.array([[[6e-2, 0], [0, 8e-2]],
                                 [[500, 100], [100, 700]]]),
        weights=numpy.array([0.85, 0.15]),
        library=lib)
    return state_pdf, measurement_pdf


def get_random_io(rng=None):
    """Random system input and output for benchmark runs (sim_base.py:188-204).  NOTE: the
    reference's y is in g/L while g() returns mg/L, so every likelihood underflows with it; benchmarks here
    use get_realistic_io instead."""
    rnd = numpy.random if rng is None else rng
    u = numpy.array([rnd.uniform(low=0, high=0.1), rnd.uniform(low=0, high=0.2)])
    y = numpy.array([rnd.uniform(low=0.25, high=0.3), rnd.uniform(low=0.8, high=0.9)])
    return u, y


def get_realistic_io(rng, x_ref):
    """u as get_random_io; z = g(x_ref) + a host draw of the measurement noise (for benchmarks / tests)."""
    u = numpy.array([rng.uniform(0, 0.1), rng.uniform(0, 0.2)])
    y = numpy.array(model.Bioreactor.static_outputs(x_ref, u), dtype=numpy.float64)
    if rng.uniform() < 0.85:
        y = y + numpy.array([0.1, 0.0]) + rng.standard_normal(2) * numpy.sqrt([6e-2, 8e-2])
    else:
        y = y + numpy.array([0.0, -0.1]) + numpy.linalg.cholesky(numpy.array([[500., 100.], [100., 700.]])) @ rng.standard_normal(2)
    return u, y


# ----------------------------------------------------------------------------------------------------------------
# Closed loop (sim_base.py:164-309).  The estimator and the controller run on the device; the plant, its noise and the
# bookkeeping are scalar host work exactly as in the reference.  ``controller=`` replaces the MPC by any callable
# (x_estimate, u_previous, y_measured) -> u, e.g. hold_controller.
# ----------------------------------------------------------------------------------------------------------------
_OUTPUTS = [0, 2]            # lin_model.outputs after select_subset (sim_base.py:69-73): Cg, Cfa
_INPUTS = [0, 1]
_U_FALLBACK = numpy.array([0.06, 0.2])
_STATE_NOISE = (numpy.zeros((2, 5)),
                numpy.array([numpy.diag([1e-4, 1e-7, 1e-3, 1e-3, 1e-7]), numpy.diag([1e-3, 1e-6, 1e-2, 1e-2, 1e-6])]),
                numpy.array([0.75, 0.25]))
_MEAS_NOISE = (numpy.array([[1e-1, 0], [0, -1e-1]]),
               numpy.array([[[6e-2, 0], [0, 8e-2]], [[500, 100], [100, 700]]]),
               numpy.array([0.85, 0.15]))


def _host_mixture_draw(rng, mixture, size):
    """size draws of a Gaussian sum on the host (the plant's own noise; the reference pulls these through the device
    class and .get(), sim_base.py:281-284 -- a host generator gives the same distribution without two syncs a tick)."""
    means, covs, weights = mixture
    comp = rng.choice(len(weights), size=size, p=weights / weights.sum())
    chol = numpy.linalg.cholesky(covs)
    z = rng.standard_normal((size, means.shape[1]))
    return means[comp] + numpy.einsum("nij,nj->ni", chol[comp], z)


def _simpson_pairs(y, x):
    """Composite Simpson's rule over an odd number of samples, interval pairs of unequal width allowed."""
    h0, h1 = numpy.diff(x)[0::2], numpy.diff(x)[1::2]
    hs, hp, hq = h0 + h1, h0 * h1, h0 / h1
    return numpy.sum(hs / 6.0 * (y[0:-2:2] * (2.0 - 1.0 / hq) + y[1:-1:2] * (hs * hs / hp) + y[2::2] * (2.0 - hq)))


def simps(y, x):
    """scipy.integrate.simps(y, x) as the reference's SciPy (>= 1.4.1, environment.yml:8) evaluates it -- the function is
    gone from SciPy 1.14 on, and scipy.integrate.simpson treats an EVEN number of samples (the loop has end_time * 10 of
    them) differently.  Default even='avg': the mean of (Simpson over the first N - 1 samples + a trapezoid on the last
    interval) and (a trapezoid on the first interval + Simpson over the last N - 1 samples)."""
    y, x = numpy.asarray(y, dtype=float), numpy.asarray(x, dtype=float)
    if len(y) % 2 == 1:
        return float(_simpson_pairs(y, x))
    first = _simpson_pairs(y[:-1], x[:-1]) + 0.5 * (x[-1] - x[-2]) * (y[-1] + y[-2])
    last = 0.5 * (x[1] - x[0]) * (y[1] + y[0]) + _simpson_pairs(y[1:], x[1:])
    return float(0.5 * (first + last))


def performance(ys, r, ts):
    """Integral of the time-weighted squared error, summed over the outputs (sim_base.py:164-185)."""
    se = (numpy.asarray(ys) - numpy.asarray(r)) ** 2
    return float(sum(simps(se_ax * ts, ts) for se_ax in numpy.rollaxis(se, 1)))


def performance_rollouts(ys, r, ts):
    """performance() of B rollouts at once: ys, r (ticks, B, outputs) -> (B,)."""
    se = (numpy.asarray(ys) - numpy.asarray(r)) ** 2 * numpy.asarray(ts)[:, None, None]
    # all columns at once, each summed along its own contiguous row: the additions (NumPy's pairwise order) of simps() on one
    # column, so the same bits -- without 2 B Python calls of it (1.7 ms of a 9.4 ms loop of 32 rollouts)
    cols = numpy.ascontiguousarray(se.reshape(len(ts), -1).T)
    x = numpy.asarray(ts, dtype=float)

    def pairs(y, xx):
        h0, h1 = numpy.diff(xx)[0::2], numpy.diff(xx)[1::2]
        hs, hp, hq = h0 + h1, h0 * h1, h0 / h1
        terms = hs / 6.0 * (y[:, 0:-2:2] * (2.0 - 1.0 / hq) + y[:, 1:-1:2] * (hs * hs / hp) + y[:, 2::2] * (2.0 - hq))
        return numpy.ascontiguousarray(terms).sum(axis=1)
    if cols.shape[1] % 2 == 1:
        per = pairs(cols, x)
    else:
        first = pairs(cols[:, :-1], x[:-1]) + 0.5 * (x[-1] - x[-2]) * (cols[:, -1] + cols[:, -2])
        last = 0.5 * (x[1] - x[0]) * (cols[:, 1] + cols[:, 0]) + pairs(cols[:, 1:], x[1:])
        per = 0.5 * (first + last)
    return per.reshape(se.shape[1], se.shape[2]).sum(axis=1)


class InjectedNoise:
    """Every random number of a closed loop as an INPUT -- the parity mode of the loop drivers below (what
    DeterministicGaussianSum is for one filter): with it a loop is a deterministic function of its arguments and can be
    compared tick by tick with a restatement of the reference's loop.

    plant_state : (ticks - 1, B, 5) fp32   what state_pdf.draw() returns at sim_base.py:282, per tick and rollout
    plant_meas  : (ticks - 1, B, 2) fp32   measurement_pdf.draw() at sim_base.py:285
    predict     : sequence, one entry per predict(): (B, N, 5) for the particle filter, (B, N, 11, 5) for the GS-UKF
    r           : sequence, one entry per resample(): (B,) uniform numbers in [0, 1)
    ancestors   : optional sequence, one entry per resample(): (B, N) integers -- the ancestors a RECORDED run chose (the
                  reference's own resample, tests/golden/closed_loop.npz); given, the estimator copies those instead of
                  computing its own, so that a run whose weights sit within rounding of a threshold somewhere is still
                  comparable to the end (the replay of the reference's recorded runs)
    (B = 1 for Simulation; the leading rollout axis may then be left out.)"""

    def __init__(self, plant_state, plant_meas, predict, r, ancestors=None):
        self.plant_state = numpy.asarray(plant_state, dtype=numpy.float32)
        self.plant_meas = numpy.asarray(plant_meas, dtype=numpy.float32)
        self.predict = [numpy.asarray(v, dtype=numpy.float32) for v in predict]
        self.r = [numpy.atleast_1d(numpy.asarray(v, dtype=numpy.float64)) for v in r]
        self.ancestors = None if ancestors is None else [numpy.asarray(a, dtype=numpy.int64) for a in ancestors]

    def given(self, k):
        """The recorded ancestors of resample k, or None."""
        return None if self.ancestors is None else self.ancestors[k]

    def state(self, tick, B):
        return self.plant_state[tick].reshape(B, 5)

    def meas(self, tick, B):
        return self.plant_meas[tick].reshape(B, 2)


def _replaying(estimator):
    """Switch an estimator to replayed state noise (its predict() then takes what DeterministicGaussianSum.preload holds)."""
    estimator.state_pdf = get_noise(deterministic=True)[0]


def hold_controller(x_estimate, u_previous, y_measured):
    """The input the reference falls back to when its MPC step fails (sim_base.py:271-273)."""
    return _U_FALLBACK.copy()


class Simulation:
    """Holds details of a closed-loop simulation (sim_base.py:207-309): plant + estimator + MPC.

    Parameters as in the reference; ``controller`` optionally replaces the MPC by a callable
    (x_estimate, u_previous, y_measured) -> u, ``seed`` seeds the plant's noise."""

    def __init__(self, N_particles, dt_control, dt_predict, end_time=50, pf=True, controller=None, seed=0, noise=None):
        self.ts = numpy.linspace(0, end_time, int(end_time * 10))
        self.dt = self.ts[1]
        self.dt_control = dt_control
        self.dt_predict = dt_predict
        self.bioreactor, self.lin_model, self.K, self.f = get_parts(dt_control=dt_control, N_particles=N_particles, pf=pf)
        self.controller = controller
        self._rng = numpy.random.default_rng(seed)
        self.noise = noise                              # InjectedNoise: every random number is an input (parity mode)
        if noise is not None:
            _replaying(self.f)

        self.us = [_U_FALLBACK.copy()]
        self.xs = [self.bioreactor.X.copy()]
        self.ys = [self.bioreactor.outputs(self.us[-1])]
        self.ys_meas = [self.bioreactor.outputs(self.us[-1])]
        self.xs_f = [self.f.point_estimate()]
        self.ys_f = [numpy.array(model.Bioreactor.static_outputs(self.xs_f[-1], self.us[-1]))]
        self.covariance_point_size = [self.f.point_covariance()]
        self.biass = []
        self.performance = None
        self.mpc_frac = None
        self.predict_count, self.update_count = 0, 0

    def _control(self):
        """One controller call (sim_base.py:262-274): the MPC in deviation variables; when its QP is not solved the
        reference substitutes [0.06, 0.2] for the DEVIATION input (the operating point is then added on top)."""
        lm = self.lin_model
        u_next = self.us[-1].copy()
        if self.controller is not None:
            return numpy.asarray(self.controller(self.xs_f[-1], self.us[-1], self.ys_meas[-1]), dtype=numpy.float64), True
        try:
            u = self.K.step(lm.xn2d(self.xs_f[-1]), lm.un2d(self.us[-1]), lm.yn2d(self.ys_meas[-1]))
            solved = True
        except ValueError:
            u, solved = _U_FALLBACK.copy(), False
        u_next[lm.inputs] = lm.ud2n(u)
        return u_next, solved

    def simulate(self):
        """Performs a simulation using the simulation parameters (sim_base.py:243-309)."""
        t_next_control, t_next_predict = 0, 0
        solved_count = failed_count = 0
        noise = self.noise
        for tick, t in enumerate(self.ts[1:]):
            if t > t_next_predict:
                if noise is not None:
                    gaussian_sum_dist.DeterministicGaussianSum.preload(noise.predict[self.predict_count])
                self.f.predict(self.us[-1], self.dt)
                self.predict_count += 1
                t_next_predict += self.dt_predict
            if t > t_next_control:
                if self.controller is None and self.K.y_predicted is not None:
                    self.biass.append(self.lin_model.yn2d(self.ys_meas[-1]) - self.K.y_predicted)
                self.f.update(self.us[-1], self.ys_meas[-1][_OUTPUTS])
                self.f.resample(r=None if noise is None else float(noise.r[self.update_count][0]),
                                ancestors=None if noise is None else noise.given(self.update_count))
                self.update_count += 1
                self.xs_f.append(self.f.point_estimate())
                u, solved = self._control()
                solved_count, failed_count = solved_count + solved, failed_count + (not solved)
                self.us.append(u.copy())
                t_next_control += self.dt_control
            else:
                self.us.append(self.us[-1])

            self.bioreactor.step(self.dt, self.us[-1])
            self.bioreactor.X += (_host_mixture_draw(self._rng, _STATE_NOISE, 1)[0] if noise is None
                                  else noise.state(tick, 1)[0].astype(numpy.float64))
            outputs = self.bioreactor.outputs(self.us[-1])
            self.ys.append(outputs.copy())
            outputs[_OUTPUTS] += (_host_mixture_draw(self._rng, _MEAS_NOISE, 1)[0] if noise is None
                                  else noise.meas(tick, 1)[0].astype(numpy.float64))
            self.ys_meas.append(outputs)
            self.xs.append(self.bioreactor.X.copy())
            estimate = self.f.point_estimate()
            self.ys_f.append(numpy.array(model.Bioreactor.static_outputs(estimate, self.us[-1])))
            self.covariance_point_size.append(self.f.point_covariance())

        for name in ("us", "xs", "ys", "ys_meas", "xs_f", "ys_f", "covariance_point_size"):
            setattr(self, name, numpy.array(getattr(self, name)))
        self.performance = performance(self.ys[:, _OUTPUTS], self.ys_f, self.ts)
        self.mpc_frac = solved_count / max(solved_count + failed_count, 1)


class BatchedSimulation:
    """B independent closed loops advanced together: the Monte-Carlo repetitions of the reference's closed-loop
    performance studies (results/pf_closedloop/bioreactor_performance_pf.py:104-118 and, with pf=False,
    results/gsf_closedloop/bioreactor_performance_gsf.py:104-118: sequential Python loops there) become the batch
    dimension of every kernel launch.  Plants are vectorised NumPy on the host; per tick there is one fused estimator
    launch and one synchronisation (the estimates of all rollouts)."""

    def __init__(self, batch, N_particles, dt_control, dt_predict, end_time=50, pf=True, controller=None, seed=0,
                 resident=False, noise=None, persistent=True, fused_tail=True, native_loop=True):
        self.B = int(batch)
        self.persistent = bool(persistent)      # resident loops of small Gaussian-sum filters: the whole loop in one launch
        self.native_loop = bool(native_loop)    # resident particle-filter loops: the ticks are enqueued from C++ (gse_pf_closed_loop)
        self.fused_tail = bool(fused_tail)      # resident particle-filter loops: moments' reduction + controller + plant in one launch
        self.noise = noise                      # InjectedNoise: every random number is an input (parity mode)
        self.resident = bool(resident)          # True: plant, controller and histories live on the device too
        if self.resident and controller is not None:
            raise ValueError("a resident loop runs the MPC on the device; host callables cannot replace it")
        self.ts = numpy.linspace(0, end_time, int(end_time * 10))
        self.dt = self.ts[1]
        self.dt_control, self.dt_predict = dt_control, dt_predict
        xss = model.Bioreactor.find_SS(numpy.array([0.06, 0.2]), numpy.array([260 / 180, 640 / 24.6, 1000 / 116, 0, 0]))
        self.lin_model, self.K = get_controller(model.Bioreactor(X0=xss.copy(), high_N=False), dt_control)
        state_pdf, measurement_pdf = get_noise()
        x0, _ = get_noise()
        x0.means += numpy.asarray(xss[numpy.newaxis, :], dtype=numpy.float32)
        batched = _filter.BatchedParticleFilter if pf else _filter.BatchedGaussianSumUnscentedKalmanFilter
        self.f = batched(model.Bioreactor.homeostatic_DEs, model.Bioreactor.static_outputs, N_particles, x0, state_pdf,
                         measurement_pdf, batch=self.B)
        self.controller = controller
        self.pf = bool(pf)
        if noise is not None:
            _replaying(self.f)
        self._rng = numpy.random.default_rng(seed)
        self.X = numpy.repeat(xss[None, :], self.B, axis=0)
        self.us = [numpy.repeat(_U_FALLBACK[None, :], self.B, axis=0)]
        self.xs, self.ys, self.ys_f, self.covariance_point_size = [self.X.copy()], [], [], []
        self.ys.append(self._outputs())
        self.ys_meas = [self.ys[-1].copy()]
        est, cov = self.f.estimates()
        self.xs_f = [est]
        self.ys_f.append(numpy.stack([est[:, 0] * 180, est[:, 2] * 116], axis=1))
        self.covariance_point_size.append(cov)
        self.performance = None
        self.mpc_frac = None

    def _control(self, est):
        """All rollouts' controller calls at once (one kernel launch); see Simulation._control."""
        if self.controller is not None:
            u = numpy.array([self.controller(est[b], self.us[-1][b], self.ys_meas[-1][b]) for b in range(self.B)])
            return u, numpy.ones(self.B, dtype=bool)
        lm = self.lin_model
        u_dev, ok = self.K.step_batch(lm.xn2d(est), lm.un2d(self.us[-1]), lm.yn2d(self.ys_meas[-1]))
        u_dev = numpy.where(ok[:, None], u_dev, _U_FALLBACK[None, :])
        u_next = self.us[-1].copy()
        u_next[:, lm.inputs] = lm.ud2n(u_dev)
        return u_next, ok

    def _outputs(self):
        return self.X * numpy.array([180, 24.6, 116, 46, 1])

    def _plant_step(self, u):
        dX = model.Bioreactor.homeostatic_DEs_rows(self.X, u, 1)
        self.X = self.X + dX * self.dt
        self.X[:, :4] = numpy.maximum(self.X[:, :4], 0)

    def simulate(self):
        if self.resident:
            return self._simulate_resident()
        t_next_control, t_next_predict = 0, 0
        solved = asked = 0
        noise, n_predict, n_update = self.noise, 0, 0
        for tick, t in enumerate(self.ts[1:]):
            if t > t_next_predict:
                if noise is not None:
                    gaussian_sum_dist.DeterministicGaussianSum.preload(noise.predict[n_predict])
                self.f.predict(self.us[-1], self.dt)
                n_predict += 1
                t_next_predict += self.dt_predict
            if t > t_next_control:
                self.f.update(self.us[-1], self.ys_meas[-1][:, _OUTPUTS])
                self.f.resample(r=None if noise is None else noise.r[n_update],
                                ancestors=None if noise is None else noise.given(n_update))
                n_update += 1
                est = self.f.point_estimate()
                self.xs_f.append(est)
                u, ok = self._control(est)
                solved += int(ok.sum())
                asked += self.B
                self.us.append(u)
                t_next_control += self.dt_control
            else:
                self.us.append(self.us[-1])
            self._plant_step(self.us[-1])
            self.X += (_host_mixture_draw(self._rng, _STATE_NOISE, self.B) if noise is None
                       else noise.state(tick, self.B).astype(numpy.float64))
            out = self._outputs()
            self.ys.append(out.copy())
            out[:, _OUTPUTS] += (_host_mixture_draw(self._rng, _MEAS_NOISE, self.B) if noise is None
                                 else noise.meas(tick, self.B).astype(numpy.float64))
            self.ys_meas.append(out)
            self.xs.append(self.X.copy())
            est, cov = self.f.estimates()
            self.ys_f.append(numpy.stack([est[:, 0] * 180, est[:, 2] * 116], axis=1))
            self.covariance_point_size.append(cov)
        for name in ("us", "xs", "ys", "ys_meas", "xs_f", "ys_f", "covariance_point_size"):
            setattr(self, name, numpy.array(getattr(self, name)))
        self.performance = numpy.array([performance(self.ys[:, b][:, _OUTPUTS], self.ys_f[:, b], self.ts)
                                        for b in range(self.B)])
        self.mpc_frac = solved / max(asked, 1)

    def _simulate_resident(self):
        """The same loop with NOTHING on the host between ticks: the estimator takes its inputs and measurements from
        device memory (GSE_OP_U_DEVICE / GSE_OP_Z_DEVICE), the plant (gse_plant_step) and the controller
        (gse_loop_control) are kernels, and every per-tick record goes into device arrays that are read once, at the
        end.  Either estimator: the particle filter, or -- pf=False -- the batch of Gaussian-sum filters (one launch per
        control tick for banks of at most 64 Gaussians, gse_gsukf_cycle_dev).  The host only enqueues -- per tick: plant + moments, and on control ticks update / resample / controller --
        and never waits.  The plant's noise comes from Philox streams of its own here (the host loop uses NumPy's); the
        bias log ``biass`` of the single-rollout class is not kept."""
        import ctypes
        import torch
        from gpu_se_amd import _lib
        import gpu_se_amd
        lib = _lib.load()
        f, K, B = self.f, self.K, self.B
        dev = f._w.device
        ticks = len(self.ts)
        f64 = dict(dtype=torch.float64, device=dev)
        # (the loop's initial values in ONE host -> device copy: state, input, outputs, the estimator's measurement)
        ys0 = numpy.asarray(self.ys[0], dtype=numpy.float64).reshape(B, 5)
        init = torch.as_tensor(numpy.ascontiguousarray(numpy.concatenate(
            [numpy.asarray(self.X, dtype=numpy.float64).reshape(B, 5), numpy.asarray(self.us[0], dtype=numpy.float64).reshape(B, 2),
             ys0, ys0[:, _OUTPUTS]], axis=1)), **f64)
        X = init[:, 0:5].contiguous()
        u = init[:, 5:7].contiguous()                                    # the current input of every rollout
        # every per-tick record lives in ONE device buffer (carved below) and comes back in one copy at the end
        widths = dict(us=2, xs=5, ys=5, ys_meas=5, moments=26)
        flat = torch.zeros((ticks * B * sum(widths.values()),), **f64)
        hist, at = {}, 0
        for name, wdt in widths.items():
            hist[name] = flat[at:at + ticks * B * wdt].view(ticks, B, wdt)
            at += ticks * B * wdt
        us, xs, ys, ys_meas, moments = (hist[k] for k in ("us", "xs", "ys", "ys_meas", "moments"))
        z = init[:, 12:14].contiguous()
        us[0], xs[0] = u, X
        ys[0] = ys_meas[0] = init[:, 7:12]
        y_pred, known, _ = K.loop_state(B, dev)
        status = torch.zeros((ticks, B), dtype=torch.int32, device=dev)      # one row per tick, written by the controller
        n_control, last_control = 0, -1
        control_ticks = [0]
        rng = _lib.Rng(seed=gpu_se_amd._new_stream_seed(), step=0, element_offset=0)
        state_pod = ctypes.byref(f.state_pdf._pod())
        meas_pod = ctypes.byref(f.measurement_pdf._pod())
        stream = torch.cuda.current_stream().cuda_stream

        def record_moments(t):
            f._moments26_into(moments[t])          # (either estimator: 26 doubles per rollout, the particle filter's layout)

        import time
        record_moments(0)
        if self._persistent_capable():
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            clock = [time.perf_counter()]
            ev0.record()
            control_ticks, n_control = self._launch_persistent(lib, f, K, X, u, z, y_pred, known, status, us, xs, ys, ys_meas,
                                                               moments, state_pod, meas_pod, rng, stream)
            return self._finish_resident(lib, torch, time, flat, widths, moments, status, control_ticks, n_control, ev0, ev1,
                                         clock, f64, stream)
        if self._native_capable():
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            clock = [time.perf_counter()]
            ev0.record()
            control_ticks, n_control = self._launch_native(lib, f, K, X, u, z, y_pred, known, status, us, xs, ys, ys_meas, moments,
                                                           rng, stream)
            return self._finish_resident(lib, torch, time, flat, widths, moments, status, control_ticks, n_control, ev0, ev1,
                                         clock, f64, stream)
        noise, n_predict, n_update = self.noise, 0, 0
        if noise is not None:                             # the plant's draws: uploaded once, sliced per tick
            inj_state = torch.as_tensor(numpy.ascontiguousarray(noise.plant_state.reshape(ticks - 1, B, 5)), device=dev)
            inj_meas = torch.as_tensor(numpy.ascontiguousarray(noise.plant_meas.reshape(ticks - 1, B, 2)), device=dev)
        # the rest of a control tick behind the particle filter's deferred moments -- their reduction, the controller call, the
        # plant -- rides ONE launch (gse_pf_moments_resampled_tail) where it can: two launches for what were five
        tail = None
        if self.pf and self.fused_tail and K.n_bounded_rows <= controller.MAX_ACTIVE:
            _, law, _keep_law = K._device_law()
            glue = K.loop_glue(_U_FALLBACK)
            addr = lambda obj: ctypes.cast(ctypes.pointer(obj), ctypes.c_void_p).value
            tail = _lib.LoopTail(law=addr(law), glue=addr(glue), u=u.data_ptr(), y_pred=y_pred.data_ptr(), known=known.data_ptr(),
                                 X=X.data_ptr(), dt=float(self.dt), state_noise=addr(f.state_pdf._pod()),
                                 meas_noise=addr(f.measurement_pdf._pod()), rng=addr(rng), z=z.data_ptr())
            rows = {k: (hist[k].data_ptr(), B * widths[k] * 8) for k in ("us", "xs", "ys", "ys_meas")}
            row = lambda k, t_: rows[k][0] + t_ * rows[k][1]
            status_base = status.data_ptr()
        t_next_control, t_next_predict = 0, 0
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()                                      # device time of the loop: first to last kernel of the ticks
        clock = [time.perf_counter()]
        for t in range(1, ticks):
            now = self.ts[t]
            fused = False
            if now > t_next_predict:
                if noise is not None:
                    gaussian_sum_dist.DeterministicGaussianSum.preload(noise.predict[n_predict])
                f.predict(u, self.dt)
                n_predict += 1
                t_next_predict += self.dt_predict
            if now > t_next_control:
                f.update(u, z)
                f.resample(r=None if noise is None else noise.r[n_update],
                           ancestors=None if noise is None else noise.given(n_update))
                n_update += 1
                # the estimate the controller acts on, and this tick's record: the moments of the resampled ensemble taken
                # from the weighted one (offspring counts), so that the resample's copy is deferred into the next tick's
                # single launch (gather + predict + update); where that does not apply, resample now and take them after
                if tail is not None:
                    rng.step = t
                    tail.y_meas, tail.status = row("ys_meas", t - 1), status_base + t * B * 4
                    tail.y, tail.y_meas_out, tail.log_x, tail.log_u = row("ys", t), row("ys_meas", t), row("xs", t), row("us", t)
                    if noise is not None:
                        tail.state_draw, tail.meas_draw = inj_state[t - 1].data_ptr(), inj_meas[t - 1].data_ptr()
                    fused = f._resampled_moments_tail(moments[t], tail)
                if not fused:
                    if not (self.pf and f._resampled_moments_into(moments[t])):
                        record_moments(t)
                    K.loop_control(moments[t], ys_meas[t - 1], u, (y_pred, known, status[t]), _U_FALLBACK)
                n_control += 1
                last_control = t
                control_ticks.append(t)
                t_next_control += self.dt_control
            rng.step = t
            if fused:
                pass                                      # (the plant ran in the tail's launch)
            elif noise is not None:
                _lib.check(lib.gse_plant_step_injected(X.data_ptr(), u.data_ptr(), float(self.dt), inj_state[t - 1].data_ptr(),
                                                       inj_meas[t - 1].data_ptr(), ys[t].data_ptr(), ys_meas[t].data_ptr(),
                                                       z.data_ptr(), xs[t].data_ptr(), us[t].data_ptr(), B, stream))
            else:
                _lib.check(lib.gse_plant_step(X.data_ptr(), u.data_ptr(), float(self.dt), state_pod, meas_pod,
                                              ctypes.byref(rng), ys[t].data_ptr(), ys_meas[t].data_ptr(), z.data_ptr(),
                                              xs[t].data_ptr(), us[t].data_ptr(), B, stream))
            if last_control != t:                         # (a control tick has recorded this state already)
                record_moments(t)
        return self._finish_resident(lib, torch, time, flat, widths, moments, status, control_ticks, n_control, ev0, ev1, clock,
                                     f64, stream)

    @property
    def moments(self):
        """The estimator's moment rows of a resident run, (ticks, B, 26) doubles (gse_pf_moments layout), fetched from the device
        on first use."""
        if getattr(self, "_moments_host", None) is None and getattr(self, "_moments_device", None) is not None:
            self._moments_host = self._moments_device.cpu().numpy()
            self._moments_device = None
        return getattr(self, "_moments_host", None)

    def _schedule(self):
        """(schedule [ticks] uint8: bit 0 a predict is due, bit 1 the controller acts -- sim_base.py:250, :255; control ticks)."""
        ticks = len(self.ts)
        sched = numpy.zeros(ticks, dtype=numpy.uint8)
        t_next_control, t_next_predict = 0, 0
        control_ticks = [0]
        for t in range(1, ticks):
            if self.ts[t] > t_next_predict:
                sched[t] |= 1
                t_next_predict += self.dt_predict
            if self.ts[t] > t_next_control:
                sched[t] |= 2
                control_ticks.append(t)
                t_next_control += self.dt_control
        return sched, control_ticks

    def _native_capable(self):
        """The ticks of a resident particle-filter loop enqueued from C++ (gse_pf_closed_loop): Philox noise, the built-in
        model, the fused tail, up to 2^21 particles.  ``native_loop = False`` keeps the Python loop below (the same launches)."""
        f = self.f
        return (self.native_loop and self.pf and self.fused_tail and self.noise is None and self.controller is None and
                self.K.n_bounded_rows <= controller.MAX_ACTIVE and
                f._model is None and f.N_particles <= (1 << 21) and f.measurement_pdf._Nd == 2 and f.lazy and
                f._island is None and not f.record_ancestors and
                (not f.cluster or f.N_particles <= _lib.GSE_CLUSTER_MAX_TILES * _lib.GSE_TILE))

    def _launch_native(self, lib, f, K, X, u, z, y_pred, known, status, us, xs, ys, ys_meas, moments, rng, stream):
        import ctypes
        from gpu_se_amd import _lib
        sched, control_ticks = self._schedule()
        n_control = len(control_ticks) - 1
        B = self.B
        f._flush()
        # the uniform numbers of the resamples: the class's own host stream, in the order resample() would draw them
        r_host = numpy.ascontiguousarray(f._host_rng.random((max(n_control, 1), B)))      # (= that many calls of random(B))
        _, law, _keep_law = K._device_law()
        glue = K.loop_glue(_U_FALLBACK)
        addr = lambda obj: ctypes.cast(ctypes.pointer(obj), ctypes.c_void_p).value
        spec = _lib.PfLoopSpec()
        spec.B, spec.ticks, spec.N, spec.ld, spec.dt = B, len(self.ts), f.N_particles, f._ld, float(self.dt)
        clustered = bool(f.cluster)                       # (the filter's policy: four workgroups per tile for mid-size ensembles)
        spec.flags = (_lib.GSE_PF_LOOP_WEIGHT_SCALE if f.weight_scale else 0) | (_lib.GSE_PF_LOOP_CLUSTER if clustered else 0)
        spec.x[0], spec.x[1], spec.w[0], spec.w[1] = f._x.data_ptr(), f._x_alt.data_ptr(), f._w.data_ptr(), f._w_alt.data_ptr()
        spec.workspace = f._ws.data_ptr()
        spec.state_pdf, spec.measurement_pdf = addr(f.state_pdf._pod()), addr(f.measurement_pdf._pod())
        spec.rng = _lib.Rng(seed=f._seed, step=f._step, element_offset=f._element_offset)
        spec.law, spec.glue = addr(law), addr(glue)
        spec.X, spec.u, spec.z, spec.y_pred, spec.known = X.data_ptr(), u.data_ptr(), z.data_ptr(), y_pred.data_ptr(), known.data_ptr()
        spec.plant_state_noise, spec.plant_meas_noise = spec.state_pdf, spec.measurement_pdf        # (the same two mixtures)
        spec.plant_rng = _lib.Rng(seed=rng.seed, step=0, element_offset=rng.element_offset, filter_offset=rng.filter_offset)
        spec.schedule, spec.r = sched.ctypes.data, r_host.ctypes.data
        spec.us, spec.xs, spec.ys, spec.ys_meas, spec.moments = (t_.data_ptr() for t_ in (us, xs, ys, ys_meas, moments))
        spec.status = status.data_ptr()
        spec.w_uniform = 1 if f._w_uniform else 0          # (the first update of a fresh filter needs no pass for its scale's anchor)
        _lib.check(lib.gse_pf_closed_loop(ctypes.byref(spec), stream))
        # the filter object carries on from where the loop left it
        if spec.x_current:
            f._x, f._x_alt = f._x_alt, f._x
        if spec.w_current:
            f._w, f._w_alt = f._w_alt, f._w
        f._step = int(spec.rng.step)
        f._pend_gather = bool(spec.pend_gather)
        f._gather_slot = int(spec.gather_slot) if spec.gather_slot >= 0 else None
        f._scan_slot = int(spec.scan_slot) if spec.scan_slot >= 0 else None
        for slot in (f._gather_slot, f._scan_slot):
            if slot is not None:
                f._slot_fmt[slot] = "q" if clustered else "t"
        f._tiles_valid = False
        f._moments_host = None
        f._samples_ok = bool(spec.samples_ok)
        f._w_uniform = bool(spec.w_uniform)
        if spec.pend_predict:
            f._pend_predict = (u, float(self.dt), None)
        for b in range(B):
            f._c_r[b] = spec.r_pending[b]
        self._keep_alive = (sched, r_host, law, glue, _keep_law)
        return control_ticks, n_control

    def _persistent_capable(self):
        """The whole loop in ONE launch (gse_gsukf_closed_loop): rollouts whose estimator is a Gaussian-sum filter of at most 64
        Gaussians on the built-in model, Philox noise.  ``persistent = False`` keeps the launch-per-piece loop."""
        from gpu_se_amd import _lib
        return (self.persistent and not self.pf and self.noise is None and self.f._model is None and
                self.K.n_bounded_rows <= controller.MAX_ACTIVE and
                self.f.N_particles <= _lib.GSE_GSUKF_CYCLE_MAX and self.controller is None)

    def _launch_persistent(self, lib, f, K, X, u, z, y_pred, known, status, us, xs, ys, ys_meas, moments, state_pod, meas_pod, rng,
                           stream):
        import ctypes
        import torch
        from gpu_se_amd import _lib
        ticks, B = len(self.ts), self.B
        # the schedule of sim_base.py:250, :255 and the uniform numbers of the resamples (the class's own host stream, in the
        # order resample() would draw them)
        sched, control_ticks = self._schedule()
        n_control = len(control_ticks) - 1
        f._flush()
        # (one call: the generator fills the array in the order n_control calls of random(B) would)
        r_host = f._host_rng.random((n_control, B)) if n_control else numpy.zeros((1, B))
        dev = X.device
        sched_d = torch.as_tensor(sched, device=dev)
        r_d = torch.as_tensor(numpy.ascontiguousarray(r_host), device=dev)
        _, law, _keep = K._device_law()
        glue = K.loop_glue(_U_FALLBACK)
        frng = _lib.Rng(seed=f._seed, step=f._step, filter_offset=f._filter_offset)
        spec = _lib.LoopSpec()
        spec.B, spec.N, spec.ticks, spec.dt = B, f.N_particles, ticks, float(self.dt)
        addr = lambda obj: ctypes.cast(ctypes.pointer(obj), ctypes.c_void_p).value
        keep_alive = (frng, law, glue, sched_d, r_d)
        spec.w_sigma = ctypes.cast(f._w_sigma_c, ctypes.c_void_p).value
        spec.state_pdf, spec.measurement_pdf = addr(f.state_pdf._pod()), addr(f.measurement_pdf._pod())
        spec.rng, spec.plant_rng = addr(frng), addr(rng)
        spec.plant_state_noise, spec.plant_meas_noise = spec.state_pdf, spec.measurement_pdf     # (the same two mixtures)
        spec.law, spec.glue = addr(law), addr(glue)
        spec.schedule, spec.r = sched_d.data_ptr(), r_d.data_ptr()
        spec.mu, spec.P, spec.w = f._means.data_ptr(), f._covs.data_ptr(), f._w.data_ptr()
        spec.X, spec.u, spec.z = X.data_ptr(), u.data_ptr(), z.data_ptr()
        spec.y_pred, spec.known = y_pred.data_ptr(), known.data_ptr()
        spec.us, spec.xs, spec.ys, spec.ys_meas, spec.moments = (t_.data_ptr() for t_ in (us, xs, ys, ys_meas, moments))
        spec.status, spec.chol_status = status.data_ptr(), f._status.data_ptr()
        args_buffer = torch.empty(_lib.GSE_LOOP_ARGS_BYTES, dtype=torch.uint8, device=dev)
        spec.args_buffer = args_buffer.data_ptr()
        keep_alive = keep_alive + (args_buffer,)
        _lib.check(lib.gse_gsukf_closed_loop(ctypes.byref(spec), stream))
        f._step += int((sched & 1).sum())
        self._keep_alive = keep_alive
        return control_ticks, n_control

    def _finish_resident(self, lib, torch, time, flat, widths, moments, status, control_ticks, n_control, ev0, ev1, clock, f64,
                         stream):
        from gpu_se_amd import _lib
        ticks, B = len(self.ts), self.B
        # the estimates of every tick and rollout from the recorded moments, still on the device (point estimate and the
        # largest singular value of the covariance: gse_moments_estimates); then one synchronisation and the copies
        estimates = torch.empty((ticks * B, 6), **f64)
        _lib.check(lib.gse_moments_estimates(moments.data_ptr(), ticks * B, estimates.data_ptr(), stream))
        ev1.record()
        clock.append(time.perf_counter())
        torch.cuda.synchronize()
        clock.append(time.perf_counter())
        device_s = ev0.elapsed_time(ev1) * 1e-3
        # (the moment rows -- 26 of a tick's 43 doubles, kept for whoever wants them -- stay on the device until somebody asks:
        #  ``moments`` fetches them; the records the reference's loop keeps come back now, in one copy)
        eager = ticks * B * sum(wdt for name, wdt in widths.items() if name != "moments")
        assert list(widths)[-1] == "moments"
        host = flat[:eager].cpu().numpy()
        est6 = estimates.cpu().numpy().reshape(ticks, B, 6)
        if self.pf:
            # a resample that met a weight sum that is not finite (the stored fp32 weights overflowed their power-of-two scale)
            # has collapsed its rollout onto one particle: never silently (GSE_WSTATUS_SUM_NOT_FINITE; round 5)
            self.f._check_status()
        clock.append(time.perf_counter())
        back, at = {}, 0
        for name, wdt in widths.items():
            if name == "moments":
                continue
            back[name] = host[at:at + ticks * B * wdt].reshape(ticks, B, wdt)
            at += ticks * B * wdt
        self._moments_device, self._moments_host = moments, None      # (rows of 26 doubles per tick and rollout: see ``moments``)
        self.us, self.xs, self.ys, self.ys_meas = back["us"], back["xs"], back["ys"], back["ys_meas"]
        self.X = self.xs[-1].copy()
        est = est6[..., :5]
        self.xs_f = est[control_ticks]                # appended at control ticks only (sim_base.py:260)
        self.ys_f = numpy.stack([est[..., 0] * 180, est[..., 2] * 116], axis=-1)
        self.covariance_point_size = est6[..., 5].copy()
        self.performance = performance_rollouts(self.ys[:, :, _OUTPUTS], self.ys_f, self.ts)
        self.mpc_frac = 1.0 - float(status.sum().item()) / max(n_control * B, 1)
        clock.append(time.perf_counter())
        self.timing = dict(zip(("enqueue", "device_drain", "copy_back", "host_post"), numpy.diff(clock)))
        self.timing["device_loop"] = device_s             # (HIP events around the ticks; >= the time the host needs to enqueue them)
