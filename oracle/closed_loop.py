"""CPU oracle for the closed loop (reference sim_base.py:207-309).  TEST INFRASTRUCTURE ONLY -- nothing under gpu_se_amd/
imports this module; only tests/ and tests/golden/make_golden.py may.

A NumPy restatement of the reference's ``Simulation``: plant (Bioreactor.step / outputs, model/BioreactorModel.py:95-122),
estimator (oracle.gse_oracle.ParticleFilterOracle or GSUKFOracle), controller (oracle.mpc_oracle.MPCOracle, or anything with
the reference's ``step(x0, um1, y0)`` contract) in the reference's order of operations, tick by tick:

    predict if due (:250-253) -> on a control tick: bias log (:256-257), update, resample (:259-261), point estimate
    (:263), MPC step with the except-branch fallback (:265-274), new input (:275-277) -> plant Euler step + clamp
    (:281) -> plant noise (:282) -> outputs (:283-284) -> measurement noise on the controlled outputs (:285-286) ->
    records (:287-296); at the end the integral performance (:164-185) and the solved fraction (:309).

Every random quantity is an INPUT (``LoopNoise``): the plant's state and measurement noise per tick, the estimator's
noise per predict, the resample's uniform number per resample -- so that a device loop fed the same numbers can be
compared tick by tick.

Parity status: PINNED -- tests/test_oracle_golden.py::test_closed_loop_oracle_against_the_reference_loop replays
tests/golden/closed_loop.npz, which tests/golden/make_golden.py recorded by running the reference's own
``Simulation.__init__`` / ``simulate`` text on the reference's NumPy classes (see there for what stands in for the absent
cupy / osqp).
"""
import numpy as np

from oracle import gse_oracle as orc

MOLAR_MASS = np.array([180, 24.6, 116, 46, 1])          # BioreactorModel.py:118-120
U_FALLBACK = np.array([0.06, 0.2])                      # sim_base.py:272 (lands where the DEVIATION input goes, :275)
U_START = np.array([0.06, 0.2])                         # sim_base.py:224


class LinearGlue:
    """What the loop needs of model.LinearModel after select_subset (LinearModel.py:126-274): the kept indices, the
    linearisation point and the deviation <-> natural conversions."""

    def __init__(self, x_bar, u_bar, y_bar, states=(0, 2), inputs=(0, 1), outputs=(0, 2)):
        self.x_bar, self.u_bar, self.y_bar = (np.asarray(v, dtype=float) for v in (x_bar, u_bar, y_bar))
        self.states, self.inputs, self.outputs = list(states), list(inputs), list(outputs)

    def xn2d(self, x):
        return np.asarray(x)[self.states] - self.x_bar

    def un2d(self, u):
        return np.asarray(u)[self.inputs] - self.u_bar

    def yn2d(self, y):
        return np.asarray(y)[self.outputs] - self.y_bar

    def ud2n(self, u_hat):
        return u_hat + self.u_bar


class LoopNoise:
    """The random inputs of one rollout.  plant_state [ticks - 1][5] and plant_meas [ticks - 1][2]: what
    ``state_pdf.draw()`` / ``measurement_pdf.draw()`` return in sim_base.py:282 / :285 (fp32 values); predict: a list of
    the estimator's noise arrays, one per predict() -- (N, 5) for the particle filter, (N, 11, 5) for the GS-UKF; r: the
    uniform number of every resample()."""

    def __init__(self, plant_state, plant_meas, predict, r, ancestors=None):
        self.plant_state = np.asarray(plant_state, dtype=np.float32)
        self.plant_meas = np.asarray(plant_meas, dtype=np.float32)
        self.predict = list(predict)
        self.r = [float(v) for v in r]
        # replay of a recorded run: the ancestors IT chose, copied instead of the oracle's own (which are still computed and
        # counted against them in ClosedLoopOracle.ancestor_flips) -- a weight within rounding of a threshold must not send
        # the rest of a long comparison down another path
        self.ancestors = None if ancestors is None else [np.asarray(a, dtype=np.int64) for a in ancestors]


def simps_avg(y, x):
    """scipy.integrate.simps(y, x) of the reference's SciPy (>= 1.4.1, environment.yml:8; gone since 1.14), default
    ``even='avg'``: composite Simpson's rule for an odd number of samples; for an even number the average of (Simpson on
    the first N - 1 samples + trapezoid on the last interval) and (trapezoid on the first interval + Simpson on the last
    N - 1 samples).  Unequal spacing handled per pair of intervals as SciPy's ``_basic_simps`` did."""
    y, x = np.asarray(y, dtype=float), np.asarray(x, dtype=float)

    def basic(y, x):                                   # odd number of samples
        h0, h1 = np.diff(x)[0::2], np.diff(x)[1::2]
        hsum, hprod, h0divh1 = h0 + h1, h0 * h1, h0 / h1
        return np.sum(hsum / 6.0 * (y[0:-2:2] * (2.0 - 1.0 / h0divh1) + y[1:-1:2] * (hsum * hsum / hprod)
                                    + y[2::2] * (2.0 - h0divh1)))

    n = len(y)
    if n % 2 == 1:
        return float(basic(y, x))
    first = basic(y[:-1], x[:-1]) + 0.5 * (x[-1] - x[-2]) * (y[-1] + y[-2])
    last = 0.5 * (x[1] - x[0]) * (y[1] + y[0]) + basic(y[1:], x[1:])
    return float((first + last) / 2.0)


def performance(ys, r, ts):
    """sim_base.py:164-185: sum over the outputs of the integral of the time-weighted squared error."""
    se = (np.asarray(ys) - np.asarray(r)) ** 2
    return float(sum(simps_avg(se_ax * ts, ts) for se_ax in np.rollaxis(se, 1)))


class ClosedLoopOracle:
    """sim_base.Simulation (sim_base.py:207-309) on the host, with every random number injected.

    estimator : ParticleFilterOracle / GSUKFOracle (its particles or means already drawn)
    mpc       : object with step(x0, um1, y0) -> u (raising on failure) and ``y_predicted``; or None = the MPC always
                fails, i.e. the reference's except branch on every control tick
    lin       : LinearGlue
    x_plant   : the plant's initial state (fp64, (5,))
    """

    def __init__(self, estimator, mpc, lin, x_plant, dt_control, dt_predict, end_time, noise):
        self.ts = np.linspace(0, end_time, int(end_time * 10))                          # :211
        self.dt = self.ts[1]
        self.dt_control, self.dt_predict = dt_control, dt_predict
        self.f, self.K, self.lin, self.noise = estimator, mpc, lin, noise
        self.X = np.array(x_plant, dtype=np.float64, copy=True)
        self.us = [U_START.copy()]
        self.xs = [self.X.copy()]
        self.ys = [self.X * MOLAR_MASS]
        self.ys_meas = [self.X * MOLAR_MASS]
        self.xs_f = [np.asarray(self.f.point_estimate(), dtype=np.float64)]
        self.ys_f = [self._static_outputs(self.xs_f[-1])]
        self.covariance_point_size = [float(self.f.point_covariance())]
        self.biass = []
        self.ancestors = []
        self.ancestor_flips = []                        # per resample, when the noise carries recorded ancestors
        self.performance = self.mpc_frac = None
        self.predict_count = self.update_count = 0

    @staticmethod
    def _static_outputs(x):
        return np.array([x[0] * 180, x[2] * 116], dtype=np.float64)                   # BioreactorModel.py:233-253

    def _predict(self, u):
        noise = self.noise.predict[self.predict_count]
        if hasattr(self.f, "predict_vec"):
            self.f.predict_vec(u, self.dt, noise=noise)
        else:
            self.f.predict(u, self.dt, noise=noise, loop=False)

    def simulate(self):
        lin = self.lin
        t_next_control, t_next_predict = 0, 0
        solved = failed = 0
        for tick, t in enumerate(self.ts[1:]):
            if t > t_next_predict:                                                      # :250-253
                self._predict(self.us[-1])
                self.predict_count += 1
                t_next_predict += self.dt_predict
            if t > t_next_control:                                                      # :255-277
                u_next = self.us[-1].copy()
                if self.K is not None and self.K.y_predicted is not None:
                    self.biass.append(lin.yn2d(self.ys_meas[-1]) - self.K.y_predicted)
                self.f.update(self.us[-1], self.ys_meas[-1][lin.outputs]) if not hasattr(self.f, "update_vec") else \
                    self.f.update_vec(self.us[-1], self.ys_meas[-1][lin.outputs])
                r = self.noise.r[self.update_count]
                if self.noise.ancestors is None:
                    self.ancestors.append(self.f.resample(r=r, loop=False))
                else:
                    own = self.f.ancestors(r)
                    forced = self.noise.ancestors[self.update_count]
                    self.ancestor_flips.append(int(np.sum(own != forced)))
                    self.ancestors.append(self.f.resample(r=r, idx=forced))
                self.update_count += 1
                self.xs_f.append(np.asarray(self.f.point_estimate(), dtype=np.float64))
                try:
                    if self.K is None:
                        raise ValueError("no controller")
                    u = self.K.step(lin.xn2d(self.xs_f[-1]), lin.un2d(self.us[-1]), lin.yn2d(self.ys_meas[-1]))
                    solved += 1
                except ValueError:
                    u = U_FALLBACK.copy()
                    failed += 1
                u_next[lin.inputs] = lin.ud2n(u)
                self.us.append(u_next.copy())
                t_next_control += self.dt_control
            else:
                self.us.append(self.us[-1])
            # plant: Bioreactor.step (BioreactorModel.py:95-108) = Euler step of the fp64 model, first four states >= 0
            self.X = self.X + np.array(orc.bioreactor_f(self.X, self.us[-1], 1.0)) * self.dt
            self.X[:4] = np.maximum(self.X[:4], 0)
            self.X = self.X + self.noise.plant_state[tick].astype(np.float64)            # :282 (fp32 draw, fp64 state)
            outputs = self.X * MOLAR_MASS                                              # :283
            self.ys.append(outputs.copy())
            outputs[lin.outputs] += self.noise.plant_meas[tick].astype(np.float64)     # :285
            self.ys_meas.append(outputs)
            self.xs.append(self.X.copy())
            self.ys_f.append(self._static_outputs(np.asarray(self.f.point_estimate(), dtype=np.float64)))
            self.covariance_point_size.append(float(self.f.point_covariance()))
        for name in ("us", "xs", "ys", "ys_meas", "xs_f", "ys_f", "covariance_point_size"):
            setattr(self, name, np.array(getattr(self, name)))
        self.performance = performance(self.ys[:, lin.outputs], self.ys_f, self.ts)
        self.mpc_frac = solved / (solved + failed) if solved + failed else None
        return self
