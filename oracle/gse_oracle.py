"""CPU oracle for the particle-filter / GS-UKF hot path.  TEST INFRASTRUCTURE ONLY.

This module is a from-scratch NumPy restatement of the reference's CPU algorithm
(AlgorithmicAmoeba/gpu_se).  Nothing in the product package ``gpu_se_amd`` imports it; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may, and there
only as the checker / the timed CPU baseline -- never as the thing shipped.

Parity status: PINNED.  Every function below is checked (tests/test_oracle_golden.py) against
golden vectors captured from the reference's own NumPy classes in the build container by
tests/golden/make_golden.py (model_fg, mixture_pdf, resample, pf_trace, gsukf_trace, constants).

Numerics restated here (reference file:line in each docstring):
  * the model functions use fp64 arithmetic on fp32-stored states -- what the reference's
    device code does (Python literals are fp64, filter/particle.py:176-191) and what its
    NumPy path does up to NumPy-2 scalar promotion (<= 3e-6 relative on dCg);
  * particles / means are stored fp32; likelihoods and cumulative sums are fp64.

Two flavours of every filter step are provided: a loop-faithful one (one Python call per
particle, like filter/particle.py:54-103 -- this is what bench.py times as the CPU baseline)
and a vectorised one (same arithmetic per element, used by the parity tests at larger N).
"""
import math

import numpy as np

NX, NU, NY = 5, 2, 2

# --------------------------------------------------------------------------------------
# Bioreactor model  (model/BioreactorModel.py:170-253)
# --------------------------------------------------------------------------------------
_CG_FEED = 5000 / 180            # glucose feed concentration            (:195)
_K_MONOD = 1e-2                  # saturation constant in Cg/(K+Cg)       (:205, :210)
_MM_X = 24.6                     # g biomass per mol
_FA_RATE = 0.25 / 116            # (:204)
_TH1_RATE = (0.4 - 0.25) / 180   # (:208)
_E_RATE = 0.025 / 46             # (:213)
_TH2_RATE = (0.1 - 0.025) / 180  # (:218)
_H_SET = 280 / 180               # (:201)
_H_GAIN = 2000.0
_H_SCALE = 0.28 / 180


def bioreactor_f(x, u, dt=1.0):
    """Euler increment of the low-nitrogen bioreactor ODE for ONE state row, fp64 math.

    Restates Bioreactor.homeostatic_DEs (model/BioreactorModel.py:170-231): Cg, Cx, Cfa, Ce are
    clamped at zero (Ch is not), V = 1 L, rX = 0.  Returns (dCg, dCx, dCfa, dCe, dCh).
    """
    cg, cx, cfa, ce, ch = (float(v) for v in x)
    cg, cx, cfa, ce = max(cg, 0.0), max(cx, 0.0), max(cfa, 0.0), max(ce, 0.0)
    fg, fm = float(u[0]), float(u[1])
    dt = float(dt)
    f_out = fg + fm
    r_x = 0.0 * cx
    r_h = _H_SET - cg
    monod = cg / (_K_MONOD + cg)
    r_fa = (_FA_RATE * cx * _MM_X * 1) * monod
    th1_max = _TH1_RATE * cx * _MM_X * 1
    th1_req = th1_max - (th1_max / _H_GAIN / _H_SCALE * r_h + 0.01 * ch)
    th1 = min(th1_max, max(0.0, th1_req)) * monod
    e_max = _E_RATE * cx * _MM_X * 1
    e_req = th1_req - th1_max
    r_e = min(e_max, max(0.0, e_req))
    th2_max = _TH2_RATE * cx * _MM_X * 1
    th2_req = th1_req - th1_max - r_e
    th2 = min(th2_max, max(0.0, th2_req))
    r_g = -r_fa * (116 / 180) - th1 - r_e * (46 / 180) - th2
    d_cg = (fg * _CG_FEED - f_out * cg + r_g) / 1 * dt
    d_cx = r_x / 1 * dt
    d_cfa = (-f_out * cfa + r_fa) / 1 * dt
    d_ce = (-f_out * ce + r_e) / 1 * dt
    d_ch = r_h / 1 * dt
    return d_cg, d_cx, d_cfa, d_ce, d_ch


def bioreactor_g(x, u=None):
    """Observation of ONE state row: (Cg*180, Cfa*116), no clamping
    (Bioreactor.static_outputs, model/BioreactorModel.py:233-253).  The products are formed in
    the state's own precision (fp32 rows give fp32 outputs, as in the reference under NumPy 2;
    the correctly rounded fp32 product is also what the reference's device code stores)."""
    cg, cfa = x[0], x[2]
    if isinstance(cg, (np.floating,)):
        return cg * cg.dtype.type(180), cfa * cfa.dtype.type(116)
    return cg * 180, cfa * 116


def bioreactor_f_vec(x, u, dt):
    """bioreactor_f applied to every row of ``x`` (..., 5); returns fp64 increments (..., 5).
    Element for element the same fp64 operations in the same order as bioreactor_f."""
    x = np.asarray(x)
    xd = x.astype(np.float64)
    cg = np.maximum(xd[..., 0], 0.0)
    cx = np.maximum(xd[..., 1], 0.0)
    cfa = np.maximum(xd[..., 2], 0.0)
    ce = np.maximum(xd[..., 3], 0.0)
    ch = xd[..., 4]
    fg, fm = float(u[0]), float(u[1])
    dt = float(dt)
    f_out = fg + fm
    r_x = 0.0 * cx
    r_h = _H_SET - cg
    monod = cg / (_K_MONOD + cg)
    r_fa = (_FA_RATE * cx * _MM_X * 1) * monod
    th1_max = _TH1_RATE * cx * _MM_X * 1
    th1_req = th1_max - (th1_max / _H_GAIN / _H_SCALE * r_h + 0.01 * ch)
    th1 = np.minimum(th1_max, np.maximum(0.0, th1_req)) * monod
    e_max = _E_RATE * cx * _MM_X * 1
    e_req = th1_req - th1_max
    r_e = np.minimum(e_max, np.maximum(0.0, e_req))
    th2_max = _TH2_RATE * cx * _MM_X * 1
    th2_req = th1_req - th1_max - r_e
    th2 = np.minimum(th2_max, np.maximum(0.0, th2_req))
    r_g = -r_fa * (116 / 180) - th1 - r_e * (46 / 180) - th2
    out = np.empty(x.shape, dtype=np.float64)
    out[..., 0] = (fg * _CG_FEED - f_out * cg + r_g) / 1 * dt
    out[..., 1] = r_x / 1 * dt
    out[..., 2] = (-f_out * cfa + r_fa) / 1 * dt
    out[..., 3] = (-f_out * ce + r_e) / 1 * dt
    out[..., 4] = r_h / 1 * dt
    return out


def bioreactor_g_vec(x):
    """bioreactor_g on every row of an fp32 array (..., 5) -> fp32 (..., 2)."""
    x = np.asarray(x, dtype=np.float32)
    return np.stack([x[..., 0] * np.float32(180), x[..., 2] * np.float32(116)], axis=-1)


def steady_state():
    """X_ss that sim_base.get_parts uses as the prior mean (sim_base.py:46-53); value pinned by
    tests/golden/constants.npz.  Closed form of the fsolve problem at u = (0.06, 0.2):
    dCh = 0 gives Cg = 280/180; Cx is held; Ce stays 0; dCfa = 0 and dCg = 0 give Cfa and Ch."""
    fg, fm = 0.06, 0.2
    cg = _H_SET
    cx = 640 / 24.6
    monod = cg / (_K_MONOD + cg)
    r_fa = _FA_RATE * cx * _MM_X * monod
    cfa = r_fa / (fg + fm)
    th1_max = _TH1_RATE * cx * _MM_X
    # dCg = 0 with rE = th2 = 0:  fg*Cg_in - F*Cg - rFA*116/180 - th1_req*monod = 0
    th1_req = (fg * _CG_FEED - (fg + fm) * cg - r_fa * (116 / 180)) / monod
    ch = (th1_max - th1_req) / 0.01
    return np.array([cg, cx, cfa, 0.0, ch])


# --------------------------------------------------------------------------------------
# Gaussian mixture  (gaussian_sum_dist/MultivariateGaussianSum.py:27-97)
# --------------------------------------------------------------------------------------
class MixtureOracle:
    """Gaussian-sum distribution on the host.

    Construction follows MultivariateGaussianSum.__init__ (:27-37): means / covariances /
    weights stored fp32, the inverse covariances are taken from the fp64 inputs, the
    normalising constants (2 pi)^(-Nx/2) / sqrt(det(cov_fp32)) stay fp32.
    """

    def __init__(self, means, covariances, weights, rng=None):
        means = np.asarray(means)
        self.means = means.astype(np.float32)
        self.covariances = np.asarray(covariances, dtype=np.float32)
        self.weights = np.asarray(weights, dtype=np.float32)
        self.inv_cov = np.linalg.inv(np.asarray(covariances, dtype=np.float64))
        self.nd, self.nx = means.shape
        self.consts = (2 * np.pi) ** (-self.nx / 2) / np.sqrt(np.linalg.det(self.covariances))
        self.rng = rng if rng is not None else np.random.default_rng(0)

    def pdf(self, x):
        """Mixture density at the rows of x (m, Nx) -> (m,) fp64 (:39-63):
        sum_d fp32(const_d * w_d) * exp(-0.5 * e_d^T inv_cov_d e_d), e_d = x - mean_d.
        Weights are used as given (they need not sum to one)."""
        x = np.atleast_2d(x)
        m = x.shape[0]
        total = np.zeros(m, dtype=np.float64)
        cw = self.consts * self.weights                      # fp32 product, as in the reference
        for d in range(self.nd):
            e = x - self.means[d][None, :]                   # fp64 if x is fp64, else fp32
            t = np.zeros((m, self.nx), dtype=np.float64)
            for j in range(self.nx):
                acc = e[:, 0] * self.inv_cov[d, 0, j]
                for i in range(1, self.nx):
                    acc = acc + e[:, i] * self.inv_cov[d, i, j]
                t[:, j] = acc
            q = t[:, 0] * e[:, 0]
            for j in range(1, self.nx):
                q = q + t[:, j] * e[:, j]
            term = cw[d] * np.exp(-0.5 * q)
            total = term if d == 0 else total + term
        return total

    def draw(self, shape=(1,)):
        """Samples of shape ``shape + (Nx,)`` fp32 (:65-97).  Like the reference the rows of one
        component are laid out contiguously (component 0 first); unlike it the normals come
        from Cholesky factors and this object's own Generator (the RNG stream itself can never
        match)."""
        if not isinstance(shape, tuple):
            shape = (shape,)
        size = int(np.prod(shape))
        comp = self.rng.choice(self.nd, size=size, p=self.weights / self.weights.sum())
        counts = np.bincount(comp, minlength=self.nd)
        out = np.empty((size, self.nx), dtype=np.float32)
        at = 0
        for d in range(self.nd):
            n = int(counts[d])
            chol = np.linalg.cholesky(self.covariances[d].astype(np.float64))
            out[at:at + n] = self.means[d] + self.rng.standard_normal((n, self.nx)) @ chol.T
            at += n
        return out.reshape(shape + (self.nx,))


class ReplayMixture(MixtureOracle):
    """A mixture whose draw() hands out pre-recorded noise arrays in order -- the role
    DeterministicGaussianSum plays in the reference's tests (DeterministicGaussianSum.py:32-65)."""

    def __init__(self, base, recorded):
        self.__dict__.update(base.__dict__)
        self._recorded = list(recorded)
        self._at = 0

    def draw(self, shape=(1,)):
        out = self._recorded[self._at]
        self._at += 1
        return np.array(out, dtype=np.float32, copy=True)


def bioreactor_noise(rng=None):
    """The two mixtures of sim_base.get_noise (sim_base.py:141-160)."""
    state = MixtureOracle(
        means=np.zeros((2, 5)),
        covariances=np.array([np.diag([1e-4, 1e-7, 1e-3, 1e-3, 1e-7]),
                              np.diag([1e-3, 1e-6, 1e-2, 1e-2, 1e-6])]),
        weights=np.array([0.75, 0.25]), rng=rng)
    meas = MixtureOracle(
        means=np.array([[1e-1, 0], [0, -1e-1]]),
        covariances=np.array([[[6e-2, 0], [0, 8e-2]], [[500, 100], [100, 700]]]),
        weights=np.array([0.85, 0.15]), rng=rng)
    return state, meas


# --------------------------------------------------------------------------------------
# Systematic resampling  (filter/particle.py:85-103 and the device variant :223-263)
# --------------------------------------------------------------------------------------
def normalised_cumsum(weights):
    """cumsum(w) / cumsum(w)[-1] in the dtype of ``weights`` (particle.py:89-90)."""
    c = np.cumsum(weights)
    c /= c[-1]
    return c


def systematic_index_loop(weights, r):
    """Sequential two-pointer sweep, literally particle.py:92-100: for each i the first k with
    cumsum[k] >= (i + r) / N."""
    c = normalised_cumsum(weights)
    n = len(c)
    idx = np.zeros(n, dtype=np.int64)
    k = 0
    for i in range(n):
        thr = (i + r) / n
        while c[k] < thr:
            k += 1
        idx[i] = k
    return idx


def systematic_index(weights, r):
    """Vectorised equivalent of systematic_index_loop: idx_i = #{k : c_k < u_i} with
    u_i = (i + r) / N evaluated in fp64 exactly as the loop does."""
    c = normalised_cumsum(weights)
    n = len(c)
    thr = (np.arange(n, dtype=np.float64) + r) / n
    idx = np.searchsorted(c, thr, side="left")
    return np.minimum(idx, n - 1).astype(np.int64)


# --------------------------------------------------------------------------------------
# Particle filter  (filter/particle.py:9-114)
# --------------------------------------------------------------------------------------
def round24(x):
    """fp64 values rounded to an fp32 MANTISSA (24 bits, round to nearest even) with their fp64 exponent range kept."""
    m, e = np.frexp(np.asarray(x, dtype=np.float64))
    return np.ldexp(m.astype(np.float32).astype(np.float64), e)


class ParticleFilterOracle:
    """Bootstrap particle filter on the host.  ``loop=True`` methods make one Python call per
    particle (the reference's CPU cost model); the ``*_vec`` methods do the same arithmetic
    with array operations."""

    def __init__(self, n, x0, state_pdf, measurement_pdf, rng=None, particles=None, weights_dtype=None):
        # weights_dtype: None = the reference's own dtypes (fp32 until the first resample, fp64 after it, particle.py:50,
        # :103); np.float32 = plain fp32 storage (sums fp64); "scaled32" = what the device classes store since round 4
        # (an fp32 MANTISSA per weight under a launch-uniform power of two, GSE_OP_WEIGHT_SCALE -- the
        # reference's exponent range, 24 bits of its mantissa; held here as fp64 values rounded to 24 bits, sums fp64) --
        # used where a free-running device loop is compared tick by tick
        self.weights_dtype = weights_dtype
        self.n = int(n)
        self.state_pdf = state_pdf
        self.measurement_pdf = measurement_pdf
        self.rng = rng if rng is not None else np.random.default_rng(0)
        self.particles = (np.array(particles, dtype=np.float32, copy=True)
                          if particles is not None else x0.draw(self.n))          # (:49)
        self.weights = np.full(self.n, 1 / self.n, dtype=np.float32)              # (:50)

    # --- predict: x_i += f(x_i, u, dt); x += noise                       (:54-67)
    def predict(self, u, dt, noise=None):
        for i in range(self.n):
            self.particles[i] += bioreactor_f(self.particles[i], u, dt)
        self.particles += self.state_pdf.draw(self.n) if noise is None else noise

    def predict_vec(self, u, dt, noise=None):
        self.particles += bioreactor_f_vec(self.particles, u, dt)      # fp64 add, rounded once to fp32
        self.particles += self.state_pdf.draw(self.n) if noise is None else np.asarray(noise, dtype=np.float32)

    # --- update: w_i *= pdf(z - g(x_i, u))                               (:69-83)
    def update(self, u, z):
        z = np.asarray(z, dtype=np.float64)
        for i in range(self.n):
            y = bioreactor_g(self.particles[i], u)
            e = z - y
            self.weights[i] *= self.measurement_pdf.pdf(e)[0]

    def likelihood_vec(self, z):
        z = np.asarray(z, dtype=np.float64)
        e = z[None, :] - bioreactor_g_vec(self.particles).astype(np.float64)
        return self.measurement_pdf.pdf(e)

    def update_vec(self, u, z):
        p = self.likelihood_vec(z)
        if self.weights_dtype == "scaled32":
            self.weights = round24(self.weights.astype(np.float64) * p)
        else:
            self.weights = (self.weights.astype(np.float64) * p).astype(self.weights.dtype)

    # --- resample                                                         (:85-103)
    def _sum_weights(self):
        """The weights as the sums see them: the reference's own dtype (fp32 weights give an fp32 cumsum and an fp32 dot
        product before the first resample, particle.py:89, :107), or -- device storage mode -- fp64 values of the stored fp32
        weights (every sum on the device is fp64)."""
        return self.weights if self.weights_dtype is None else self.weights.astype(np.float64)

    def ancestors(self, r, loop=False):
        w = self._sum_weights()
        return systematic_index_loop(w, r) if loop else systematic_index(w, r)

    def resample(self, r=None, loop=True, idx=None):
        """``idx``: ancestors to copy INSTEAD of the ones the weights give (replay of a recorded run; the caller compares)."""
        if r is None:
            r = self.rng.random()
        if idx is None:
            idx = self.ancestors(r, loop)
        self.particles = self.particles[idx]
        if self.weights_dtype == "scaled32":
            self.weights = np.full(self.n, np.float64(np.float32(1 / self.n)))
        else:
            self.weights = np.full(self.n, 1 / self.n, dtype=self.weights_dtype or np.float64)   # fp64 from here on in the reference
        return idx

    # --- moments                                                          (:105-114)
    def point_estimate(self):
        return self._sum_weights() @ self.particles

    def point_covariance(self):
        w = self._sum_weights()
        d = self.particles - (w @ self.particles)
        cov = d.T @ (d * w[:, None])
        return np.linalg.svd(cov, compute_uv=False)[0]

    def moments21(self):
        """[sum w, sum w x (5), upper triangle of sum w x x^T (15)] in fp64 -- the quantity
        the sharded filter all-reduces."""
        w = self.weights.astype(np.float64)
        x = self.particles.astype(np.float64)
        s2 = x.T @ (x * w[:, None])
        iu = np.triu_indices(NX)
        return np.concatenate([[w.sum()], w @ x, s2[iu]])


# --------------------------------------------------------------------------------------
# Gaussian-sum unscented Kalman filter  (filter/gs_ukf.py:9-183)
# --------------------------------------------------------------------------------------
def sigma_weights(nx=NX):
    """w_0 = 1/(1 + 5/4 Nx), w_j = 1/(2 Nx + 8/5), fp32 (gs_ukf.py:62-67)."""
    w = np.full(2 * nx + 1, 1 / (2 * nx + 8 / 5), dtype=np.float32)
    w[0] = 1 / (1 + 5 / 4 * nx)
    return w


class GSUKFOracle:
    """Bank of N Gaussians, each an unscented Kalman filter with 2 Nx + 1 unscaled sigma points;
    mixture weights are updated and resampled like particle weights."""

    def __init__(self, n, x0, state_pdf, measurement_pdf, rng=None, means=None, covariances=None, f_rows=None, g_rows=None):
        # f_rows(x (..., Nx), u, dt) -> increments, g_rows(x (..., Nx), u) -> outputs (..., 2): the model evaluated on every
        # row (default: the bioreactor; the reference calls self.f / self.g per sigma point, gs_ukf.py:91-93, :113-114)
        self.f_rows = f_rows if f_rows is not None else bioreactor_f_vec
        self.g_rows = g_rows if g_rows is not None else (lambda x, u: bioreactor_g_vec(x))
        self._builtin = f_rows is None
        self.n = int(n)
        self.state_pdf = state_pdf
        self.measurement_pdf = measurement_pdf
        self.rng = rng if rng is not None else np.random.default_rng(0)
        self.means = (np.array(means, dtype=np.float32, copy=True) if means is not None
                      else x0.draw(self.n))                                            # (:50)
        self.covariances = (np.array(covariances, dtype=np.float32, copy=True) if covariances is not None
                            else np.repeat(state_pdf.covariances[0][None], self.n, axis=0))  # (:52)
        self.weights = np.full(self.n, 1 / self.n, dtype=np.float32)                  # (:54)
        self.nx = self.means.shape[1]
        self.ny = NY
        self.n_sigma = 2 * self.nx + 1
        self.w_sigma = sigma_weights(self.nx)

    def sigma_points(self):
        """sigma_0 = mu, sigma_j = mu + L[:, j-1], sigma_{Nx+j} = mu - L[:, j-1] with L the lower
        Cholesky factor (retry with 1e-10 I on failure) -- gs_ukf.py:69-80."""
        try:
            low = np.linalg.cholesky(self.covariances)
        except np.linalg.LinAlgError:
            low = np.linalg.cholesky(self.covariances + 1e-10 * np.eye(self.nx))
        cols = np.swapaxes(low, 1, 2)                       # row j of cols = column j of L
        sig = np.repeat(self.means[:, None, :], self.n_sigma, axis=1)
        sig[:, 1:self.nx + 1, :] += cols
        sig[:, self.nx + 1:, :] -= cols
        return sig

    def predict(self, u, dt, noise=None, loop=True):
        """gs_ukf.py:82-103: propagate sigma points through f, add a noise DRAW to every sigma
        point (not Q), then mean = average_w(sigma), P = sum_j w_j d_j d_j^T."""
        sig = self.sigma_points()
        if loop and self._builtin:
            for a in range(self.n):
                for j in range(self.n_sigma):
                    sig[a, j] += bioreactor_f(sig[a, j], u, dt)
        else:
            sig += self.f_rows(sig, u, dt)
        sig += self.state_pdf.draw((self.n, self.n_sigma)) if noise is None else np.asarray(noise, dtype=np.float32)
        self.means = np.average(sig, axis=1, weights=self.w_sigma)
        sig -= self.means[:, None, :]
        self.covariances = np.swapaxes(sig, 1, 2) @ (sig * self.w_sigma[:, None])

    def update(self, u, z):
        """gs_ukf.py:105-149: local UKF update per Gaussian (P_yy carries no R; gain uses the
        pseudo-inverse), then the global weight update w *= pdf(z - g(mu_new))."""
        z = np.asarray(z, dtype=np.float64)
        sig = self.sigma_points()
        eta = self.g_rows(sig, u).astype(np.float64)                      # (N, 11, 2)
        eta_mean = np.average(eta, axis=1, weights=self.w_sigma)
        sig -= self.means[:, None, :]
        eta -= eta_mean[:, None, :]
        p_xy = np.swapaxes(sig, 1, 2) @ (eta * self.w_sigma[:, None])
        p_yy = np.swapaxes(eta, 1, 2) @ (eta * self.w_sigma[:, None])
        gain = p_xy @ np.linalg.pinv(p_yy)
        innov = z - eta_mean
        self.means += (gain @ innov[:, :, None]).squeeze(-1)
        self.covariances -= gain @ p_yy @ np.swapaxes(gain, 1, 2)
        e = z[None, :] - self.g_rows(self.means, u).astype(np.float64)
        self.weights *= self.measurement_pdf.pdf(e)

    def ancestors(self, r, loop=False):
        return systematic_index_loop(self.weights, r) if loop else systematic_index(self.weights, r)

    def resample(self, r=None, loop=True, idx=None):
        """gs_ukf.py:151-170.  ``idx``: ancestors to copy instead of the ones the weights give (replay of a recorded run)."""
        if r is None:
            r = self.rng.random()
        if idx is None:
            idx = self.ancestors(r, loop)
        self.means = self.means[idx]
        self.covariances = self.covariances[idx]
        self.weights = np.full(self.n, 1 / self.n)
        return idx

    def point_estimate(self):
        return self.weights @ self.means                                  # (:172-174)

    def point_covariance(self):
        """Largest singular value of sum_i w_i P_i + sum_i w_i (mu_i - mu)(mu_i - mu)^T (:176-183)."""
        cov_cov = np.sum(self.weights[:, None, None] * self.covariances, axis=0)
        d = self.means - (self.weights @ self.means)
        cov = cov_cov + d.T @ (d * self.weights[:, None])
        return np.linalg.svd(cov, compute_uv=False)[0]


def effective_sample_size(weights):
    w = np.asarray(weights, dtype=np.float64)
    return float(w.sum() ** 2 / np.sum(w * w))


__all__ = [n for n in dir() if not n.startswith("_") and n not in ("math", "np")]


# --------------------------------------------------------------------------------------
# Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11)
# and the word -> sample mapping of the device draw kernels (gpu_se_amd/csrc/gse_device.h), restated with
# NumPy integer arithmetic so that the tests can pin the device RNG bit for bit.
# --------------------------------------------------------------------------------------
_PHILOX_M0, _PHILOX_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_PHILOX_W0, _PHILOX_W1 = 0x9E3779B9, 0xBB67AE85
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(counter, key):
    """counter: (..., 4) uint32, key: (2,) or (..., 2) uint32 -> (..., 4) uint32."""
    c = np.asarray(counter, dtype=np.uint64) & _MASK32
    k = np.broadcast_to(np.asarray(key, dtype=np.uint64) & _MASK32, c.shape[:-1] + (2,)).copy()
    c0, c1, c2, c3 = c[..., 0], c[..., 1], c[..., 2], c[..., 3]
    k0, k1 = k[..., 0], k[..., 1]
    for _ in range(10):
        p0 = _PHILOX_M0 * c0
        p1 = _PHILOX_M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK32
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0 = (k0 + np.uint64(_PHILOX_W0)) & _MASK32
        k1 = (k1 + np.uint64(_PHILOX_W1)) & _MASK32
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


RNG_PURPOSE = {"prior": 0, "predict": 1, "draw": 2, "gsukf": 3}


def rng_words8(seed, step, purpose, elements, filter_id=0):
    """The 8 random words per element that the device kernels consume: two Philox blocks with counter
    (elem_lo, elem_hi, step_lo, (step_hi & 0xff) << 24 | purpose << 20 | sub << 16 | filter), key = seed."""
    e = np.asarray(elements, dtype=np.uint64)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint64)
    out = []
    for sub in (0, 1):
        hi = ((((step >> 32) & 0xFF) << 24) | ((purpose & 0xF) << 20) | (filter_id & 0xFFFF)) | (sub << 16)
        ctr = np.stack([e & _MASK32, e >> np.uint64(32), np.full_like(e, step & 0xFFFFFFFF), np.full_like(e, hi)], axis=-1)
        out.append(philox4x32_10(ctr, key))
    return np.concatenate(out, axis=-1)


def words_to_draw(words, mixture):
    """Map (M, 8) words to mixture samples the way mixture_draw5 does (Box-Muller on word pairs (0,1),
    (2,3), (4,5); component from the top 24 bits of word 7), in fp64 -- the device uses fp32 fast
    transcendental instructions, so compare with an absolute tolerance of a few 1e-6 sigma."""
    w = np.asarray(words, dtype=np.uint64)
    n = np.zeros((w.shape[0], 6))
    for p in range(3):
        u1 = (w[:, 2 * p].astype(np.float64) + 0.5) / 2.0 ** 32
        ang = w[:, 2 * p + 1].astype(np.float64) / 2.0 ** 32
        r = np.sqrt(-2.0 * np.log(u1))
        n[:, 2 * p] = r * np.cos(2 * np.pi * ang)
        n[:, 2 * p + 1] = r * np.sin(2 * np.pi * ang)
    uc = (w[:, 7] >> np.uint64(8)).astype(np.float64) / 2.0 ** 24
    cw = np.cumsum(mixture.weights.astype(np.float64))
    cw = (cw / cw[-1]).astype(np.float32).astype(np.float64)
    comp = np.zeros(w.shape[0], dtype=np.int64)
    for d in range(mixture.nd - 1):
        comp += (uc >= cw[d]).astype(np.int64)
    out = np.zeros((w.shape[0], mixture.nx))
    for d in range(mixture.nd):
        chol = np.linalg.cholesky(mixture.covariances[d].astype(np.float64)).astype(np.float32).astype(np.float64)
        sel = comp == d
        out[sel] = mixture.means[d].astype(np.float64) + n[sel, :mixture.nx] @ chol.T
    return out, comp


def gsukf_predict_noise(seed, step, n_gaussians, mixture, filter_id=0, n_sigma=11):
    """The state noise the Gaussian-sum filter's predict adds to the sigma points of Gaussians 0 .. n_gaussians - 1 at Philox
    step `step` (one draw per sigma point, filter/gs_ukf.py:87-91): (noise fp64 (N, n_sigma, nx), component (N, n_sigma)).

    Mapping of gsukf_point_noise / gsukf_pair_noise (gse_gsukf_kernels.h): the sigma points of Gaussian g come in pairs
    (2i, 2i + 1) that share three Philox blocks with counter (elem_lo, elem_hi, step_lo, (step_hi & 0xff) << 24 | purpose << 20
    | sub << 16 | filter), elem = g * n_sigma + 2i: block A (sub 0) is two Box-Muller pairs = normals 0..3 of point 2i,
    block C (sub 2) the same for point 2i + 1, block B (sub 1) one Box-Muller pair (words 0, 1) whose cosine half is normal 4 of
    point 2i and whose sine half normal 4 of point 2i + 1; word 3 of B selects the component of point 2i, word 2 that of point
    2i + 1 (top 24 bits).  The last point, without a partner, is the even half of a pair."""
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint64)
    g = np.arange(n_gaussians, dtype=np.uint64)

    def block(elem, sub):
        hi = ((((step >> 32) & 0xFF) << 24) | ((RNG_PURPOSE["gsukf"] & 0xF) << 20) | (filter_id & 0xFFFF)) | (sub << 16)
        ctr = np.stack([elem & _MASK32, elem >> np.uint64(32), np.full_like(elem, step & 0xFFFFFFFF), np.full_like(elem, hi)],
                       axis=-1)
        return philox4x32_10(ctr, key).astype(np.float64)

    def box_muller(a, b):
        r = np.sqrt(-2.0 * np.log((a + 0.5) / 2.0 ** 32))
        ang = 2 * np.pi * b / 2.0 ** 32
        return r * np.cos(ang), r * np.sin(ang)

    cw = np.cumsum(mixture.weights.astype(np.float64))
    cw = (cw / cw[-1]).astype(np.float32).astype(np.float64)
    chols = [np.linalg.cholesky(mixture.covariances[d].astype(np.float64)).astype(np.float32).astype(np.float64)
             for d in range(mixture.nd)]
    out = np.zeros((n_gaussians, n_sigma, mixture.nx))
    comps = np.zeros((n_gaussians, n_sigma), dtype=np.int64)
    for j in range(n_sigma):
        odd = j & 1
        elem = g * np.uint64(n_sigma) + np.uint64(j & ~1)
        own, shared = block(elem, 2 * odd), block(elem, 1)
        normals = np.zeros((n_gaussians, 5))
        normals[:, 0], normals[:, 1] = box_muller(own[:, 0], own[:, 1])
        normals[:, 2], normals[:, 3] = box_muller(own[:, 2], own[:, 3])
        normals[:, 4] = box_muller(shared[:, 0], shared[:, 1])[odd]
        sel = shared[:, 2 if odd else 3].astype(np.uint64)
        uc = (sel >> np.uint64(8)).astype(np.float64) / 2.0 ** 24
        comp = np.zeros(n_gaussians, dtype=np.int64)
        for d in range(mixture.nd - 1):
            comp += (uc >= cw[d]).astype(np.int64)
        comps[:, j] = comp
        for d in range(mixture.nd):
            rows = comp == d
            out[rows, j] = mixture.means[d].astype(np.float64) + normals[rows, :mixture.nx] @ chols[d].T
    return out, comps


def predict_noise(seed, step, elements, mixture, filter_id=0, element_offset=0):
    """The state noise the particle filter's predict stage adds to the particles with LOCAL indices `elements` (within
    their filter / shard, whose first particle has the global index `element_offset`) at Philox step `step`:
    (noise fp64 (M, nx), component (M,)).

    Mapping of pf_step_kernel (gse_device.h rng_block_of_group + mixture_draw5x4): the four particles  t + 256 p
    (p = 0..3) of a tile of 1024 form a group, group id = tile * 256 + t, and share SIX Philox blocks with counter
    (group | offset bits 32..35 << 28, offset low 32 bits, step_lo, (step_hi & 0xff) << 24 | purpose << 20 | sub << 16 |
    filter), sub = 0..5: blocks 0..4 are ten Box-Muller pairs = normals 0..19, particle p takes normals 5p .. 5p+4 and
    its component from word p of block 5 (top 24 bits)."""
    e = np.asarray(elements, dtype=np.uint64)
    group = (e >> np.uint64(10)) * np.uint64(256) + (e & np.uint64(255))
    slot = ((e >> np.uint64(8)) & np.uint64(3)).astype(np.int64)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint64)
    off = int(element_offset)
    c0 = (group | np.uint64(((off >> 32) & 0xF) << 28)) & _MASK32
    blocks = []
    for sub in range(6):
        hi = ((((step >> 32) & 0xFF) << 24) | ((RNG_PURPOSE["predict"] & 0xF) << 20) | (filter_id & 0xFFFF)) | (sub << 16)
        ctr = np.stack([c0, np.full_like(e, off & 0xFFFFFFFF), np.full_like(e, step & 0xFFFFFFFF), np.full_like(e, hi)],
                       axis=-1)
        blocks.append(philox4x32_10(ctr, key))
    words = np.concatenate(blocks[:5], axis=-1).astype(np.float64)            # (M, 20): ten Box-Muller pairs
    normals = np.zeros((len(e), 20))
    for pair in range(10):
        u1 = (words[:, 2 * pair] + 0.5) / 2.0 ** 32
        ang = words[:, 2 * pair + 1] / 2.0 ** 32
        r = np.sqrt(-2.0 * np.log(u1))
        normals[:, 2 * pair] = r * np.cos(2 * np.pi * ang)
        normals[:, 2 * pair + 1] = r * np.sin(2 * np.pi * ang)
    rows = np.arange(len(e))
    mine = normals[rows[:, None], 5 * slot[:, None] + np.arange(5)[None, :]]  # (M, 5)
    uc = (blocks[5][rows, slot] >> np.uint64(8)).astype(np.float64) / 2.0 ** 24
    cw = np.cumsum(mixture.weights.astype(np.float64))
    cw = (cw / cw[-1]).astype(np.float32).astype(np.float64)
    comp = np.zeros(len(e), dtype=np.int64)
    for d in range(mixture.nd - 1):
        comp += (uc >= cw[d]).astype(np.int64)
    out = np.zeros((len(e), mixture.nx))
    for d in range(mixture.nd):
        chol = np.linalg.cholesky(mixture.covariances[d].astype(np.float64)).astype(np.float32).astype(np.float64)
        sel = comp == d
        out[sel] = mixture.means[d].astype(np.float64) + mine[sel, :mixture.nx] @ chol.T
    return out, comp
