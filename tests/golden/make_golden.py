#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE NumPy classes.

Runs only in the build container (needs /root/reference); the .npz files it writes are
committed and travel to the GPU box, this script's inputs (the reference) never do.

The reference imports ``cupy``, ``numba``, ``numba.cuda`` and ``osqp`` unconditionally at
module top (filter/particle.py:1-6, gaussian_sum_dist/MultivariateGaussianSum.py:2-3,
controller.py:6) although its NumPy (CPU) classes never call into them.  Those four
packages are not installed here, so inert placeholder modules are registered first; every
number written below is computed by the reference's own NumPy code paths
(ParticleFilter, GaussianSumUnscentedKalmanFilter, MultivariateGaussianSum(library=numpy),
Bioreactor.homeostatic_DEs / static_outputs, sim_base.get_parts(gpu=False) / get_noise).

Fixture list:
  G1 model_fg.npz       f, g on 256 states x 4 (u, dt)
  G2 mixture_pdf.npz    pdf of both mixtures on 256 points (down to exp(-700) underflow)
  G3 resample.npz       systematic resample cases with recorded r -> ancestor index
  G4 pf_trace.npz       ParticleFilter N=1024, 20 cycles: inputs, recorded noise/r, states
  G5 gsukf_trace.npz    GS-UKF N in {1,7,64}: sigma points, predict, update, resample, estimates
  G6 pf_stat.npz        posterior-mean envelope over independent seeds (statistical parity)
  G7 constants.npz      mixture constants, inverse covariances, sigma weights, X_ss
  G8 linear_model.npz   LinearModel.create_LinearModel of the bioreactor (get_parts wiring), dt_control in {1, 0.1}
  G10 closed_loop.npz   sim_base.Simulation.simulate (the reference's own loop text) on the NumPy classes: recorded random
                        inputs and every record of the loop, particle filter and GS-UKF, control every tick / every 10th
  G11c loop_envelope_gsf.npz  256 more seeds of G11's gsf64_c01 (GS-UKF, control every tick): a sample for a two-sample test
  G11 loop_envelope.npz the reference's Simulation run to end_time = 50 for 32 seeds per configuration (statistical envelope
                        of the closed loop: ITSE, per-tick estimation error, outputs, inputs)
  G9 mpc_qp.npz         controller.MPC: the QP it hands to OSQP (H, q, A, l, u) for several horizons / bounds, and the
                        l, u vectors after step() has written x0, u_{-1} and the bias into them.  (osqp is not
                        installed: the SOLUTION of these QPs is not a reference output and is not stored.)
"""
import os
import sys
import types

import numpy

REF = os.environ.get("GSE_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_placeholders():
    cupy = types.ModuleType("cupy")
    numba = types.ModuleType("numba")
    cuda = types.ModuleType("numba.cuda")

    def _passthrough(*args, **kwargs):
        if len(args) == 1 and callable(args[0]) and not kwargs:
            return args[0]
        return lambda fn: fn

    cuda.jit = _passthrough
    numba.guvectorize = _passthrough
    numba.cuda = cuda
    osqp = types.ModuleType("osqp")

    class OSQP:  # MPC is constructed by get_parts but never stepped here
        def setup(self, *a, **k):
            pass

        def update(self, *a, **k):
            pass

        def solve(self):
            raise RuntimeError("osqp is not available offline")

    osqp.OSQP = OSQP
    sys.modules.update({"cupy": cupy, "numba": numba, "numba.cuda": cuda, "osqp": osqp})


_install_placeholders()
sys.path.insert(0, REF)
import sim_base  # noqa: E402  (reference)
import filter as ref_filter  # noqa: E402  (reference)
import gaussian_sum_dist as ref_gsd  # noqa: E402
import model as ref_model  # noqa: E402

f_ref = ref_model.Bioreactor.homeostatic_DEs
g_ref = ref_model.Bioreactor.static_outputs


def save(name, **arrays):
    path = os.path.join(OUT, name)
    numpy.savez_compressed(path, **arrays)
    print("wrote", name, {k: getattr(v, "shape", None) for k, v in arrays.items()})


def x_ss():
    bio, _, _, _ = sim_base.get_parts(N_particles=2, gpu=False)
    return numpy.array(bio.X, dtype=numpy.float64)


class Recorder:
    """Wraps a mixture's draw() so that every returned noise array is kept."""

    def __init__(self, dist):
        self.dist = dist
        self.log = []
        self._orig = dist.draw
        dist.draw = self._draw

    def _draw(self, *a, **k):
        out = self._orig(*a, **k)
        self.log.append(numpy.array(out, copy=True))
        return out


class RandRecorder:
    def __init__(self):
        self.log = []
        self._orig = numpy.random.rand

    def __enter__(self):
        def rand(*a):
            v = self._orig(*a)
            self.log.append(v)
            return v

        numpy.random.rand = rand
        return self

    def __exit__(self, *exc):
        numpy.random.rand = self._orig


# --------------------------------------------------------------------------- G7
def gen_constants(X):
    sp, mp = sim_base.get_noise(lib=numpy)
    gf = ref_filter.GaussianSumUnscentedKalmanFilter(
        f=f_ref, g=g_ref, N_particles=2, x0=sp, state_pdf=sp, measurement_pdf=mp)
    save("constants.npz",
         x_ss=X,
         state_means=sp.means, state_covs=sp.covariances, state_weights=sp.weights,
         state_consts=sp._constants, state_inv=sp._inverse_covariances,
         meas_means=mp.means, meas_covs=mp.covariances, meas_weights=mp.weights,
         meas_consts=mp._constants, meas_inv=mp._inverse_covariances,
         w_sigma=gf._w_sigma)


# --------------------------------------------------------------------------- G1
def gen_model(X):
    rng = numpy.random.default_rng(101)
    xs = X[None, :] * (1 + 0.3 * rng.uniform(-1, 1, size=(256, 5)))
    xs[:, 3] = rng.uniform(-0.05, 0.3, size=256)          # Ce is 0 at X_ss
    xs[:16, 0] *= -1                                        # negative Cg  -> clamp
    xs[16:32, 2] *= -1                                      # negative Cfa -> clamp
    xs[32:40, 1] = 0.0                                      # Cx = 0
    xs[40:48, 0] = 0.0                                      # Cg = 0 (0/(0.01+0))
    xs[48:56, 4] *= -1                                      # Ch not clamped
    xs[56:64, 0] = rng.uniform(0, 0.02, size=8)             # near the Monod knee
    xs[64:96, 4] = rng.uniform(-40, 60, size=32)            # drive theta1_req across its clamps
    xs32 = xs.astype(numpy.float32)
    cases = [(numpy.array([0.06, 0.2]), 0.1),
             (numpy.array([0.05, 0.15]), 1.0),
             (numpy.array([0.0, 0.0]), 0.1002),
             (numpy.array([0.0997, 0.1993]), 1.0)]
    us = numpy.array([c[0] for c in cases])
    dts = numpy.array([c[1] for c in cases])
    # f is evaluated on float32 rows exactly as ParticleFilter.predict does (particle.py:65-66)
    df = numpy.zeros((len(cases), 256, 5), dtype=numpy.float64)
    gy = numpy.zeros((len(cases), 256, 2), dtype=numpy.float64)
    x_after = numpy.zeros((len(cases), 256, 5), dtype=numpy.float32)
    for c, (u, dt) in enumerate(cases):
        work = xs32.copy()
        for i in range(256):
            inc = f_ref(xs32[i], u, dt)
            df[c, i] = numpy.array([float(v) for v in inc])
            work[i] += inc
            gy[c, i] = numpy.array([float(v) for v in g_ref(xs32[i], u)])
        x_after[c] = work
    # same states handed over as float64 (the SURVEY's known-answer form)
    df64 = numpy.zeros_like(df)
    for c, (u, dt) in enumerate(cases):
        for i in range(256):
            df64[c, i] = numpy.array([float(v) for v in f_ref(xs32[i].astype(numpy.float64), u, dt)])
    save("model_fg.npz", x=xs32, u=us, dt=dts, f_inc=df, f_inc_f64in=df64, g_out=gy, x_after=x_after)


# --------------------------------------------------------------------------- G2
def gen_pdf():
    sp, mp = sim_base.get_noise(lib=numpy)
    rng = numpy.random.default_rng(202)
    # measurement mixture (Nx=2): residuals from tiny to far enough for exp() underflow
    scale = numpy.concatenate([numpy.full(64, 0.3), numpy.full(64, 5.0),
                               numpy.full(64, 80.0), numpy.full(64, 1500.0)])
    e_meas = rng.normal(size=(256, 2)) * scale[:, None]
    e_meas[:5] = numpy.array([[0.1, 0], [0, -0.1], [3, -2], [30, -40], [0.3, -0.5]])
    p_meas = mp.pdf(e_meas)
    # and the float32 route the GPU class takes (particle.py:290-293): es is float32
    p_meas_f32 = mp.pdf(e_meas.astype(numpy.float32))
    sd = numpy.sqrt(numpy.array([1e-3, 1e-6, 1e-2, 1e-2, 1e-6]))
    mult = numpy.concatenate([numpy.full(128, 1.0), numpy.full(64, 6.0), numpy.full(64, 40.0)])
    e_state = rng.normal(size=(256, 5)) * sd[None, :] * mult[:, None]
    p_state = sp.pdf(e_state)
    # non-normalised weights as in tests/ParticleFilter_test.py:41-52
    toy = ref_gsd.MultivariateGaussianSum(
        means=numpy.array([[1, 0], [0, -1]]),
        covariances=numpy.array([[[0.1, 0], [0, 0.1]], [[0.2, 0.01], [0.01, 0.005]]]),
        weights=numpy.array([0.7, 0.9]), library=numpy)
    e_toy = rng.normal(size=(64, 2))
    p_toy = toy.pdf(e_toy)
    save("mixture_pdf.npz", e_meas=e_meas, p_meas=p_meas, p_meas_f32=p_meas_f32,
         e_state=e_state, p_state=p_state,
         toy_means=toy.means, toy_covs=toy.covariances, toy_weights=toy.weights,
         toy_consts=toy._constants, toy_inv=toy._inverse_covariances, e_toy=e_toy, p_toy=p_toy)


# --------------------------------------------------------------------------- G3
def ref_resample_idx(weights, r):
    """Runs ParticleFilter.resample (particle.py:85-103) on a filter whose particles are
    their own row numbers, so that the gathered particles reveal the ancestor index."""
    n = len(weights)
    pf = ref_filter.ParticleFilter.__new__(ref_filter.ParticleFilter)
    pf.N_particles = n
    pf.particles = numpy.arange(n, dtype=numpy.int64)[:, None].repeat(5, axis=1)
    pf.weights = numpy.array(weights, copy=True)
    orig = numpy.random.rand
    numpy.random.rand = lambda *a: r
    try:
        pf.resample()
    finally:
        numpy.random.rand = orig
    return pf.particles[:, 0].copy()


def gen_resample():
    rng = numpy.random.default_rng(303)
    out = {}
    cases = []
    cases.append(("known8", numpy.array([.05, .30, 0, .10, .25, 0, .20, .10]), 0.6964691855978616))
    cases.append(("uniform8", numpy.full(8, 1 / 8), 0.0))
    cases.append(("uniform1000", numpy.full(1000, 1 / 1000), 0.37))
    w = rng.random(1000)
    cases.append(("random1000", w / w.sum(), rng.random()))
    w = numpy.zeros(1000)
    w[617] = 1.0
    cases.append(("onehot1000", w, 0.5))
    w = rng.random(4096)
    w[::2] = 0
    cases.append(("zeros_interleaved4096", w, rng.random()))
    w = rng.random(4096) ** 8
    cases.append(("skewed4096", w, rng.random()))
    w = numpy.exp(-0.5 * ((numpy.arange(4096) - 1000) / 3.0) ** 2) + 1e-30
    cases.append(("peaked4096", w, 0.999999))
    w = rng.random(5000)
    cases.append(("unnormalised5000", w * 37.5, rng.random()))
    cases.append(("single1", numpy.array([0.3]), 0.25))
    w = rng.random(2049)
    cases.append(("random2049", w, rng.random()))
    names = []
    for name, w, r in cases:
        # weights that are exactly float32-representable, handed to the reference as float64
        # (the dtype ParticleFilter.weights has after its first resample, particle.py:103)
        w32 = w.astype(numpy.float32)
        idx = ref_resample_idx(w32.astype(numpy.float64), float(r))
        out[name + "_w"] = w32
        out[name + "_r"] = numpy.float64(r)
        out[name + "_idx"] = idx
        names.append(name)
    out["names"] = numpy.array(names)
    save("resample.npz", **out)


# --------------------------------------------------------------------------- G4
def plant_step(x_true, u, dt, sp, mp):
    """A realistic measurement sequence: Euler plant step + state noise, z = g(x) + noise."""
    x_true = x_true + numpy.array([float(v) for v in f_ref(x_true, u, dt)])
    x_true = x_true + sp.draw().squeeze().astype(numpy.float64)
    z = numpy.array([float(v) for v in g_ref(x_true, u)]) + mp.draw().squeeze().astype(numpy.float64)
    return x_true, z


def gen_pf_trace(X, n=1024, cycles=20, seed=404):
    numpy.random.seed(seed)
    _, _, _, pf = sim_base.get_parts(N_particles=n, gpu=False)
    plant_sp, plant_mp = sim_base.get_noise(lib=numpy)
    rec = Recorder(pf.state_pdf)
    x_true = X.copy()
    keys = dict(x0=pf.particles.copy(), w0=pf.weights.copy())
    us, zs, dts, rs = [], [], [], []
    p_pred, p_res, w_upd, noises, idxs, est, cov = [], [], [], [], [], [], []
    for c in range(cycles):
        u = numpy.array([numpy.random.uniform(0.03, 0.09), numpy.random.uniform(0.1, 0.3)])
        dt = 0.1 if c % 2 == 0 else 1.0
        x_true, z = plant_step(x_true, u, dt, plant_sp, plant_mp)
        pf.predict(u, dt)
        p_pred.append(pf.particles.copy())
        noises.append(rec.log[-1])
        pf.update(u, z)
        w_upd.append(numpy.array(pf.weights, dtype=numpy.float64))
        # ancestor index: rerun the reference's own resample on row numbers with the same r
        with RandRecorder() as rr:
            pf.resample()
        r = float(rr.log[-1])
        idxs.append(ref_resample_idx(w_upd[-1] if c > 0 else pf_weights_f32(w_upd[-1]), r))
        p_res.append(pf.particles.copy())
        est.append(numpy.array(pf.point_estimate(), dtype=numpy.float64))
        cov.append(float(pf.point_covariance()))
        us.append(u), zs.append(z), dts.append(dt), rs.append(r)
    keys.update(u=numpy.array(us), z=numpy.array(zs), dt=numpy.array(dts), r=numpy.array(rs),
                noise=numpy.array(noises), after_predict=numpy.array(p_pred),
                w_after_update=numpy.array(w_upd), idx=numpy.array(idxs),
                after_resample=numpy.array(p_res), estimate=numpy.array(est), cov_sv=numpy.array(cov))
    save("pf_trace.npz", **keys)


def pf_weights_f32(w):
    # in the first cycle ParticleFilter.weights is still float32 (particle.py:50)
    return numpy.asarray(w, dtype=numpy.float32)


# --------------------------------------------------------------------------- G5
def gen_gsukf(X, seed=505):
    out = {}
    for n in (1, 7, 64):
        numpy.random.seed(seed + n)
        _, _, _, gf = sim_base.get_parts(N_particles=n, gpu=False, pf=False)
        plant_sp, plant_mp = sim_base.get_noise(lib=numpy)
        rec = Recorder(gf.state_pdf)
        x_true = X.copy()
        tag = "n%d_" % n
        out[tag + "means0"] = gf.means.copy()
        out[tag + "covs0"] = gf.covariances.copy()
        out[tag + "w0"] = gf.weights.copy()
        out[tag + "sigmas0"] = gf._get_sigma_points().copy()
        cyc = 6
        store = {k: [] for k in ("u", "z", "dt", "r", "noise", "m_pred", "P_pred", "sig_upd", "m_upd",
                                 "P_upd", "w_upd", "idx", "m_res", "P_res", "est", "cov_sv")}
        for c in range(cyc):
            u = numpy.array([numpy.random.uniform(0.03, 0.09), numpy.random.uniform(0.1, 0.3)])
            dt = 0.1 if c % 2 == 0 else 1.0
            x_true, z = plant_step(x_true, u, dt, plant_sp, plant_mp)
            gf.predict(u, dt)
            store["noise"].append(rec.log[-1])
            store["m_pred"].append(gf.means.copy())
            store["P_pred"].append(gf.covariances.copy())
            store["sig_upd"].append(gf._get_sigma_points().copy())
            gf.update(u, z)
            store["m_upd"].append(gf.means.copy())
            store["P_upd"].append(gf.covariances.copy())
            store["w_upd"].append(numpy.array(gf.weights, dtype=numpy.float64))
            w_before = gf.weights.copy()
            with RandRecorder() as rr:
                gf.resample()
            r = float(rr.log[-1])
            store["idx"].append(ref_resample_idx(w_before, r))
            store["m_res"].append(gf.means.copy())
            store["P_res"].append(gf.covariances.copy())
            store["est"].append(numpy.array(gf.point_estimate(), dtype=numpy.float64))
            store["cov_sv"].append(float(gf.point_covariance()))
            store["u"].append(u), store["z"].append(z), store["dt"].append(dt), store["r"].append(r)
        for k, v in store.items():
            out[tag + k] = numpy.array(v)
    # the SURVEY's single-Gaussian known answer (update then noise-free predict)
    sp, mp = sim_base.get_noise(lib=numpy)
    x0, _ = sim_base.get_noise(lib=numpy)
    x0.means = X[None, :].astype(numpy.float32)
    numpy.random.seed(1)
    gf = ref_filter.GaussianSumUnscentedKalmanFilter(f=f_ref, g=g_ref, N_particles=2, x0=x0,
                                                      state_pdf=sp, measurement_pdf=mp)
    gf.means = numpy.repeat(X[None, :].astype(numpy.float32), 2, axis=0)
    u = numpy.array([0.06, 0.2])
    z = numpy.array([280.3, 610.95])
    gf.update(u, z)
    out["ka_m_upd"], out["ka_P_upd"], out["ka_w_upd"] = gf.means.copy(), gf.covariances.copy(), gf.weights.copy()
    zero = numpy.zeros((2, 11, 5), dtype=numpy.float32)
    sp.draw = lambda *a, **k: zero
    gf.predict(u, 0.1)
    out["ka_m_pred"], out["ka_P_pred"] = gf.means.copy(), gf.covariances.copy()
    out["ka_u"], out["ka_z"] = u, z
    save("gsukf_trace.npz", **out)


# --------------------------------------------------------------------------- G6
def _stat_worker(job):
    """One seed of the reference particle filter on the fixed (u, z) sequence (runs in a worker process)."""
    n, s, us, zs = job
    numpy.random.seed(7000 + 13 * s + n)
    _, _, _, pf = sim_base.get_parts(N_particles=n, gpu=False)
    est, cov = numpy.zeros((len(us), 5)), numpy.zeros(len(us))
    for c in range(len(us)):
        pf.predict(us[c], 0.1)
        pf.update(us[c], zs[c])
        pf.resample()
        est[c] = pf.point_estimate()
        cov[c] = pf.point_covariance()
    return n, s, est, cov


def gen_pf_stat(X, seed=606):
    """Posterior-mean envelope of the reference PF over independent seeds on ONE fixed
    (u, z) sequence; the Philox-driven HIP filter must land inside it (3 sigma).  SURVEY 8c: 64 seeds, 50 cycles,
    N = 1024 and N = 16384 (the reference's per-particle Python loops: 25 s per seed at 16384, spread over the cores)."""
    import multiprocessing
    rng = numpy.random.default_rng(seed)
    cycles, seeds = 50, 64
    plant_sp, plant_mp = sim_base.get_noise(lib=numpy)
    numpy.random.seed(seed)
    x_true = X.copy()
    us, zs = [], []
    for c in range(cycles):
        u = numpy.array([rng.uniform(0.03, 0.09), rng.uniform(0.1, 0.3)])
        x_true, z = plant_step(x_true, u, 0.1, plant_sp, plant_mp)
        us.append(u), zs.append(z)
    us, zs = numpy.array(us), numpy.array(zs)
    out = dict(u=us, z=zs, dt=numpy.float64(0.1))
    jobs = [(n, s, us, zs) for n in (16384, 1024) for s in range(seeds)]
    with multiprocessing.Pool(min(8, os.cpu_count() or 1)) as pool:
        done = 0
        res = {}
        for n, s, est, cov in pool.imap_unordered(_stat_worker, jobs):
            res[(n, s)] = (est, cov)
            done += 1
            if done % 8 == 0:
                print("  stat %d/%d" % (done, len(jobs)), flush=True)
    for n in (1024, 16384):
        out["est_n%d" % n] = numpy.array([res[(n, s)][0] for s in range(seeds)])
        out["cov_n%d" % n] = numpy.array([res[(n, s)][1] for s in range(seeds)])
    save("pf_stat.npz", **out)


def _ref_linear_model(dt_control):
    import model.LinearModel  # noqa: F401  (reference)
    ss = ref_model.Bioreactor.find_SS
    shape = numpy.array([260 / 180, 640 / 24.6, 1000 / 116, 0, 0])
    bioreactor = ref_model.Bioreactor(X0=ss(numpy.array([0.06, 0.2]), shape), high_N=False)
    lin = ref_model.LinearModel.create_LinearModel(bioreactor, x_bar=ss(numpy.array([0.04, 0.1]), shape),
                                                   u_bar=numpy.array([0.04, 0.1]), T=dt_control)
    return bioreactor, lin


def gen_linear_model():
    out = {}
    for tag, dt_control in (("dt1", 1.0), ("dt01", 0.1)):
        _, lin = _ref_linear_model(dt_control)
        for name in ("A", "B", "C", "D", "x_bar", "u_bar", "f_bar", "y_bar"):
            out["%s_full_%s" % (tag, name)] = numpy.array(getattr(lin, name), dtype=numpy.float64)
        lin.select_subset(states=[0, 2], inputs=[0, 1], outputs=[0, 2])
        for name in ("A", "B", "C", "D", "x_bar", "u_bar", "f_bar", "y_bar"):
            out["%s_sub_%s" % (tag, name)] = numpy.array(getattr(lin, name), dtype=numpy.float64)
        out["%s_ysp" % tag] = lin.yn2d(numpy.array([280, 850]), subselect=False)
    save("linear_model.npz", **out)


def gen_mpc_qp():
    import controller  # noqa: E402  (reference; osqp is the inert placeholder above)
    _, lin = _ref_linear_model(1.0)
    lin.select_subset(states=[0, 2], inputs=[0, 1], outputs=[0, 2])
    ysp = lin.yn2d(numpy.array([280, 850]), subselect=False)
    ub_default = [numpy.array([0, numpy.inf]) - lin.u_bar[0], numpy.array([0, numpy.inf]) - lin.u_bar[1]]
    cases = {
        "small": dict(P=6, M=3, u_bounds=ub_default),
        "square": dict(P=5, M=5, u_bounds=ub_default),
        "free": dict(P=8, M=4),
        "boxed": dict(P=8, M=4, u_bounds=[(-0.03, 0.05), (-0.08, 0.1)], u_step_bounds=[(-0.01, 0.01), (-0.02, 0.02)],
                      y_bounds=[(-3.0, 3.0), (-400.0, 400.0)]),
        "parts": dict(P=300, M=200, u_bounds=ub_default),          # get_parts(dt_control=1), sim_base.py:75-86
    }
    rng = numpy.random.default_rng(909)
    out = {"names": numpy.array(sorted(cases))}
    for name, kw in cases.items():
        K = controller.MPC(Q=numpy.diag([0.1, 1]), R=numpy.diag([1, 1]), lin_model=lin, ysp=ysp, **kw)
        A = K.A_matrix.tocoo()
        H = K.H.tocoo()
        out[name + "_PM"] = numpy.array([kw["P"], kw["M"]])
        out[name + "_shape"] = numpy.array(A.shape)
        for tag, mat in (("A", A), ("H", H)):
            out["%s_%s_row" % (name, tag)] = mat.row.astype(numpy.int32)
            out["%s_%s_col" % (name, tag)] = mat.col.astype(numpy.int32)
            out["%s_%s_val" % (name, tag)] = mat.data.astype(numpy.float64)
        out[name + "_q"] = numpy.array(K.q)
        out[name + "_l0"] = K.l_matrix.copy()
        out[name + "_u0"] = K.u_matrix.copy()
        # two consecutive step() calls: the placeholder's solve() raises after l, u have been written
        x0 = rng.normal(size=2) * numpy.array([0.05, 0.5])
        um1 = rng.normal(size=2) * 0.01
        y0 = rng.normal(size=2) * numpy.array([5.0, 50.0])
        for k, y_pred in enumerate((None, numpy.array([1.5, -20.0]))):
            K.y_predicted = y_pred
            try:
                K.step(x0, um1, y0)
            except RuntimeError:
                pass
            out["%s_step%d_l" % (name, k)] = K.l_matrix.copy()
            out["%s_step%d_u" % (name, k)] = K.u_matrix.copy()
        out[name + "_x0"] = x0
        out[name + "_um1"] = um1
        out[name + "_y0"] = y0
        out[name + "_ypred"] = numpy.array([1.5, -20.0])
    save("mpc_qp.npz", **out)


# --------------------------------------------------------------------------- G10
class _HostArray(numpy.ndarray):
    """What a cupy array is to the reference's loop: something with .get() (sim_base.py:281-284)."""

    def get(self):
        return numpy.asarray(self)


class _KktOsqp:
    """Stands where osqp.OSQP stands in controller.py:232-236, :264-267 (osqp >= 0.6.1 is not installed and cannot be
    fetched).  mode 'exact': solve() returns the exact minimiser of the QP the reference handed to setup() / update()
    (oracle.mpc_oracle.KKTSolver) with status 1; mode 'fail': solve() reports a non-solved status, which drives the
    reference's own except branch (sim_base.py:271-273).  The SOLUTION is therefore not a reference output; everything
    around it -- what the loop feeds the controller, what it does with the answer, the bias bookkeeping -- is."""
    mode = "exact"

    def setup(self, H, q, A, l, u, verbose=False):
        ROOT = os.path.dirname(os.path.dirname(OUT))
        if ROOT not in sys.path:
            sys.path.append(ROOT)
        from oracle import mpc_oracle
        self._solver = mpc_oracle.KKTSolver(H, q, A, l, u)
        self._l, self._u = numpy.array(l, dtype=float), numpy.array(u, dtype=float)

    def update(self, l=None, u=None):
        self._l, self._u = numpy.array(l, dtype=float), numpy.array(u, dtype=float)

    def solve(self):
        info = types.SimpleNamespace(status_val=1, status="solved")
        if self.mode == "fail":
            return types.SimpleNamespace(x=None, info=types.SimpleNamespace(status_val=-2, status="maximum iterations reached"))
        return types.SimpleNamespace(x=self._solver.solve(self._l, self._u), info=info)


def _host_noise(lib=numpy, deterministic=False, _orig=sim_base.get_noise):
    """sim_base.get_noise on NumPy whose draws answer .get() like device arrays do."""
    pair = _orig(numpy, deterministic)
    for dist in pair:
        inner = dist.draw
        dist.draw = (lambda inner: lambda *a, **k: numpy.asarray(inner(*a, **k)).view(_HostArray))(inner)
    return pair


def gen_closed_loop(seed=1010):
    """The reference's own Simulation.__init__ / simulate (sim_base.py:207-309) executed on its NumPy classes: get_parts
    is called with gpu=False (the loop hard-wires the default gpu=True, :215-219), get_noise hands out NumPy mixtures whose
    draws have .get(), scipy.integrate.simps (removed in SciPy 1.14; the reference pins >= 1.4.1) is the rule of
    oracle/closed_loop.simps_avg, and osqp is _KktOsqp above.  Recorded: every random input and every record the loop keeps."""
    import controller  # noqa: F401  (reference)
    import scipy.integrate
    ROOT = os.path.dirname(os.path.dirname(OUT))
    if ROOT not in sys.path:
        sys.path.append(ROOT)
    from oracle import closed_loop
    keep = (sim_base.get_parts, sim_base.get_noise, sys.modules["osqp"].OSQP, getattr(scipy.integrate, "simps", None))
    sim_base.get_noise = _host_noise
    sim_base.get_parts = lambda dt_control=1, N_particles=2 * 15, gpu=False, pf=True, _orig=keep[0]: _orig(
        dt_control=dt_control, N_particles=N_particles, gpu=False, pf=pf)
    sys.modules["osqp"].OSQP = _KktOsqp
    controller.osqp.OSQP = _KktOsqp
    scipy.integrate.simps = lambda y, x: closed_loop.simps_avg(y, x)
    cases = {
        #           N    dt_control dt_predict end pf     osqp
        "pf_c1":   (256, 1.0,       0.1,       6,  True,  "exact"),     # predict every tick, control every 10th
        "pf_c1p1": (256, 1.0,       1.0,       6,  True,  "exact"),     # predict and control every 10th tick
        "pf_c01":  (256, 0.1,       0.1,       3,  True,  "exact"),     # control every tick (P = 2999, M = 1999)
        "pf_fail": (256, 1.0,       0.1,       4,  True,  "fail"),      # the except branch on every control tick
        "gsf_c1":  (8,   1.0,       0.1,       4,  False, "exact"),
        "gsf_c01": (6,   0.1,       0.1,       2,  False, "exact"),
    }
    out = {"names": numpy.array(sorted(cases))}
    try:
        for k, (name, (n, dtc, dtp, end, pf, mode)) in enumerate(sorted(cases.items())):
            _KktOsqp.mode = mode
            numpy.random.seed(seed + k)
            sim = sim_base.Simulation(n, dtc, dtp, end_time=end, pf=pf)
            est0 = dict(particles0=numpy.array(sim.f.particles)) if pf else dict(
                means0=numpy.array(sim.f.means), covs0=numpy.array(sim.f.covariances))
            rec_f, rec_ps, rec_pm = Recorder(sim.f.state_pdf), Recorder(sim.state_pdf), Recorder(sim.measurement_pdf)
            # the ancestors of every resample: the reference's own resample rerun on row numbers with the same weights and r
            idx_log, r_log = [], []

            def recording_resample(_orig=sim.f.resample, _f=sim.f):
                w_before = numpy.array(_f.weights, copy=True)
                with RandRecorder() as rr:
                    _orig()
                r_log.append(float(rr.log[-1]))
                idx_log.append(ref_resample_idx(w_before, r_log[-1]))

            sim.f.resample = recording_resample
            sim.simulate()
            tag = name + "_"
            out[tag + "ancestors"] = numpy.array(idx_log, dtype=numpy.int32)
            out[tag + "cfg"] = numpy.array([n, dtc, dtp, end, float(pf)])
            for key, val in est0.items():
                out[tag + key] = val
            out[tag + "predict_noise"] = numpy.array(rec_f.log)
            out[tag + "plant_state"] = numpy.array(rec_ps.log).reshape(len(rec_ps.log), 5)
            out[tag + "plant_meas"] = numpy.array(rec_pm.log).reshape(len(rec_pm.log), 2)
            out[tag + "r"] = numpy.array(r_log)
            for key in ("us", "xs", "ys", "ys_meas", "xs_f", "ys_f", "covariance_point_size"):
                out[tag + key] = numpy.array(getattr(sim, key), dtype=numpy.float64)
            out[tag + "biass"] = numpy.array(sim.biass, dtype=numpy.float64).reshape(len(sim.biass), 2)
            out[tag + "performance"] = numpy.float64(sim.performance)
            out[tag + "mpc_frac"] = numpy.float64(sim.mpc_frac)
            out[tag + "counts"] = numpy.array([sim.predict_count, sim.update_count])
            print("  closed loop %s: %d predicts, %d updates, mpc_frac %.2f, performance %.4g" %
                  (name, sim.predict_count, sim.update_count, sim.mpc_frac, sim.performance), flush=True)
    finally:
        sim_base.get_parts, sim_base.get_noise = keep[0], keep[1]
        sys.modules["osqp"].OSQP = keep[2]
        controller.osqp.OSQP = keep[2]
        if keep[3] is None:
            del scipy.integrate.simps
    save("closed_loop.npz", **out)


# --------------------------------------------------------------------------- G11
def _patch_loop_environment():
    """What gen_closed_loop installs around the reference's Simulation (see there), for a worker process."""
    import controller  # noqa: F401  (reference)
    import scipy.integrate
    ROOT = os.path.dirname(os.path.dirname(OUT))
    if ROOT not in sys.path:
        sys.path.append(ROOT)
    from oracle import closed_loop
    orig_parts = sim_base.get_parts
    sim_base.get_noise = _host_noise
    sim_base.get_parts = lambda dt_control=1, N_particles=2 * 15, gpu=False, pf=True, _orig=orig_parts: _orig(
        dt_control=dt_control, N_particles=N_particles, gpu=False, pf=pf)
    _KktOsqp.mode = "exact"
    sys.modules["osqp"].OSQP = _KktOsqp
    controller.osqp.OSQP = _KktOsqp
    scipy.integrate.simps = lambda y, x: closed_loop.simps_avg(y, x)


def _envelope_worker(job):
    tag, n, dtc, dtp, end, pf, seed = job
    numpy.random.seed(seed)
    sim = sim_base.Simulation(n, dtc, dtp, end_time=end, pf=pf)
    sim.simulate()
    outs = list(sim.lin_model.outputs)
    return (tag, seed, float(sim.performance), float(sim.mpc_frac),
            (numpy.asarray(sim.ys_f) - numpy.asarray(sim.ys)[:, outs]).astype(numpy.float32),
            numpy.asarray(sim.ys)[:, outs].astype(numpy.float32), numpy.asarray(sim.us).astype(numpy.float32),
            numpy.asarray(sim.covariance_point_size, dtype=numpy.float64))


def gen_loop_envelope_large(seed=2222, seeds=16):
    """G11b: the same for ONE larger ensemble, N = 16384 particles, control every tick -- 16 seeds of the reference's own
    Simulation (four minutes of its per-particle Python loops each): what the ITSE does as N grows (the device loops run
    2^18 particles per rollout in BASELINE config 5; the reference cannot, but its trend from 256 over 1024 to 16384 can be
    checked).  Stored like G11, in loop_envelope_large.npz."""
    import multiprocessing
    tag, cfg = "pf16384_c01", (16384, 0.1, 0.1, 50, True)
    jobs = [(tag,) + cfg + (seed + s,) for s in range(seeds)]
    runs = []
    with multiprocessing.Pool(min(7, os.cpu_count() or 1), initializer=_patch_loop_environment) as pool:
        for k, r in enumerate(pool.imap_unordered(_envelope_worker, jobs)):
            runs.append(r[1:])
            print("  large envelope %d/%d: ITSE %.4g" % (k + 1, len(jobs), r[2]), flush=True)
    runs.sort()
    out = {"names": numpy.array([tag]), tag + "_cfg": numpy.array([cfg[0], cfg[1], cfg[2], cfg[3], float(cfg[4])]),
           tag + "_seeds": numpy.array([r[0] for r in runs]), tag + "_performance": numpy.array([r[1] for r in runs]),
           tag + "_mpc_frac": numpy.array([r[2] for r in runs]), tag + "_err": numpy.array([r[3] for r in runs]),
           tag + "_ys": numpy.array([r[4] for r in runs]), tag + "_us": numpy.array([r[5] for r in runs]),
           tag + "_cov": numpy.array([r[6] for r in runs])}
    print("  %s: ITSE median %.4g (10%% %.4g, 90%% %.4g)" % (tag, numpy.median(out[tag + "_performance"]),
                                                         *numpy.quantile(out[tag + "_performance"], [0.1, 0.9])), flush=True)
    save("loop_envelope_large.npz", **out)


def gen_loop_envelope_gsf(seed=3333, seeds=256):
    """G11c (round 5): ``seeds`` more independent runs of the reference's own loop for ONE configuration, the Gaussian-sum UKF
    with 64 Gaussians and control every tick (gsf64_c01 of G11, whose 32 seeds put the standard error of the median ITSE at
    4e4 -- too coarse to say whether a device median of 1.4e6 or 1.6e6 is the reference's): a sample large enough for a
    two-sample test against the Philox-driven device loop (tests/test_gpu_mpc.py).  Stored per run: ITSE, solved fraction, the
    estimation error of the two controlled outputs per tick.  loop_envelope_gsf.npz."""
    import multiprocessing
    tag, cfg = "gsf64_c01", (64, 0.1, 0.1, 50, False)
    jobs = [(tag,) + cfg + (seed + s,) for s in range(seeds)]
    runs = []
    with multiprocessing.Pool(min(7, os.cpu_count() or 1), initializer=_patch_loop_environment) as pool:
        for k, r in enumerate(pool.imap_unordered(_envelope_worker, jobs)):
            runs.append(r[1:5])
            if (k + 1) % 32 == 0:
                print("  gsf envelope %d/%d" % (k + 1, len(jobs)), flush=True)
    runs.sort(key=lambda r: r[0])
    out = {"names": numpy.array([tag]), tag + "_cfg": numpy.array([cfg[0], cfg[1], cfg[2], cfg[3], float(cfg[4])]),
           tag + "_seeds": numpy.array([r[0] for r in runs]), tag + "_performance": numpy.array([r[1] for r in runs]),
           tag + "_mpc_frac": numpy.array([r[2] for r in runs]), tag + "_err": numpy.array([r[3] for r in runs])}
    p = out[tag + "_performance"]
    print("  %s: %d seeds, ITSE median %.4g (10%% %.4g, 90%% %.4g), mean %.4g, sd %.4g" % (
        tag, len(p), numpy.median(p), *numpy.quantile(p, [0.1, 0.9]), p.mean(), p.std()), flush=True)
    save("loop_envelope_gsf.npz", **out)


def gen_loop_envelope(seed=1111, seeds=32):
    """G11: the reference's own Simulation (sim_base.py:207-309, the protocol of
    results/pf_closedloop/bioreactor_performance_pf.py:19-58: end_time = 50, dt_predict = 0.1) run to the end on its NumPy
    classes for ``seeds`` independent seeds per configuration -- particle filter N in {256, 1024} and GS-UKF N = 64, control
    every tenth tick and every tick -- with the stand-ins of gen_closed_loop for what the container lacks.  Stored per run:
    the ITSE ``performance`` (sim_base.py:164-185), the solved fraction, per tick the estimation error of the two controlled
    outputs ys_f - ys[:, outputs], the outputs themselves, the inputs and the covariance size.  The Philox-driven device
    loops must land inside the spread of these runs (tests/test_gpu_mpc.py)."""
    import multiprocessing
    cases = {
        #             N     dt_control dt_predict end pf
        "pf256_c1":  (256,  1.0, 0.1, 50, True),
        "pf256_c01": (256,  0.1, 0.1, 50, True),
        "pf1024_c1": (1024, 1.0, 0.1, 50, True),
        "pf1024_c01": (1024, 0.1, 0.1, 50, True),
        "gsf64_c1":  (64,   1.0, 0.1, 50, False),
        "gsf64_c01": (64,   0.1, 0.1, 50, False),
    }
    jobs = [(tag,) + cfg + (seed + 1000 * k + s,) for k, (tag, cfg) in enumerate(sorted(cases.items())) for s in range(seeds)]
    jobs.sort(key=lambda j: -j[1] * (10 if j[2] < 1 else 1))         # the long runs first
    res = {}
    with multiprocessing.Pool(min(7, os.cpu_count() or 1), initializer=_patch_loop_environment) as pool:
        for k, r in enumerate(pool.imap_unordered(_envelope_worker, jobs)):
            res.setdefault(r[0], []).append(r[1:])
            if (k + 1) % 8 == 0:
                print("  envelope %d/%d" % (k + 1, len(jobs)), flush=True)
    out = {"names": numpy.array(sorted(cases))}
    for tag, cfg in cases.items():
        runs = sorted(res[tag])
        out[tag + "_cfg"] = numpy.array([cfg[0], cfg[1], cfg[2], cfg[3], float(cfg[4])])
        out[tag + "_seeds"] = numpy.array([r[0] for r in runs])
        out[tag + "_performance"] = numpy.array([r[1] for r in runs])
        out[tag + "_mpc_frac"] = numpy.array([r[2] for r in runs])
        out[tag + "_err"] = numpy.array([r[3] for r in runs])            # (seeds, ticks, 2)  ys_f - ys[:, outputs]
        out[tag + "_ys"] = numpy.array([r[4] for r in runs])             # (seeds, ticks, 2)
        out[tag + "_us"] = numpy.array([r[5] for r in runs])             # (seeds, ticks, 2)
        out[tag + "_cov"] = numpy.array([r[6] for r in runs])            # (seeds, ticks)
        print("  %s: ITSE median %.4g (10%% %.4g, 90%% %.4g), mpc_frac min %.2f" % (
            tag, numpy.median(out[tag + "_performance"]), *numpy.quantile(out[tag + "_performance"], [0.1, 0.9]),
            out[tag + "_mpc_frac"].min()), flush=True)
    save("loop_envelope.npz", **out)


if __name__ == "__main__":
    which = set(sys.argv[1:]) or {"constants", "model", "pdf", "resample", "pf", "gsukf", "stat", "linear", "mpc", "loop", "envelope", "envelope_large"}
    Xss = x_ss()
    print("X_ss =", Xss)
    if "constants" in which:
        gen_constants(Xss)
    if "model" in which:
        gen_model(Xss)
    if "pdf" in which:
        gen_pdf()
    if "resample" in which:
        gen_resample()
    if "pf" in which:
        gen_pf_trace(Xss)
    if "gsukf" in which:
        gen_gsukf(Xss)
    if "stat" in which:
        gen_pf_stat(Xss)
    if "linear" in which:
        gen_linear_model()
    if "mpc" in which:
        gen_mpc_qp()
    if "loop" in which:
        gen_closed_loop()
    if "envelope" in which:
        gen_loop_envelope()
    if "envelope_large" in which:
        gen_loop_envelope_large()
    if "envelope_gsf" in which:
        gen_loop_envelope_gsf()
