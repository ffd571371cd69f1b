"""Pins the CPU oracle (oracle/gse_oracle.py) against the golden vectors captured from the
reference's NumPy classes (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import gse_oracle as orc


def test_constants(golden):
    g = golden("constants.npz")
    sp, mp = orc.bioreactor_noise()
    np.testing.assert_array_equal(sp.consts, g["state_consts"])
    np.testing.assert_array_equal(mp.consts, g["meas_consts"])
    np.testing.assert_array_equal(sp.inv_cov, g["state_inv"])
    np.testing.assert_array_equal(mp.inv_cov, g["meas_inv"])
    np.testing.assert_array_equal(orc.sigma_weights(), g["w_sigma"])
    np.testing.assert_allclose(orc.steady_state(), g["x_ss"], rtol=1e-9, atol=1e-12)


def test_model_f_fp64_inputs_bitwise(golden):
    """With fp64 inputs the reference's f is pure fp64 arithmetic: the restatement must agree to the bit."""
    g = golden("model_fg.npz")
    x = g["x"]
    for c in range(len(g["dt"])):
        ours = orc.bioreactor_f_vec(x, g["u"][c], g["dt"][c])
        np.testing.assert_array_equal(ours, g["f_inc_f64in"][c])
        for i in (0, 17, 33, 41, 50, 60, 70, 255):
            np.testing.assert_array_equal(np.array(orc.bioreactor_f(x[i], g["u"][c], g["dt"][c])), ours[i])


def test_model_f_vs_reference_mixed_precision(golden):
    """Against the reference evaluated on fp32 rows (NumPy-2 scalar promotion keeps some terms fp32):
    tolerance rtol 1e-5 relative to the size of the state derivative terms."""
    g = golden("model_fg.npz")
    x = g["x"]
    for c in range(len(g["dt"])):
        ours = orc.bioreactor_f_vec(x, g["u"][c], g["dt"][c])
        scale = np.abs(g["f_inc"][c]).max(axis=0, keepdims=True) + 1e-30
        assert np.max(np.abs(ours - g["f_inc"][c]) / scale) < 1e-5
        after = (x.astype(np.float64) + ours).astype(np.float32)
        # one fp32 rounding of x + 1e-5 of the increment (the mixed-precision slack)
        bound = 2e-7 * np.abs(g["x_after"][c]) + 1e-5 * np.abs(g["f_inc"][c]) + 1e-12
        assert np.all(np.abs(after.astype(np.float64) - g["x_after"][c]) <= bound)


def test_model_g(golden):
    g = golden("model_fg.npz")
    ours = orc.bioreactor_g_vec(g["x"])
    for c in range(len(g["dt"])):
        np.testing.assert_array_equal(ours.astype(np.float64), g["g_out"][c])


def test_known_answers():
    """SURVEY 8a known answers."""
    # (the survey's numbers were taken on the fp32-rounded state: 0.01 -> 0.0099999998)
    inc = orc.bioreactor_f(np.array([1.2, 25, 5, 0.01, 14], dtype=np.float32), [0.05, 0.15], 0.1)
    np.testing.assert_allclose(inc, (-9.552687259968e-4, 0, 3.1447705942284e-2, -1.9999999553e-4,
                                     3.5555550787184e-2), rtol=3e-6, atol=1e-15)
    xss = orc.steady_state()
    assert np.all(np.abs(orc.bioreactor_f(xss, [0.06, 0.2], 1.0)) < 2e-12)
    np.testing.assert_allclose(orc.bioreactor_g(xss), (280.0, 611.4538406938659), rtol=1e-12)
    _, mp = orc.bioreactor_noise()
    e = np.array([[0.1, 0], [0, -0.1], [3, -2], [30, -40], [0.3, -0.5]])
    np.testing.assert_allclose(mp.pdf(e), [1.9526639513, 1.6876964058, 4.0389400817e-05,
                                           3.5361479426e-06, 2.9331147948e-01], rtol=1e-9)
    idx = orc.systematic_index_loop(np.array([.05, .30, 0, .10, .25, 0, .20, .10]), 0.6964691855978616)
    np.testing.assert_array_equal(idx, [1, 1, 1, 4, 4, 6, 6, 7])


def test_mixture_pdf(golden):
    g = golden("mixture_pdf.npz")
    sp, mp = orc.bioreactor_noise()
    np.testing.assert_allclose(mp.pdf(g["e_meas"]), g["p_meas"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(mp.pdf(g["e_meas"].astype(np.float32)), g["p_meas_f32"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(sp.pdf(g["e_state"]), g["p_state"], rtol=1e-11, atol=0)
    assert (g["p_meas"] == 0).any(), "fixture must contain the exp() underflow edge"
    toy = orc.MixtureOracle(g["toy_means"], g["toy_covs"].astype(np.float64), g["toy_weights"])
    np.testing.assert_allclose(toy.pdf(g["e_toy"]), g["p_toy"], rtol=1e-5)   # fp32-rounded covariances as input


def test_resample_index(golden):
    g = golden("resample.npz")
    for name in g["names"]:
        w = g[name + "_w"].astype(np.float64)
        r = float(g[name + "_r"])
        np.testing.assert_array_equal(orc.systematic_index_loop(w, r), g[name + "_idx"], err_msg=name)
        np.testing.assert_array_equal(orc.systematic_index(w, r), g[name + "_idx"], err_msg=name)


def _mixtures():
    return orc.bioreactor_noise()


def assert_cov_close(p, ref, rel):
    """|dP_ij| <= rel * sqrt(P_ii P_jj).  Covariances are built from fp32 sigma-point deviations; for Cx
    (value 26, fp32 ulp 1.9e-6, sigma 3e-4) one ulp of a sigma point is already 0.6 % of the deviation,
    so entries are only meaningful relative to the scale of their row and column."""
    ref = np.asarray(ref, dtype=np.float64)
    d = np.sqrt(np.abs(np.einsum("...ii->...i", ref)))
    scale = d[..., :, None] * d[..., None, :] + 1e-30
    err = np.max(np.abs(np.asarray(p, dtype=np.float64) - ref) / scale)
    assert err <= rel, err


def test_pf_trace_per_step(golden):
    """Each step is started from the reference's own state and fed the reference's recorded noise
    and r: particles agree to one fp32 rounding, weights to 2e-6, ancestors exactly."""
    g = golden("pf_trace.npz")
    sp, mp = _mixtures()
    n = len(g["w0"])
    pf = orc.ParticleFilterOracle(n, None, sp, mp, particles=g["x0"])
    for c in range(len(g["dt"])):
        pf.predict(g["u"][c], g["dt"][c], noise=g["noise"][c])
        np.testing.assert_allclose(pf.particles, g["after_predict"][c], rtol=5e-7, atol=1e-7)   # (3e-6 of the increments: SURVEY 8c)
        pf.particles = g["after_predict"][c].copy()
        pf.update(g["u"][c], g["z"][c])
        np.testing.assert_allclose(pf.weights, g["w_after_update"][c], rtol=2e-6)
        pf.weights = g["w_after_update"][c].astype(pf.weights.dtype)
        idx = pf.resample(r=float(g["r"][c]))
        np.testing.assert_array_equal(idx, g["idx"][c])
        np.testing.assert_array_equal(pf.particles, g["after_resample"][c])
        np.testing.assert_allclose(pf.point_estimate(), g["estimate"][c], rtol=1e-7)
        np.testing.assert_allclose(pf.point_covariance(), g["cov_sv"][c], rtol=1e-6)


def test_pf_trace_free_running(golden):
    """Free-running over the 10 recorded cycles (no re-seeding from the reference).  At N = 1024 this
    filter is strongly degenerate (effective sample size 2-14 in most cycles) and a one-ulp difference
    in a particle moves its likelihood by up to ~5e-4 relative (y ~ 280 has an fp32 ulp of 3e-5 against
    a measurement sigma of 0.245), so ancestors may flip and only the posterior summaries are compared,
    loosely; the exact comparison is test_pf_trace_per_step."""
    g = golden("pf_trace.npz")
    sp, mp = _mixtures()
    n = len(g["w0"])
    pf = orc.ParticleFilterOracle(n, None, sp, mp, particles=g["x0"])
    for c in range(10):                                   # (beyond that one flipped ancestor has grown into 0.4 % of the weight sum)
        pf.predict(g["u"][c], g["dt"][c], noise=g["noise"][c])
        pf.update(g["u"][c], g["z"][c])
        np.testing.assert_allclose(np.sum(pf.weights, dtype=np.float64),
                                   np.sum(g["w_after_update"][c], dtype=np.float64), rtol=1e-3)
        pf.resample(r=float(g["r"][c]))
        np.testing.assert_allclose(pf.point_estimate(), g["estimate"][c], rtol=1e-3, atol=5e-3)


def test_pf_vectorised_equals_loop(golden):
    g = golden("pf_trace.npz")
    sp, mp = _mixtures()
    a = orc.ParticleFilterOracle(len(g["w0"]), None, sp, mp, particles=g["x0"])
    b = orc.ParticleFilterOracle(len(g["w0"]), None, sp, mp, particles=g["x0"])
    for c in range(3):
        a.predict(g["u"][c], g["dt"][c], noise=g["noise"][c])
        b.predict_vec(g["u"][c], g["dt"][c], noise=g["noise"][c])
        np.testing.assert_array_equal(a.particles, b.particles)
        a.update(g["u"][c], g["z"][c])
        b.update_vec(g["u"][c], g["z"][c])
        np.testing.assert_allclose(a.weights, b.weights, rtol=1e-15 if a.weights.dtype == np.float64 else 0)
        ia = a.resample(r=float(g["r"][c]), loop=True)
        ib = b.resample(r=float(g["r"][c]), loop=False)
        np.testing.assert_array_equal(ia, ib)


# Covariance after predict against the reference's fp32 accumulation (measured: <= 8.6e-4 at N = 64, 1.3e-4 at N = 7,
# 5e-5 at N = 1 -- the reference sums 11 fp32 outer products whose cancellation grows with the spread of the bank)
PRED_COV_TOL = {1: 5e-4, 2: 5e-4, 7: 5e-4, 64: 3e-3}


@pytest.mark.parametrize("n", [1, 7, 64])
def test_gsukf_trace(golden, n):
    g = golden("gsukf_trace.npz")
    t = "n%d_" % n
    sp, mp = _mixtures()
    gf = orc.GSUKFOracle(n, None, sp, mp, means=g[t + "means0"], covariances=g[t + "covs0"])
    np.testing.assert_allclose(gf.sigma_points(), g[t + "sigmas0"], rtol=1e-6, atol=1e-9)
    for c in range(len(g[t + "dt"])):
        # per-step: start every step from the reference's own state
        gf.predict(g[t + "u"][c], g[t + "dt"][c], noise=g[t + "noise"][c])
        np.testing.assert_allclose(gf.means, g[t + "m_pred"][c], rtol=1e-6, atol=1e-8)
        assert_cov_close(gf.covariances, g[t + "P_pred"][c], PRED_COV_TOL[n])
        gf.means, gf.covariances = g[t + "m_pred"][c].copy(), g[t + "P_pred"][c].copy()
        np.testing.assert_allclose(gf.sigma_points(), g[t + "sig_upd"][c], rtol=1e-6, atol=1e-9)
        gf.update(g[t + "u"][c], g[t + "z"][c])
        np.testing.assert_allclose(gf.means, g[t + "m_upd"][c], rtol=1e-6, atol=1e-8)
        assert_cov_close(gf.covariances, g[t + "P_upd"][c], 1e-4)
        np.testing.assert_allclose(gf.weights, g[t + "w_upd"][c], rtol=1e-4)
        gf.means, gf.covariances = g[t + "m_upd"][c].copy(), g[t + "P_upd"][c].copy()
        gf.weights = g[t + "w_upd"][c].astype(gf.weights.dtype)
        idx = gf.resample(r=float(g[t + "r"][c]))
        np.testing.assert_array_equal(idx, g[t + "idx"][c])
        np.testing.assert_array_equal(gf.means, g[t + "m_res"][c])
        np.testing.assert_allclose(gf.point_estimate(), g[t + "est"][c], rtol=1e-6)
        np.testing.assert_allclose(gf.point_covariance(), g[t + "cov_sv"][c], rtol=1e-5)


def test_gsukf_known_answer(golden):
    g = golden("gsukf_trace.npz")
    sp, mp = _mixtures()
    xss = orc.steady_state().astype(np.float32)
    gf = orc.GSUKFOracle(2, None, sp, mp, means=np.repeat(xss[None], 2, axis=0))
    gf.update(g["ka_u"], g["ka_z"])
    np.testing.assert_allclose(gf.means, g["ka_m_upd"], rtol=1e-6, atol=1e-9)
    assert_cov_close(gf.covariances, g["ka_P_upd"], 1e-5)
    np.testing.assert_allclose(gf.weights, g["ka_w_upd"], rtol=1e-6)
    np.testing.assert_allclose(gf.means[0], [1.5572222, 26.016260, 5.2668104, 0, 15.188571], rtol=1e-6, atol=1e-9)
    gf.predict(g["ka_u"], 0.1, noise=np.zeros((2, 11, 5), np.float32))
    np.testing.assert_allclose(gf.means, g["ka_m_pred"], rtol=1e-6, atol=1e-8)
    assert_cov_close(gf.covariances, g["ka_P_pred"], 5e-4)


# ---------------------------------------------------------------------------------------------------------------------
# The closed loop: oracle/closed_loop.py against the reference's own Simulation.simulate, run on the
# reference's NumPy classes by tests/golden/make_golden.py (G10).
# ---------------------------------------------------------------------------------------------------------------------
from loop_util import loop_oracle_from_golden  # noqa: E402


@pytest.mark.parametrize("name", ["pf_c1", "pf_c1p1", "pf_c01", "pf_fail", "gsf_c1", "gsf_c01"])
def test_closed_loop_oracle_against_the_reference_loop(golden, name):
    """Fed the recorded random inputs, the oracle loop reproduces every record of the reference's loop over the whole run
    (no re-seeding of states from the recording): the order of operations of sim_base.py:243-309 (predict-if-due, bias log,
    update, resample, estimate, MPC with the except branch, plant step + clamp, the two noise draws, per-tick estimate and
    covariance), the performance integral and the solved fraction.  The reference's f keeps some terms fp32 under NumPy 2
     where the oracle restates the device's fp64 arithmetic, so particles differ by fp32 roundings and weights
    by 2e-6: the oracle's OWN ancestors are counted against the recorded ones (at most 2 flips per resample, a weight
    within rounding of a threshold -- measured: 8 of 7 424 over the 29 resamples of pf_c01, none elsewhere) and the
    recorded ones are copied, so that one flip does not send the rest of the run down another path.  Tolerances are the
    measured deviations rounded up: estimates 2e-6, inputs 1e-6, covariance sizes 1e-5."""
    g = golden("closed_loop.npz")
    sim = loop_oracle_from_golden(g, name).simulate()
    assert (sim.predict_count, sim.update_count) == tuple(g[name + "_counts"])
    assert max(sim.ancestor_flips) <= 2 and sum(sim.ancestor_flips) <= 8, sim.ancestor_flips
    np.testing.assert_allclose(sim.us, g[name + "_us"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(sim.xs, g[name + "_xs"], rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(sim.ys, g[name + "_ys"], rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(sim.ys_meas, g[name + "_ys_meas"], rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(sim.xs_f, g[name + "_xs_f"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(sim.ys_f, g[name + "_ys_f"], rtol=2e-6)
    np.testing.assert_allclose(sim.covariance_point_size, g[name + "_covariance_point_size"], rtol=1e-5)
    np.testing.assert_allclose(np.array(sim.biass).reshape(-1, 2), g[name + "_biass"], rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(sim.performance, g[name + "_performance"], rtol=1e-5)
    assert sim.mpc_frac == g[name + "_mpc_frac"]


def test_simps_rule_of_the_reference_scipy():
    """oracle.closed_loop.simps_avg on an odd sample count is Simpson's rule (= scipy.integrate.simpson there); on an even
    count it is the published even='avg' rule: exact for cubics only up to the trapezoid ends."""
    import scipy.integrate
    from oracle import closed_loop
    x = np.linspace(0, 3, 31)
    y = np.sin(x) + x ** 3
    np.testing.assert_allclose(closed_loop.simps_avg(y, x), scipy.integrate.simpson(y, x=x), rtol=1e-14)
    x = np.linspace(0, 3, 30)
    y = 2 * x ** 2 - x
    first = scipy.integrate.simpson(y[:-1], x=x[:-1]) + 0.5 * (x[-1] - x[-2]) * (y[-1] + y[-2])
    last = 0.5 * (x[1] - x[0]) * (y[1] + y[0]) + scipy.integrate.simpson(y[1:], x=x[1:])
    np.testing.assert_allclose(closed_loop.simps_avg(y, x), (first + last) / 2, rtol=1e-14)
    from gpu_se_amd import sim_base
    np.testing.assert_allclose(sim_base.simps(y, x), closed_loop.simps_avg(y, x), rtol=1e-15)


def test_loop_envelope_fixture_and_the_oracle_loop_inside_it(golden):
    """G11 (loop_envelope.npz): 32 runs of the reference's own Simulation per configuration.  The fixture is complete, and
    the oracle loop (oracle/closed_loop.py) driven by NumPy generators -- free-running, its own random numbers -- lands inside
    the spread of those runs: 8 seeds of the particle filter, N = 256, control every tenth tick, the whole 50 minutes.
    (The device loops are held to the same envelope in tests/test_gpu_mpc.py.)"""
    import loop_util
    from oracle import closed_loop
    g = golden("loop_envelope.npz")
    for tag in g["names"]:
        assert g[tag + "_performance"].shape == (32,) and g[tag + "_err"].shape == (32, 500, 2)
        assert np.all(g[tag + "_mpc_frac"] == 1.0) and np.all(np.isfinite(g[tag + "_performance"]))
    big = golden("loop_envelope_large.npz")                       # G11b: 16 seeds at N = 16384
    assert big["pf16384_c01_performance"].shape == (16,) and np.all(big["pf16384_c01_mpc_frac"] == 1.0)
    # the reference's own trend: its ITSE does not fall as the ensemble grows (the estimate lags the plant by one tick)
    assert np.median(big["pf16384_c01_performance"]) > np.median(g["pf256_c01_performance"])
    tag = "pf256_c1"
    n, dtc, dtp, end, _ = g[tag + "_cfg"]
    n, end = int(n), int(end)
    perf, err = [], []
    xss = orc.steady_state()
    for seed in range(8):
        rng = np.random.default_rng(4000 + seed)
        sp, mp = orc.bioreactor_noise(rng)
        x0, _ = orc.bioreactor_noise(rng)
        x0.means = x0.means + xss[None, :].astype(np.float32)
        est = orc.ParticleFilterOracle(n, x0, sp, mp, rng=rng)
        ticks = end * 10
        noise = closed_loop.LoopNoise(sp.draw(ticks - 1), mp.draw(ticks - 1), [sp.draw(n) for _ in range(ticks - 1)],
                                      rng.random(ticks))
        sim = loop_util.loop_oracle(est, float(dtc), float(dtp), end, xss, noise).simulate()
        assert sim.mpc_frac == 1.0
        perf.append(sim.performance)
        err.append(sim.ys_f - sim.ys[:, [0, 2]])
    q = np.quantile(g[tag + "_performance"], [0.1, 0.9])
    assert 0.8 * q[0] <= np.median(perf) <= 1.2 * q[1], (np.median(perf), q)
    rms_o = np.sqrt(np.mean(np.array(err)[:, 100:] ** 2, axis=(0, 1)))
    rms_r = np.sqrt(np.mean(g[tag + "_err"][:, 100:].astype(np.float64) ** 2, axis=(0, 1)))
    assert np.all(rms_o / rms_r > 0.75) and np.all(rms_o / rms_r < 1.35), (rms_o, rms_r)
