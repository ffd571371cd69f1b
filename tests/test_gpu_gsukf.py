"""Parity of the HIP Gaussian-sum UKF against the CPU oracle and the golden vectors (-m gpu)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

import gpu_se_amd                                            # noqa: E402
from gpu_se_amd import sim_base                              # noqa: E402
from gpu_se_amd.filter import BatchedGaussianSumUnscentedKalmanFilter as BatchedGSF   # noqa: E402
from gpu_se_amd.filter import ParallelGaussianSumUnscentedKalmanFilter as GSF   # noqa: E402
from gpu_se_amd.gaussian_sum_dist import DeterministicGaussianSum               # noqa: E402
from gpu_se_amd.model import Bioreactor                      # noqa: E402
from oracle import gse_oracle as orc                         # noqa: E402
from tests.test_oracle_golden import assert_cov_close, PRED_COV_TOL        # noqa: E402


def assert_mean_close(mu, ref, cov, rel_sigma=5e-5, rtol=3e-7):
    """|d mu_k| <= rel_sigma * sqrt(P_kk) + rtol |mu_k|.  A one-ulp difference in a sigma point (the device
    factorises with its own fp32 Cholesky, the oracle with LAPACK) moves y = 180 Cg by an fp32 ulp of 280
    (3e-5) against deviations of ~1.8, i.e. the Kalman gain by ~1e-5 relative; the update of each state is a
    fraction of its standard deviation, hence the sigma-relative tolerance."""
    sd = np.sqrt(np.abs(np.einsum("...ii->...i", np.asarray(cov, dtype=np.float64))))
    bound = rel_sigma * sd + rtol * np.abs(ref) + 1e-12
    err = np.abs(np.asarray(mu, dtype=np.float64) - ref)
    assert np.all(err <= bound), float(np.max(err / bound))


def make(n, means=None, covs=None, deterministic=False):
    gpu_se_amd.seed(21)
    state_pdf, measurement_pdf = sim_base.get_noise(deterministic=deterministic)
    x0, _ = sim_base.get_noise()
    x0.means += orc.steady_state()[None, :].astype(np.float32)
    gf = GSF(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, n, x0, state_pdf, measurement_pdf)
    if means is not None:
        gf.means = means
    if covs is not None:
        gf.covariances = covs
    return gf


def test_constants(golden):
    g = golden("constants.npz")
    gf = make(4)
    np.testing.assert_array_equal(gf._w_sigma.get(), g["w_sigma"])
    assert gf._N_sigmas == 11 and gf._Nx == 5 and gf._Ny == 2
    assert tuple(gf.means.shape) == (4, 5) and tuple(gf.covariances.shape) == (4, 5, 5)
    np.testing.assert_array_equal(gf.covariances.get()[2], g["state_covs"][0])


@pytest.mark.parametrize("n", [1, 7, 64])
def test_gsukf_trace_per_step(golden, n):
    """Golden G5: the reference's GS-UKF trace, every step started from the reference's own state with its
    recorded noise and r.  Tolerances (SURVEY 8c gives none for G5; these are the measured deviations rounded up, each with
    its cause): sigma points rtol 1e-6 (fp32 Cholesky here, LAPACK's there); means after predict rtol 1e-6; covariance after
    predict PRED_COV_TOL = 5e-4 .. 3e-3 of the entry's scale (an fp32 sum of eleven outer products of deviations about a mean
    that itself carries an fp32 rounding: NumPy sums pairwise in another order); means after update 5e-5 of the prior
    standard deviation + 3e-7 relative (assert_mean_close: one fp32 ulp of y = 180 Cg moves the Kalman gain by 1e-5);
    covariance after update 1e-4; weights rtol 2e-4 (the likelihood at the updated mean); ancestors, gathered rows: exact;
    estimate rtol 1e-6, covariance size rtol 1e-5."""
    g = golden("gsukf_trace.npz")
    t = "n%d_" % n
    gf = make(n, g[t + "means0"], g[t + "covs0"], deterministic=True)
    np.testing.assert_allclose(gf._get_sigma_points().get(), g[t + "sigmas0"], rtol=1e-6, atol=1e-8)
    for c in range(len(g[t + "dt"])):
        DeterministicGaussianSum.preload(g[t + "noise"][c])
        gf.predict(g[t + "u"][c], g[t + "dt"][c])
        np.testing.assert_allclose(gf.means.get(), g[t + "m_pred"][c], rtol=1e-6, atol=1e-8)
        assert_cov_close(gf.covariances.get(), g[t + "P_pred"][c], PRED_COV_TOL[n])
        gf.means, gf.covariances = g[t + "m_pred"][c], g[t + "P_pred"][c]
        np.testing.assert_allclose(gf._get_sigma_points().get(), g[t + "sig_upd"][c], rtol=1e-6, atol=1e-8)
        gf.update(g[t + "u"][c], g[t + "z"][c])
        assert_mean_close(gf.means.get(), g[t + "m_upd"][c], g[t + "P_pred"][c])
        assert_cov_close(gf.covariances.get(), g[t + "P_upd"][c], 1e-4)
        np.testing.assert_allclose(gf.weights.get(), g[t + "w_upd"][c], rtol=2e-4)
        gf.means, gf.covariances, gf.weights = g[t + "m_upd"][c], g[t + "P_upd"][c], g[t + "w_upd"][c]
        idx = gf.resample(r=float(g[t + "r"][c]), return_index=True).get()
        w32 = g[t + "w_upd"][c].astype(np.float32).astype(np.float64)
        np.testing.assert_array_equal(idx, orc.systematic_index(w32, float(g[t + "r"][c])))
        np.testing.assert_array_equal(idx, g[t + "idx"][c])
        np.testing.assert_array_equal(gf.means.get(), g[t + "m_res"][c])
        np.testing.assert_array_equal(gf.covariances.get(), g[t + "P_res"][c])
        np.testing.assert_allclose(gf.point_estimate(), g[t + "est"][c], rtol=1e-6)
        np.testing.assert_allclose(gf.point_covariance(), g[t + "cov_sv"][c], rtol=1e-5)


@pytest.mark.parametrize("n", [5, 64, 1000])
def test_gsukf_against_oracle(n):
    """Same inputs to the HIP path and the NumPy oracle over several free-running cycles with injected noise."""
    rng = np.random.default_rng(n)
    sp, mp = orc.bioreactor_noise()
    xss = orc.steady_state()
    means = (xss[None, :] + rng.standard_normal((n, 5)) * np.sqrt([1e-4, 1e-7, 1e-3, 1e-3, 1e-7])).astype(np.float32)
    o = orc.GSUKFOracle(n, None, sp, mp, means=means)
    gf = make(n, means, deterministic=True)
    for c in range(3):
        u = [0.05 + 0.01 * c, 0.2]
        z = np.array(orc.bioreactor_g(xss)) + [0.2, -0.3]
        noise = (rng.standard_normal((n, 11, 5)) * np.sqrt([1e-4, 1e-7, 1e-3, 1e-3, 1e-7])).astype(np.float32)
        DeterministicGaussianSum.preload(noise)
        gf.predict(u, 0.1)
        o.predict(u, 0.1, noise=noise, loop=False)
        np.testing.assert_allclose(gf.means.get(), o.means, rtol=3e-7, atol=1e-9)
        assert_cov_close(gf.covariances.get(), o.covariances, 2e-5)
        gf.means, gf.covariances = o.means, o.covariances
        prior_cov = o.covariances.copy()
        gf.update(u, z)
        o.update(u, z)
        assert_mean_close(gf.means.get(), o.means, prior_cov)
        assert_cov_close(gf.covariances.get(), o.covariances, 1e-4)
        # the weight factor is pdf(z - g(mu_new)): an fp32 ulp of y = 180 Cg (3e-5) against the 0.245 measurement
        # sigma moves the likelihood by up to a few 1e-4 relative
        np.testing.assert_allclose(gf.weights.get(), o.weights, rtol=1e-3)
        gf.means, gf.covariances, gf.weights = o.means, o.covariances, o.weights
        r = float(rng.random())
        w32 = np.asarray(o.weights, dtype=np.float32).astype(np.float64)
        idx = gf.resample(r=r, return_index=True).get()
        o.weights = w32
        np.testing.assert_array_equal(idx, o.resample(r=r, loop=False))
        np.testing.assert_allclose(gf.point_estimate(), o.point_estimate(), rtol=1e-6)
        np.testing.assert_allclose(gf.point_covariance(), o.point_covariance(), rtol=1e-5)


def test_gsukf_known_answer(golden):
    g = golden("gsukf_trace.npz")
    xss = orc.steady_state().astype(np.float32)
    gf = make(2, np.repeat(xss[None], 2, axis=0), deterministic=True)
    gf.update(g["ka_u"], g["ka_z"])
    np.testing.assert_allclose(gf.means.get(), g["ka_m_upd"], rtol=1e-6, atol=1e-9)
    assert_cov_close(gf.covariances.get(), g["ka_P_upd"], 1e-5)
    np.testing.assert_allclose(gf.weights.get(), g["ka_w_upd"], rtol=1e-6)
    DeterministicGaussianSum.preload(np.zeros((2, 11, 5), np.float32))
    gf.means, gf.covariances = g["ka_m_upd"], g["ka_P_upd"]
    gf.predict(g["ka_u"], 0.1)
    np.testing.assert_allclose(gf.means.get(), g["ka_m_pred"], rtol=1e-6, atol=1e-8)
    assert_cov_close(gf.covariances.get(), g["ka_P_pred"], 5e-4)


def test_row_helpers_on_the_gsukf():
    """f_vectorize / g_vectorize of the GS-UKF class (the reference's gsf_run_seq.py:220, :285, :315 call them on the
    filter object): f on the sigma points, g on the means, against the fp64 oracle."""
    gf = make(64)
    u = [0.05, 0.18]
    sig = gf._get_sigma_points()
    inc = gf.f_vectorize(sig, u, 0.1).get()
    assert inc.shape == (64, 11, 5)
    ref = orc.bioreactor_f_vec(sig.get().reshape(-1, 5), u, 0.1).astype(np.float32).reshape(64, 11, 5)
    np.testing.assert_array_equal(inc, ref)
    ys = gf.g_vectorize(gf.means, u, gf._y_dummy).get()
    np.testing.assert_array_equal(ys, orc.bioreactor_g_vec(gf.means.get()))


@pytest.mark.parametrize("n", [64, 300, 5000])
def test_gsukf_philox_predict_against_the_oracle(n):
    """The Philox noise of the Gaussian-sum filter's predict is pinned in the oracle (oracle.gsukf_predict_noise: sigma points
    in pairs sharing three Philox blocks): the device's predict in its own random-number mode against the NumPy oracle fed
    the restated noise, for the one-launch / group / lane kernels (64, 300, 5000 Gaussians), two steps.  The device's normals
    are fp32 fast-math (2e-5 sigma from the fp64 ones): the predicted means
    agree to 1e-3 sigma of the noise / sqrt(11) and a rounding."""
    sp, mp = orc.bioreactor_noise()
    xss = orc.steady_state()
    rng = np.random.default_rng(n)
    means = (xss[None, :] + rng.standard_normal((n, 5)) * np.sqrt([1e-4, 1e-7, 1e-3, 1e-3, 1e-7])).astype(np.float32)
    gf = make(n, means)
    o = orc.GSUKFOracle(n, None, sp, mp, means=means)
    o.covariances = gf.covariances.get().copy()
    sd = np.sqrt(np.diagonal(sp.covariances, axis1=1, axis2=2)).max(axis=0)
    for step in range(2):
        u = [0.05 + 0.01 * step, 0.2]
        noise, comp = orc.gsukf_predict_noise(gf._seed, step, n, sp, filter_id=gf._filter_offset)
        assert set(np.unique(comp)) <= {0, 1} and 0.6 < np.mean(comp == 0) < 0.9          # (weights 0.75 / 0.25)
        gf.predict(u, 0.1)
        o.predict(u, 0.1, noise=noise.astype(np.float32), loop=False)
        err = np.abs(gf.means.get() - o.means)
        assert np.all(err <= 1e-3 * sd + 2.4e-7 * np.abs(o.means)), float(np.max(err / (1e-3 * sd + 2.4e-7 * np.abs(o.means))))
        assert_cov_close(gf.covariances.get(), o.covariances, 2e-3)
        gf.means, gf.covariances = o.means, o.covariances


def test_gsukf_philox_cycle_and_errors():
    gpu_se_amd.seed(5)
    _, _, _, gf = sim_base.get_parts(N_particles=64, pf=False)
    xss = orc.steady_state()
    z = np.array(orc.bioreactor_g(xss))
    for c in range(20):
        gf.predict([0.06, 0.2], 0.1)
        gf.update([0.06, 0.2], z)
        gf.resample()
    est = gf.point_estimate()
    np.testing.assert_allclose(est[[0, 2]], xss[[0, 2]], rtol=2e-3)
    assert np.isfinite(gf.point_covariance())
    # a covariance that is not positive definite even after the 1e-10 retry is reported, like the reference's
    # uncaught LinAlgError / RuntimeError (gs_ukf.py:72-75, :336-339)
    bad = gf.covariances.get().copy()
    bad[3] = -np.eye(5, dtype=np.float32)
    gf.covariances = bad
    gf.predict([0.06, 0.2], 0.1)
    with pytest.raises(np.linalg.LinAlgError):
        gf.point_estimate()


def test_large_bank_lane_kernels_against_the_oracle_and_the_group_kernels():
    """2^16 Gaussians: the class runs one lane per Gaussian with predict + update in ONE launch.  Against the same calls
    one kernel each (lazy = False: bit-identical), against the 16-lanes-per-Gaussian kernels through the C ABI
    (GSE_GSUKF_LANES is read once per process, so those are reached with a bank of 64, the largest they still serve: same
    arithmetic, same element numbers), and against the NumPy oracle on a slice."""
    n = 1 << 16
    xss = orc.steady_state()
    z = np.array(orc.bioreactor_g(xss)) + [0.2, -0.1]
    rng = np.random.default_rng(3)
    means = (xss[None, :] + rng.standard_normal((n, 5)) * np.sqrt([1e-4, 1e-7, 1e-3, 1e-3, 1e-7])).astype(np.float32)
    banks = []
    for lazy in (True, False):
        gf = make(n, means)
        gf.lazy = lazy
        banks.append(gf)
    m_small = 64                                                  # at most 64 Gaussians: the group kernels
    small = make(m_small, means[:m_small])
    small._seed = banks[0]._seed
    for c in range(2):
        u = [0.05 + 0.01 * c, 0.2]
        for gf in banks + [small]:
            gf.predict(u, 0.1)
            gf.update(u, z)
        np.testing.assert_array_equal(banks[0].means.get(), banks[1].means.get())
        np.testing.assert_array_equal(banks[0].covariances.get(), banks[1].covariances.get())
        np.testing.assert_array_equal(banks[0].weights.get(), banks[1].weights.get())
        np.testing.assert_array_equal(banks[0].means.get()[:m_small], small.means.get())
        np.testing.assert_array_equal(banks[0].covariances.get()[:m_small], small.covariances.get())
        # (the smaller bank starts from weights 1 / m_small: same likelihood factors, another scale)
        np.testing.assert_allclose(banks[0].weights.get()[:m_small] * (n / float(m_small)), small.weights.get(), rtol=3e-7)
    # oracle: one cycle with injected noise on a fresh pair
    m = 4096
    sp, mp = orc.bioreactor_noise()
    o = orc.GSUKFOracle(m, None, sp, mp, means=means[:m])
    gf = make(n, means, deterministic=True)
    noise = (rng.standard_normal((n, 11, 5)) * np.sqrt([1e-4, 1e-7, 1e-3, 1e-3, 1e-7])).astype(np.float32)
    DeterministicGaussianSum.preload(noise)
    gf.predict([0.05, 0.2], 0.1)
    gf.update([0.05, 0.2], z)
    o.predict([0.05, 0.2], 0.1, noise=noise[:m], loop=False)
    prior_cov = o.covariances.copy()
    o.update([0.05, 0.2], z)
    assert_mean_close(gf.means.get()[:m], o.means, prior_cov, rel_sigma=2e-4)
    assert_cov_close(gf.covariances.get()[:m], o.covariances, 2e-4)
    np.testing.assert_allclose(gf.weights.get()[:m] * (n / m), o.weights, rtol=2e-3)


def test_lane_kernels_against_the_oracle_on_the_whole_bank_of_config_size():
    """2^18 Gaussians (the size of the bench's GS-UKF block): predict + update in one launch of the lane kernels, then the
    resample through ancestors_kernel, against the vectorised NumPy oracle on EVERY Gaussian, with injected noise.  Means to
    2e-4 sigma, covariances to 6e-4 of their scale, weights to 1e-6 (round 2 compared 4096 of 2^16 Gaussians, weights to
    2e-3), ancestors from the DEVICE's weights exact up to last-bit ties of the cumulative sums."""
    n = 1 << 18
    xss = orc.steady_state()
    z = np.array(orc.bioreactor_g(xss)) + [0.2, -0.1]
    rng = np.random.default_rng(18)
    means = (xss[None, :] + rng.standard_normal((n, 5)) * np.sqrt([1e-4, 1e-7, 1e-3, 1e-3, 1e-7])).astype(np.float32)
    sp, mp = orc.bioreactor_noise()
    o = orc.GSUKFOracle(n, None, sp, mp, means=means)
    gf = make(n, means, deterministic=True)
    noise = (rng.standard_normal((n, 11, 5)) * np.sqrt([1e-4, 1e-7, 1e-3, 1e-3, 1e-7])).astype(np.float32)
    u = [0.05, 0.2]
    DeterministicGaussianSum.preload(noise)
    gf.predict(u, 0.1)
    gf.update(u, z)
    o.predict(u, 0.1, noise=noise, loop=False)
    prior_cov = o.covariances.copy()
    o.update(u, z)
    w_dev = gf.weights.get()
    assert_mean_close(gf.means.get(), o.means, prior_cov, rel_sigma=2e-4)
    assert_cov_close(gf.covariances.get(), o.covariances, 6e-4)                  # (measured over the bank: 3.6e-4)
    np.testing.assert_allclose(w_dev, o.weights, rtol=1e-6, atol=1e-30)         # (measured: 3.3e-7, median 0)
    r = 0.6180339887
    idx = gf.resample(r=r, return_index=True).get()
    own = orc.systematic_index(w_dev.astype(np.float64), r)
    assert np.sum(idx != own) <= 2, int(np.sum(idx != own))
    # second cycle: the deferred row copy rides the next predict + update launch (the benchmarked sequence)
    noise2 = (rng.standard_normal((n, 11, 5)) * np.sqrt([1e-4, 1e-7, 1e-3, 1e-3, 1e-7])).astype(np.float32)
    DeterministicGaussianSum.preload(noise2)
    gf.predict(u, 0.1)
    gf.update(u, z)
    o.resample(r=r, idx=idx.astype(np.int64))
    o.predict(u, 0.1, noise=noise2, loop=False)
    prior_cov = o.covariances.copy()
    o.update(u, z)
    # (free-running into the second cycle: the first cycle's fp32 differences in P are inputs now -- measured 1.8e-3 on the
    #  covariances, 1.0e-4 on the worst weight of the bank, median 2.5e-8)
    assert_mean_close(gf.means.get(), o.means, prior_cov, rel_sigma=2e-3)
    assert_cov_close(gf.covariances.get(), o.covariances, 4e-3)
    np.testing.assert_allclose(gf.weights.get(), o.weights, rtol=3e-4, atol=1e-30)


@pytest.mark.parametrize("n", [65, 1000, 1024, 1025, 4095, 4096, 4097, 9000, 70001, (1 << 18) + 5])
def test_one_launch_resample_gives_the_ancestors_of_the_three_kernel_path(n):
    """Banks of at most 4096 Gaussians find their ancestors in ONE launch (small_resample_kernel: tile sums, prefix, cumulative
    weights in LDS, search); larger ones from their tile sums without materialised cumulative weights (ancestors_kernel; the
    eager form keeps cumsum + gather).  All against the library's stand-alone pieces --
    gse_cumsum_normalised (the three-kernel path's cumulative weights) + gse_systematic_index -- bit for bit, for smooth and
    for degenerate weights, in both the deferred (lazy) and the eager form; and the rows really are the ancestors' rows."""
    from gpu_se_amd import _lib
    from tests.gpu_util import stream, workspace
    xss = orc.steady_state()
    rng = np.random.default_rng(n)
    means = (xss[None, :] + rng.standard_normal((n, 5)) * np.sqrt([1e-4, 1e-7, 1e-3, 1e-3, 1e-7])).astype(np.float32)
    for kind in ("smooth", "degenerate", "one", "stretch"):
        w = rng.random(n).astype(np.float32)
        if kind == "degenerate":
            w = (w ** 40).astype(np.float32)
        if kind == "one":
            w[:] = 0
            w[n // 3] = 1.0
        if kind == "stretch":                         # long weightless stretches between a few heavy Gaussians, and tiny weights
            w[:] = 0
            w[rng.integers(0, n, 7)] = rng.random(7).astype(np.float32)
            w[rng.integers(0, n, 50)] += np.float32(1e-12)
        r = float(rng.random())
        wd = torch.as_tensor(w, device="cuda")
        c = torch.empty(n, dtype=torch.float64, device="cuda")
        want = torch.empty(n, dtype=torch.int64, device="cuda")
        _lib.call("gse_cumsum_normalised", wd.data_ptr(), c.data_ptr(), n, workspace(1, n).data_ptr(), stream())
        _lib.call("gse_systematic_index", c.data_ptr(), want.data_ptr(), n, r, stream())
        want = want.cpu().numpy()
        # ... and the oracle's ancestors of the WHOLE bank (NumPy's left-to-right fp64 cumsum against the tile-wise one:
        # a cumulative weight may differ in its last bit, which shows only where it ties with a threshold)
        own = orc.systematic_index(w.astype(np.float64), r)
        assert np.sum(own != want) <= 2, (kind, int(np.sum(own != want)))
        for lazy in (True, False):
            gf = make(n, means)
            gf.lazy = lazy
            gf.weights = w
            idx = gf.resample(r=r, return_index=True).get()
            np.testing.assert_array_equal(idx, want)
            np.testing.assert_array_equal(gf.means.get(), means[want])
            np.testing.assert_array_equal(gf.weights.get(), np.full(n, np.float32(1.0 / n)))


@pytest.mark.parametrize("n", [4097, 70001, (1 << 18) + 5])
def test_inverse_map_ancestors_on_adversarial_weights(n):
    """The deferred resample of a large bank finds its ancestors by the inverse map (ancestors_scatter_kernel: every key writes
    itself to the outputs it owns; keys with many offspring are left to the whole workgroup): on weights chosen to hurt --
    exact ties, one Gaussian owning every output, heavy Gaussians at tile edges, weightless stretches, steps of four decades,
    sums that are zero or NaN -- and with r = 0, bit for bit the ancestors of the eager route (cumsum_kernel + the searching
    gather_kernel), which the tests above hold against the oracle."""
    from tests.gpu_util import adversarial_weights
    xss = orc.steady_state()
    rng = np.random.default_rng(n + 1)
    means = (xss[None, :] + rng.standard_normal((n, 5)) * np.sqrt([1e-4, 1e-7, 1e-3, 1e-3, 1e-7])).astype(np.float32)
    for name, w in adversarial_weights(n, rng).items():
        w = w.astype(np.float32)
        for r in (0.0, float(rng.random()), 1.0 - 2.0 ** -53):
            got = {}
            for lazy in (True, False):
                gf = make(n, means)
                gf.lazy = lazy
                gf.weights = w
                got[lazy] = gf.resample(r=r, return_index=True).get()
            np.testing.assert_array_equal(got[True], got[False], err_msg="%s r=%r" % (name, r))
            if np.isfinite(w.sum()) and w.sum() > 0:
                own = orc.systematic_index(w.astype(np.float64), r)
                assert np.sum(own != got[True]) <= 2, (name, r, int(np.sum(own != got[True])))
            else:
                assert not got[True].any(), name


def test_large_bank_deferred_resample_equals_the_eager_calls():
    """Large banks defer the row copy of resample() into the next predict + update launch (gse_gsukf_resample_begin +
    gse_gsukf_predict_update_from): over whole cycles bit-identical to one kernel per method (lazy = False), ancestors
    included; accessors in between see the resampled bank (the copy then runs on its own, gse_gsukf_gather_rows)."""
    n = (1 << 16) + 77
    xss = orc.steady_state()
    rng = np.random.default_rng(5)
    means = (xss[None, :] + rng.standard_normal((n, 5)) * np.sqrt([1e-4, 1e-7, 1e-3, 1e-3, 1e-7])).astype(np.float32)
    banks = []
    for lazy in (True, False):
        gf = make(n, means)
        gf.lazy = lazy
        banks.append(gf)
    banks[1]._seed = banks[0]._seed
    z0 = np.array(orc.bioreactor_g(xss))
    for c in range(5):
        u = [0.05 + 0.005 * c, 0.2]
        z = z0 + rng.normal(0, 0.3, 2)
        r = float(rng.random())
        idx = []
        for gf in banks:
            gf.predict(u, 0.1)
            gf.update(u, z)
            idx.append(gf.resample(r=r, return_index=(c % 2 == 0)))
        assert banks[0]._pend_gather is not None
        if c % 2 == 0:
            np.testing.assert_array_equal(idx[0].get(), idx[1].get())
        if c == 2:                                    # look at the bank between resample and predict: the copy runs now
            np.testing.assert_array_equal(banks[0].means.get(), banks[1].means.get())
            assert banks[0]._pend_gather is None
        if c == 3:
            np.testing.assert_array_equal(banks[0].point_estimate(), banks[1].point_estimate())
    np.testing.assert_array_equal(banks[0].means.get(), banks[1].means.get())
    np.testing.assert_array_equal(banks[0].covariances.get(), banks[1].covariances.get())
    np.testing.assert_array_equal(banks[0].weights.get(), banks[1].weights.get())


@pytest.mark.parametrize("n,batch,deterministic", [((1 << 16) + 100, None, False), (12 * 1024 + 4, 2, False),
                                                   (1 << 18, None, False), (20 * 1024, None, True), (300, None, False),
                                                   (2048 + 4, 2, False)])
def test_one_launch_cycle_of_a_large_bank_equals_the_two_launches(n, batch, deterministic):
    """gse_gsukf_cycle_lanes (round 4): resample + predict + update of a large bank in ONE launch -- the workgroups of the
    one-lane-per-Gaussian kernel find the ancestors of their own 256 Gaussians with the gather stage of the particle filter's
    cluster kernels, from the quarter sums and quad prefixes the previous launch left -- against the two launches it replaces
    (ancestors_scatter_kernel, then predict + update reading every Gaussian from its ancestor's row): the same means,
    covariances and weights bit for bit over whole cycles, on ragged banks, a batch, Philox and injected noise, through a
    measurement that leaves a handful of Gaussians with all the weight (ancestor ranges many tiles wide: the look-up path),
    and with accessors in between (the pending resample then runs through its own kernels)."""
    xss = orc.steady_state()
    rng = np.random.default_rng(11)
    B = batch or 1
    means = (xss[None, :] + rng.standard_normal((n, 5)) * np.sqrt([1e-4, 1e-7, 1e-3, 1e-3, 1e-7])).astype(np.float32)
    banks = []
    for one_launch in (True, False):
        if batch is None:
            gf = make(n, means, deterministic=deterministic)
        else:
            gf = make_batch(batch, n, np.stack([means, means[::-1]][:batch]), deterministic=deterministic)
        # (the class's policy: while the launch is at most 512 workgroups -- here the entry point is exercised up to its limit)
        assert gf._cycle_lanes == (4 * ((n + 1023) // 1024) * B <= 512)
        gf._cycle_lanes = one_launch
        banks.append(gf)
    banks[1]._seed = banks[0]._seed
    z0 = np.array(orc.bioreactor_g(xss))
    shape = (lambda v: v[0]) if batch is None else (lambda v: v)
    for c in range(7):
        u = np.repeat([[0.05 + 0.005 * c, 0.2]], B, axis=0)
        z = z0[None, :] + rng.normal(0, 0.3, (B, 2))
        if c == 3:
            z[:, 0] += 60.0                           # far out: a few Gaussians carry the weight afterwards
        r = rng.random(B)
        noise = (rng.standard_normal((n, 11, 5)) * np.sqrt([1e-4, 1e-7, 1e-3, 1e-3, 1e-7])).astype(np.float32)
        for gf in banks:
            if deterministic:
                DeterministicGaussianSum.preload(noise)
            gf.predict(shape(u), 0.1)
            gf.update(shape(u), shape(z))
            gf.resample(r=shape(r))
        assert banks[0]._pend_resample is not None and banks[0]._pend_gather is None       # nothing has run for the resample yet
        assert banks[1]._pend_resample is None and banks[1]._pend_gather is not None       # the ancestors are there already
        if c == 4:                                    # an accessor in between: the pending resample runs through its own kernels
            np.testing.assert_array_equal(banks[0].means.get(), banks[1].means.get())
            assert banks[0]._pend_resample is None
        if c == 5:
            np.testing.assert_array_equal(banks[0].point_estimate(), banks[1].point_estimate())
    np.testing.assert_array_equal(banks[0].means.get(), banks[1].means.get())
    np.testing.assert_array_equal(banks[0].covariances.get(), banks[1].covariances.get())
    np.testing.assert_array_equal(banks[0].weights.get(), banks[1].weights.get())
    assert int(banks[0]._status.item()) == int(banks[1]._status.item())


@pytest.mark.parametrize("n", [1, 7, 33, 64])
def test_fused_cycle_equals_separate_launches(n):
    """predict + update + resample of a small bank in ONE launch (gse_gsukf_cycle) vs the three methods run one kernel
    each: bit-identical means, covariances, weights and ancestors; with Philox noise and with injected noise."""
    xss = orc.steady_state()
    z = np.array(orc.bioreactor_g(xss)) + [0.2, -0.1]
    for deterministic in (False, True):
        banks = []
        for lazy in (True, False):
            gf = make(n, deterministic=deterministic)
            gf.lazy = lazy
            banks.append(gf)
        rng = np.random.default_rng(n)
        for c in range(4):
            u = [0.05 + 0.005 * c, 0.2]
            r = float(rng.random())
            noise = (rng.standard_normal((n, 11, 5)) * np.sqrt([1e-4, 1e-7, 1e-3, 1e-3, 1e-7])).astype(np.float32)
            idx = []
            for gf in banks:
                if deterministic:
                    DeterministicGaussianSum.preload(noise)
                gf.predict(u, 0.1)
                gf.update(u, z)
                idx.append(gf.resample(r=r, return_index=True).get())
            np.testing.assert_array_equal(idx[0], idx[1])
            np.testing.assert_array_equal(banks[0].means.get(), banks[1].means.get())
            np.testing.assert_array_equal(banks[0].covariances.get(), banks[1].covariances.get())
            np.testing.assert_array_equal(banks[0].weights.get(), banks[1].weights.get())
        np.testing.assert_array_equal(banks[0].point_estimate(), banks[1].point_estimate())


# ------------------------------------------------------------------------------ B banks in one launch
def make_batch(B, n, means, deterministic=False):
    gpu_se_amd.seed(21)
    state_pdf, measurement_pdf = sim_base.get_noise(deterministic=deterministic)
    x0, _ = sim_base.get_noise()
    x0.means += orc.steady_state()[None, :].astype(np.float32)
    gf = BatchedGSF(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, n, x0, state_pdf, measurement_pdf, batch=B)
    gf.means = means
    return gf


@pytest.mark.parametrize("n", [7, 64, 300, 400, 1 << 15])
def test_batch_of_banks_equals_single_banks(n):
    """B independent banks through the batch dimension of every launch (per-bank u, z, r and Philox stream) give, bank by
    bank, bit-identical results to B separate filters: the one-launch cycle (n <= 64: one workgroup per bank) and the lane
    kernels with the one-launch (300, 400) and the three-kernel (2^15) search for the ancestors."""
    B = 3 if n < (1 << 15) else 2
    xss = orc.steady_state()
    rng = np.random.default_rng(n)
    means = (xss[None, None, :] + rng.standard_normal((B, n, 5)) * np.sqrt([1e-4, 1e-7, 1e-3, 1e-3, 1e-7])).astype(np.float32)
    batch = make_batch(B, n, means)
    assert tuple(batch.means.shape) == (B, n, 5) and tuple(batch.covariances.shape) == (B, n, 5, 5)
    assert tuple(batch.weights.shape) == (B, n) and batch.batch == B
    singles = []
    for b in range(B):
        gf = make(n, means[b])
        gf._seed, gf._filter_offset = batch._seed, b
        singles.append(gf)
    for c in range(3):
        u = rng.uniform([0.03, 0.1], [0.09, 0.3], size=(B, 2))
        z = np.array(orc.bioreactor_g(xss))[None, :] + rng.normal(0, 0.3, size=(B, 2))
        r = rng.random(B)
        batch.predict(u, 0.1)
        batch.update(u, z)
        idx = batch.resample(r=r, return_index=True).get()
        for b, gf in enumerate(singles):
            gf.predict(u[b], 0.1)
            gf.update(u[b], z[b])
            np.testing.assert_array_equal(idx[b], gf.resample(r=r[b], return_index=True).get())
            np.testing.assert_array_equal(batch.means.get()[b], gf.means.get())
            np.testing.assert_array_equal(batch.covariances.get()[b], gf.covariances.get())
            np.testing.assert_array_equal(batch.weights.get()[b], gf.weights.get())
    est, cov = batch.estimates()
    assert est.shape == (B, 5) and cov.shape == (B,)
    for b, gf in enumerate(singles):
        np.testing.assert_array_equal(est[b], gf.point_estimate())
        np.testing.assert_allclose(cov[b], gf.point_covariance(), rtol=1e-12)
    assert not np.array_equal(est[0], est[1])
    sig = batch._get_sigma_points().get()
    np.testing.assert_array_equal(sig[1], singles[1]._get_sigma_points().get())


def test_inverse_map_ancestors_of_a_batch_of_banks():
    """Three banks of 70001 Gaussians in one launch of the inverse-map ancestors kernel, each with weights of another kind --
    one Gaussian owning every output (the whole-workgroup stores), a bump (16 lanes / a wavefront per key), exact ties -- and a
    uniform number of its own: every bank's ancestors are those of the single bank (the eager route, itself held against the
    oracle above) and the rows are the ancestors' rows."""
    from tests.gpu_util import adversarial_weights
    B, n = 3, 70001
    xss = orc.steady_state()
    rng = np.random.default_rng(77)
    means = (xss[None, None, :] + rng.standard_normal((B, n, 5)) * np.sqrt([1e-4, 1e-7, 1e-3, 1e-3, 1e-7])).astype(np.float32)
    pats = adversarial_weights(n, rng)
    ws = np.stack([pats[k] for k in ("one_hot_tile_edge", "bump", "uniform")]).astype(np.float32)
    r = np.array([0.0, 0.37, 1.0 - 2.0 ** -53])
    batch = make_batch(B, n, means)
    assert batch.lazy
    batch.weights = ws
    idx = batch.resample(r=r, return_index=True).get()
    for b in range(B):
        gf = make(n, means[b])
        gf.lazy = False
        gf.weights = ws[b]
        want = gf.resample(r=float(r[b]), return_index=True).get()
        np.testing.assert_array_equal(idx[b], want, err_msg=str(b))
        np.testing.assert_array_equal(batch.means.get()[b], means[b][want])
    np.testing.assert_array_equal(batch.weights.get(), np.full((B, n), np.float32(1.0 / n)))


def test_batch_of_banks_with_injected_noise_tracks_the_oracle():
    """The batched class in the parity mode: every bank against its own NumPy oracle fed the same noise."""
    B, n = 2, 40
    xss = orc.steady_state()
    rng = np.random.default_rng(8)
    means = (xss[None, None, :] + rng.standard_normal((B, n, 5)) * np.sqrt([1e-4, 1e-7, 1e-3, 1e-3, 1e-7])).astype(np.float32)
    batch = make_batch(B, n, means, deterministic=True)
    sp, mp = orc.bioreactor_noise()
    oracles = [orc.GSUKFOracle(n, None, sp, mp, means=means[b]) for b in range(B)]
    u = np.array([[0.05, 0.2], [0.07, 0.15]])
    z = np.array(orc.bioreactor_g(xss))[None, :] + np.array([[0.2, -0.1], [-0.3, 0.2]])
    noise = (rng.standard_normal((B, n, 11, 5)) * np.sqrt([1e-4, 1e-7, 1e-3, 1e-3, 1e-7])).astype(np.float32)
    DeterministicGaussianSum.preload(noise)
    batch.predict(u, 0.1)
    batch.update(u, z)
    for b, o in enumerate(oracles):
        o.predict(u[b], 0.1, noise=noise[b], loop=False)
        prior_cov = o.covariances.copy()
        o.update(u[b], z[b])
        assert_mean_close(batch.means.get()[b], o.means, prior_cov, rel_sigma=2e-4)
        assert_cov_close(batch.covariances.get()[b], o.covariances, 2e-4)
        np.testing.assert_allclose(batch.weights.get()[b], o.weights, rtol=2e-3)


# ------------------------------------------------------------------------------ K cycles enqueued from C++
@pytest.mark.parametrize("n,batch", [(7, None), (64, None), (300, None), (4096, None), (1 << 16, None), (1 << 18, None),
                                     (5000, None), (64, 5), (2048, 3), (1 << 14, 2)])
def test_gsukf_run_cycles_equals_the_three_methods(n, batch):
    """ParallelGaussianSumUnscentedKalmanFilter.run_cycles (gse_gsukf_cycles: K cycles enqueued from C++, round 5) against K
    rounds of predict / update / resample from Python -- the loop of results/gsf_openloop/gsf_run_seq.py:16-134: the same
    launches in the class's three regimes (<= 64 Gaussians: one workgroup per bank; up to 2^17: one lane per Gaussian, one
    launch per cycle; beyond: two launches), so banks, weights and estimates bit for bit; a second batch starts from what the
    first left pending; and the filter carries on identically afterwards.  5000 Gaussians with B = 1 is outside the
    one-launch range (not a multiple of... the two-launch regime); batches share every launch."""
    banks = []
    for _ in range(2):
        if batch is None:
            banks.append(make(n))
        else:
            rng0 = np.random.default_rng(5)
            means = (orc.steady_state()[None, None, :] * (1 + 0.01 * rng0.standard_normal((batch, n, 5)))).astype(np.float32)
            banks.append(make_batch(batch, n, means))
    a, b_ = banks
    rng = np.random.default_rng(3)
    xss = orc.steady_state()
    K, B = 9, batch or 1
    us = np.stack([rng.uniform(0.03, 0.09, (K, B)), rng.uniform(0.1, 0.3, (K, B))], axis=-1)
    zs = np.array(orc.bioreactor_g(xss))[None, None, :] + rng.normal(0, 0.3, (K, B, 2))
    shape = (lambda v: v[0]) if batch is None else (lambda v: v)
    for part in (slice(0, 4), slice(4, 9)):                 # (the second call starts from what the first left pending)
        a.run_cycles(us[part] if batch else us[part, 0], zs[part] if batch else zs[part, 0], 0.1)
        for k in range(part.start, part.stop):
            b_.predict(shape(us[k]), 0.1)
            b_.update(shape(us[k]), shape(zs[k]))
            b_.resample()
        assert (a._pend_resample is None) == (b_._pend_resample is None)
        assert (a._pend_gather is None) == (b_._pend_gather is None)
    assert a._step == b_._step
    np.testing.assert_array_equal(a.means.get(), b_.means.get())
    np.testing.assert_array_equal(a.covariances.get(), b_.covariances.get())
    np.testing.assert_array_equal(host_w(a), host_w(b_))
    for f in (a, b_):
        f.predict(shape(us[0]), 0.1)
        f.update(shape(us[0]), shape(zs[0]))
        f.resample()
    np.testing.assert_array_equal(a.point_estimate(), b_.point_estimate())
    np.testing.assert_array_equal(a.means.get(), b_.means.get())
    assert int(a._status.item()) == 0


def host_w(f):
    f._flush()
    return f._w.cpu().numpy()
