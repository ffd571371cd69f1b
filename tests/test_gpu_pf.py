"""Parity of the HIP particle-filter path against the CPU oracle and the golden vectors.  All calls go
through the C ABI (ctypes) or the drop-in classes built on it.  Run with -m gpu on an MI355X."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():          # collected everywhere, executed only with a device
    pytest.skip("no HIP device", allow_module_level=True)

import gpu_se_amd                                            # noqa: E402
from gpu_se_amd import _lib, sim_base                        # noqa: E402
from gpu_se_amd.filter import ParallelParticleFilter         # noqa: E402
from gpu_se_amd.gaussian_sum_dist import DeterministicGaussianSum, MultivariateGaussianSum   # noqa: E402
from gpu_se_amd.model import Bioreactor                      # noqa: E402
from oracle import gse_oracle as orc                         # noqa: E402
from tests.gpu_util import adversarial_weights, dev, host, padded, stream, to_state_major, workspace   # noqa: E402


def test_library_is_the_hip_build():
    assert gpu_se_amd.version() == 100
    assert _lib.load()._name.endswith("libgse_hip.so")


# ------------------------------------------------------------------------------ RNG
def test_rng_words_bit_exact():
    """Device Philox4x32-10 words == the published algorithm restated in NumPy, for several streams."""
    for seed, step, purpose, off in [(1, 0, 1, 0), (0xDEADBEEFCAFEF00D, 77, 2, 0),
                                     (12345, (3 << 32) + 5, 3, (1 << 33) + 17)]:
        m = 3000
        out = torch.empty((m, 8), dtype=torch.int32, device="cuda")
        rng = _lib.Rng(seed=seed, step=step, element_offset=off, filter_offset=0)
        _lib.call("gse_rng_raw", out.data_ptr(), m, ctypes.byref(rng), purpose, stream())
        got = host(out).view(np.uint32)
        want = orc.rng_words8(seed, step, purpose, np.arange(m, dtype=np.uint64) + np.uint64(off))
        np.testing.assert_array_equal(got, want)


def test_mixture_draw_matches_word_mapping():
    """draw() == mean + L n with n = Box-Muller(words); fp32 fast transcendentals vs fp64: 2e-5 sigma."""
    gpu_se_amd.seed(4242)
    sp, _ = sim_base.get_noise()
    osp, _ = orc.bioreactor_noise()
    m = 100000
    x = sp.draw(m).get()
    words = orc.rng_words8(sp._seed, 0, orc.RNG_PURPOSE["draw"], np.arange(m))
    want, comp = orc.words_to_draw(words, osp)
    sd = np.sqrt(np.array([np.diag(osp.covariances[c]) for c in comp]))
    assert np.max(np.abs(x - want) / sd) < 2e-4
    assert x.dtype == np.float32 and x.shape == (m, 5)
    # shape handling like the reference: tuple shapes append Nx
    assert tuple(sp.draw((7, 11)).shape) == (7, 11, 5)
    assert tuple(sp.draw().shape) == (1, 5)


def test_mixture_draw_statistics():
    gpu_se_amd.seed(99)
    _, mp = sim_base.get_noise()
    x = mp.draw(400000).get().astype(np.float64)
    mean = 0.85 * np.array([0.1, 0]) + 0.15 * np.array([0, -0.1])
    cov = (0.85 * (np.diag([6e-2, 8e-2]) + np.outer([0.1, 0], [0.1, 0]))
           + 0.15 * (np.array([[500, 100], [100, 700.]]) + np.outer([0, -0.1], [0, -0.1])) - np.outer(mean, mean))
    np.testing.assert_allclose(np.cov(x.T), cov, rtol=0.03, atol=0.02)
    assert np.all(np.abs(x.mean(axis=0) - mean) < 5 * np.sqrt(np.diag(cov) / len(x)))


# ------------------------------------------------------------------------------ model f, g, pdf
def test_model_f_g_rows(golden):
    g = golden("model_fg.npz")
    x = dev(g["x"])
    for c in range(len(g["dt"])):
        out = torch.empty_like(x)
        _lib.call("gse_model_f", x.data_ptr(), out.data_ptr(), x.shape[0], _lib.doubles(g["u"][c], 2), float(g["dt"][c]), stream())
        want = orc.bioreactor_f_vec(g["x"], g["u"][c], g["dt"][c])
        # bit-exact against the fp64 oracle rounded once to fp32
        np.testing.assert_array_equal(host(out), want.astype(np.float32))
        # and within the stated tolerance of the reference's own NumPy evaluation (mixed precision)
        scale = np.abs(g["f_inc"][c]).max(axis=0, keepdims=True) + 1e-30
        assert np.max(np.abs(host(out) - g["f_inc"][c]) / scale) < 1e-5
    y = torch.empty((x.shape[0], 2), dtype=torch.float32, device="cuda")
    _lib.call("gse_model_g", x.data_ptr(), y.data_ptr(), x.shape[0], stream())
    np.testing.assert_array_equal(host(y).astype(np.float64), g["g_out"][0])


def test_mixture_pdf(golden):
    g = golden("mixture_pdf.npz")
    sp, mp = sim_base.get_noise()
    # fp64 exp on the device vs NumPy's: a few ulp
    # (atol: results in the fp64 denormal range, below 2.3e-308, may be flushed or rounded differently)
    np.testing.assert_allclose(mp.pdf(g["e_meas"]).get(), g["p_meas"], rtol=1e-13, atol=1e-300)
    np.testing.assert_allclose(mp.pdf(g["e_meas"].astype(np.float32)).get(), g["p_meas_f32"], rtol=1e-13, atol=1e-300)
    np.testing.assert_allclose(sp.pdf(g["e_state"]).get(), g["p_state"], rtol=1e-11, atol=0)
    assert (mp.pdf(g["e_meas"]).get() == 0).any()
    toy = MultivariateGaussianSum(g["toy_means"], g["toy_covs"].astype(np.float64), g["toy_weights"])
    np.testing.assert_allclose(toy.pdf(g["e_toy"]).get(), g["p_toy"], rtol=1e-5)
    # single point, host list (tests/MultivariateGaussianSum_test.py)
    assert tuple(mp.pdf([0.1, 0.0]).shape) == (1,)


# ------------------------------------------------------------------------------ predict
@pytest.mark.parametrize("n", [1, 3, 1000, 1024, 4097])
def test_predict_injected_bit_exact(n):
    rng = np.random.default_rng(n)
    xss = orc.steady_state()
    x = (xss[None, :] * (1 + 0.2 * rng.uniform(-1, 1, (n, 5)))).astype(np.float32)
    x[:, 3] = rng.uniform(-0.02, 0.2, n)
    noise = (rng.standard_normal((n, 5)) * np.sqrt([1e-3, 1e-6, 1e-2, 1e-2, 1e-6])).astype(np.float32)
    u, dt = [0.047, 0.163], 0.1
    sp, mp = orc.bioreactor_noise()
    o = orc.ParticleFilterOracle(n, None, sp, mp, particles=x)
    o.predict_vec(u, dt, noise=noise)
    xd, ld = to_state_major(x)
    nd, _ = to_state_major(noise)
    _lib.call("gse_pf_predict_injected", xd.data_ptr(), ld, 1, n, _lib.doubles(u, 2), dt, nd.data_ptr(), stream())
    np.testing.assert_array_equal(host(xd)[:, :n].T, o.particles)


def test_predict_rng_matches_word_mapping():
    """The fused predict kernel = oracle ODE step + the draw mapped from the same Philox words."""
    n = 5000
    gpu_se_amd.seed(7)
    _, _, _, pf = sim_base.get_parts(N_particles=n)
    x0 = pf.particles.get().copy()
    u, dt = [0.05, 0.2], 1.0
    pf.predict(u, dt)
    osp, omp = orc.bioreactor_noise()
    o = orc.ParticleFilterOracle(n, None, osp, omp, particles=x0)
    noise, comp = orc.predict_noise(pf._seed, 0, np.arange(n), osp)
    o.predict_vec(u, dt, noise=noise.astype(np.float32))
    sd = np.sqrt(np.array([np.diag(osp.covariances[c]) for c in comp]))
    assert np.max(np.abs(pf.particles.get() - o.particles) / sd) < 1e-3   # fp32 add of a 2e-5-sigma-accurate normal


# ------------------------------------------------------------------------------ update
@pytest.mark.parametrize("n", [1, 5, 1024, 3001])
def test_update_against_oracle(n):
    rng = np.random.default_rng(100 + n)
    xss = orc.steady_state()
    x = (xss[None, :] * (1 + 0.002 * rng.standard_normal((n, 5)))).astype(np.float32)
    w = rng.random(n).astype(np.float32)
    z = np.array(orc.bioreactor_g(xss)) + [0.3, -0.4]
    sp, mp = orc.bioreactor_noise()
    o = orc.ParticleFilterOracle(n, None, sp, mp, particles=x)
    o.weights = w.copy()
    o.update_vec([0.06, 0.2], z)
    xd, ld = to_state_major(x)
    wd = padded(w, ld)
    ws = workspace(1, n)
    _, pmp = sim_base.get_noise()
    pod = pmp._pod()
    _lib.call("gse_pf_update", xd.data_ptr(), wd.data_ptr(), ld, 1, n, _lib.doubles([0.06, 0.2], 2), _lib.doubles(z, 2),
              ctypes.byref(pod), ws.data_ptr(), stream())
    got = host(wd)[:n]
    # fp64 likelihood (device exp within a few ulp of NumPy's), one rounding to fp32: at most one fp32 ulp apart
    np.testing.assert_allclose(got, o.weights, rtol=1.2e-7, atol=0)
    # the tile sums left in the workspace are the fp64 sums of the new fp32 weights
    t = -(-n // 1024)
    sums = host(ws[:8 * t].view(torch.float64))
    want = [got[i * 1024:(i + 1) * 1024].astype(np.float64).sum() for i in range(t)]
    np.testing.assert_allclose(sums, want, rtol=1e-14)


# ------------------------------------------------------------------------------ resample
def run_resample(w, r, x=None, flags=0):
    n = len(w)
    if x is None:
        x = np.arange(n, dtype=np.float32)[:, None] + np.arange(5, dtype=np.float32)[None, :] * 0.125
    xd, ld = to_state_major(x)
    wd = padded(np.asarray(w, dtype=np.float32), ld)
    xo = torch.full_like(xd, float("nan"))
    wo = torch.full_like(wd, float("nan"))
    idx = torch.full((ld,), -1, dtype=torch.int32, device="cuda")
    ws = workspace(1, n)
    _lib.call("gse_pf_resample", xd.data_ptr(), wd.data_ptr(), xo.data_ptr(), wo.data_ptr(), idx.data_ptr(), ld, 1, n,
              _lib.doubles([r], 1), flags, None, ws.data_ptr(), stream())
    return host(idx)[:n], host(xo)[:, :n].T, host(wo)[:n], x


def test_resample_golden_indices(golden):
    """Ancestor indices are integer work: bit-exact against the reference's own resample (golden) and the oracle."""
    g = golden("resample.npz")
    for name in g["names"]:
        w, r = g[name + "_w"], float(g[name + "_r"])
        idx, xo, wo, x = run_resample(w, r)
        np.testing.assert_array_equal(idx, g[name + "_idx"], err_msg=str(name))
        np.testing.assert_array_equal(idx, orc.systematic_index(w.astype(np.float64), r), err_msg=str(name))
        np.testing.assert_array_equal(xo, x[idx], err_msg=str(name))
        np.testing.assert_array_equal(wo, np.full(len(w), np.float32(1.0 / len(w))))


@pytest.mark.parametrize("n", [2, 1023, 1025, 2048, 5000, 65536 + 3])
def test_resample_random_weights_vs_oracle(n):
    rng = np.random.default_rng(n)
    for kind in range(4):
        w = rng.random(n).astype(np.float32)
        if kind == 1:
            w = (w ** 12).astype(np.float32)
        if kind == 2:
            w[rng.random(n) < 0.7] = 0
            w[n // 2] = 1
        if kind == 3:
            w = np.exp(-0.5 * ((np.arange(n) - 0.3 * n) / (0.01 * n + 1)) ** 2).astype(np.float32)
        r = float(rng.random())
        idx, xo, wo, x = run_resample(w, r)
        want = orc.systematic_index(w.astype(np.float64), r)
        # the device scans tile-wise, the oracle left to right: their fp64 cumulative sums may differ in the
        # last bit, which can move an ancestor by one only when a threshold lands within ~1e-16 of a
        # cumulative value -- none expected at these sizes
        np.testing.assert_array_equal(idx, want)
        np.testing.assert_array_equal(xo, x[idx])


def test_resample_degenerate_weights():
    n = 3000
    for w in (np.zeros(n, np.float32), np.full(n, np.nan, np.float32)):
        idx, xo, wo, x = run_resample(w, 0.3)
        np.testing.assert_array_equal(idx, np.zeros(n, dtype=np.int32))     # the CPU class's sweep never advances
        np.testing.assert_array_equal(xo, np.repeat(x[:1], n, axis=0))
    w = np.zeros(n, np.float32)
    w[-1] = 5.0
    idx, *_ = run_resample(w, 0.999)
    np.testing.assert_array_equal(idx, np.full(n, n - 1))
    with pytest.raises(_lib.GseError):
        run_resample(np.ones(8, np.float32), 1.0)                            # r outside [0, 1)


def test_resample_full_size_properties():
    """N = 2^20 (BASELINE config 2): size-independent properties of systematic resampling."""
    n = 1 << 20
    rng = np.random.default_rng(20)
    w = rng.random(n).astype(np.float32) ** 3
    r = float(rng.random())
    idx, xo, wo, x = run_resample(w, r)
    assert np.all(np.diff(idx) >= 0) and idx.min() >= 0 and idx.max() < n          # sorted, in range
    counts = np.bincount(idx, minlength=n)
    expect = w.astype(np.float64) / w.astype(np.float64).sum() * n
    assert np.all(np.abs(counts - expect) < 1.0 + 1e-6)                            # |count - N w| < 1
    np.testing.assert_array_equal(xo, x[idx])
    want = orc.systematic_index(w.astype(np.float64), r)
    assert np.sum(idx != want) <= 2                                                 # ties on the last bit only
    # uniform weights, any r: identity
    idx, *_ = run_resample(np.full(n, 1.0 / n, np.float32), 0.5)
    np.testing.assert_array_equal(idx, np.arange(n))


def test_resample_scan_paths_agree():
    """The tile prefix is computed inside the resample kernel up to 4096 tiles and by a scan kernel of its own
    beyond; both must give the oracle's ancestors (the two differ at most in the last bit of a cumulative sum)."""
    rng = np.random.default_rng(77)
    for n in (5000, (1 << 18) + 11, 4100 * 1024 + 3):           # the last one has > 4096 tiles
        w = (rng.random(n) ** 2).astype(np.float32)
        r = float(rng.random())
        want = orc.systematic_index(w.astype(np.float64), r)
        for flags in (0, _lib.GSE_RESAMPLE_SEPARATE_SCAN):
            idx, xo, wo, x = run_resample(w, r, flags=flags)
            assert np.sum(idx != want) <= 2, (n, flags)
            assert np.all(np.diff(idx) >= 0)
            np.testing.assert_array_equal(xo, x[idx])


# ------------------------------------------------------------------------------ moments
@pytest.mark.parametrize("n", [1, 7, 1024, 50001])
def test_moments(n):
    rng = np.random.default_rng(n + 5)
    xss = orc.steady_state()
    x = (xss[None, :] + rng.standard_normal((n, 5)) * np.sqrt([1e-3, 1e-6, 1e-2, 1e-2, 1e-6])).astype(np.float32)
    w = (rng.random(n) / n).astype(np.float32)
    sp, mp = orc.bioreactor_noise()
    o = orc.ParticleFilterOracle(n, None, sp, mp, particles=x)
    o.weights = w.astype(np.float64)
    gpu_se_amd.seed(1)
    _, _, _, pf = sim_base.get_parts(N_particles=n)
    pf.particles = x
    pf.weights = w
    np.testing.assert_allclose(pf.point_estimate(), o.point_estimate(), rtol=1e-12)
    if n > 1:
        np.testing.assert_allclose(pf.point_covariance(), o.point_covariance(), rtol=1e-9)


# ------------------------------------------------------------------------------ the class, golden trace
def make_filter(n, particles, deterministic=False):
    gpu_se_amd.seed(11)
    state_pdf, measurement_pdf = sim_base.get_noise(deterministic=deterministic)
    x0, _ = sim_base.get_noise()
    pf = ParallelParticleFilter(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, n, x0, state_pdf, measurement_pdf)
    pf.particles = particles
    return pf


def test_pf_trace_per_step(golden):
    """The reference's own 20-cycle trace (golden G4), each step started from the reference's state with its
    recorded noise and r, through the drop-in class: particles to one fp32 rounding, weights 2e-6,
    ancestors exact."""
    g = golden("pf_trace.npz")
    n = len(g["w0"])
    pf = make_filter(n, g["x0"], deterministic=True)
    for c in range(len(g["dt"])):
        DeterministicGaussianSum.preload(g["noise"][c])
        pf.predict(g["u"][c], g["dt"][c])
        np.testing.assert_allclose(pf.particles.get(), g["after_predict"][c], rtol=5e-7, atol=1e-7)   # (3e-6 of the increments)
        pf.particles = g["after_predict"][c]
        pf.update(g["u"][c], g["z"][c])
        np.testing.assert_allclose(pf.weights.get(), g["w_after_update"][c], rtol=2e-6)
        pf.weights = g["w_after_update"][c]
        idx = pf.resample(r=float(g["r"][c]), return_index=True)
        w32 = g["w_after_update"][c].astype(np.float32).astype(np.float64)
        np.testing.assert_array_equal(idx.get(), orc.systematic_index(w32, float(g["r"][c])))
        # vs the reference's ancestors (it resampled fp64 weights; ours are their fp32 roundings): flips only where
        # a threshold falls within 6e-8 relative of a cumulative weight
        assert np.sum(idx.get() != g["idx"][c]) <= 2
        pf.particles = g["after_resample"][c]
        np.testing.assert_allclose(pf.point_estimate(), g["estimate"][c], rtol=1e-6)
        np.testing.assert_allclose(pf.point_covariance(), g["cov_sv"][c], rtol=1e-5)


def test_api_surface():
    gpu_se_amd.seed(3)
    bioreactor, lin_model, K, pf = sim_base.get_parts(N_particles=2000)
    np.testing.assert_allclose(bioreactor.X, orc.steady_state(), rtol=1e-12)
    assert tuple(pf.particles.shape) == (2000, 5) and pf.particles.dtype == torch.float32
    assert tuple(pf.weights.shape) == (2000,)
    assert pf.N_particles == 2000 and pf._tpb == 1024 and pf._bpg == 2
    u, z = [0.06, 0.2], np.array([280.1, 611.0])
    pf.predict(u, 0.1)
    pf.update(u, z)
    pf.resample()
    est = pf.point_estimate()
    assert est.shape == (5,) and np.all(np.isfinite(est))
    assert np.isfinite(pf.point_covariance())
    # pieces the benchmark scripts use (pf_run_seq.py:172, :223, :282-287)
    inc = pf.f_vectorize(pf.particles, u, 1.0)
    assert tuple(inc.shape) == (2000, 5)
    ys = pf.g_vectorize(pf.particles, u, pf._y_dummy)
    np.testing.assert_array_equal(ys.get(), orc.bioreactor_g_vec(pf.particles.get()))
    w = np.random.default_rng(0).random(2000)
    pf.weights = w / w.sum()                       # host array assignment (pf_run_seq.py:124-125)
    c = torch.empty(2000, dtype=torch.float64, device="cuda")
    _lib.call("gse_cumsum_normalised", pf.weights.data_ptr(), c.data_ptr(), 2000, pf._ws.data_ptr(), stream())
    w32 = pf.weights.get().astype(np.float64)
    np.testing.assert_allclose(host(c), np.cumsum(w32) / w32.sum(), rtol=1e-14)
    sample_index = torch.zeros(2000, dtype=torch.int64, device="cuda")
    ParallelParticleFilter._parallel_resample[pf._bpg, pf._tpb](c, sample_index, 0.25, 2000)
    np.testing.assert_array_equal(host(sample_index), orc.systematic_index(w32, 0.25))
    with pytest.raises(NotImplementedError):
        ParallelParticleFilter(lambda x, u, dt: x, Bioreactor.static_outputs, 8, *[sim_base.get_noise()[0]] * 2,
                               sim_base.get_noise()[1])


def test_statistical_parity_with_reference(golden):
    """Philox-driven HIP filter vs the reference's NumPy filter (MT19937) on one fixed (u, z) sequence: the
    posterior mean must fall inside the envelope of the reference's own seed-to-seed scatter (golden G6)."""
    g = golden("pf_stat.npz")                           # (64 seeds x 50 cycles at N = 1024 and N = 16384)
    for n, key in ((1024, "est_n1024"), (16384, "est_n16384")):
        ref = g[key]                                     # (seeds, cycles, 5)
        mu, sd = ref.mean(axis=0), ref.std(axis=0, ddof=1)
        runs = []
        for s in range(8):
            gpu_se_amd.seed(1000 + s)
            _, _, _, pf = sim_base.get_parts(N_particles=n)
            est = []
            for c in range(len(g["u"])):
                pf.predict(g["u"][c], float(g["dt"]))
                pf.update(g["u"][c], g["z"][c])
                pf.resample()
                est.append(pf.point_estimate())
            runs.append(np.array(est))
        runs = np.array(runs)
        zscore = (runs.mean(axis=0) - mu) / (sd * np.sqrt(1 / 8 + 1 / ref.shape[0]) + 1e-9)
        zscore = zscore[:, [0, 2, 3, 4]]                 # Cx has no dynamics and no information: skip
        assert np.mean(np.abs(zscore) < 3.5) > 0.97, np.abs(zscore).max()
        assert np.abs(zscore).max() < 6


def test_sharded_filter_single_rank_bookkeeping():
    """ShardedParticleFilter on one rank: the device-side island bookkeeping (log-weight accumulation, last
    weight sum) and the normalised posterior merge."""
    from gpu_se_amd.distributed import ShardedParticleFilter
    gpu_se_amd.seed(31)
    spf = ShardedParticleFilter.from_get_parts(n_local=6000)
    xss = orc.steady_state()
    z = np.array(orc.bioreactor_g(xss)) + [0.1, 0.2]
    log_acc = 0.0
    for c in range(3):
        spf.predict([0.06, 0.2], 0.1)
        spf.update([0.06, 0.2], z)
        w = spf.local.weights.get().astype(np.float64)
        x = spf.local.particles.get().astype(np.float64)
        if c == 2:
            want_mean = (w / w.sum()) @ x
            np.testing.assert_allclose(spf.point_estimate(), want_mean, rtol=1e-9)
            dd = x - want_mean
            want_cov = dd.T @ (dd * (w / w.sum())[:, None])
            np.testing.assert_allclose(spf.point_covariance(), np.linalg.svd(want_cov, compute_uv=False)[0], rtol=1e-6)
        spf.resample()
        spf.local._flush()                     # the gather (deferred) is where the bookkeeping happens
        log_acc += np.log(w.sum())
        isl = host(spf.local._island)
        np.testing.assert_allclose(isl[0, 0], log_acc, rtol=1e-12)
        np.testing.assert_allclose(isl[1, 0], w.sum(), rtol=1e-13)


# ------------------------------------------------------------------------------ fused / deferred execution
def _two_filters(n, seed):
    out = []
    for lazy in (True, False):
        gpu_se_amd.seed(seed)
        _, _, _, pf = sim_base.get_parts(N_particles=n)
        pf.lazy = lazy
        out.append(pf)
    return out


# 2^21 = one shard of BASELINE config 4 and the largest N whose cycle kernel rebuilds the tile prefix in every workgroup (2048
# tiles); one more tile and the prefix comes from a one-workgroup scan (GSE_OP_PREFIX_GLOBAL: windows of it staged in LDS):
# both sides are covered, and 2^22 + 1024
@pytest.mark.parametrize("n", [7, 1024, 5003, 1 << 17, 1 << 20, 1 << 21, (1 << 21) + 1024, (1 << 22) + 1024])
def test_fused_cycle_equals_eager_calls(n):
    """resample / predict / update deferred into ONE fused pass must give bit-identical particles and weights to the
    same calls executed one kernel each (lazy=False), for any interleaving of calls and accessors."""
    fused, eager = _two_filters(n, 77)
    np.testing.assert_array_equal(fused.particles.get(), eager.particles.get())
    xss = orc.steady_state()
    rng = np.random.default_rng(n)
    for c in range(6):
        u = [rng.uniform(0.03, 0.09), rng.uniform(0.1, 0.3)]
        z = np.array(orc.bioreactor_g(xss)) + rng.normal(0, 0.3, 2)
        r = float(rng.random())
        for pf in (fused, eager):
            pf.predict(u, 0.1)
            if c == 2:
                pf.predict(u, 0.05)                # two predicts in a row
            pf.update(u, z)
            if c == 3:
                pf.update(u, z)                    # two updates in a row
            pf.resample(r=r)
            if c == 4:
                pf.point_estimate()                # estimate between resample and predict (closed loop order)
        if c in (0, 3, 5):
            np.testing.assert_array_equal(fused.particles.get(), eager.particles.get())
            np.testing.assert_array_equal(fused.weights.get(), eager.weights.get())
    np.testing.assert_array_equal(fused.point_estimate(), eager.point_estimate())


@pytest.mark.parametrize("n", [5003, 1 << 20, 1 << 21, (1 << 21) + 1024, (1 << 22) + 1024])
def test_single_launch_philox_cycle_against_oracle(n):
    """The benchmarked kernel variant itself -- pf_step_kernel<gather, predict (Philox), update, scan in + out>, one
    launch per cycle -- against the oracle fed the same Philox words (oracle.predict_noise), at BASELINE config 2's size,
    at config 4's shard size, and one tile beyond it where the class switches to cumsum + fused step.  Per cycle:
    ancestors vs oracle.systematic_index on the device's fp32 weights (<= 2 last-bit ties of the fp64 cumulative sums),
    particles within 1e-3 sigma (+ one fp32 rounding) of the oracle's ODE step + mapped draw, weights to 2e-6 of the oracle's update on
    the device's particles."""
    gpu_se_amd.seed(2020)
    _, _, _, pf = sim_base.get_parts(N_particles=n)
    assert pf.lazy and pf._prefix_global == (n > (1 << 21))
    osp, omp = orc.bioreactor_noise()
    xss = orc.steady_state()
    rng = np.random.default_rng(n)
    x = pf.particles.get().copy()
    w = pf.weights.get().copy()
    dt = 1.0
    for c in range(3):
        u = [rng.uniform(0.03, 0.09), rng.uniform(0.1, 0.3)]
        z = np.array(orc.bioreactor_g(xss)) + rng.normal(0, 0.3, 2)
        r = float(rng.random())
        want_idx = None
        if c > 0:
            pf.resample(r=r)                       # deferred: runs inside the launch that carries the update
            pf._want_index = True                  # (test hook: have that launch write the ancestors out too)
            want_idx = orc.systematic_index(w.astype(np.float64), r)
        pf.predict(u, dt)
        pf.update(u, z)
        assert pf._scan_slot is not None
        x_new = pf.particles.get().copy()
        w_new = pf.weights.get().copy()
        good = np.ones(n, dtype=bool)
        if c > 0:
            idx = host(pf._last_index)[0, :n]
            good = idx == want_idx
            assert np.sum(~good) <= 2, int(np.sum(~good))
            x = x[idx]
            w = np.full(n, np.float32(1.0 / n), dtype=np.float32)
        o = orc.ParticleFilterOracle(n, None, osp, omp, particles=x)
        noise, comp = orc.predict_noise(pf._seed, c, np.arange(n), osp)
        o.predict_vec(u, dt, noise=noise.astype(np.float32))
        sd = np.sqrt(np.diagonal(osp.covariances, axis1=1, axis2=2))[comp]
        # (1e-3 sigma of the noise component; plus one rounding of the fp32 state where the 2e-5-sigma difference between
        #  the device's fp32 fast-math normal and the fp64 one flips the last bit of the sum -- Cx = 26 has an fp32 ulp of
        #  1.9e-6 against a noise sigma of 3e-4)
        bound = 1e-3 * sd + 1.2e-7 * np.abs(o.particles)
        assert np.all(np.abs(x_new - o.particles) <= bound), float(np.max(np.abs(x_new - o.particles) / bound))
        o.particles = x_new
        o.weights = w.copy()
        o.update_vec(u, z)
        np.testing.assert_allclose(w_new, o.weights, rtol=2e-6, atol=1.2e-38)
        x, w = x_new, w_new


@pytest.mark.parametrize("n", [8, 1000, 4096, 100003, 1 << 20, (1 << 22) + 1024])
def test_inverse_map_resample_on_adversarial_weights(n):
    """The single-launch cycle's resample (inverse map from the weights, tools of DESIGN.md section 3) on weight patterns
    chosen to hurt: exact ties (one-hot, interleaved zeros: runs of equal cumulative weights), a single heavy particle at
    tile edges, a bump over the index (ancestor ranges of a few keys next to ranges of many tiles: the census path), a
    long weightless stretch over a floor, contiguous lineages a few hundred times lighter than the rest (round 3: the work lists
    beyond GSE_SEARCH_FROM tiles, where every output looks its ancestor up), weights that sum to zero or to NaN, r at both ends of [0, 1) -- r = 0 puts
    every threshold ON an integer of the offspring count, i.e. into the guard band, where the count is settled with the
    reference's own comparison.  All particles sit on one state, so the update multiplies the assigned weights by ONE
    likelihood and leaves the pattern in place (plus the tile sums and prefixes the next launch resamples from); the
    ancestors of that launch must be those of the materialised-cumsum route exactly, and the oracle's up to last-bit ties."""
    gpu_se_amd.seed(17)
    _, _, _, pf = sim_base.get_parts(N_particles=n)
    xss = orc.steady_state().astype(np.float32)
    u, z = [0.06, 0.2], np.array(orc.bioreactor_g(orc.steady_state()))
    rng = np.random.default_rng(n)
    patterns = adversarial_weights(n, rng)
    for name, w in patterns.items():
        for r in (0.0, 0.37, 1.0 - 2.0 ** -53):
            pf.particles = np.repeat(xss[None, :], n, axis=0)
            pf.weights = (w / np.nansum(w) if np.nansum(w) > 0 else w).astype(np.float32)
            pf.update(u, z)                                    # weights <- weights * one likelihood; tile sums + prefixes out
            assert pf._scan_slot is not None
            stored32 = host(pf._w)[0, :n].copy()                # (not through the attribute: that would drop the prefixes)
            stored = stored32.astype(np.float64)
            total = stored.sum()
            want = orc.systematic_index(stored, r) if (total > 0 and np.isfinite(total)) else np.zeros(n, dtype=np.int64)
            # ... and this library's other route to the same ancestors: materialised cumulative weights (cumsum_kernel, the
            # same tile-wise association of the sums) + the staged search of gse_pf_resample -- bit for bit
            eager = run_resample(stored32, r)[0] if r < 1.0 else None
            pf.resample(r=r)
            assert pf._pend_gather and pf._gather_slot is not None
            pf._want_index = True
            pf.predict(u, 0.1)
            pf.update(u, z)                                    # the single launch: gather (inverse map) + predict + update
            got = host(pf._last_index)[0, :n]
            np.testing.assert_array_equal(got, eager, err_msg="%s, r = %r" % (name, r))
            # against the oracle's left-to-right cumulative sum: equal, except where a threshold meets a cumulative weight
            # to the last bit (the symmetric bump puts c = 1/2 exactly ON a threshold) and the two associations round
            # that bit differently -- then the neighbour
            off = np.flatnonzero(got != want)
            if len(off):
                c_or = orc.normalised_cumsum(stored)
                lo, hi = np.minimum(got[off], want[off]), np.maximum(got[off], want[off])
                # (every key between the two answers carries the same cumulative weight to the last bits: a plateau of
                #  weightless particles, or neighbours)
                tie = c_or[hi] - c_or[lo] <= 4.5e-16
                # ... or, among millions of keys, a threshold that falls between the two sums' ROUNDING of one key: NumPy adds
                # 4.2 M terms one after the other (error growing like sqrt(n) eps ~ 2e-13), the device per thread, wavefront,
                # tile and prefix (~1e-15); with thresholds 2.4e-7 apart that happens to about one key in three patterns
                near = np.abs(c_or[lo] - (off + r) / n) <= 2e-12 if n > (1 << 21) else np.zeros(len(off), dtype=bool)
                assert len(off) <= 2 and np.all(tie | near), "%s, r = %r: %s" % (name, r, off[:10])


def test_in_place_writes_through_the_attribute_views_are_seen():
    """The reference's attributes are plain arrays that callers may change in place (pf.weights[:] = ...).  The views this
    class hands out write through to the device buffers; whatever was cached from the old contents (moments, the tile sums
    and prefixes a deferred resample would use) must not survive."""
    n = 6000
    gpu_se_amd.seed(3)
    _, _, _, pf = sim_base.get_parts(N_particles=n)
    xss = orc.steady_state()
    pf.predict([0.06, 0.2], 0.1)
    pf.update([0.06, 0.2], np.array(orc.bioreactor_g(xss)))
    first = pf.point_estimate()
    rng = np.random.default_rng(0)
    w_new = (rng.random(n) ** 5).astype(np.float32)
    view = pf.weights
    view[:] = torch.as_tensor(w_new, device="cuda")                  # in place, not through the setter
    np.testing.assert_array_equal(pf.weights.get(), w_new)
    assert not np.allclose(pf.point_estimate(), first)               # the cached estimate was dropped
    want = orc.systematic_index(w_new.astype(np.float64), 0.3)
    idx = pf.resample(r=0.3, return_index=True).get()
    np.testing.assert_array_equal(idx, want)                         # the resample used the new weights


def test_fused_step_against_oracle():
    """gather + predict(injected) + update in one launch (gse_pf_step) vs the oracle's three separate steps."""
    n = 6000
    rng = np.random.default_rng(8)
    xss = orc.steady_state()
    x = (xss[None, :] * (1 + 0.01 * rng.standard_normal((n, 5)))).astype(np.float32)
    x[:, 3] = rng.uniform(0, 0.05, n)
    w = (rng.random(n) ** 4).astype(np.float32)
    noise = (rng.standard_normal((n, 5)) * np.sqrt([1e-4, 1e-7, 1e-3, 1e-3, 1e-7])).astype(np.float32)
    u, dt, r = [0.055, 0.21], 0.1, 0.3721
    z = np.array(orc.bioreactor_g(xss)) + [0.15, -0.2]
    sp, mp = orc.bioreactor_noise()
    o = orc.ParticleFilterOracle(n, None, sp, mp, particles=x)
    o.weights = w.astype(np.float64)
    idx_want = o.resample(r=r, loop=False)
    o.weights = np.full(n, np.float32(1.0 / n), dtype=np.float32)          # the device keeps fp32 weights
    o.predict_vec(u, dt, noise=noise)
    o.update_vec(u, z)

    xd, ld = to_state_major(x)
    wd = padded(w, ld)
    nd, _ = to_state_major(noise)
    xo, wo = torch.full_like(xd, float("nan")), torch.full_like(wd, float("nan"))
    idx = torch.full((ld,), -1, dtype=torch.int32, device="cuda")
    ws = workspace(1, n)
    _, pmp = sim_base.get_noise()
    pod = pmp._pod()
    _lib.call("gse_pf_resample_begin", wd.data_ptr(), ld, 1, n, 0, None, ws.data_ptr(), stream())
    _lib.call("gse_pf_step", _lib.GSE_OP_GATHER | _lib.GSE_OP_PREDICT | _lib.GSE_OP_UPDATE, xd.data_ptr(), xo.data_ptr(),
              wd.data_ptr(), wo.data_ptr(), idx.data_ptr(), ld, 1, n, _lib.doubles(u, 2), dt, _lib.doubles(z, 2),
              _lib.doubles([r], 1), None, None, nd.data_ptr(), ctypes.byref(pod), None, ws.data_ptr(), stream())
    np.testing.assert_array_equal(host(idx)[:n], idx_want)
    np.testing.assert_array_equal(host(xo)[:, :n].T, o.particles)
    np.testing.assert_allclose(host(wo)[:n], o.weights, rtol=1.2e-7)


def test_resampled_moments_equal_the_moments_after_the_copy():
    """gse_pf_moments_resampled: the moments of the ensemble a pending resample will produce, from the weighted ensemble
    and the offspring counts of the inverse map -- against gse_pf_moments of the particles once they have been copied
    (different summation order: equal to fp64 rounding).  One filter and a batch; also after a second cycle, when the
    single launch gather + predict + update has consumed the deferred resample."""
    from gpu_se_amd.filter import BatchedParticleFilter
    from gpu_se_amd.filter.particle import moments_to_mean_cov
    xss = orc.steady_state()
    from gpu_se_amd.gaussian_sum_dist import MultivariateGaussianSum
    # (the last four: measurement noise so wide that the weights stay near uniform -- from "every particle has its one
    #  offspring" to "a third of them have": the kernel's list of particles with offspring overflows into repeated flushes)
    for n, B, wide in ((5003, 1, 0), (1 << 18, 1, 0), (40000, 3, 0), (70001, 2, 1e6), (1 << 16, 1, 30.0), (50000, 2, 3.0), (30000, 1, 0.5)):
        gpu_se_amd.seed(11)
        sp, mp = sim_base.get_noise()
        if wide:
            mp = MultivariateGaussianSum(means=np.zeros((2, 2)), covariances=np.array([np.eye(2) * wide, np.eye(2) * 2 * wide]),
                                         weights=np.array([0.5, 0.5]))
        x0, _ = sim_base.get_noise()
        x0.means += xss[None, :].astype(np.float32)
        pf = BatchedParticleFilter(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, n, x0, sp, mp, batch=B)
        rng = np.random.default_rng(n)
        for c in range(3):
            u = np.repeat([[rng.uniform(0.03, 0.09), rng.uniform(0.1, 0.3)]], B, axis=0)
            z = np.array(orc.bioreactor_g(xss))[None, :] + rng.normal(0, 0.3, (B, 2))
            pf.predict(u, 0.1)
            pf.update(u, z)
            pf.resample(r=rng.random(B))
            out = torch.empty((B, 26), dtype=torch.float64, device="cuda")
            assert pf._resampled_moments_into(out)
            assert pf._pend_gather                                  # nothing was copied
            ahead = out.cpu().numpy()
            after = np.array([np.concatenate([[s0], s1, s2[np.triu_indices(5)], pv]) for s0, s1, s2, pv in pf._raw_moments_all()])
            assert not pf._pend_gather                              # (that one did copy)
            for b in range(B):
                s0a, ma, ca = moments_to_mean_cov(ahead[b, 0], ahead[b, 1:6], ahead[b, 6:21], ahead[b, 21:26])
                s0b, mb, cb = moments_to_mean_cov(after[b, 0], after[b, 1:6], after[b, 6:21], after[b, 21:26])
                np.testing.assert_allclose(s0a, s0b, rtol=1e-12)
                np.testing.assert_allclose(ma, mb, rtol=1e-12)
                np.testing.assert_allclose(ca, cb, rtol=1e-9, atol=1e-18)


# ------------------------------------------------------------------------------ batches of independent filters
def test_batch_of_filters_equals_single_filters():
    """B independent filters in one launch (leading dimension of every array; per-filter u, z, r, Philox stream) give,
    filter by filter, bit-identical results to B separate single-filter calls -- over two fused single-launch cycles."""
    B, n = 3, 2500
    ld = -(-n // 4) * 4
    rng = np.random.default_rng(4)
    xss = orc.steady_state()
    x0 = (xss[None, None, :] * (1 + 0.01 * rng.standard_normal((B, n, 5)))).astype(np.float32)
    sp, mp = sim_base.get_noise()
    pod_s, pod_m = sp._pod(), mp._pod()
    us = rng.uniform([0.03, 0.1], [0.09, 0.3], size=(2, B, 2))
    zs = np.array(orc.bioreactor_g(xss))[None, None, :] + rng.normal(0, 0.3, size=(2, B, 2))
    rs = rng.random((2, B))
    OPS = _lib.GSE_OP_PREDICT | _lib.GSE_OP_UPDATE | _lib.GSE_OP_SCAN_OUT

    def run(filters, nb):
        """filters: list of filter ids handled by this call sequence (one call with nb = len(filters))."""
        xb = np.full((nb, 5, ld), np.nan, np.float32)
        for i, f in enumerate(filters):
            xb[i, :, :n] = x0[f].T
        xd = dev(xb)
        xa = torch.empty_like(xd)
        wd = torch.zeros((nb, ld), dtype=torch.float32, device="cuda")
        wd[:, :n] = 1.0 / n
        wa = torch.empty_like(wd)
        ws = workspace(nb, n)
        fl = lambda a: _lib.doubles(np.asarray(a).ravel(), np.asarray(a).size)
        # cycle 0: predict + update (+ scans out, slot 0)
        r0 = _lib.Rng(seed=99, step=0, element_offset=0, filter_offset=filters[0])
        _lib.call("gse_pf_step", OPS, xd.data_ptr(), xd.data_ptr(), wd.data_ptr(), wd.data_ptr(), None, ld, nb, n,
                  fl(us[0, filters]), 0.1, fl(zs[0, filters]), None, ctypes.byref(pod_s), ctypes.byref(r0), None,
                  ctypes.byref(pod_m), None, ws.data_ptr(), stream())
        # cycle 1: gather (scans in, slot 0) + predict + update (scans out, slot 1) -- one launch
        r1 = _lib.Rng(seed=99, step=1, element_offset=0, filter_offset=filters[0])
        _lib.call("gse_pf_step", OPS | _lib.GSE_OP_GATHER | _lib.GSE_OP_SCAN_IN | _lib.GSE_OP_SCAN_OUT_SLOT1,
                  xd.data_ptr(), xa.data_ptr(), wd.data_ptr(), wa.data_ptr(), None, ld, nb, n, fl(us[1, filters]), 0.1,
                  fl(zs[1, filters]), fl(rs[1, filters]), ctypes.byref(pod_s), ctypes.byref(r1), None,
                  ctypes.byref(pod_m), None, ws.data_ptr(), stream())
        return host(xa)[:, :, :n], host(wa)[:, :n]

    xb, wb = run([0, 1, 2], 3)
    for f in range(B):
        x1, w1 = run([f], 1)
        np.testing.assert_array_equal(xb[f], x1[0])
        np.testing.assert_array_equal(wb[f], w1[0])
    assert not np.array_equal(xb[0], xb[1])


def test_random_call_sequences_lazy_equals_eager():
    """State-machine soak: 150 random calls (predict / update / resample / reading particles, weights, estimates /
    assigning weights) on a deferred-execution filter and on an eager one must leave bit-identical state at every
    point where it is observed."""
    n = 3000
    fused, eager = _two_filters(n, 123)
    rng = np.random.default_rng(99)
    xss = orc.steady_state()
    for step in range(150):
        op = rng.choice(["predict", "update", "resample", "particles", "weights", "estimate", "set_weights"],
                        p=[0.3, 0.25, 0.2, 0.07, 0.07, 0.07, 0.04])
        u = [rng.uniform(0.03, 0.09), rng.uniform(0.1, 0.3)]
        if op == "predict":
            dt = float(rng.choice([0.05, 0.1]))
            for pf in (fused, eager):
                pf.predict(u, dt)
        elif op == "update":
            z = np.array(orc.bioreactor_g(xss)) + rng.normal(0, 0.5, 2)
            for pf in (fused, eager):
                pf.update(u, z)
        elif op == "resample":
            r = float(rng.random())
            for pf in (fused, eager):
                pf.resample(r=r)
        elif op == "particles":
            np.testing.assert_array_equal(fused.particles.get(), eager.particles.get(), err_msg="step %d" % step)
        elif op == "weights":
            np.testing.assert_array_equal(fused.weights.get(), eager.weights.get(), err_msg="step %d" % step)
        elif op == "estimate":
            np.testing.assert_array_equal(fused.point_estimate(), eager.point_estimate(), err_msg="step %d" % step)
        else:
            w = rng.random(n)
            for pf in (fused, eager):
                pf.weights = w / w.sum()
    np.testing.assert_array_equal(fused.particles.get(), eager.particles.get())
    np.testing.assert_array_equal(fused.weights.get(), eager.weights.get())


def test_cycles_are_hipgraph_capturable_and_replay_bit_identically():
    """include/gse.h: every call is enqueued on the caller's stream without synchronising -- so a stream capture records
    it.  Two cycles of the class (the period of its buffer and scan-slot alternation) captured into a HIP graph and
    replayed twice, against the same four cycles launched eagerly: bit-identical particles and weights.  (What a replay
    re-uses is what a launch carries by value: u, z, r and the Philox step -- the eager twin is handed the same ones.)"""
    n = 1 << 16
    pfs = []
    for _ in range(2):
        gpu_se_amd.seed(404)
        _, _, _, pf = sim_base.get_parts(N_particles=n)
        pfs.append(pf)
    graphed, eager = pfs
    xss = orc.steady_state()
    io = [([0.05 + 0.01 * c, 0.2], np.array(orc.bioreactor_g(xss)) + [0.1 * c, -0.1], 0.3 + 0.2 * c) for c in range(2)]

    def two_cycles(pf):
        for u, z, r in io:
            pf.predict(u, 0.1)
            pf.update(u, z)
            pf.resample(r=r)

    for pf in pfs:                                   # one cycle up front: the steady-state launch is the one captured
        pf.predict(io[0][0], 0.1)
        pf.update(io[0][0], io[0][1])
        pf.resample(r=0.5)
    torch.cuda.synchronize()
    step0 = graphed._step
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            two_cycles(graphed)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    for rep in range(2):
        g.replay()
        eager._step = step0                          # (the replay carries the Philox steps of the capture)
        two_cycles(eager)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(host(graphed._x), host(eager._x))
        np.testing.assert_array_equal(host(graphed._w), host(eager._w))
    np.testing.assert_array_equal(graphed.point_estimate(), eager.point_estimate())


# ------------------------------------------------------------------------------ the reference's weight range
def _filter_and_reference_dtype_oracle(n, seed):
    gpu_se_amd.seed(seed)
    state_pdf, measurement_pdf = sim_base.get_noise(deterministic=True)
    x0, _ = sim_base.get_noise()
    xss = orc.steady_state()
    x0.means += xss[None, :].astype(np.float32)
    pf = ParallelParticleFilter(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, n, x0, state_pdf, measurement_pdf)
    osp, omp = orc.bioreactor_noise()
    # weights_dtype=None: the REFERENCE's dtypes -- fp32 until the first resample, fp64 from then on (particle.py:50, :103)
    o = orc.ParticleFilterOracle(n, None, osp, omp, particles=pf.particles.get(), weights_dtype=None)
    return pf, o, xss


@pytest.mark.parametrize("n,lazy", [(1 << 12, True), (1 << 16, True), (1 << 16, False)])
def test_weight_range_of_the_reference_after_a_measurement_jump(n, lazy):
    """The reference re-creates its weights as fp64 after every resample (filter/particle.py:103, :316), so ``weights *= pdf``
    (:294) survives likelihoods down to 1e-308.  A jump of the measured glucose by 350 mg/L puts every particle's likelihood
    near 1e-59: as plain fp32 products the new weights are ALL zero, their sum is zero, and the resample degenerates to
    copies of particle 0 (what this build did until round 4; checked below with weight_scale = False).  With the
    launch-uniform power-of-two scale (GSE_OP_WEIGHT_SCALE) the stored weights times 2^k are the reference's weights to one
    fp32 rounding, the un-normalised point estimate between update and resample is the reference's, and the ancestors are
    those of the oracle run in the reference's dtypes.  Both ways of launching: the fused cycle and one kernel per method.

    Ancestors: the scale restores the reference's RANGE, not its mantissa -- a stored weight is the reference's fp64 weight
    rounded to 24 bits, so a cumulative weight is off by about 2^-25 / sqrt(N_eff) of the total (N_eff = (sum w)^2 / sum w^2
    heavy particles, each rounded independently), and with N thresholds among N keys about N^2 2^-25 / sqrt(N_eff) of them
    fall between a key and its rounded twin: none at N = 4096 (asserted: at most 1), a handful at 2^16 right after the jump,
    when a few hundred particles carry the posterior (asserted: that estimate x 4, + 2, and every one of them moves its
    offspring to the NEIGHBOURING particle).  Exact equality would need fp64 weight storage (SURVEY 7 recommends fp32)."""
    pf, o, xss = _filter_and_reference_dtype_oracle(n, 77)
    pf.lazy = lazy
    plain, _, _ = _filter_and_reference_dtype_oracle(n, 77)          # the same filter without the scale
    plain.weight_scale = False
    plain.particles = pf.particles.get()
    rng = np.random.default_rng(21)
    sd = np.sqrt([1e-4, 1e-7, 1e-3, 1e-3, 1e-7])
    u = [0.06, 0.2]
    total_flips = 0
    for cycle, jump in enumerate([0.0, 350.0, 350.0, 340.0]):
        noise = (rng.standard_normal((n, 5)) * sd).astype(np.float32)
        z = np.array(orc.bioreactor_g(xss)) + [0.2 + jump, -0.1]
        r = float(rng.random())
        for f in (pf, plain):
            DeterministicGaussianSum.preload(noise)
            f.predict(u, 0.1)
            f.update(u, z)
        o.predict_vec(u, 0.1, noise=noise)
        o.update_vec(u, z)
        w, k = pf._scaled_weights()
        w_plain = plain._w[0, :n].cpu().numpy()
        if cycle == 1:
            assert k[0] < -150 and o.weights.max() < 1e-50, (k, o.weights.max())
            assert w_plain.sum() == 0.0                                 # what plain fp32 weights make of this update
            # reading the attribute now hands out the TRUE weights in fp64 and leaves the filter alone (round 5; round 4 folded
            # the exponent into the fp32 numbers: all zero, and the resample below went to particle 0)
            got = pf.weights.get()
            assert got.dtype == np.float64 and 0 < got.max() < 1e-45
            np.testing.assert_array_equal(got, np.ldexp(w[0].astype(np.float64), int(k[0])))
            assert pf.weight_log2_scale[0] == k[0] and np.array_equal(pf._scaled_weights()[0], w)
        elif cycle == 0:
            assert k[0] == 0                                            # likelihoods fp32 holds: the launch is bit for bit
            np.testing.assert_array_equal(w[0], w_plain)                # what it is without the scale
        heavy = o.weights > o.weights.max() * 1e-25                     # (beyond fp32's span below the heaviest: denormal / 0)
        np.testing.assert_allclose(np.ldexp(w[0].astype(np.float64), int(k[0]))[heavy], o.weights[heavy], rtol=2.5e-7)
        assert np.ldexp(w[0].astype(np.float64), int(k[0]))[~heavy].max(initial=0.0) <= o.weights.max() * 2e-25
        # point_estimate() between update and resample is the reference's un-normalised weights @ particles (:318-320)
        np.testing.assert_allclose(pf.point_estimate(), o.point_estimate(), rtol=1e-6)
        idx = pf.resample(r=r, return_index=True).get().astype(np.int64)
        # (before its first resample the reference's weights are still fp32 and so is its cumulative sum, particle.py:50, :89;
        #  the device sums in fp64 always -- DESIGN.md section 2 -- so cycle 0 is compared on fp64 sums of the same weights;
        #  from cycle 1 on that IS the reference's arithmetic)
        own = orc.systematic_index(o.weights.astype(np.float64), r)
        if cycle > 0:
            np.testing.assert_array_equal(own, o.ancestors(r))
        flips = int(np.sum(own != idx))
        wo = o.weights.astype(np.float64)
        n_eff = wo.sum() ** 2 / np.sum(wo * wo)
        allowed = 2 + 4.0 * n * n * 2.0 ** -25 / np.sqrt(n_eff) if n > 4096 else 1
        assert flips <= allowed and np.max(np.abs(own - idx)) <= 1, (cycle, flips, allowed, n_eff)
        total_flips += flips
        o.resample(r=r, idx=idx)
        np.testing.assert_array_equal(pf.particles.get(), o.particles)
        assert pf.weight_log2_scale[0] == 0                            # offspring weigh 1 / N: the exponent is spent
        plain.resample(r=r)
        if cycle == 1:                                                  # (zero weights: every offspring copies particle 0)
            assert np.all(plain.particles.get() == plain.particles.get()[0])
            plain.particles = pf.particles.get()                        # (carry on from the same ensemble)
    assert total_flips <= (2 if n <= 4096 else 40)


@pytest.mark.parametrize("n,batch", [(1 << 17, None), (1 << 16, 2), (3 * (1 << 16), None)])
def test_carried_scale_samples_give_what_the_fetched_ones_give(n, batch):
    """GSE_OP_SAMPLES (round 4): a launch whose update fills a scan slot leaves the weight scale's 64 sample particles beside
    it and the next launch takes them from there instead of fetching them from x_in / w_in.  Speed only: with and without,
    the particles are the same bits and so are the true weights (stored weight x 2^scale), through a jump of the measurement
    that engages the scale; the three methods and run_cycles (whose second call inherits the samples of the first through the
    spec); touching the particles in between withdraws the word."""
    from gpu_se_amd import _lib
    from gpu_se_amd.filter import BatchedParticleFilter
    fs = []
    for _ in range(3):
        gpu_se_amd.seed(91)
        state_pdf, measurement_pdf = sim_base.get_noise()
        x0, _ = sim_base.get_noise()
        x0.means += orc.steady_state()[None, :].astype(np.float32)
        if batch is None:
            f = ParallelParticleFilter(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, n, x0, state_pdf, measurement_pdf)
        else:
            f = BatchedParticleFilter(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, n, x0, state_pdf, measurement_pdf,
                                      batch=batch)
        f.cluster = False                                   # (the tile kernels are the ones that carry samples)
        fs.append(f)
    carried, fetched, looped = fs
    seen = {id(f): [] for f in fs}
    for f in (carried, fetched):
        def spy(ops, *args, _f=f, _inner=f._fn_step):
            seen[id(_f)].append(int(ops))
            return _inner(ops, *args)
        f._fn_step = spy
    rng = np.random.default_rng(8)
    xss = orc.steady_state()
    K, B = 10, batch or 1
    us = np.stack([rng.uniform(0.03, 0.09, (K, B)), rng.uniform(0.1, 0.3, (K, B))], axis=-1)
    zs = np.array(orc.bioreactor_g(xss))[None, None, :] + rng.normal(0, 0.3, (K, B, 2))
    zs[6:, :, 0] -= 500.0                                   # the measured glucose jumps: likelihoods around 1e-100
    rs = rng.random((K, B))
    shape = (lambda v: v[0]) if batch is None else (lambda v: v)
    true_weights = lambda f: np.ldexp(f._w[:, :n].cpu().numpy().astype(np.float64), f._wscale.cpu().numpy().astype(int)[:, None])
    engaged = False
    for k in range(K):
        for f in (carried, fetched):
            f.predict(shape(us[k]), 0.1)
            if f is fetched:
                f._samples_ok = False
            f.update(shape(us[k]), shape(zs[k]))
        engaged = engaged or bool(np.any(carried._wscale.cpu().numpy() < -100))
        np.testing.assert_array_equal(carried._x.cpu().numpy(), fetched._x.cpu().numpy())
        np.testing.assert_array_equal(true_weights(carried), true_weights(fetched))
        assert np.all(np.abs(carried._wscale.cpu().numpy() - fetched._wscale.cpu().numpy()) <= 1)
        for f in (carried, fetched):
            f.resample(r=shape(rs[k]))
    assert engaged
    with_word = [bool(o & _lib.GSE_OP_SAMPLES) for o in seen[id(carried)] if o & _lib.GSE_OP_SCAN_IN]
    assert len(with_word) == K - 1 and all(with_word), seen[id(carried)]
    assert not any(o & _lib.GSE_OP_SAMPLES for o in seen[id(fetched)])
    # run_cycles: two calls, the second starting from the first's pending resample and its samples
    for part in (slice(0, 4), slice(4, K)):
        looped.run_cycles(us[part] if batch else us[part, 0], zs[part] if batch else zs[part, 0], 0.1,
                          rs=rs[part] if batch else rs[part, 0])
        assert looped._samples_ok
    for f in (carried, looped):
        f._flush()
    np.testing.assert_array_equal(looped._x.cpu().numpy(), carried._x.cpu().numpy())
    np.testing.assert_array_equal(looped._w.cpu().numpy(), carried._w.cpu().numpy())
    # the accessors withdraw the word (a caller may write through the view they hand out)
    carried.predict(shape(us[0]), 0.1)
    carried.update(shape(us[0]), shape(zs[0]))
    assert carried._samples_ok
    carried.particles
    assert not carried._samples_ok


@pytest.mark.parametrize("n,cluster,lazy", [(1 << 13, True, True), (1 << 13, False, True), (1 << 13, False, False),
                                            (1 << 16, False, True)])
def test_scale_of_an_update_without_a_resample_looks_at_every_particle(n, cluster, lazy):
    """An update that is NOT preceded by a resample finds weights that may span any number of binades, and the heaviest
    particle need not stand anywhere near one of the 64 sample particles the scale's anchor looks at after a resample: with
    all but one particle at the bottom of fp32 (2^-149) and the one heavy particle between two samples, a k taken from the
    samples would scale the heavy one past the end of fp32 (inf; round 4 until this test).  Such launches take the anchor
    from a pass over ALL particles (scale_exact_kernel): the new weights are the reference's fp64 products to one fp32
    rounding -- with likelihoods fp32 can hold (k = 0) and through a measurement jump (k far below)."""
    pf, o, xss = _filter_and_reference_dtype_oracle(n, 83)
    pf.cluster, pf.lazy = cluster, lazy
    rng = np.random.default_rng(4)
    sd = np.sqrt([1e-4, 1e-7, 1e-3, 1e-3, 1e-7])
    u = [0.06, 0.2]
    heavy_at = 5 + 3 * (n // 64)                           # (between the samples, which stand at multiples of n / 64)
    x = pf.particles.get().copy()
    for jump in (0.0, 350.0):
        w0 = np.full(n, np.float32(2.0 ** -149))
        w0[heavy_at] = 1.0
        pf.weights = w0
        noise = (rng.standard_normal((n, 5)) * sd).astype(np.float32)
        z = np.array(orc.bioreactor_g(xss)) + [0.2 + jump, -0.1]
        DeterministicGaussianSum.preload(noise)
        pf.predict(u, 0.1)
        pf.update(u, z)
        w, k = pf._scaled_weights()
        stored = w[0].astype(np.float64)
        assert np.all(np.isfinite(stored)), (jump, int(k[0]))
        # the reference's arithmetic on the same particles: fp64 products of the fp32 weights with the fp64 likelihoods
        o.particles = x.copy()
        o.predict_vec(u, 0.1, noise=noise)
        e = z[None, :] - orc.bioreactor_g_vec(o.particles).astype(np.float64)
        want = w0.astype(np.float64) * o.measurement_pdf.pdf(e)
        got = np.ldexp(stored, int(k[0]))
        np.testing.assert_allclose(got[heavy_at], want[heavy_at], rtol=2.5e-7)
        assert (k[0] == 0) == (jump == 0.0) and (jump == 0.0 or k[0] < -100), (jump, k)
        light = np.arange(n) != heavy_at
        assert got[light].max() <= want[heavy_at] * 2.0 ** -120           # (the light ones: denormal or zero, as they should be)
        x = pf.particles.get().copy()
        np.testing.assert_array_equal(x, o.particles)


def test_weight_scale_travels_with_the_accessors():
    """``weights`` hands out the TRUE weights (fp64 while an exponent is engaged, the fp32 view otherwise) without touching the
    filter, assigning weights clears the exponent, and the island bookkeeping of a shard (log of the TRUE weight sum) includes
    it."""
    n = 8192
    pf, o, xss = _filter_and_reference_dtype_oracle(n, 78)
    pf._island = torch.zeros((2, 1), dtype=torch.float64, device="cuda")
    u = [0.06, 0.2]
    z = np.array(orc.bioreactor_g(xss)) + [200.0, -0.1]
    noise = np.zeros((n, 5), np.float32)
    DeterministicGaussianSum.preload(noise)
    pf.predict(u, 0.1)
    pf.update(u, z)
    o.predict_vec(u, 0.1, noise=noise)
    o.update_vec(u, z)
    w, k = pf._scaled_weights()
    assert k[0] < -40
    true_sum = float(np.sum(np.ldexp(w[0].astype(np.float64), int(k[0]))))
    np.testing.assert_allclose(true_sum, o.weights.astype(np.float64).sum(), rtol=1e-6)
    pf.resample(r=0.25)
    pf._flush()
    isl = pf._island.cpu().numpy()
    np.testing.assert_allclose(isl[0, 0], np.log(true_sum), rtol=1e-12)
    np.testing.assert_allclose(isl[1, 0], true_sum, rtol=1e-12)
    # a second update, then the accessor (round 5): with an exponent engaged the true weights are below fp32's range, so an
    # fp64 COPY of them is handed out -- what the reference holds after its first resample (filter/particle.py:103, :316) -- and
    # the filter's state is not touched: the exponent stays, and the resample that follows finds the ancestors the stored
    # weights call for (round 4 folded the exponent into the fp32 numbers here: every weight flushed to zero, every ancestor 0)
    DeterministicGaussianSum.preload(noise)
    pf.predict(u, 0.1)
    pf.update(u, z)
    k_before = pf.weight_log2_scale[0]
    assert k_before < -40
    w_stored, _ = pf._scaled_weights()
    got = pf.weights.get()
    assert got.dtype == np.float64 and pf.weight_log2_scale[0] == k_before
    np.testing.assert_array_equal(got, np.ldexp(w_stored[0].astype(np.float64), int(k_before)))
    assert 0 < got.max() < 1e-20
    assert np.array_equal(pf._scaled_weights()[0], w_stored)
    idx = pf.resample(r=0.6, return_index=True).get()
    o.weights = w_stored[0].astype(np.float64)                      # (ancestors are invariant under the power of two)
    np.testing.assert_array_equal(idx, o.resample(r=0.6, loop=False))
    assert len(np.unique(idx)) > 10
    # assigning plain numbers clears the exponent
    DeterministicGaussianSum.preload(noise)
    pf.predict(u, 0.1)
    pf.update(u, z)
    assert pf.weight_log2_scale[0] < -40
    pf.weights = np.full(n, 1.0 / n, dtype=np.float32)
    assert pf.weight_log2_scale[0] == 0
    # ... and while no exponent is engaged the attribute is the writable fp32 view it always was
    v = pf.weights
    assert v.get().dtype == np.float32
    v[:] = 2.0 / n
    assert pf.weights.get()[5] == np.float32(2.0 / n)


# ------------------------------------------------------------------------------ four workgroups per tile (mid-size ensembles)
def _pair_of_filters(n, seed, batch=None):
    out = []
    for cluster in (True, False):
        gpu_se_amd.seed(seed)
        state_pdf, measurement_pdf = sim_base.get_noise()
        x0, _ = sim_base.get_noise()
        x0.means += orc.steady_state()[None, :].astype(np.float32)
        if batch is None:
            f = ParallelParticleFilter(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, n, x0, state_pdf, measurement_pdf)
        else:
            from gpu_se_amd.filter import BatchedParticleFilter
            f = BatchedParticleFilter(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, n, x0, state_pdf, measurement_pdf,
                                      batch=batch)
        assert f.cluster                                        # (the policy takes these sizes)
        f.cluster = cluster
        out.append(f)
    return out


@pytest.mark.parametrize("n,lazy", [(5003, True), (1 << 14, True), ((1 << 16) + 17, True), (1 << 18, True), (1 << 15, False)])
def test_cluster_cycle_equals_the_tile_cycle(n, lazy):
    """GSE_OP_CLUSTER (gse_pf_cluster_kernels.h: four workgroups per tile, one particle per thread, scan slots in quarter
    format) against the one-workgroup-per-tile kernels on the same filter: Philox noise, 12 cycles from the prior (the first
    ones have the wide ancestor ranges that take the look-up path), a measurement jump in between (the weight scale) --
    particles, weights, ancestors and estimates bit for bit after every cycle, fused launches and one launch per method.  The
    policy switch between the two kernel families changes the time of a cycle, not its result."""
    a, b_ = _pair_of_filters(n, 61)
    rng = np.random.default_rng(5)
    xss = orc.steady_state()
    for f in (a, b_):
        f.lazy = lazy
    for c in range(12):
        u = [float(rng.uniform(0.03, 0.09)), float(rng.uniform(0.1, 0.3))]
        # (the inputs wander and every third step is a whole minute: the cloud drifts 50 .. 150 mg/L from the measurements, wide
        #  ancestor ranges in most cycles; at cycle 7 the measurement is another 500 mg/L away: likelihoods near 2^-600)
        z = np.array(orc.bioreactor_g(xss)) + rng.normal(0, 0.3, 2) + ([-500.0, 0.0] if c == 7 else [0.0, 0.0])
        r = float(rng.random())
        idx = []
        for f in (a, b_):
            f.predict(u, 0.1 if c % 3 else 1.0)
            f.update(u, z)
            if c in (2, 7):                                     # between update and resample: stored weights and their exponent
                idx.append(f._scaled_weights())
                continue
            idx.append(f.resample(r=r, return_index=True).get())
        if c in (2, 7):
            np.testing.assert_array_equal(idx[0][0], idx[1][0])
            np.testing.assert_array_equal(idx[0][1], idx[1][1])
            if c == 7:
                assert idx[0][1][0] < -400 and idx[0][0].max() > 1e-30, (idx[0][1], idx[0][0].max())
            for f in (a, b_):
                f.resample(r=r)
        else:
            np.testing.assert_array_equal(idx[0], idx[1], err_msg="ancestors, cycle %d" % c)
        if c % 4 == 3:
            np.testing.assert_array_equal(a.particles.get(), b_.particles.get(), err_msg="particles, cycle %d" % c)
            np.testing.assert_array_equal(a.point_estimate(), b_.point_estimate())
    assert a._slot_fmt.count("q") >= 1 and "q" not in b_._slot_fmt
    np.testing.assert_array_equal(a.particles.get(), b_.particles.get())
    np.testing.assert_array_equal(a.weights.get(), b_.weights.get())


def test_cluster_cycle_on_adversarial_weights_and_in_batches():
    """The same on weight patterns built to hurt a systematic resample (assigned through the attribute: the resample that
    follows forms its cumulative weights itself, the update after it leaves a quarter-format slot again), and with several
    filters per launch."""
    n = 1 << 15
    a, b_ = _pair_of_filters(n, 62)
    rng = np.random.default_rng(6)
    xss = orc.steady_state()
    z = np.array(orc.bioreactor_g(xss)) + [0.2, -0.1]
    for name, w in adversarial_weights(n, rng).items():
        if name in ("zero_sum", "nan"):
            continue
        r = float(rng.random())
        got = []
        for f in (a, b_):
            f.weights = (w / w.sum()).astype(np.float32)
            f.resample(r=r)
            f.predict([0.06, 0.2], 0.1)
            f.update([0.06, 0.2], z)
            got.append(f.resample(r=0.5 * r, return_index=True).get())
        np.testing.assert_array_equal(got[0], got[1], err_msg=name)
        np.testing.assert_array_equal(a.particles.get(), b_.particles.get(), err_msg=name)
    B, n = 3, 40000
    a, b_ = _pair_of_filters(n, 63, batch=B)
    u = np.tile([0.06, 0.2], (B, 1))
    zb = np.tile(z, (B, 1)) + np.arange(B)[:, None] * 0.3
    for c in range(5):
        rr = rng.random(B)
        got = []
        for f in (a, b_):
            f.predict(u, 0.1)
            f.update(u, zb)
            got.append(f.resample(r=rr, return_index=True).get())
        np.testing.assert_array_equal(got[0], got[1])
    np.testing.assert_array_equal(a.particles.get(), b_.particles.get())
    np.testing.assert_array_equal(a.point_estimate(), b_.point_estimate())


@pytest.mark.parametrize("n,batch,jump", [(1 << 14, None, False), (70001, None, False), (1 << 20, None, False), (9000, 3, False),
                                          (1 << 14, None, True), (1 << 20, None, True), (9000, 3, True)])
def test_run_cycles_equals_the_three_methods(n, batch, jump):
    """ParallelParticleFilter.run_cycles (gse_pf_cycles: K cycles enqueued from C++) against K rounds of predict / update /
    resample from Python: the same launches, so particles, weights and estimates bit for bit -- mid-size filters (cluster
    launches), 2^20 (the tile kernels), a batch; and the filter carries on identically afterwards."""
    fs = []
    for _ in range(2):
        gpu_se_amd.seed(44)
        state_pdf, measurement_pdf = sim_base.get_noise()
        x0, _ = sim_base.get_noise()
        x0.means += orc.steady_state()[None, :].astype(np.float32)
        if batch is None:
            fs.append(ParallelParticleFilter(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, n, x0, state_pdf, measurement_pdf))
        else:
            from gpu_se_amd.filter import BatchedParticleFilter
            fs.append(BatchedParticleFilter(Bioreactor.homeostatic_DEs, Bioreactor.static_outputs, n, x0, state_pdf,
                                            measurement_pdf, batch=batch))
    a, b_ = fs
    rng = np.random.default_rng(3)
    xss = orc.steady_state()
    K, B = 9, batch or 1
    us = np.stack([rng.uniform(0.03, 0.09, (K, B)), rng.uniform(0.1, 0.3, (K, B))], axis=-1)
    zs = np.array(orc.bioreactor_g(xss))[None, None, :] + rng.normal(0, 0.3, (K, B, 2))
    if jump:
        # a measurement jump on the FIRST cycle (round 5): that launch updates without a copy in front of it, and
        # the C++ loop now vouches for the uniform weights as the class does (GSE_OP_UNIFORM_WEIGHTS: the scale's anchor from the
        # 64 samples instead of a pass over all particles) -- the same binary exponent, the same stored weights
        zs[0, :, 0] += 300.0
    shape = (lambda v: v[0]) if batch is None else (lambda v: v)
    for part in (slice(0, 4), slice(4, 9)):                 # (the second call starts from a pending resample)
        a.run_cycles(us[part] if batch else us[part, 0], zs[part] if batch else zs[part, 0], 0.1)
        for k in range(part.start, part.stop):
            b_.predict(shape(us[k]), 0.1)
            b_.update(shape(us[k]), shape(zs[k]))
            b_.resample()
        if jump and part.start == 0:
            assert np.array_equal(a.weight_log2_scale, b_.weight_log2_scale)
    np.testing.assert_array_equal(a.particles.get(), b_.particles.get())
    np.testing.assert_array_equal(a.weights.get(), b_.weights.get())
    for f in (a, b_):
        f.predict(shape(us[0]), 0.1)
        f.update(shape(us[0]), shape(zs[0]))
    if jump:
        assert np.all(a.weight_log2_scale < -100)
    np.testing.assert_array_equal(a._scaled_weights()[0], b_._scaled_weights()[0])
    np.testing.assert_array_equal(a.weight_log2_scale, b_.weight_log2_scale)
    np.testing.assert_array_equal(a.point_estimate(), b_.point_estimate())
