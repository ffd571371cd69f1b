"""Loop-level parity of the closed-loop drivers, -m gpu: gpu_se_amd.sim_base.Simulation,
BatchedSimulation and BatchedSimulation(resident=True), either estimator, against oracle/closed_loop.py -- the restatement of
the reference's sim_base.py:243-309 that tests/golden/closed_loop.npz pins to the reference's own loop -- tick by tick, with
every random number injected (sim_base.InjectedNoise <-> closed_loop.LoopNoise), and against the recorded reference runs
themselves."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
if not torch.cuda.is_available():
    pytest.skip("no HIP device", allow_module_level=True)

import gpu_se_amd                                            # noqa: E402
from gpu_se_amd import sim_base                              # noqa: E402
from oracle import closed_loop                               # noqa: E402
from oracle import gse_oracle as orc                         # noqa: E402
from gpu_util import host                                    # noqa: E402
import loop_util                                             # noqa: E402


def schedule(end_time, dt_control, dt_predict):
    """(ticks, predicts, updates) of sim_base.py:243-277 for these periods."""
    ts = np.linspace(0, end_time, int(end_time * 10))
    t_c = t_p = 0
    n_p = n_u = 0
    for t in ts[1:]:
        if t > t_p:
            n_p += 1
            t_p += dt_predict
        if t > t_c:
            n_u += 1
            t_c += dt_control
    return len(ts), n_p, n_u


def draw_noise(seed, ticks, n_predict, n_update, B, n, pf):
    """Random inputs with the distributions the loop draws them from (host generator; fp32 like the mixtures' draws)."""
    rng = np.random.default_rng(seed)
    sp, mp = orc.bioreactor_noise(rng)
    shape = (B, n) if pf else (B, n, 11)
    return sim_base.InjectedNoise(
        plant_state=sp.draw((ticks - 1, B)), plant_meas=mp.draw((ticks - 1, B)),
        predict=[sp.draw(shape) for _ in range(n_predict)], r=[rng.random(B) for _ in range(n_update)])


def oracle_rollout(noise, b, est0, pf, dt_control, dt_predict, end_time, x_plant, ancestors=None):
    sp, mp = orc.bioreactor_noise()
    if pf:
        n = est0.shape[0]
        est = orc.ParticleFilterOracle(n, None, sp, mp, particles=est0, weights_dtype="scaled32")
    else:
        means, covs = est0
        est = orc.GSUKFOracle(means.shape[0], None, sp, mp, means=means, covariances=covs)
    ln = closed_loop.LoopNoise(noise.plant_state[:, b], noise.plant_meas[:, b], [p[b] for p in noise.predict],
                               [r[b] for r in noise.r], ancestors=ancestors)
    return loop_util.loop_oracle(est, dt_control, dt_predict, end_time, x_plant, ln).simulate()


def run_driver(driver, B, n, dt_control, dt_predict, end_time, pf, noise):
    kw = dict(N_particles=n, dt_control=dt_control, dt_predict=dt_predict, end_time=end_time, pf=pf, noise=noise)
    if driver == "host":
        assert B == 1
        sim = sim_base.Simulation(**kw)
    else:
        sim = sim_base.BatchedSimulation(batch=B, resident=(driver == "resident"), **kw)
    sim.f.record_ancestors = True
    if pf:
        est0 = host(sim.f.particles).reshape(B, n, 5).copy()
    else:
        est0 = (host(sim.f.means).reshape(B, n, 5).copy(), host(sim.f.covariances).reshape(B, n, 5, 5).copy())
    sim.simulate()
    sim.f._flush()
    anc = [host(a)[:, :n].astype(np.int64) for a in sim.f.ancestor_log]
    return sim, est0, anc


def records(sim, b, B, driver):
    """The records of rollout b in the shapes of the single-rollout class."""
    if driver == "host":
        return {k: np.asarray(getattr(sim, k)) for k in ("us", "xs", "ys", "ys_meas", "xs_f", "ys_f", "covariance_point_size")}
    return {k: np.asarray(getattr(sim, k))[:, b] for k in ("us", "xs", "ys", "ys_meas", "xs_f", "ys_f", "covariance_point_size")}


PF_CASES = [
    # driver,   B, N,    dt_control, dt_predict, end_time
    ("host", 1, 1024, 0.1, 0.1, 3),
    ("host", 1, 1024, 1.0, 0.1, 4),
    ("batched", 3, 1024, 0.1, 0.1, 3),
    ("batched", 3, 2048, 1.0, 0.1, 4),
    ("resident", 3, 1024, 0.1, 0.1, 3),
    ("resident", 3, 2048, 1.0, 0.1, 4),
    ("resident", 2, 5003, 1.0, 1.0, 4),        # predict only when the controller acts; a ragged last tile
]


@pytest.mark.parametrize("driver,B,n,dt_control,dt_predict,end_time", PF_CASES)
def test_particle_filter_loops_tick_by_tick_following_the_devices_ancestors(driver, B, n, dt_control, dt_predict, end_time):
    """Over the whole run, no state re-seeded: the drivers' records against the oracle loop fed the same random inputs.
    Ancestors: the contract on the device's likelihoods is one fp32 ulp of the oracle's (device exp against NumPy's; in the
    runs examined they were bit-equal), and with N thresholds among N cumulative weights a shift of 1e-7 would move a
    threshold across a key once in a few resamples (N^2 x 1e-7): the oracle computes ITS ancestors from ITS weights, they
    are counted against the device's (at most 2 entries per resample, the standard of test_pf_trace_per_step; measured: 0
    everywhere), and the device's are then copied so that one such tie could not send the rest of a run down another path.
    With that: inputs, plant states and outputs to 1e-9 (MPC law against the oracle's sparse KKT solve, fp64 plant; measured
    1e-12); estimates to 1e-11 (measured: bit-equal), covariance sizes to 1e-7; the solved fraction equal.  The resident loop runs the single-launch cycle (gather + predict + update
    with the recorded noise) and takes its estimates from gse_pf_moments_resampled; the other two run one stage group per
    launch."""
    gpu_se_amd.seed(100 + n + int(10 * dt_control))
    ticks, n_p, n_u = schedule(end_time, dt_control, dt_predict)
    assert ticks >= 25 or dt_control == 1.0
    noise = draw_noise(7 * n + B, ticks, n_p, n_u, B, n, pf=True)
    sim, est0, anc = run_driver(driver, B, n, dt_control, dt_predict, end_time, True, noise)
    assert len(anc) == n_u
    x_plant = np.asarray(sim.xs)[0] if driver == "host" else np.asarray(sim.xs)[0, 0]
    for b in range(B):
        o = oracle_rollout(noise, b, est0[b], True, dt_control, dt_predict, end_time, x_plant, ancestors=[a[b] for a in anc])
        got = records(sim, b, B, driver)
        assert max(o.ancestor_flips) <= 2 and sum(o.ancestor_flips) <= max(2, n_u // 2), o.ancestor_flips
        # (an input on its limit u = 0 comes back within the dual sweeps' stopping tolerance of it: 1e-12 relative of the row)
        np.testing.assert_allclose(got["us"], o.us, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(got["xs"], o.xs, rtol=1e-9, atol=1e-11)
        np.testing.assert_allclose(got["ys"], o.ys, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(got["ys_meas"], o.ys_meas, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(got["xs_f"], o.xs_f, rtol=1e-11, atol=1e-13)
        np.testing.assert_allclose(got["ys_f"], o.ys_f, rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(got["covariance_point_size"], o.covariance_point_size, rtol=1e-7, atol=1e-16)
        if driver == "host":
            np.testing.assert_allclose(np.array(sim.biass).reshape(-1, 2), np.array(o.biass).reshape(-1, 2), rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(sim.performance, o.performance, rtol=1e-9)
            assert (sim.predict_count, sim.update_count) == (o.predict_count, o.update_count)
        else:
            np.testing.assert_allclose(sim.performance[b], o.performance, rtol=1e-9)
        assert sim.mpc_frac == o.mpc_frac == 1.0


GSF_CASES = [
    ("host", 1, 16, 0.1, 0.1, 3),
    ("host", 1, 64, 1.0, 0.1, 4),
    ("batched", 3, 64, 0.1, 0.1, 3),
    ("resident", 3, 64, 0.1, 0.1, 3),
    ("resident", 3, 64, 1.0, 0.1, 4),
    ("resident", 2, 200, 0.1, 0.1, 3),         # large banks: the lane kernels, the deferred row copy
    ("batched", 2, 200, 1.0, 0.1, 4),
]


@pytest.mark.parametrize("driver,B,n,dt_control,dt_predict,end_time", GSF_CASES)
def test_gaussian_sum_filter_loops_tick_by_tick_following_the_devices_ancestors(driver, B, n, dt_control, dt_predict, end_time):
    """The same with the GS-UKF as the estimator.  Its covariances are fp32 sums of eleven outer products (the device keeps
    another order than NumPy: 2e-5 per step, tests/test_gpu_gsukf.py), so weights agree to 1e-4 and a Gaussian whose
    cumulative weight sits that close to a threshold may go either way: the oracle copies the device's ancestors and counts
    its own against them (at most 1 in 200 may differ per resample); everything else is free-running."""
    gpu_se_amd.seed(300 + n)
    ticks, n_p, n_u = schedule(end_time, dt_control, dt_predict)
    noise = draw_noise(11 * n + B, ticks, n_p, n_u, B, n, pf=False)
    sim, est0, anc = run_driver(driver, B, n, dt_control, dt_predict, end_time, False, noise)
    assert len(anc) == n_u
    x_plant = np.asarray(sim.xs)[0] if driver == "host" else np.asarray(sim.xs)[0, 0]
    for b in range(B):
        o = oracle_rollout(noise, b, (est0[0][b], est0[1][b]), False, dt_control, dt_predict, end_time, x_plant,
                           ancestors=[a[b] for a in anc])
        got = records(sim, b, B, driver)
        assert sum(o.ancestor_flips) <= max(1, n_u * n // 200), o.ancestor_flips
        np.testing.assert_allclose(got["us"], o.us, rtol=2e-4, atol=2e-5)
        np.testing.assert_allclose(got["xs"], o.xs, rtol=1e-4, atol=1e-6)
        np.testing.assert_allclose(got["xs_f"], o.xs_f, rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(got["ys_f"], o.ys_f, rtol=1e-4)
        np.testing.assert_allclose(got["covariance_point_size"], o.covariance_point_size, rtol=5e-3)
        assert sim.mpc_frac == o.mpc_frac == 1.0


# ---------------------------------------------------------------------------------------------------------------------
# The recorded runs of the reference's own loop (tests/golden/closed_loop.npz, G10), all six, on every driver.
# ---------------------------------------------------------------------------------------------------------------------
RECORDED = ["pf_c1", "pf_c1p1", "pf_c01", "pf_fail", "gsf_c1", "gsf_c01"]


@pytest.mark.parametrize("name", RECORDED)
@pytest.mark.parametrize("driver", ["host", "batched", "resident"])
def test_drivers_replay_the_recorded_reference_runs(golden, name, driver):
    """tests/golden/closed_loop.npz holds six runs of the reference's own Simulation.simulate on its NumPy classes: initial
    ensemble, every random input, every record, the ancestors of every resample.  The drivers -- Simulation,
    BatchedSimulation, BatchedSimulation(resident=True) -- started from the same ensemble and fed the same inputs must
    reproduce the records AT THE TOLERANCES THE ORACLE LOOP IS HELD TO (tests/test_oracle_golden.py:
    test_closed_loop_oracle_against_the_reference_loop): inputs 1e-6, plant 2e-6, estimates 2e-6, covariance sizes 1e-5,
    performance 1e-5, the solved fraction equal -- over the whole run, nothing re-seeded.

    Every resample runs on the device in full (InjectedNoise.ancestors + r): the filter forms its cumulative weights and
    its OWN ancestors from ITS weights, they are counted against the recorded ones (ancestor_flips; the oracle's standard:
    at most 2 per resample, 8 per run -- the reference's mixed-precision weights resolve eight ties of pf_c01 differently),
    and then the recorded ones are copied so that one tie cannot fork the rest of a run.

      pf_c1p1  the filter predicts a tenth of the elapsed time (the reference passes dt, not dt_predict, sim_base.py:251),
               loses the plant, and its likelihoods fall to 1e-60 and below: fp32 weights flush to zero there; the
               reference's fp64 weights (particle.py:103, :316) do not.  Held by the power-of-two weight scale
               (GSE_OP_WEIGHT_SCALE): the ancestors are the recorded ones.
      pf_c01   control every tick: the P = 2999 horizon (16 000 variables in the reference's QP).
      pf_fail  the reference's except branch on every control tick (sim_base.py:271-273): the device law reports "not
               solved" (gse_mpc_law.report_unsolved), u = [0.06, 0.2] + u_bar, mpc_frac = 0.
    GS-UKF: fp32 covariances summed in another order than NumPy's (2e-5 per step), so estimates to 2e-4 there."""
    g = golden("closed_loop.npz")
    n, dtc, dtp, end, is_pf = g[name + "_cfg"]
    n, end, is_pf = int(n), int(end), bool(is_pf)
    gpu_se_amd.seed(5)
    ticks = len(g[name + "_us"])
    noise = sim_base.InjectedNoise(g[name + "_plant_state"][:, None], g[name + "_plant_meas"][:, None],
                                   [p[None] for p in g[name + "_predict_noise"]], [np.array([r]) for r in g[name + "_r"]],
                                   ancestors=[a[None] for a in g[name + "_ancestors"]])
    kw = dict(N_particles=n, dt_control=float(dtc), dt_predict=float(dtp), end_time=end, pf=is_pf, noise=noise)
    if driver == "host":
        sim = sim_base.Simulation(**kw)
    else:
        sim = sim_base.BatchedSimulation(batch=1, resident=(driver == "resident"), **kw)
    if name == "pf_fail":
        sim.K.report_unsolved = True
    lead = (n,) if driver == "host" else (1, n)
    if is_pf:
        sim.f.particles = g[name + "_particles0"].reshape(lead + (5,))
    else:
        sim.f.means = g[name + "_means0"].reshape(lead + (5,))
        sim.f.covariances = g[name + "_covs0"].reshape(lead + (5, 5))
    sim.simulate()
    pick = (lambda a: np.asarray(a)) if driver == "host" else (lambda a: np.asarray(a)[:, 0])
    flips = [int(f.sum().item()) for f in sim.f.ancestor_flips]
    assert len(flips) == len(g[name + "_r"]) == int(g[name + "_counts"][1])
    if is_pf:
        assert max(flips) <= 2 and sum(flips) <= 8, flips
        tol_u, tol_x, tol_e = dict(rtol=1e-6, atol=1e-6), dict(rtol=2e-6, atol=1e-9), dict(rtol=2e-6, atol=2e-6)
        tol_yf, tol_c, tol_p = dict(rtol=2e-6), dict(rtol=1e-5), 1e-5
    else:
        assert sum(flips) <= max(1, len(flips) * n // 200), flips
        tol_u, tol_x, tol_e = dict(rtol=2e-4, atol=2e-5), dict(rtol=1e-4, atol=1e-6), dict(rtol=1e-4, atol=2e-4)
        tol_yf, tol_c, tol_p = dict(rtol=2e-4), dict(rtol=5e-3), 2e-3
    assert pick(sim.us).shape == (ticks, 2)
    np.testing.assert_allclose(pick(sim.us), g[name + "_us"], **tol_u)
    np.testing.assert_allclose(pick(sim.xs), g[name + "_xs"], **tol_x)
    np.testing.assert_allclose(pick(sim.ys), g[name + "_ys"], **tol_x)
    np.testing.assert_allclose(pick(sim.ys_meas), g[name + "_ys_meas"], **tol_x)
    # (the records before the first tick belong to the ensemble the constructor drew, replaced above)
    np.testing.assert_allclose(pick(sim.xs_f)[1:], g[name + "_xs_f"][1:], **tol_e)
    np.testing.assert_allclose(pick(sim.ys_f)[1:], g[name + "_ys_f"][1:], **tol_yf)
    np.testing.assert_allclose(pick(sim.covariance_point_size)[1:], g[name + "_covariance_point_size"][1:], **tol_c)
    # the integral performance includes the first sample (t = 0, weight 0 in the time-weighted integrand)
    perf = sim.performance if driver == "host" else sim.performance[0]
    np.testing.assert_allclose(perf, g[name + "_performance"], rtol=tol_p)
    assert sim.mpc_frac == g[name + "_mpc_frac"]
    if name == "pf_fail":
        assert sim.mpc_frac == 0.0
        np.testing.assert_allclose(pick(sim.us)[11], np.array([0.06, 0.2]) + sim.lin_model.u_bar, rtol=0, atol=1e-15)
    if driver == "host":
        np.testing.assert_allclose(np.array(sim.biass).reshape(-1, 2), g[name + "_biass"], rtol=1e-6, atol=1e-4)
        assert (sim.predict_count, sim.update_count) == tuple(g[name + "_counts"])
