"""MPC on the device: gse_mpc_step through the C ABI against the NumPy checker of the same law and
against an independent exact solve of the reference's QP; the controller inside the closed loop."""
import os

import numpy as np
import pytest
import torch

import gpu_se_amd
from gpu_se_amd import controller, sim_base
from oracle import mpc_oracle as orc
from test_oracle_mpc import CASES, GOLD, LIN, Q, R, full_vector, golden_dense, lin_model, thetas_for

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["small", "square", "free", "boxed", "parts"])
def test_device_law_equals_checker_and_qp_optimum(name):
    K = controller.MPC(Q=Q, R=R, lin_model=lin_model(), ysp=LIN["dt1_ysp"], max_sweeps=2000, **CASES[name])
    rng = np.random.default_rng(21)
    thetas = thetas_for(name, rng, 24)
    du0, y1, status = K.step_device(torch.as_tensor(thetas, device="cuda"))
    du0, y1, status = du0.cpu().numpy(), y1.cpu().numpy(), status.cpu().numpy()
    H, q, A = golden_dense(name, "H"), GOLD[name + "_q"], golden_dense(name, "A")
    lay = orc.layout(K.P, K.M, K.Nx, K.Ni, K.No)
    solved = 0
    for b, theta in enumerate(thetas):
        ref_du0, ref_y1, nu, ok = orc.hildreth(K._law, theta, max_sweeps=2000)
        assert (status[b] == 0) == ok
        if not ok:
            continue
        solved += 1
        np.testing.assert_allclose(du0[b], ref_du0, rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(y1[b], ref_y1, rtol=1e-9, atol=1e-9)
        if b < (2 if name == "parts" else 6):          # and the optimum of the reference's QP itself
            l, u = orc.write_step(GOLD[name + "_l0"], GOLD[name + "_u0"], K.P, K.Nx, K.Ni, K.No, theta[:2], theta[2:4], theta[4:])
            z, _, _ = orc.solve_qp(H, q, A, l, u)
            np.testing.assert_allclose(du0[b], z[lay["du"]:lay["du"] + K.Ni], rtol=1e-6, atol=1e-8)
            np.testing.assert_allclose(y1[b], z[lay["y"]:lay["y"] + K.No], rtol=1e-6, atol=1e-5)
    assert solved >= (8 if name == "boxed" else 24)


def test_step_follows_the_reference_contract():
    """step(): deviation variables in, first input out, y_predicted advanced by the first predicted output minus the
    bias (controller.py:264-279); a QP without solution raises ValueError and leaves y_predicted alone."""
    K = controller.MPC(Q=Q, R=R, lin_model=lin_model(), ysp=LIN["dt1_ysp"], **CASES["parts"])
    x0, um1, y0 = np.array([0.01, -0.2]), np.array([0.02, 0.1]), np.array([1.8, -23.0])
    assert K.y_predicted is None
    u1 = K.step(x0, um1, y0)
    du0, y1, nu, ok = orc.hildreth(K._law, np.concatenate([x0, um1, np.zeros(2)]))
    np.testing.assert_allclose(u1, du0 + um1, rtol=1e-9)
    np.testing.assert_allclose(K.y_predicted, y1, rtol=1e-9)
    first = K.y_predicted.copy()
    u2 = K.step(x0, u1, y0 + 0.5)
    bias = y0 + 0.5 - first
    du0, y1, nu, ok = orc.hildreth(K._law, np.concatenate([x0, u1, bias]))
    np.testing.assert_allclose(u2, du0 + u1, rtol=1e-9)
    np.testing.assert_allclose(K.y_predicted, y1 - bias, rtol=1e-9)
    boxed = controller.MPC(Q=Q, R=R, lin_model=lin_model(), ysp=LIN["dt1_ysp"], max_sweeps=200, **CASES["boxed"])
    with pytest.raises(ValueError):
        boxed.step(np.array([2.5, 0.0]), um1 * 0, y0)
    assert boxed.y_predicted is None
    with pytest.raises(NotImplementedError):            # 600 bounded output rows: more than the device solver takes
        controller.MPC(Q=Q, R=R, lin_model=lin_model(), ysp=LIN["dt1_ysp"], P=300, M=200, y_bounds=[(-3, 3), (-400, 400)])._law


def test_closed_loop_with_mpc_drives_the_outputs_to_the_setpoint():
    """get_parts wiring (sim_base.py:75-86): the fatty-acid set point is 850 mg/L, the plant starts at 611 mg/L."""
    gpu_se_amd.seed(5)
    sim = sim_base.Simulation(N_particles=1 << 13, dt_control=1, dt_predict=0.1, end_time=40, seed=2)
    sim.simulate()
    assert sim.mpc_frac == 1.0 and len(sim.biass) == sim.update_count - 1
    assert np.all(sim.us >= -1e-12) and np.any(np.abs(sim.us) < 1e-9)      # the input limit of the controller, reached
    cfa = sim.ys[:, 2]
    assert cfa[0] < 620 and cfa[-1] > cfa[0] + 60                # moving towards 850 mg/L
    assert abs(sim.ys[-1, 0] - 280) < 25                         # glucose held near its set point
    assert np.isfinite(sim.performance)


def test_batched_closed_loops_with_mpc():
    gpu_se_amd.seed(6)
    sim = sim_base.BatchedSimulation(batch=3, N_particles=1 << 12, dt_control=1, dt_predict=0.1, end_time=12, seed=4)
    sim.simulate()
    assert sim.mpc_frac == 1.0 and sim.us.shape == (len(sim.ts), 3, 2)
    assert np.all(sim.us >= -1e-12) and len(np.unique(np.round(sim.us[-1, :, 0], 9))) == 3
    assert np.all(sim.ys[-1, :, 2] > sim.ys[0, :, 2])
