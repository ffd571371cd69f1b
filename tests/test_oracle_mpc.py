"""MPC, CPU side: the oracle's QP against the reference's own matrices, the product's condensation
against the same matrices, and the explicit law against an independent QP solver.  No GPU needed."""
import os

import numpy as np
import pytest
import scipy.sparse

from gpu_se_amd import controller, model
from oracle import mpc_oracle as orc

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "mpc_qp.npz"))
LIN = np.load(os.path.join(os.path.dirname(__file__), "golden", "linear_model.npz"))
Q, R = np.diag([0.1, 1]), np.diag([1, 1])
U_BAR = LIN["dt1_sub_u_bar"]
UB_DEFAULT = [np.array([0, np.inf]) - U_BAR[0], np.array([0, np.inf]) - U_BAR[1]]
CASES = {
    "small": dict(P=6, M=3, u_bounds=UB_DEFAULT),
    "square": dict(P=5, M=5, u_bounds=UB_DEFAULT),
    "free": dict(P=8, M=4),
    "boxed": dict(P=8, M=4, u_bounds=[(-0.03, 0.05), (-0.08, 0.1)], u_step_bounds=[(-0.01, 0.01), (-0.02, 0.02)],
                  y_bounds=[(-3.0, 3.0), (-400.0, 400.0)]),
    "parts": dict(P=300, M=200, u_bounds=UB_DEFAULT),
}


def lin_model():
    return model.LinearModel(LIN["dt1_sub_A"], LIN["dt1_sub_B"], LIN["dt1_sub_C"], LIN["dt1_sub_D"], 1.0,
                             LIN["dt1_sub_x_bar"], LIN["dt1_sub_u_bar"], LIN["dt1_sub_f_bar"], LIN["dt1_sub_y_bar"])


def golden_dense(name, tag):
    shape = tuple(GOLD[name + "_shape"]) if tag == "A" else (len(GOLD[name + "_q"]),) * 2
    return scipy.sparse.coo_matrix((GOLD["%s_%s_val" % (name, tag)], (GOLD["%s_%s_row" % (name, tag)],
                                                                     GOLD["%s_%s_col" % (name, tag)])), shape=shape).toarray()


def oracle_qp(name):
    lm = lin_model()
    return orc.assemble_qp(Q=Q, R=R, A=lm.A, B=lm.B, C=lm.C, D=lm.D, ysp=LIN["dt1_ysp"], **CASES[name])


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_qp_equals_the_reference_matrices(name):
    H, q, A, l, u = oracle_qp(name)
    np.testing.assert_allclose(H, golden_dense(name, "H"), rtol=0, atol=1e-15)
    np.testing.assert_allclose(q, GOLD[name + "_q"], rtol=1e-14, atol=0)
    np.testing.assert_allclose(A, golden_dense(name, "A"), rtol=1e-14, atol=1e-16)
    np.testing.assert_array_equal(l, GOLD[name + "_l0"])
    np.testing.assert_array_equal(u, GOLD[name + "_u0"])
    P = CASES[name]["P"]
    x0, um1, y0, ypred = (GOLD[name + s] for s in ("_x0", "_um1", "_y0", "_ypred"))
    for k, bias in enumerate((np.zeros(2), y0 - ypred)):               # controller.py:248-256
        ls, us = orc.write_step(l, u, P, 2, 2, 2, x0, um1, bias)
        np.testing.assert_allclose(ls, GOLD["%s_step%d_l" % (name, k)], rtol=1e-15, atol=0)
        np.testing.assert_allclose(us, GOLD["%s_step%d_u" % (name, k)], rtol=1e-15, atol=0)


def full_vector(K, d, theta):
    """z of the reference's QP for moves d and parameters theta."""
    ys, dmus = K.response(d[:, None], theta[:, None])
    return np.concatenate([theta[:K.Nx], dmus[:, 0], ys[:, 0], theta[K.Nx:K.Nx + K.Ni], d])


@pytest.mark.parametrize("name", sorted(CASES))
def test_condensed_problem_is_the_reference_qp(name):
    """Any (d, theta) mapped to the full variable vector satisfies the reference's equality rows, and the condensed
    cost differs from the reference's objective by a constant in d."""
    K = controller.MPC(Q=Q, R=R, lin_model=lin_model(), ysp=LIN["dt1_ysp"], **CASES[name])
    H, q, A = golden_dense(name, "H"), GOLD[name + "_q"], golden_dense(name, "A")
    rng = np.random.default_rng(3)
    P, Nx, Ni, No = K.P, K.Nx, K.Ni, K.No
    n_eq = Ni + (P + 1) * Nx + P * No
    theta = rng.normal(size=Nx + Ni + No) * np.array([0.05, 0.5, 0.01, 0.01, 5.0, 50.0])
    l, u = orc.write_step(GOLD[name + "_l0"], GOLD[name + "_u0"], P, Nx, Ni, No, theta[:Nx], theta[Nx:Nx + Ni], theta[Nx + Ni:])
    costs_full, costs_cond = [], []
    c = K._condensed
    for _ in range(3):
        d = rng.normal(size=(K.M + 1) * Ni) * 0.02
        z = full_vector(K, d, theta)
        np.testing.assert_allclose((A @ z)[:n_eq], l[:n_eq], rtol=0, atol=1e-9 * (1 + np.abs(z).max()))
        costs_full.append(0.5 * z @ H @ z + q @ z)
        costs_cond.append(0.5 * d @ c["Hc"] @ d + (c["F"] @ theta + c["f0"]) @ d)
        # the bounded rows of the law are the finite inequality rows of the reference, in its order
        ineq = (A @ z)[n_eq:]
        finite = np.isfinite(l[n_eq:]) | np.isfinite(u[n_eq:])
        np.testing.assert_allclose(c["Cin"] @ d + c["Pin"] @ theta, ineq[finite], rtol=1e-12, atol=1e-12)
        np.testing.assert_array_equal(K._law["lo"], l[n_eq:][finite])
        np.testing.assert_array_equal(K._law["hi"], u[n_eq:][finite])
    diffs_full, diffs_cond = np.diff(costs_full), np.diff(costs_cond)
    np.testing.assert_allclose(diffs_cond, diffs_full, rtol=1e-9, atol=1e-9)


def thetas_for(name, rng, count):
    """Parameter vectors (x0, u_-1, bias): ordinary ones, and ones far enough from the operating point that the
    unconstrained first input would be negative (the limit u >= 0 of sim_base.py:82-85 becomes active)."""
    scale = np.array([0.05, 0.5, 0.01, 0.01, 5.0, 50.0])
    out = rng.normal(size=(count, 6)) * scale
    out[::2, 0] += 2.5                      # a large glucose excess: input 0 wants to go below zero
    out[::4, 1] -= 5.0                      # ... and a fatty-acid deficit: input 1 too
    if name == "boxed":
        out[1::2] *= 0.2                    # (the even ones violate this case's output limits outright: infeasible)
    return out


@pytest.mark.parametrize("name", ["small", "square", "free", "boxed", "parts"])
def test_explicit_law_reaches_the_qp_optimum(name):
    """du_0, y_1 of the condensed law + Hildreth sweeps (NumPy checker of the device kernel) against an independent
    full-space solve of the reference's QP."""
    K = controller.MPC(Q=Q, R=R, lin_model=lin_model(), ysp=LIN["dt1_ysp"], **CASES[name])
    H, q, A = golden_dense(name, "H"), GOLD[name + "_q"], golden_dense(name, "A")
    lay = orc.layout(K.P, K.M, K.Nx, K.Ni, K.No)
    rng = np.random.default_rng(11)
    active_seen = infeasible_seen = 0
    for theta in thetas_for(name, rng, 2 if name == "parts" else 6):
        l, u = orc.write_step(GOLD[name + "_l0"], GOLD[name + "_u0"], K.P, K.Nx, K.Ni, K.No, theta[:2], theta[2:4], theta[4:])
        du0, y1, nu, ok = orc.hildreth(K._law, theta)
        try:
            z, objective, info = orc.solve_qp(H, q, A, l, u)
        except ValueError:                      # contradictory limits: the sweeps must not claim a solution
            assert not ok
            infeasible_seen += 1
            continue
        assert ok
        active_seen += int(np.any(nu != 0))
        np.testing.assert_allclose(du0, z[lay["du"]:lay["du"] + K.Ni], rtol=1e-6, atol=1e-8)
        np.testing.assert_allclose(y1, z[lay["y"]:lay["y"] + K.No], rtol=1e-6, atol=1e-6)
        # optimality of the law's point in the reference's own terms: feasible and not worse than the solver's optimum
        d = K.moves(theta[None, :], nu[None, :])[0]
        zf = full_vector(K, d, theta)
        Az = A @ zf
        tol = 1e-8 * (1 + np.abs(zf).max())
        assert np.all(Az >= l - tol) and np.all(Az <= u + tol)
        mine = 0.5 * zf @ H @ zf + q @ zf
        assert mine <= objective + 1e-8 * (1 + abs(objective))
    if name != "free":
        assert active_seen > 0          # the bounds were exercised
    assert (infeasible_seen > 0) == (name == "boxed")


@pytest.mark.parametrize("name", ["small", "square", "free", "parts"])
def test_kkt_active_set_solver_equals_the_dense_exact_solver(name):
    """oracle.mpc_oracle.KKTSolver (sparse KKT systems over enumerated active sets: what the closed-loop oracle's MPC and the
    recording of tests/golden/closed_loop.npz solve the reference's QP with) against solve_qp (null space + NNLS), bounds
    inactive and active.  Two exact methods that share no step."""
    sparse = orc.assemble_qp(Q=Q, R=R, A=lin_model().A, B=lin_model().B, C=lin_model().C, D=lin_model().D, ysp=LIN["dt1_ysp"],
                             sparse=True, **CASES[name])
    H, q, A, l0, u0 = oracle_qp(name)
    np.testing.assert_array_equal(sparse[0].toarray(), H)
    np.testing.assert_array_equal(sparse[2].toarray(), A)
    solver = orc.KKTSolver(*sparse)
    P, M = CASES[name]["P"], CASES[name]["M"]
    rng = np.random.default_rng(5)
    active = 0
    for theta in thetas_for(name, rng, 2 if name == "parts" else 8):
        l, u = orc.write_step(l0, u0, P, 2, 2, 2, theta[:2], theta[2:4], theta[4:])
        z_dense, objective, info = orc.solve_qp(H, q, A, l, u)
        z = solver.solve(l, u)
        active += info["active"]
        assert 0.5 * z @ H @ z + q @ z <= objective + 1e-9 * (1 + abs(objective))
        lay = orc.layout(P, M, 2, 2, 2)
        np.testing.assert_allclose(z[lay["du"]:lay["du"] + 2], z_dense[lay["du"]:lay["du"] + 2], rtol=1e-6, atol=1e-9)
        # (y_1 ~ 285: the dense route's SVD of the 1204 x 1606 equality block of "parts" is good to 1e-8 relative of that;
        #  the tight comparison of this solver is the next test, 1e-9 against the product's condensation)
        np.testing.assert_allclose(z[lay["y"]:lay["y"] + 2], z_dense[lay["y"]:lay["y"] + 2], rtol=1e-7, atol=1e-5)
    assert (active > 0) == (name != "free")


def test_mpc_oracle_step_is_the_explicit_law_at_the_long_horizon():
    """get_parts(dt_control = 0.1): P = 2999, M = 1999 -- 16 000 variables, out of reach of the dense solvers.  MPCOracle
    (sparse QP of the reference + KKTSolver) against the product's condensation + the NumPy checker of its device sweeps,
    over consecutive steps (bias bookkeeping included), bounds inactive and active."""
    lin = {k: LIN["dt01_sub_" + k] for k in ("A", "B", "C", "D", "x_bar", "u_bar", "f_bar", "y_bar")}
    lm = model.LinearModel(lin["A"], lin["B"], lin["C"], lin["D"], 0.1, lin["x_bar"], lin["u_bar"], lin["f_bar"], lin["y_bar"])
    ub = [np.array([0, np.inf]) - lin["u_bar"][0], np.array([0, np.inf]) - lin["u_bar"][1]]
    K = controller.MPC(P=2999, M=1999, Q=Q, R=R, lin_model=lm, ysp=LIN["dt01_ysp"], u_bounds=ub)
    O = orc.MPCOracle(2999, 1999, Q, R, lin["A"], lin["B"], lin["C"], lin["D"], LIN["dt01_ysp"], u_bounds=ub)
    rng = np.random.default_rng(8)
    y_pred, active = None, 0
    for k, theta in enumerate(thetas_for("parts", rng, 6)):
        x0, um1, y0 = theta[:2] * 0.2, theta[2:4], theta[4:]
        bias = y0 - y_pred if y_pred is not None else np.zeros(2)
        du0, y1, nu, ok = orc.hildreth(K._law, np.concatenate([x0, um1, bias]))
        assert ok
        active += int(np.any(nu != 0))
        ctrl = O.step(x0, um1, y0)
        # (round 3: the law's matrices come from sparse KKT solves in the reference's variable space -- the condensed
        #  Hessian of this horizon has a condition number of 4.5e14, through it du_0 was good to 1e-6 only once a bias had
        #  accumulated)
        np.testing.assert_allclose(du0 + um1, ctrl, rtol=1e-12, atol=1e-13)
        np.testing.assert_allclose(y1 - bias, O.y_predicted, rtol=1e-12, atol=1e-11)
        y_pred = y1 - bias
    assert active > 0
