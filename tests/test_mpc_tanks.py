"""The reference's own MPC tests (tests/mpc_tests/tank_simulation_test.py, diagtank_simulation_test.py,
linkedtank_simulation_test.py), restated against gpu_se_amd.controller.MPC: the same plants (tests/tank_plants.py), the
same linearisation, horizons (P = 20, M = 8), weights (Q = 10 I, R = 0), set points and 80-second closed loops, and
EXACTLY the assertions the reference makes (bias of the output prediction, steady-state error below 1e-3).  The
reference solves its QP with OSQP, which is not installed here; these assertions are what it holds the controller to,
so they pin the solution side of the MPC.

Two routes: the host checker of the condensed law (oracle/mpc_oracle.hildreth; CPU), and the device law through
gse_mpc_step (MPC.step; -m gpu)."""
import numpy as np
import pytest

from gpu_se_amd import controller
from gpu_se_amd.model import LinearModel
from oracle import mpc_oracle as orc
from tank_plants import DiagTanks, LinkedTanks, Tank

END_TIME, DT_CONTROL = 80, 1
TS = np.linspace(0, END_TIME, END_TIME * 100)


class CheckerStep:
    """MPC.step (controller.py:236-279 of the reference) with the host checker in place of the device kernel."""

    def __init__(self, K):
        self.K, self.y_predicted = K, None

    def step(self, x0, um1, y0):
        x0, um1, y0 = (np.atleast_1d(np.asarray(v, dtype=float)) for v in (x0, um1, y0))
        bias = y0 - self.y_predicted if self.y_predicted is not None else np.zeros_like(y0)
        du0, y1, _, ok = orc.hildreth(self.K._law, np.concatenate([x0, um1, bias]))
        assert ok
        self.y_predicted = y1 - bias
        return du0 + um1


class DeviceStep:
    def __init__(self, K):
        self.K = K

    @property
    def y_predicted(self):
        return self.K.y_predicted

    def step(self, x0, um1, y0):
        return self.K.step(np.atleast_1d(x0), np.atleast_1d(um1), np.atleast_1d(y0))


def closed_loop(plant, x_bar, u_bar, r, route):
    """The loop of the reference's tests (tank_simulation_test.py:74-108): plant Euler steps of 0.01 s, a controller
    call once a second, the bias log taken just before each call."""
    dt = TS[1]
    lin = LinearModel.create_LinearModel(plant, x_bar=np.array(x_bar, dtype=float), u_bar=np.array(u_bar, dtype=float),
                                         T=DT_CONTROL)
    n = len(np.atleast_1d(r))
    K = controller.MPC(P=20, M=8, Q=np.diag([10.0] * n), R=np.diag([0.1] * n) * 0, lin_model=lin, ysp=lin.yn2d(np.array(r, dtype=float)))
    ctl = route(K)
    x_op, u_op, y_op = lin.x_bar, lin.u_bar, lin.y_bar
    us, ys, biass = [np.atleast_1d(lin.u_bar).astype(float)], [np.atleast_1d(lin.y_bar).astype(float)], []
    t_next = 0
    for t in TS[1:]:
        plant.step(dt, us[-1])
        ys.append(np.atleast_1d(plant.outputs(us[-1])).astype(float))
        if t > t_next:
            if ctl.y_predicted is not None:
                biass.append(ys[-1] - y_op - ctl.y_predicted)
            u = ctl.step(plant.X - x_op, us[-1] - u_op, ys[-1] - y_op)
            us.append(np.atleast_1d(u + u_op))
            t_next += DT_CONTROL
        else:
            us.append(us[-1])
    return np.array(ys), np.array(us), np.array(biass)


def check_single_tank(route):
    r = np.array([100.0])
    ys_lin, _, bias_lin = closed_loop(Tank([50.0], linear=True), [50.0], [10.0], r, route)
    ys_non, _, bias_non = closed_loop(Tank([50.0], linear=False), [50.0], [10.0], r, route)
    assert np.max(bias_lin) < 0.1                                            # test_linear_tank_bias (:115-116)
    a = bias_non[100:]                                                       # test_nonlinear_tank_bias (:119-121) -- the log has
    assert a - (np.average(a) if len(a) else 0) == pytest.approx(0)          #   79 entries, so the reference's slice is empty
    assert ys_lin[500:] - r == pytest.approx(0, abs=1e-3)                    # test_linear_tank_SS (:124-126)
    assert ys_non[500:] - r == pytest.approx(0, abs=1e-3)                    # test_nonlinear_tank_SS (:129-131)
    # what the reference's empty slice was after: the plant-model mismatch of the nonlinear tank settles to a constant
    tail = bias_non[40:]
    assert np.max(np.abs(tail - tail.mean(axis=0))) < 1e-6 and np.max(np.abs(bias_lin[40:])) < 1e-6


def check_diag_tanks(route):
    r = np.array([100.0, 30.0])
    ys, _, biass = closed_loop(DiagTanks([50.0, 60.0]), [50.0, 60.0], [10.0, 12.0], r, route)
    assert np.max(biass[:, 0]) == pytest.approx(0, abs=1e-3)                 # test_diag_tank_bias (diagtank_simulation_test.py:75-78)
    a = biass[:, 1][100:]
    assert a - (np.average(a) if len(a) else 0) == pytest.approx(0)
    assert ys[500:] - r == pytest.approx(0, abs=1e-3)                        # test_diag_tank_SS (:81-83)


def check_linked_tanks(route):
    r = np.array([100.0, 30.0])
    ys, _, biass = closed_loop(LinkedTanks([50.0, 60.0]), [50.0, 60.0], [10.0, 12.0], r, route)
    a = biass[100:]                                                          # test_linked_tank_bias (linkedtank_simulation_test.py:75-77)
    assert a - (np.average(a, axis=0) if len(a) else 0) == pytest.approx(0)
    assert ys[1000:] - r == pytest.approx(0, abs=1e-3)                       # test_linked_tank_SS (:80-82)
    tail = biass[40:]
    assert np.max(np.abs(tail - tail.mean(axis=0))) < 1e-5                   # (the intent of the empty slice)


@pytest.mark.parametrize("check", [check_single_tank, check_diag_tanks, check_linked_tanks])
def test_reference_mpc_assertions_host_checker(check):
    check(CheckerStep)


@pytest.mark.gpu
@pytest.mark.parametrize("check", [check_single_tank, check_diag_tanks, check_linked_tanks])
def test_reference_mpc_assertions_device_law(check):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    check(DeviceStep)
