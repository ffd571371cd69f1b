"""Discrete linear state-space model obtained by linearising a nonlinear plant (reference model/LinearModel.py).

    x_{k+1} = A (x_k - x_bar) + B (u_k - u_bar) + x_bar        y_k = C (x_k - x_bar) + D (u_k - u_bar) + y_bar

Host-side set-up code (runs once per controller); the MPC built on it is gpu_se_amd.controller.MPC.
Pinned by tests/golden/linear_model.npz (the reference's own matrices for the get_parts wiring).
"""
import numpy
import scipy.signal


def _central_gradient(fun, tol=1e-8, h=0.1):
    """d fun / d s at s = 0 by central differences, halving the step until two successive estimates agree to
    ``tol`` in the max norm (the procedure -- start step, halving, stopping rule -- of LinearModel.py:79-95, so the
    finite-difference noise matches the reference's to the last digits)."""
    estimate = (fun(h) - fun(-h)) / 2 / h
    change = tol + 1
    while change > tol:
        h /= 2
        refined = (fun(h) - fun(-h)) / 2 / h
        change = numpy.max(numpy.abs(refined - estimate))
        estimate = refined
    return estimate


class LinearModel:
    """Matrices A, B, C, D of the discrete model with sampling interval ``T``, its linearisation point (x_bar, u_bar)
    and the plant's derivative / output there (f_bar, y_bar).  ``states`` / ``inputs`` / ``outputs`` index the
    variables kept by :meth:`select_subset`."""

    def __init__(self, A, B, C, D, dt, x_bar, u_bar, f_bar, y_bar):
        self.A, self.B, self.C, self.D = (numpy.atleast_2d(m) for m in (A, B, C, D))
        self.T = dt
        self.x_bar, self.u_bar, self.f_bar, self.y_bar = x_bar, u_bar, f_bar, y_bar
        self.Nx, self.Ni, self.No = self.A.shape[0], self.B.shape[1], self.C.shape[0]
        self.states, self.inputs, self.outputs = range(self.Nx), range(self.Ni), range(self.No)

    @staticmethod
    def create_LinearModel(nonlinear_model, x_bar, u_bar, T):
        """Linearise ``nonlinear_model`` (an object with ``X``, ``DEs(u)`` and ``outputs(u)``) about (x_bar, u_bar)
        and discretise with a zero-order hold of length ``T`` (LinearModel.py:57-124)."""
        x_bar = numpy.asarray(x_bar, dtype=float)
        u_bar = numpy.asarray(u_bar, dtype=float)
        keep = nonlinear_model.X

        def jacobian(fun, wrt_state):
            point = x_bar if wrt_state else u_bar
            columns = []
            for k in range(len(point)):
                def along(step):
                    moved = point.copy()
                    moved[k] += step
                    if wrt_state:
                        nonlinear_model.X = moved
                        value = fun(u_bar)
                        nonlinear_model.X = x_bar
                        return value
                    return fun(moved)
                nonlinear_model.X = x_bar
                columns.append(_central_gradient(along))
            return numpy.array(columns).T

        A = jacobian(nonlinear_model.DEs, True)
        B = jacobian(nonlinear_model.DEs, False)
        C = jacobian(nonlinear_model.outputs, True)
        D = jacobian(nonlinear_model.outputs, False)
        Ad, Bd, Cd, Dd, _ = scipy.signal.cont2discrete((A, B, C, D), T)
        nonlinear_model.X = x_bar
        f_bar = nonlinear_model.DEs(u_bar)
        y_bar = nonlinear_model.outputs(u_bar)
        nonlinear_model.X = keep
        return LinearModel(Ad, Bd, Cd, Dd, T, x_bar, u_bar, f_bar, y_bar)

    def select_subset(self, states, inputs, outputs):
        """Keep only the listed states, inputs and outputs (LinearModel.py:126-152)."""
        self.states, self.inputs, self.outputs = states, inputs, outputs
        self.A = self.A[states][:, states]
        self.B = self.B[states][:, inputs]
        self.C = self.C[outputs][:, states]
        self.D = self.D[outputs][:, inputs]
        self.x_bar, self.f_bar = self.x_bar[states], self.f_bar[states]
        self.u_bar = self.u_bar[inputs]
        self.y_bar = self.y_bar[outputs]
        self.Nx, self.Ni, self.No = len(states), len(inputs), len(outputs)

    # deviation <-> natural variables (LinearModel.py:154-274); ``subselect`` picks the kept entries of a full vector
    def xd2n(self, x_hat):
        return x_hat + self.x_bar

    def xn2d(self, x, subselect=True):
        return (x[..., self.states] if subselect else x) - self.x_bar

    def yd2n(self, y_hat):
        return y_hat + self.y_bar

    def yn2d(self, y, subselect=True):
        return (y[..., self.outputs] if subselect else y) - self.y_bar

    def ud2n(self, u_hat):
        return u_hat + self.u_bar

    def un2d(self, u, subselect=True):
        return (u[..., self.inputs] if subselect else u) - self.u_bar
