"""Test plants for the MPC: the tank models the reference's own controller tests drive (tests/mpc_tests/TankModel.py:5-64,
DiagTank.py:6-61, LinkedTanks.py:5-68), restated -- same physics, same Euler step, same interface (X, DEs, step, outputs)
as `model.LinearModel.create_LinearModel` expects.  Test infrastructure only."""
import numpy


class _Plant:
    def __init__(self, X0, t0=0.0):
        self.X = numpy.array(X0, dtype=float)
        self.t = t0

    def step(self, dt, inputs):
        self.t += dt
        self.X = self.X + self.DEs(inputs) * dt
        return self.outputs(inputs)

    def outputs(self, inputs):
        return numpy.array(self.X, dtype=float).copy()


class Tank(_Plant):
    """One tank of cross-section A = 2 draining through k = 0.1: dh = (F_in - k h A) / A (linear) or
    (F_in - k sqrt(h A)) / A (TankModel.py:29-34).  Output: the level."""

    def __init__(self, X0, linear=False):
        super().__init__(X0)
        self.linear = linear

    def DEs(self, inputs):
        h, = self.X
        f_in, = numpy.atleast_1d(inputs)
        k, area = 0.1, 2.0
        drain = k * h * area if self.linear else k * numpy.sqrt(h * area)
        return numpy.array([(f_in - drain) / area])

    def outputs(self, inputs):
        return numpy.array(self.X[0])


class DiagTanks(_Plant):
    """Two independent tanks; the reference evaluates BOTH with the linear law (DiagTank.py:29-30 calls
    ``self.linear_tank.DEs`` twice) on the level of its first sub-tank object, which never moves from X0[0] = 50: the
    derivative of either state depends on its input only, dh_i = (u_i - 0.1 * 50 * 2) / 2."""

    def DEs(self, inputs):
        u = numpy.atleast_1d(inputs)
        return numpy.array([(u[0] - 0.1 * 50.0 * 2.0) / 2.0, (u[1] - 0.1 * 50.0 * 2.0) / 2.0])


class LinkedTanks(_Plant):
    """Two tanks joined by a pipe (LinkedTanks.py:24-38): F_12 = 0.05 (h1 - h2); dh1 = (F1 - 0.1 sqrt(2 h1) + F_12) / 2 (the
    nonlinear variant the reference's test builds; the sign of F_12 is the reference's), dh2 = (F2 - 0.3 * 8 h2) / 8."""

    def DEs(self, inputs):
        h1, h2 = self.X
        f1, f2 = inputs
        k1, k2, k_link = 0.1, 0.3, 0.05
        a1, a2 = 2.0, 8.0
        f12 = k_link * (h1 - h2)
        return numpy.array([(f1 - k1 * numpy.sqrt(h1 * a1) + f12) / a1, (f2 - k2 * h2 * a2) / a2])
