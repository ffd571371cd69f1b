"""Host-side bioreactor model: the plant object sim_base hands out and the two static functions that
identify the built-in device model.

Only what the estimator path needs from the reference's model/BioreactorModel.py is mirrored: the
low-nitrogen ("homeostatic") regime used by sim_base.get_parts (high_N=False, sim_base.py:46-53), the
state, Euler stepping, outputs, and the steady-state used as the filters' prior mean.  The per-particle
evaluation of these functions happens in the HIP kernels (gpu_se_amd/csrc/gse_device.h); the Python bodies
below serve the scalar plant simulation and act as the tokens ParallelParticleFilter dispatches on.
"""
import numpy

_MOLAR_MASS = numpy.array([180, 24.6, 116, 46, 1])


class Bioreactor:
    """Fed-batch fumaric-acid bioreactor, production phase (model/BioreactorModel.py:6-43).

    Parameters
    ----------
    X0 : array (5,)  initial state [Cg, Cx, Cfa, Ce, Ch]
    t : float        initial time
    high_N : bool    only the low-nitrogen regime (False) is available here
    """

    def __init__(self, X0, t=0, high_N=False):
        if high_N:
            raise NotImplementedError("only the low-nitrogen regime (high_N=False) that the estimators "
                                      "run on is provided (sim_base.py:46-53)")
        self.X = numpy.array(X0, dtype=numpy.float64)
        self.t = t
        self.high_N = high_N

    # ---- the f and g handed to the filters (model/BioreactorModel.py:170-253) ----
    @staticmethod
    def homeostatic_DEs(x, u, dt=1):
        """Euler increment (dCg, dCx, dCfa, dCe, dCh) of the production-phase ODE over dt."""
        cg, cx, cfa, ce, ch = (float(v) for v in x)
        cg, cx, cfa, ce = max(cg, 0.0), max(cx, 0.0), max(cfa, 0.0), max(ce, 0.0)
        fg, fm = float(u[0]), float(u[1])
        outflow = fg + fm
        sat = cg / (1e-2 + cg)
        r_h = 280 / 180 - cg
        r_fa = 0.25 / 116 * cx * 24.6 * sat
        cap1 = (0.4 - 0.25) / 180 * cx * 24.6
        want1 = cap1 - (cap1 / 2000 / (0.28 / 180) * r_h + 0.01 * ch)
        th1 = min(cap1, max(0.0, want1)) * sat
        r_e = min(0.025 / 46 * cx * 24.6, max(0.0, want1 - cap1))
        th2 = min((0.1 - 0.025) / 180 * cx * 24.6, max(0.0, want1 - cap1 - r_e))
        r_g = -r_fa * (116 / 180) - th1 - r_e * (46 / 180) - th2
        return ((fg * (5000 / 180) - outflow * cg + r_g) * dt,
                0.0 * cx * dt,
                (-outflow * cfa + r_fa) * dt,
                (-outflow * ce + r_e) * dt,
                r_h * dt)

    @staticmethod
    def homeostatic_DEs_rows(X, U, dt=1):
        """homeostatic_DEs for B plants at once: X (B, 5), U (B, 2) -> increments (B, 5); NumPy, same expressions
        (the host side of BatchedSimulation)."""
        X = numpy.asarray(X, dtype=numpy.float64)
        U = numpy.asarray(U, dtype=numpy.float64)
        cg, cx, cfa, ce = (numpy.maximum(X[:, k], 0.0) for k in range(4))
        ch = X[:, 4]
        fg, fm = U[:, 0], U[:, 1]
        outflow = fg + fm
        sat = cg / (1e-2 + cg)
        r_h = 280 / 180 - cg
        r_fa = 0.25 / 116 * cx * 24.6 * sat
        cap1 = (0.4 - 0.25) / 180 * cx * 24.6
        want1 = cap1 - (cap1 / 2000 / (0.28 / 180) * r_h + 0.01 * ch)
        th1 = numpy.minimum(cap1, numpy.maximum(0.0, want1)) * sat
        r_e = numpy.minimum(0.025 / 46 * cx * 24.6, numpy.maximum(0.0, want1 - cap1))
        th2 = numpy.minimum((0.1 - 0.025) / 180 * cx * 24.6, numpy.maximum(0.0, want1 - cap1 - r_e))
        r_g = -r_fa * (116 / 180) - th1 - r_e * (46 / 180) - th2
        return numpy.stack([(fg * (5000 / 180) - outflow * cg + r_g) * dt, 0.0 * cx * dt,
                            (-outflow * cfa + r_fa) * dt, (-outflow * ce + r_e) * dt, r_h * dt], axis=1)

    @staticmethod
    def static_outputs(x, u):
        """(mass of glucose, mass of fumaric acid) = (Cg*180, Cfa*116)."""
        return x[0] * 180, x[2] * 116

    # ---- plant (model/BioreactorModel.py:45-135) ----
    def DEs(self, inputs):
        return numpy.array(Bioreactor.homeostatic_DEs(self.X, inputs, 1))

    def step(self, dt, inputs):
        self.t += dt
        self.X += self.DEs(inputs) * dt
        self.X[:4] = numpy.maximum(self.X[:4], 0)

    def outputs(self, inputs):
        return self.X * _MOLAR_MASS

    def raw_outputs(self, inputs):
        return self.X

    @staticmethod
    def find_SS(U_op, X0):
        """Steady state near X0 with Cx held at X0[1] (model/BioreactorModel.py:137-168).  The reference
        solves this with fsolve; for the production-phase model the solution is closed-form as long as
        the ethanol and overflow branches are inactive, which is verified before returning."""
        fg, fm = float(U_op[0]), float(U_op[1])
        cx = float(X0[1])
        cg = 280 / 180
        sat = cg / (1e-2 + cg)
        r_fa = 0.25 / 116 * cx * 24.6 * sat
        cfa = r_fa / (fg + fm)
        cap1 = (0.4 - 0.25) / 180 * cx * 24.6
        want1 = (fg * (5000 / 180) - (fg + fm) * cg - r_fa * (116 / 180)) / sat
        if not 0.0 <= want1 <= cap1:
            raise ValueError("find_SS: no steady state on the respiration-limited branch for these inputs")
        ch = (cap1 - want1) / 0.01
        res = numpy.array([cg, cx, cfa, 0.0, ch])
        assert max(abs(v) for v in Bioreactor.homeostatic_DEs(res, U_op, 1)) < 1e-9
        return res
