"""Model functions for the user-model tests (plain Python, the form gpu_se_amd.usermodel translates)."""
import math


def two_state_f(x, u, dt):
    """The structure of the pair in the reference's tests/ParticleFilter_test.py:6-20 (two states, one input), written
    as an increment: the filter adds what f returns (particle.py:265-277)."""
    x1, x2 = x
    u, = u
    dx1 = x1 - x2 * u
    dx2 = x1 / u + 2 * x2
    return dx1 * dt, dx2 * dt


def two_state_g(x, u):
    x1, x2 = x
    u, = u
    return x1 * x2, x2 + u


def three_state_f(x, u, dt):
    """Uses every supported construct: subscripts, augmented assignment, conditional expression, functions."""
    a, b = u
    rate = 0.3 * max(x[0], 0.0) / (0.5 + abs(x[0]))
    rate *= 1.0 if x[2] < 4.0 else 0.25
    decay = math.exp(-0.1 * x[1]) - 1.0
    d0 = -rate + a
    d1 = rate - 0.2 * x[1] + b * decay
    d2 = min(0.05 * x[1], 1.5) - 0.01 * math.sqrt(abs(x[2]) + 1.0)
    return d0 * dt, d1 * dt, d2 * dt


def three_state_g(x, u):
    return 10.0 * x[0] + u[0], x[1] * x[2] - 0.5 * u[1]


def loop_f(x, u, dt):
    out = []
    for v in x:
        out.append(v * dt)
    return tuple(out)


# ------------------------------------------------------------------------------ beyond the fast kernels' bounds (round 5)
def six_state_f(x, u, dt):
    """Six states, three inputs: a chain of first-order lags with a saturating product and a clipped decay."""
    x1, x2, x3, x4, x5, x6 = x
    r = x2 * x3 / (1.0 + x3)
    d1 = (-0.1 * x1 + u[0]) * dt
    d2 = (x1 - 0.2 * x2) * dt
    d3 = (r - 0.05 * x3) * dt
    d4 = (0.3 * x5 - x4 * u[1]) * dt
    d5 = (u[2] - 0.3 * x5) * dt
    d6 = (0.01 * x1 - 0.1 * min(x6, 2.0)) * dt
    return d1, d2, d3, d4, d5, d6


def six_state_g(x, u):
    """Three outputs (one of them reads an input)."""
    return x[0] + x[1], x[2] * x[3], x[5] + 0.1 * u[2]
