"""Shared set-up of the closed-loop parity tests: oracle loops (oracle/closed_loop.py) wired as sim_base.get_parts wires the
reference's, with the linear model taken from the reference's own matrices (tests/golden/linear_model.npz)."""
import os

import numpy as np

from oracle import closed_loop, mpc_oracle
from oracle import gse_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_LIN = None


def linear(dt_control):
    """(tag, arrays) of the reference's LinearModel for dt_control in {1, 0.1} (after select_subset)."""
    global _LIN
    if _LIN is None:
        _LIN = np.load(os.path.join(GOLDEN, "linear_model.npz"))
    if dt_control not in (1, 1.0, 0.1):
        raise ValueError("the golden linear models cover dt_control = 1 and 0.1")
    tag = "dt1" if dt_control == 1 else "dt01"
    return {k: _LIN["%s_sub_%s" % (tag, k)] for k in ("A", "B", "C", "D", "x_bar", "u_bar", "y_bar")} | {"ysp": _LIN[tag + "_ysp"]}


_MPC_CACHE = {}


def mpc_oracle_for(dt_control):
    """The MPC of get_parts (sim_base.py:75-86) as an oracle object; the sparse KKT factors are shared between instances
    of one dt_control (assembling P = 2999 takes a second)."""
    lm = linear(dt_control)
    if dt_control not in _MPC_CACHE:
        u_bar = lm["u_bar"]
        _MPC_CACHE[dt_control] = mpc_oracle.MPCOracle(
            int(300 // dt_control), max(int(200 // dt_control), 1), np.diag([0.1, 1]), np.diag([1, 1]), lm["A"], lm["B"], lm["C"],
            lm["D"], lm["ysp"], u_bounds=[np.array([0, np.inf]) - u_bar[0], np.array([0, np.inf]) - u_bar[1]])
    K = _MPC_CACHE[dt_control]
    fresh = object.__new__(mpc_oracle.MPCOracle)
    fresh.__dict__.update(K.__dict__)
    fresh.y_predicted = None
    return fresh


def loop_oracle(estimator, dt_control, dt_predict, end_time, x_plant, noise, mpc=True):
    lm = linear(dt_control)
    glue = closed_loop.LinearGlue(lm["x_bar"], lm["u_bar"], lm["y_bar"])
    return closed_loop.ClosedLoopOracle(estimator, mpc_oracle_for(dt_control) if mpc else None, glue, x_plant, dt_control,
                                        dt_predict, end_time, noise)


def loop_oracle_from_golden(g, name, weights_dtype=None, follow=True):
    """A ClosedLoopOracle set up as the recorded reference run ``name`` of closed_loop.npz was; follow: copy the recorded
    ancestors (the oracle's own are counted against them)."""
    n, dtc, dtp, end, is_pf = g[name + "_cfg"]
    n, end = int(n), int(end)
    sp, mp = orc.bioreactor_noise()
    if is_pf:
        est = orc.ParticleFilterOracle(n, None, sp, mp, particles=g[name + "_particles0"], weights_dtype=weights_dtype)
    else:
        est = orc.GSUKFOracle(n, None, sp, mp, means=g[name + "_means0"], covariances=g[name + "_covs0"])
    noise = closed_loop.LoopNoise(g[name + "_plant_state"], g[name + "_plant_meas"], g[name + "_predict_noise"], g[name + "_r"],
                                  ancestors=g[name + "_ancestors"] if follow else None)
    return loop_oracle(est, float(dtc), float(dtp), end, g[name + "_xs"][0], noise, mpc=not name.startswith("pf_fail"))
