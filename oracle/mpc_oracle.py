"""CPU oracle for the MPC step (reference controller.py).  TEST INFRASTRUCTURE ONLY -- nothing under gpu_se_amd/ imports
this module; only tests/ may.

Two pieces:

  assemble_qp   a from-scratch restatement of the QP the reference builds for OSQP (controller.py:63-229), in the
                reference's own variable and row order, as dense arrays:
                    minimise 1/2 z' H z + q' z   subject to   l <= A z <= u
                    z = [dmu_0 .. dmu_P | y_1 .. y_P | u_-1 | du_0 .. du_M]
                Parity status of this piece: PINNED -- tests/test_oracle_mpc.py compares H, q, A, l, u entry by entry
                with the matrices of the reference's own controller.MPC (tests/golden/mpc_qp.npz, incl. the l, u that
                its step() writes, controller.py:243-262).
  solve_qp      an independent EXACT solver for such QPs (null-space elimination of the equality rows, then the
                least-distance problem through non-negative least squares, Lawson & Hanson ch. 23).
                Parity status of this piece: UNPINNED -- the reference hands the QP to osqp >= 0.6.1
                (environment.yml:21; default tolerances eps_abs = eps_rel = 1e-3), which is not installed here and
                cannot be fetched; no reference output of a solve exists, and OSQP's iterates are only approximate
                minimisers in any case.  The device controller is therefore checked through optimality: its point
                must be feasible for the reference's rows and its objective must not exceed this solver's optimum.
"""
import numpy as np


def _pairs(bounds, n):
    if bounds is None:
        return np.full(n, -np.inf), np.full(n, np.inf)
    lo, hi = (np.asarray(v, dtype=float) for v in zip(*bounds))
    return lo, hi


def layout(P, M, Nx, Ni, No):
    """Offsets of the variable blocks in z (controller.py:99-110)."""
    o_mu = 0
    o_y = (P + 1) * Nx
    o_um1 = o_y + P * No
    o_du = o_um1 + Ni
    return dict(mu=o_mu, y=o_y, um1=o_um1, du=o_du, n=o_du + (M + 1) * Ni)


def assemble_qp(P, M, Q, R, A, B, C, D, ysp, y_bounds=None, u_bounds=None, u_step_bounds=None, sparse=False):
    """H, q, A, l, u of controller.py:63-229 for x0 = 0, u_-1 = 0 -- dense arrays, or (sparse=True) scipy.sparse CSC
    matrices built block by block in the same order (the get_parts horizons for dt_control = 0.1, P = 2999 and M = 1999,
    give 16 000 variables and 28 000 rows: dense is out of reach there)."""
    if sparse:
        import scipy.sparse
        zeros = lambda shape: scipy.sparse.lil_matrix(shape)
    else:
        zeros = np.zeros
    Nx, Ni = B.shape
    No = C.shape[0]
    lay = layout(P, M, Nx, Ni, No)
    n = lay["n"]
    y_lo, y_hi = _pairs(y_bounds, No)
    u_lo, u_hi = _pairs(u_bounds, Ni)
    s_lo, s_hi = _pairs(u_step_bounds, Ni)
    mu = lambda k: slice(lay["mu"] + k * Nx, lay["mu"] + (k + 1) * Nx)          # dmu_k, k = 0..P  (dmu_0 = x0)
    yy = lambda k: slice(lay["y"] + (k - 1) * No, lay["y"] + k * No)            # y_k,   k = 1..P
    um1 = slice(lay["um1"], lay["um1"] + Ni)
    du = lambda j: slice(lay["du"] + j * Ni, lay["du"] + (j + 1) * Ni)          # du_j,  j = 0..M

    H = zeros((n, n))
    q = np.zeros(n)
    for k in range(1, P + 1):                                                   # :99-110
        H[yy(k), yy(k)] = Q
        q[yy(k)] = -Q @ ysp
    for j in range(M + 1):
        H[du(j), du(j)] = R

    rows, lows, ups = [], [], []

    def block(nrows):
        blk = zeros((nrows, n))
        rows.append(blk)
        return blk

    # u_-1 pinned (:112-117)
    blk = block(Ni)
    blk[:, um1] = np.eye(Ni)
    lows.append(np.zeros(Ni)); ups.append(np.zeros(Ni))
    # state recursion (:119-152): -dmu_0 = -x0;  (A - I) dmu_0 - dmu_1 + B (u_-1 + du_0) = 0;
    #                             A dmu_{k-1} - dmu_k + B du_{k-1} = 0 (2 <= k <= M);  A dmu_{k-1} - dmu_k = 0 (k > M)
    blk = block((P + 1) * Nx)
    r = lambda k: slice(k * Nx, (k + 1) * Nx)
    blk[r(0), mu(0)] = -np.eye(Nx)
    blk[r(1), mu(0)] = A - np.eye(Nx)
    blk[r(1), mu(1)] = -np.eye(Nx)
    blk[r(1), um1] = B
    blk[r(1), du(0)] = B
    for k in range(2, P + 1):
        blk[r(k), mu(k - 1)] = A
        blk[r(k), mu(k)] = -np.eye(Nx)
        if k <= M:
            blk[r(k), du(k - 1)] = B
    lows.append(np.zeros((P + 1) * Nx)); ups.append(np.zeros((P + 1) * Nx))
    # outputs (:154-178): C (dmu_0 + dmu_1) - y_1 + D (u_-1 + du_0 + du_1) = -bias;
    #                     C dmu_k - y_k + y_{k-1} + D du_k [k <= M] = -bias (k >= 2)
    blk = block(P * No)
    r = lambda k: slice((k - 1) * No, k * No)
    blk[r(1), mu(0)] = C
    blk[r(1), mu(1)] = C
    blk[r(1), yy(1)] = -np.eye(No)
    blk[r(1), um1] = D
    blk[r(1), du(0)] = D
    blk[r(1), du(1)] = (blk[r(1), du(1)].toarray() if sparse else blk[r(1), du(1)]) + D
    for k in range(2, P + 1):
        blk[r(k), mu(k)] = C
        blk[r(k), yy(k)] = -np.eye(No)
        blk[r(k), yy(k - 1)] = np.eye(No)
        if k <= M:
            blk[r(k), du(k)] = D
    lows.append(np.zeros(P * No)); ups.append(np.zeros(P * No))
    # y limits (:180-187)
    blk = block(P * No)
    for k in range(1, P + 1):
        blk[r(k), yy(k)] = np.eye(No)
    lows.append(np.tile(y_lo, P)); ups.append(np.tile(y_hi, P))
    # move limits (:189-195)
    blk = block((M + 1) * Ni)
    for j in range(M + 1):
        blk[j * Ni:(j + 1) * Ni, du(j)] = np.eye(Ni)
    lows.append(np.tile(s_lo, M + 1)); ups.append(np.tile(s_hi, M + 1))
    # limit on the first input only (:209-219)
    blk = block(Ni)
    blk[:, um1] = np.eye(Ni)
    blk[:, du(0)] = np.eye(Ni)
    lows.append(u_lo); ups.append(u_hi)
    if sparse:
        return H.tocsc(), q, scipy.sparse.vstack(rows).tocsc(), np.concatenate(lows), np.concatenate(ups)
    return H, q, np.vstack(rows), np.concatenate(lows), np.concatenate(ups)


def write_step(l, u, P, Nx, Ni, No, x0, um1, bias):
    """What step() writes into l and u before solving (controller.py:243-262)."""
    l, u = l.copy(), u.copy()
    for v in (l, u):
        v[:Ni] = um1
        v[Ni:Ni + Nx] = -x0
        v[Ni + (P + 1) * Nx: Ni + (P + 1) * Nx + P * No] = np.tile(-bias, P)
    return l, u


def solve_qp(H, q, A, l, u):
    """Exact solution of  min 1/2 z'Hz + q'z,  l <= Az <= u  by textbook linear algebra that shares nothing with the
    product's condensation or its dual sweeps:
      1. rows with l == u are eliminated by the null-space method (SVD of the equality block: z = z_p + Z w);
      2. the reduced Hessian Z'HZ is positive definite here (R > 0 on the moves); with its Cholesky factor the
         problem becomes a least-distance problem  min |v|  s.t.  E v >= f  over the finite inequality sides;
      3. that is solved through non-negative least squares (Lawson & Hanson, "Solving Least Squares Problems", ch. 23:
         LDP via NNLS; scipy.optimize.nnls is their active-set algorithm, finite and exact).
    Returns (z, objective, info).  Raises ValueError if the rows are infeasible."""
    import scipy.linalg
    import scipy.optimize
    eq = l == u
    Ae, be = A[eq], l[eq]
    # null space and a particular solution of the equality rows
    U, sv, Vt = np.linalg.svd(Ae, full_matrices=True)
    rank = int(np.sum(sv > 1e-12 * sv[0]))
    z_p = Vt[:rank].T @ ((U[:, :rank].T @ be) / sv[:rank])
    Z = Vt[rank:].T
    Hr = Z.T @ H @ Z
    gr = Z.T @ (H @ z_p + q)
    L = np.linalg.cholesky(Hr)
    # inequality sides  a'z >= l  and  -a'z >= -u
    Ai = A[~eq]
    li, ui = l[~eq], u[~eq]
    G = np.vstack([Ai[np.isfinite(li)], -Ai[np.isfinite(ui)]])
    h = np.concatenate([li[np.isfinite(li)], -ui[np.isfinite(ui)]])
    w_free = -scipy.linalg.cho_solve((L, True), gr)
    if len(h) == 0:
        w = w_free
        active = 0
    else:
        # v = L'(w - w_free):  1/2 w'Hr w + gr'w = 1/2 |v|^2 + const;   G(z_p + Z w) >= h  <=>  E v >= f
        GZ = G @ Z
        E = scipy.linalg.solve_triangular(L, GZ.T, lower=True).T
        f = h - G @ z_p - GZ @ w_free
        M = np.vstack([E.T, f[None, :]])
        e = np.zeros(M.shape[0]); e[-1] = 1.0
        lam, _ = scipy.optimize.nnls(M, e, maxiter=50 * M.shape[1])
        r = M @ lam - e
        if np.linalg.norm(r) < 1e-12:
            raise ValueError("the QP's rows are infeasible")
        v = -r[:-1] / r[-1]
        w = w_free + scipy.linalg.solve_triangular(L.T, v, lower=False)
        active = int(np.sum(lam > 0))
    z = z_p + Z @ w
    return z, 0.5 * z @ H @ z + q @ z, dict(active=active)


def hildreth(law, theta, max_sweeps=4000, eps_abs=1e-14, eps_rel=1e-12):
    """The explicit law of include/gse.h evaluated in NumPy for ONE parameter vector (checker for the device kernel:
    same sweeps, same stopping rule).  law: dict with Gc, gc0, S, lo, hi, Gu, gu0, Tu, Gy, gy0, Ty."""
    m = len(law["lo"])
    c0 = law["Gc"] @ theta + law["gc0"] if m else np.zeros(0)
    c, nu = c0.copy(), np.zeros(m)
    ok = m == 0
    for _ in range(max_sweeps):
        if ok:
            break
        for i in range(m):
            sii = law["S"][i, i]
            if sii <= 0:
                continue
            r = c[i] + sii * nu[i]
            want = (r - law["hi"][i]) / sii if r > law["hi"][i] else ((r - law["lo"][i]) / sii if r < law["lo"][i] else 0.0)
            delta = want - nu[i]
            nu[i] = want
            if delta != 0.0:
                c -= law["S"][:, i] * delta
        res = np.where(nu > 0, np.abs(c - law["hi"]), np.where(nu < 0, np.abs(c - law["lo"]),
                                                              np.maximum(np.maximum(c - law["hi"], law["lo"] - c), 0)))
        with np.errstate(invalid="ignore"):
            scale = np.abs(c0) + np.where(nu > 0, np.abs(law["hi"]), np.where(nu < 0, np.abs(law["lo"]),
                                          np.where(c > law["hi"], np.abs(law["hi"]), np.where(c < law["lo"], np.abs(law["lo"]), 0.0))))
        ok = bool(np.all(res <= eps_abs + eps_rel * scale))
    du0 = law["Gu"] @ theta + law["gu0"] - (law["Tu"] @ nu if m else 0)
    y1 = law["Gy"] @ theta + law["gy0"] - (law["Ty"] @ nu if m else 0)
    return du0, y1, nu, ok


# ----------------------------------------------------------------------------------------------------------------------
# The controller of the closed loop as an object (controller.py:9-279), for oracle/closed_loop.py.  The QP is the
# reference's (assemble_qp, pinned); what solves it is NOT OSQP (not installed, not fetchable: osqp >= 0.6.1,
# environment.yml:21) but an exact method that shares nothing with the product's condensation + Hildreth sweeps:
# sparse KKT systems over an enumerated active set.
# ----------------------------------------------------------------------------------------------------------------------
class KKTSolver:
    """min 1/2 z'Hz + q'z, l <= Az <= u with l, u rewritten between solves (the equality PATTERN l == u is fixed at
    construction, as controller.py:243-262 only rewrites right-hand sides of equality rows).  Every subset of the finite
    inequality sides is a candidate active set; for a candidate the stationarity + equality + active rows form ONE sparse
    symmetric indefinite system (scipy.sparse.linalg.splu, factor cached per subset, one round of iterative refinement);
    the candidate is optimal iff the inactive sides hold and the multipliers of the active ones have the right sign
    (KKT conditions of a convex QP: necessary and sufficient).  Exact and finite; meant for the handful of bounded rows
    the closed loop has (two: u_0 >= 0 for both inputs, sim_base.py:82-85)."""

    MAX_SIDES = 10

    def __init__(self, H, q, A, l, u):
        import scipy.sparse
        self.H = scipy.sparse.csc_matrix(H)
        self.q = np.asarray(q, dtype=float)
        self.A = scipy.sparse.csr_matrix(A)
        self.eq = np.asarray(l) == np.asarray(u)
        self.Aeq = self.A[np.flatnonzero(self.eq)]
        ineq = np.flatnonzero(~self.eq)
        self.sides = [(int(i), -1) for i in ineq if np.isfinite(l[i])] + [(int(i), +1) for i in ineq if np.isfinite(u[i])]
        # (more sides than can be enumerated -- every output bounded over the whole horizon, controller.py:180-207: solve()
        #  then walks active sets instead, see _solve_by_active_sets; the KKT conditions it ends on are the same proof)
        self.enumerate = len(self.sides) <= self.MAX_SIDES
        self._lu = {}

    def _factor(self, subset):
        import scipy.sparse
        import scipy.sparse.linalg
        if subset not in self._lu:
            rows = [self.Aeq] + [self.A[self.sides[s][0]] for s in subset]
            Ar = scipy.sparse.vstack(rows).tocsc()
            m = Ar.shape[0]
            K = scipy.sparse.bmat([[self.H, Ar.T], [Ar, scipy.sparse.csc_matrix((m, m))]], format="csc")
            self._lu[subset] = (scipy.sparse.linalg.splu(K), K)
        return self._lu[subset]

    def _solve_by_active_sets(self, l, u, max_steps=2000):
        """Many bounded rows: a walk over active sets instead of their enumeration -- solve the KKT system of the current set;
        drop the active side whose multiplier has the wrong sign most; otherwise add the most violated side; stop when
        neither exists.  What is returned satisfies the KKT conditions of the convex QP (checked here, to the tolerances of
        solve()), which is the proof of optimality whatever the walk was; a walk that does not end raises."""
        import scipy.sparse
        import scipy.sparse.linalg
        n, n_eq = self.H.shape[0], int(self.eq.sum())
        beq = l[self.eq]
        active = []                                                      # [(row, sign)]
        for _ in range(max_steps):
            rows = [self.Aeq] + [self.A[i] for i, _ in active]
            Ar = scipy.sparse.vstack(rows).tocsc()
            m = Ar.shape[0]
            K = scipy.sparse.bmat([[self.H, Ar.T], [Ar, scipy.sparse.csc_matrix((m, m))]], format="csc")
            rhs = np.concatenate([-self.q, beq, [l[i] if sgn < 0 else u[i] for i, sgn in active]])
            try:
                lu = scipy.sparse.linalg.splu(K)
            except RuntimeError:
                # active rows that depend on each other through the equalities (a move limit and an output limit that say the
                # same thing): factor with a whiff of dual regularisation, refine against the true system
                reg = scipy.sparse.diags(np.concatenate([np.zeros(n), np.full(m, -1e-11)]))
                lu = scipy.sparse.linalg.splu((K + reg).tocsc())
            sol = lu.solve(rhs)
            for _ in range(3):
                sol += lu.solve(rhs - K @ sol)
            z, lam = sol[:n], sol[n + n_eq:]
            wrong = [(lam[k] if sgn < 0 else -lam[k]) for k, (i, sgn) in enumerate(active)]     # > 0: wrong sign
            if wrong and max(wrong) > 1e-12:
                active.pop(int(np.argmax(wrong)))
                continue
            Az = self.A @ z
            scale = 1e-9 * (1.0 + np.abs(Az))
            low, high = (l - scale) - Az, Az - (u + scale)
            low[self.eq], high[self.eq] = -np.inf, -np.inf
            worst_low, worst_high = int(np.argmax(low)), int(np.argmax(high))
            if low[worst_low] <= 0 and high[worst_high] <= 0:
                return z
            taken = {i for i, _ in active}
            pick = (worst_low, -1) if low[worst_low] >= high[worst_high] else (worst_high, +1)
            if pick[0] in taken:
                raise ValueError("the QP's bounds on row %d contradict each other" % pick[0])
            active.append(pick)
        raise ValueError("the active-set walk did not end in %d steps" % max_steps)

    def solve(self, l, u):
        import itertools
        l, u = np.asarray(l, dtype=float), np.asarray(u, dtype=float)
        if not self.enumerate:
            return self._solve_by_active_sets(l, u)
        n = self.H.shape[0]
        n_eq = int(self.eq.sum())
        beq = l[self.eq]
        best = None
        for size in range(len(self.sides) + 1):
            for subset in itertools.combinations(range(len(self.sides)), size):
                rows = [self.sides[s][0] for s in subset]
                if len(set(rows)) < len(rows):
                    continue                                            # both sides of one row active: only if l == u
                lu, K = self._factor(subset)
                rhs = np.concatenate([-self.q, beq, [l[i] if sgn < 0 else u[i] for i, sgn in (self.sides[s] for s in subset)]])
                sol = lu.solve(rhs)
                sol += lu.solve(rhs - K @ sol)                          # one round of iterative refinement
                z, lam = sol[:n], sol[n + n_eq:]
                Az = self.A @ z
                scale = 1e-9 * (1.0 + np.abs(Az))
                feasible = np.all(Az >= l - scale) and np.all(Az <= u + scale)
                # stationarity: Hz + q + A'lam = 0; a row at its LOWER bound needs lam <= 0, at its UPPER bound lam >= 0
                signs = all((lam[k] <= 1e-12) if self.sides[s][1] < 0 else (lam[k] >= -1e-12) for k, s in enumerate(subset))
                if feasible and signs:
                    return z
                if best is None and feasible:
                    best = z
        raise ValueError("the QP has no KKT point among the enumerated active sets (rows infeasible)")


class MPCOracle:
    """controller.MPC restated around the pinned QP: same constructor arguments (the linear model as its four matrices),
    same step() contract -- clip to +-1e10 (controller.py:238-241), write u_-1, -x0 and the tiled -bias into l and u
    (:243-262), solve, first input = du_0 + u_-1 (:272-275), y_predicted = y_1 - bias (:277); raises ValueError where the
    reference raises on a non-solved status."""

    def __init__(self, P, M, Q, R, A, B, C, D, ysp, y_bounds=None, u_bounds=None, u_step_bounds=None):
        self.P, self.M = int(P), int(M)
        self.Nx, self.Ni = B.shape
        self.No = C.shape[0]
        H, q, Am, l, u = assemble_qp(P, M, np.asarray(Q, float), np.asarray(R, float), A, B, C, D, np.asarray(ysp, float),
                                     y_bounds, u_bounds, u_step_bounds, sparse=True)
        self.l0, self.u0 = l, u
        self.solver = KKTSolver(H, q, Am, l, u)
        self.y_predicted = None

    def step(self, x0, um1, y0):
        clip = lambda v: np.maximum(np.minimum(np.asarray(v, dtype=float), 1e10), -1e10)
        x0, um1, y0 = clip(x0), clip(um1), clip(y0)
        bias = y0 - self.y_predicted if self.y_predicted is not None else np.zeros_like(y0)      # :248-251
        l, u = write_step(self.l0, self.u0, self.P, self.Nx, self.Ni, self.No, x0, um1, bias)
        z = self.solver.solve(l, u)
        m = (self.P + 1) * self.Nx + self.P * self.No + self.Ni
        ctrl = z[m:m + self.Ni] + um1
        self.y_predicted = z[(self.P + 1) * self.Nx:(self.P + 1) * self.Nx + self.No] - bias
        return ctrl
