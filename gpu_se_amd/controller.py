"""Model-predictive controller of the closed loop (reference controller.py:9-279), evaluated on the device.

The reference states the controller as a sparse QP over (state deltas, outputs, u_{-1}, input moves) and calls OSQP on
every control tick.  Between ticks only theta = (x0, u_{-1}, bias) changes (controller.py:236-262), so here the QP is
reduced ONCE, on the host, in fp64, to an explicit law in theta and the multipliers of the bounded rows (its matrices
through sparse KKT solves in the reference's variable space, _law_from_kkt; the dense condensed form below is what the law
means and what the tests inspect):

    * the equality rows (controller.py:119-178: u_{-1} pinned, the delta-state recursion, the accumulated outputs)
      define every state and output as a linear function of the moves d = (du_0 .. du_M) and of theta:
      y = Phi d + Psi theta;
    * the cost (controller.py:99-110) becomes  1/2 d' Hc d + (F theta + f0)' d  with the dense, constant,
      positive-definite  Hc = Phi' (I x Q) Phi + I x R;
    * of the inequality rows (controller.py:180-219) only those with a finite bound are kept:
      lo <= Cin d + Pin theta <= hi.

What is left per tick is the explicit law documented at ``gse_mpc_step`` (include/gse.h): a handful of small
matrix-vector products and Hildreth's dual sweeps over the bounded rows, evaluated for B rollouts by one kernel
launch.  There is no host fall-back: without libgse_hip the controller does not step.

Parity: the QP this class condenses is checked against the reference's own matrices (tests/golden/mpc_qp.npz) --
every (d, theta) is mapped to the full variable vector, which must satisfy the reference's equality rows exactly and
reproduce its objective; OSQP itself is not available offline, so the SOLUTION is verified through optimality
(feasible, and no worse than an independent solver's optimum), see tests/test_oracle_mpc.py, tests/test_gpu_mpc.py.
"""
import ctypes

import numpy

from gpu_se_amd import _lib

MAX_ROWS = 2048          # GSE_MPC_MAX_ROWS_TOTAL: bounded rows in all (beyond 64 a working set of at most 64 rows, GSE_MPC_MAX_ROWS)
MAX_ACTIVE = 64          # GSE_MPC_MAX_ROWS: rows that can be active together; the most the fused loop kernels take at all


class _Law(ctypes.Structure):
    _fields_ = [("n_theta", ctypes.c_int32), ("n_u", ctypes.c_int32), ("n_y", ctypes.c_int32), ("n_rows", ctypes.c_int32),
                ("max_sweeps", ctypes.c_int32), ("report_unsolved", ctypes.c_int32),
                ("eps_abs", ctypes.c_double), ("eps_rel", ctypes.c_double)] + \
               [(name, ctypes.c_void_p) for name in ("Gc", "gc0", "S", "lo", "hi", "Gu", "gu0", "Tu", "Gy", "gy0", "Ty")]


class _Glue(ctypes.Structure):
    """struct gse_loop_glue (include/gse.h)."""
    _fields_ = [("n_state_full", ctypes.c_int32), ("n_input_full", ctypes.c_int32), ("n_output_full", ctypes.c_int32),
                ("reserved", ctypes.c_int32),
                ("states", ctypes.c_int32 * 8), ("inputs", ctypes.c_int32 * 8), ("outputs", ctypes.c_int32 * 8),
                ("x_bar", ctypes.c_double * 8), ("u_bar", ctypes.c_double * 8), ("y_bar", ctypes.c_double * 8),
                ("u_fallback", ctypes.c_double * 8)]


def _bounds(pairs, n):
    if pairs is None:
        return numpy.full(n, -numpy.inf), numpy.full(n, numpy.inf)
    lo, hi = (numpy.asarray(v, dtype=float) for v in zip(*pairs))
    return lo, hi


class MPC:
    """Deterministic reformulation of a linear chance-constrained MPC on a discrete linear model
    (controller.py:9-62).  Same constructor, attributes (``P``, ``M``, ``Q``, ``R``, ``model``, ``ysp``,
    ``y_predicted``) and ``step`` contract as the reference; ``step_batch`` advances B rollouts at once."""

    def __init__(self, P, M, Q, R, lin_model, ysp, y_bounds=None, u_bounds=None, u_step_bounds=None,
                 max_sweeps=4000, eps_abs=1e-14, eps_rel=1e-12):
        if M > P or M < 1:
            raise ValueError("MPC needs 1 <= M <= P")
        self.P, self.M = int(P), int(M)
        self.Q, self.R = numpy.asarray(Q, dtype=float), numpy.asarray(R, dtype=float)
        self.model = lin_model
        self.ysp = numpy.asarray(ysp, dtype=float)
        self.y_predicted = None
        self._known = None
        Nx, Ni = lin_model.B.shape
        No = lin_model.C.shape[0]
        self.Nx, self.Ni, self.No = Nx, Ni, No
        self._y_lim = _bounds(y_bounds, No)
        self._u_lim = _bounds(u_bounds, Ni)
        self._du_lim = _bounds(u_step_bounds, Ni)
        self._solver = (int(max_sweeps), float(eps_abs), float(eps_rel))
        # fault injection: True -> every step reports "not solved", as OSQP does when it returns a non-solved status
        # (controller.py:268-269); step() then raises and the loops take the reference's fallback (sim_base.py:271-273)
        self.report_unsolved = False
        self._built = None          # the explicit law: built on first use (callers that only want the estimator never pay)
        self._dense = None          # the condensed problem as dense arrays (inspection / tests / singular R)
        self._dev = None

    @property
    def _law(self):
        if self._built is None:
            self._build()
        return self._built[0]

    @property
    def _condensed(self):
        """The condensed problem as dense arrays (Hc, F, f0, Cin, Pin, Phi, Psi): for inspection and tests; the law itself
        is built from it only when R is singular (see _build)."""
        if self._dense is None:
            self._dense = self._condense()
        return self._dense

    # ------------------------------------------------------------------------------------------------------------
    # host set-up
    # ------------------------------------------------------------------------------------------------------------
    def response(self, d, theta):
        """Outputs y_1 .. y_P (stacked, [P*No, cols]) and state deltas for move columns ``d`` [(M+1)*Ni, cols] and
        parameter columns ``theta`` [Nx+Ni+No, cols] -- the reference's equality rows solved forwards
        (controller.py:119-178):  dmu_1 = (A - I) x0 + B (u_-1 + du_0),  dmu_k = A dmu_{k-1} + B du_{k-1} (k <= M),
        y_1 = C (x0 + dmu_1) + D (u_-1 + du_0 + du_1) + bias,  y_k = y_{k-1} + C dmu_k + D du_k + bias."""
        A, B, C, D = self.model.A, self.model.B, self.model.C, self.model.D
        Nx, Ni, P, M = self.Nx, self.Ni, self.P, self.M
        x0, um1, bias = theta[:Nx], theta[Nx:Nx + Ni], theta[Nx + Ni:]
        move = lambda j: d[j * Ni:(j + 1) * Ni]
        dmu = (A - numpy.eye(Nx)) @ x0 + B @ (um1 + move(0))
        y = C @ (x0 + dmu) + D @ (um1 + move(0) + move(1)) + bias
        ys, dmus = [y], [dmu]
        for k in range(2, P + 1):
            dmu = A @ dmu
            if k <= M:
                dmu = dmu + B @ move(k - 1)
                y = y + C @ dmu + D @ move(k) + bias
            else:
                y = y + C @ dmu + bias
            ys.append(y)
            dmus.append(dmu)
        return numpy.vstack(ys), numpy.vstack(dmus)

    def _bounded_rows(self):
        """The inequality rows of the reference that carry a finite bound, in its order (controller.py:180-219: outputs,
        input moves, first input): [(kind, step, index, lo, hi)]."""
        rows = []
        for k in range(self.P):
            for o in range(self.No):
                rows.append(("y", k, o, self._y_lim[0][o], self._y_lim[1][o]))
        for j in range(self.M + 1):
            for i in range(self.Ni):
                rows.append(("du", j, i, self._du_lim[0][i], self._du_lim[1][i]))
        for i in range(self.Ni):          # u_-1 + du_0 (controller.py:209-219 keeps the bound on the first input only)
            rows.append(("u", 0, i, self._u_lim[0][i], self._u_lim[1][i]))
        rows = [r for r in rows if numpy.isfinite(r[3]) or numpy.isfinite(r[4])]
        if len(rows) > MAX_ROWS:
            raise NotImplementedError("this MPC has %d bounded rows; the device solver takes at most %d (of which %d active "
                                      "together): bound fewer outputs / moves, or shorten the horizons"
                                      % (len(rows), MAX_ROWS, MAX_ACTIVE))
        return rows

    def _build(self):
        """The explicit law.  R positive definite (every wiring of sim_base.get_parts): through sparse KKT systems in the
        reference's full variable space -- the condensed Hessian of the long horizons is hopelessly conditioned (4.5e14 at
        P = 2999: the outputs accumulate), the KKT matrix is not.  R singular (the reference's tank tests): through the dense
        condensed problem and a pseudo-inverse."""
        if numpy.all(numpy.linalg.eigvalsh((self.R + self.R.T) / 2) > 0):
            self._built = (self._law_from_kkt(),)
        else:
            self._built = (self._law_from_condensed(),)

    def _law_from_kkt(self):
        """z = [dmu_0 .. dmu_P | y_1 .. y_P | u_-1 | du_0 .. du_M] (the reference's variable order, controller.py:99-110).
        The minimiser of the equality-constrained QP is linear in (theta, 1, nu): z = Zt theta + z0 + Zn nu, one sparse
        symmetric-indefinite solve per column (Nx + Ni + No + 1 + bounded rows of them, one LU).  The law's matrices are rows
        of those columns: du_0, y_1 and the bounded rows."""
        import scipy.sparse as sp
        import scipy.sparse.linalg
        A, B, C, D = (numpy.asarray(m, dtype=float) for m in (self.model.A, self.model.B, self.model.C, self.model.D))
        Nx, Ni, No, P, M = self.Nx, self.Ni, self.No, self.P, self.M
        o_y, o_u, o_d = (P + 1) * Nx, (P + 1) * Nx + P * No, (P + 1) * Nx + P * No + Ni
        n = o_d + (M + 1) * Ni
        rr, cc, vv = [], [], []

        def put(r0, c0, blk):
            blk = numpy.atleast_2d(blk)
            ri, ci = numpy.nonzero(blk)
            rr.append(r0 + ri); cc.append(c0 + ci); vv.append(blk[ri, ci])

        I_x, I_y, I_u = numpy.eye(Nx), numpy.eye(No), numpy.eye(Ni)
        r_state, r_out = Ni, Ni + (P + 1) * Nx
        put(0, o_u, I_u)                                                   # u_-1 = theta_u
        put(r_state, 0, -I_x)                                              # -dmu_0 = -x0
        for k in range(1, P + 1):                                          # dmu_k = A dmu_{k-1} (+ B moves)
            r0 = r_state + k * Nx
            put(r0, (k - 1) * Nx, A - I_x if k == 1 else A)
            put(r0, k * Nx, -I_x)
            if k == 1:
                put(r0, o_u, B)
            if k <= M:
                put(r0, o_d + (k - 1) * Ni, B)
        for k in range(1, P + 1):                                          # y_k = y_{k-1} + C dmu_k + D du_k + bias
            r0 = r_out + (k - 1) * No
            put(r0, k * Nx, C)
            put(r0, o_y + (k - 1) * No, -I_y)
            if k == 1:
                put(r0, 0, C)
                put(r0, o_u, D)
                put(r0, o_d, D)
                put(r0, o_d + Ni, D)
            else:
                put(r0, o_y + (k - 2) * No, I_y)
                if k <= M:
                    put(r0, o_d + k * Ni, D)
        n_eq = Ni + (P + 1) * Nx + P * No
        Aeq = sp.coo_matrix((numpy.concatenate(vv), (numpy.concatenate(rr), numpy.concatenate(cc))), shape=(n_eq, n)).tocsc()
        H = sp.block_diag([sp.csc_matrix((o_y, o_y)), sp.kron(sp.eye(P), self.Q), sp.csc_matrix((Ni, Ni)),
                           sp.kron(sp.eye(M + 1), self.R)], format="csc")
        q = numpy.zeros(n)
        q[o_y:o_u] = numpy.tile(-self.Q @ self.ysp, P)
        rows = self._bounded_rows()
        m, nt = len(rows), Nx + Ni + No
        a_rows = numpy.zeros((m, n))
        for i, (kind, step, idx, _, _) in enumerate(rows):
            if kind == "y":
                a_rows[i, o_y + step * No + idx] = 1.0
            elif kind == "du":
                a_rows[i, o_d + step * Ni + idx] = 1.0
            else:
                a_rows[i, o_u + idx] = a_rows[i, o_d + idx] = 1.0
        rhs = numpy.zeros((n + n_eq, nt + 1 + m))
        for k in range(Nx):
            rhs[n + r_state + k, k] = -1.0                                 # -dmu_0 = -x0
        for k in range(Ni):
            rhs[n + k, Nx + k] = 1.0                                       # u_-1
        for k in range(No):
            rhs[n + r_out + k:n + n_eq:No, Nx + Ni + k] = -1.0             # ... = -bias on every output row
        rhs[:n, nt] = -q
        rhs[:n, nt + 1:] = -a_rows.T                                        # H z + q + Aeq' lam + a_i nu_i = 0
        K = sp.bmat([[H, Aeq.T], [Aeq, None]], format="csc")
        lu = scipy.sparse.linalg.splu(K)
        Z = lu.solve(rhs)
        Z += lu.solve(rhs - K @ Z)                                          # one round of iterative refinement
        Z = Z[:n]
        Zt, z0, Zn = Z[:, :nt], Z[:, nt], Z[:, nt + 1:]
        du0, y1 = slice(o_d, o_d + Ni), slice(o_y, o_y + No)
        self._moves_of = (Zt[o_d:], z0[o_d:], Zn[o_d:])
        return dict(Gc=a_rows @ Zt, gc0=a_rows @ z0, S=-(a_rows @ Zn), lo=numpy.array([r[3] for r in rows], dtype=float),
                    hi=numpy.array([r[4] for r in rows], dtype=float), Gu=Zt[du0], gu0=z0[du0], Tu=-Zn[du0].reshape(Ni, m),
                    Gy=Zt[y1], gy0=z0[y1], Ty=-Zn[y1].reshape(No, m))

    def _condense(self):
        Nx, Ni, No, P, M = self.Nx, self.Ni, self.No, self.P, self.M
        nd, nt = (M + 1) * Ni, Nx + Ni + No
        Phi, _ = self.response(numpy.eye(nd), numpy.zeros((nt, nd)))
        Psi, _ = self.response(numpy.zeros((nd, nt)), numpy.eye(nt))
        QPhi = numpy.einsum("ij,pjk->pik", self.Q, Phi.reshape(P, No, nd)).reshape(P * No, nd)      # (I x Q) Phi
        Hc = QPhi.T @ Phi
        for j in range(M + 1):
            Hc[j * Ni:(j + 1) * Ni, j * Ni:(j + 1) * Ni] += self.R
        F = QPhi.T @ Psi
        f0 = -QPhi.T @ numpy.tile(self.ysp, P)
        rows = self._bounded_rows()
        m = len(rows)
        Cin, Pin = numpy.zeros((m, nd)), numpy.zeros((m, nt))
        for i, (kind, step, idx, _, _) in enumerate(rows):
            if kind == "y":
                Cin[i], Pin[i] = Phi[step * No + idx], Psi[step * No + idx]
            elif kind == "du":
                Cin[i, step * Ni + idx] = 1.0
            else:
                Cin[i, idx], Pin[i, Nx + idx] = 1.0, 1.0
        return dict(Hc=Hc, F=F, f0=f0, Cin=Cin, Pin=Pin, Phi=Phi, Psi=Psi, lo=numpy.array([r[3] for r in rows], dtype=float),
                    hi=numpy.array([r[4] for r in rows], dtype=float))

    def _law_from_condensed(self):
        """R = 0 with D = 0 (the reference's tank tests): the last move du_M enters no state and no output, the cost is flat
        along it and Hc is only positive SEMI-definite.  OSQP copes (any minimiser will do); here the minimiser of least norm
        is taken, after checking that the choice cannot show: the flat directions must not reach du_0 or y_1 (all the loop
        uses), and a bounded row that touches flat moves only costs nothing to satisfy and is dropped."""
        c = self._condensed
        Hc, F, f0, Cin, Pin, Phi, Psi = (c[k] for k in ("Hc", "F", "f0", "Cin", "Pin", "Phi", "Psi"))
        Ni, No = self.Ni, self.No
        evals, evecs = numpy.linalg.eigh((Hc + Hc.T) / 2)
        curved = evals > 1e-12 * evals.max()
        if not curved.any() or evals.min() < -1e-9 * evals.max():
            raise numpy.linalg.LinAlgError("the condensed MPC Hessian is not positive semi-definite")
        flat = evecs[:, ~curved]
        lo, hi = c["lo"], c["hi"]
        if flat.shape[1]:
            if numpy.max(numpy.abs(flat[:Ni])) > 1e-9 or numpy.max(numpy.abs(Phi[:No] @ flat)) > 1e-9:
                raise NotImplementedError("the MPC cost is flat along a direction that moves du_0 or y_1 (R is singular "
                                          "there): the minimiser is not unique where the loop looks; give those moves a weight")
            along = numpy.abs(Cin @ flat).max(axis=1) > 1e-9 if len(lo) else numpy.zeros(0, dtype=bool)
            only_flat = along & (numpy.abs(Cin @ evecs[:, curved]).max(axis=1) <= 1e-9) & numpy.all(Pin == 0, axis=1)
            if numpy.any(along & ~only_flat):
                raise NotImplementedError("a bounded row acts along a direction in which the MPC cost is flat (R is "
                                          "singular there) as well as along curved ones: give those moves a weight")
            keep = ~only_flat                       # rows on cost-free moves alone (a limit on du_M): any value inside will do
            Cin, Pin, lo, hi = Cin[keep], Pin[keep], lo[keep], hi[keep]
        pinv = (evecs[:, curved] / evals[curved]) @ evecs[:, curved].T
        HiF, Hif0, HiCt = pinv @ F, pinv @ f0, pinv @ Cin.T
        m = len(lo)
        self._moves_of = (-HiF, -Hif0, -HiCt)
        return dict(
            Gc=-Cin @ HiF + Pin, gc0=-Cin @ Hif0, S=Cin @ HiCt, lo=lo, hi=hi,
            Gu=-HiF[:Ni], gu0=-Hif0[:Ni], Tu=HiCt[:Ni].reshape(Ni, m),
            Gy=-Phi[:No] @ HiF + Psi[:No], gy0=-Phi[:No] @ Hif0, Ty=(Phi[:No] @ HiCt).reshape(No, m))

    def moves(self, theta, nu):
        """All input moves d* for parameter rows ``theta`` [B, n_theta] and multipliers ``nu`` [B, n_rows] (host;
        for inspection and tests -- the loop itself needs only du_0 and y_1)."""
        self._law                                    # (built on first use)
        Zt, z0, Zn = self._moves_of
        return (Zt @ theta.T + z0[:, None] + Zn @ nu.T).T

    # ------------------------------------------------------------------------------------------------------------
    # device evaluation
    # ------------------------------------------------------------------------------------------------------------
    def _device_law(self):
        import torch
        if self._dev is None:
            lib = _lib.load()
            dev = torch.device("cuda", torch.cuda.current_device())
            keep = {k: torch.as_tensor(numpy.ascontiguousarray(v, dtype=numpy.float64), device=dev) for k, v in self._law.items()}
            law = _Law()
            law.n_theta, law.n_u, law.n_y = self.Nx + self.Ni + self.No, self.Ni, self.No
            law.n_rows = len(self._law["lo"])
            law.max_sweeps, law.eps_abs, law.eps_rel = self._solver
            for k, t in keep.items():
                setattr(law, k, t.data_ptr() if t.numel() else None)
            self._dev = (lib, law, keep)
        self._dev[1].report_unsolved = 1 if self.report_unsolved else 0
        return self._dev

    @property
    def n_bounded_rows(self):
        """Bounded rows of the law (beyond 64 the fused loop kernels do not take it: gse_mpc_step / gse_loop_control do)."""
        return len(self._law["lo"])

    def step_device(self, theta, du0=None, y1=None, status=None):
        """theta: device tensor [B, Nx+Ni+No] fp64 -> (du0 [B, Ni], y1 [B, No], status [B] int32), all on the device,
        enqueued on the current stream without synchronising."""
        import torch
        lib, law, _ = self._device_law()
        B = theta.shape[0]
        assert theta.dtype == torch.float64 and theta.is_contiguous() and theta.shape[1] == law.n_theta
        du0 = torch.empty((B, self.Ni), dtype=torch.float64, device=theta.device) if du0 is None else du0
        y1 = torch.empty((B, self.No), dtype=torch.float64, device=theta.device) if y1 is None else y1
        status = torch.empty((B,), dtype=torch.int32, device=theta.device) if status is None else status
        _lib.check(lib.gse_mpc_step(ctypes.byref(law), theta.data_ptr(), B, du0.data_ptr(), y1.data_ptr(),
                                    status.data_ptr(), torch.cuda.current_stream().cuda_stream))
        return du0, y1, status

    def loop_state(self, B, device):
        """Device-side bookkeeping of ``loop_control`` for B rollouts: (y_predicted [B, No], known [B], status [B])."""
        import torch
        return (torch.zeros((B, self.No), dtype=torch.float64, device=device),
                torch.zeros((B,), dtype=torch.int32, device=device), torch.zeros((B,), dtype=torch.int32, device=device))

    def loop_glue(self, u_fallback, full_dims=(5, 2, 5)):
        """struct gse_loop_glue for this controller's linear model (cached)."""
        lm = self.model
        key = (tuple(full_dims), tuple(float(v) for v in u_fallback))
        if getattr(self, "_glue_key", None) != key:             # the glue only depends on the model: build it once
            glue = _Glue()
            glue.n_state_full, glue.n_input_full, glue.n_output_full = full_dims
            for name, idx in (("states", lm.states), ("inputs", lm.inputs), ("outputs", lm.outputs)):
                for k, v in enumerate(idx):
                    getattr(glue, name)[k] = int(v)
            for name, vec in (("x_bar", lm.x_bar), ("u_bar", lm.u_bar), ("y_bar", lm.y_bar), ("u_fallback", u_fallback)):
                for k, v in enumerate(vec):
                    getattr(glue, name)[k] = float(v)
            self._glue_key, self._glue = key, glue
        return self._glue

    def loop_control(self, moments, y_meas, u, state, u_fallback, full_dims=(5, 2, 5), theta_out=None):
        """The controller call of the closed loop on the device (``gse_loop_control``): estimator moments [B, 26],
        measured outputs [B, n_out] and the current inputs ``u`` [B, n_in] (updated in place) are device tensors;
        ``state`` comes from ``loop_state``.  Nothing is synchronised."""
        import torch
        lib, law, _ = self._device_law()
        glue = self.loop_glue(u_fallback, full_dims)
        y_pred, known, status = state
        _lib.check(lib.gse_loop_control(ctypes.byref(law), ctypes.byref(glue), moments.data_ptr(), y_meas.data_ptr(),
                                        u.data_ptr(), y_pred.data_ptr(), known.data_ptr(), status.data_ptr(),
                                        theta_out.data_ptr() if theta_out is not None else None, u.shape[0],
                                        torch.cuda.current_stream().cuda_stream))

    def step_batch(self, x0, um1, y0):
        """B rollouts at once.  x0 [B, Nx], um1 [B, Ni], y0 [B, No] in deviation variables (host arrays).  Returns
        (ctrl [B, Ni], ok [B] bool); ``y_predicted`` ([B, No]) advances for the rollouts that solved, as the
        reference's does when ``step`` returns normally (controller.py:270-279)."""
        import torch
        clip = lambda v: numpy.maximum(numpy.minimum(numpy.asarray(v, dtype=float), 1e10), -1e10)   # controller.py:238-241
        x0, um1, y0 = clip(x0), clip(um1), clip(y0)
        if self.y_predicted is None:
            predicted, known = numpy.zeros_like(y0), numpy.zeros(len(y0), dtype=bool)
        else:
            predicted, known = self.y_predicted, self._known
        bias = numpy.where(known[:, None], y0 - predicted, 0.0)      # no bias before the first solved step (:248-251)
        theta = torch.as_tensor(numpy.ascontiguousarray(numpy.hstack([x0, um1, bias])), device="cuda")
        du0, y1, status = self.step_device(theta)
        ok = status.cpu().numpy() == 0
        ctrl = du0.cpu().numpy() + um1
        self.y_predicted = numpy.where(ok[:, None], y1.cpu().numpy() - bias, predicted)
        self._known = known | ok
        return ctrl, ok

    def step(self, x0, um1, y0):
        """The MPC control input for one system (controller.py:236-279): raises ValueError when the QP has no
        solution the solver can reach, exactly where the reference raises on a non-solved OSQP status."""
        keep = self.y_predicted
        if keep is not None:
            self.y_predicted, self._known = numpy.atleast_2d(keep), numpy.ones(1, dtype=bool)
        try:
            ctrl, ok = self.step_batch(numpy.atleast_2d(x0), numpy.atleast_2d(um1), numpy.atleast_2d(y0))
        except Exception:
            self.y_predicted = keep
            raise
        if not ok[0]:
            self.y_predicted = keep
            raise ValueError("the MPC's QP was not solved (bounds infeasible or sweeps exhausted)")
        self.y_predicted = self.y_predicted[0]
        return ctrl[0]
