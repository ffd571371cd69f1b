// gse_loop_device.h -- device functions of the closed loop shared by its kernels (gse_mpc.hip: one launch per piece;
// gse_loop.hip: the whole loop of a small Gaussian-sum filter in ONE persistent launch): the explicit MPC law, the
// controller call, the plant.
#pragma once

#include "gse_host.h"

namespace gse {

struct MpcLawDev {
    int32_t n_theta, n_u, n_y, n_rows, max_sweeps, report_unsolved;
    double eps_abs, eps_rel;
    const double* __restrict__ Gc;
    const double* __restrict__ gc0;
    const double* __restrict__ S;
    const double* __restrict__ lo;
    const double* __restrict__ hi;
    const double* __restrict__ Gu;
    const double* __restrict__ gu0;
    const double* __restrict__ Tu;
    const double* __restrict__ Gy;
    const double* __restrict__ gy0;
    const double* __restrict__ Ty;
};

constexpr int kMpcWavesPerBlock = 4;

// Values a long-running loop should recompute per iteration rather than keep: after these the compiler cannot tell that they
// are the loop's invariants, so nothing derived from them (masks of unrolled bounds checks, addresses) is hoisted and kept
// live -- in scalar registers that spill into vector lanes, then in scratch -- across every phase of the iteration.
__device__ __forceinline__ int per_iteration_uniform(int v) { asm volatile("" : "+s"(v)); return v; }
__device__ __forceinline__ int per_iteration(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ void per_iteration_uniform(MpcLawDev& l) {
    asm volatile("" : "+s"(l.n_theta), "+s"(l.n_u), "+s"(l.n_y), "+s"(l.n_rows), "+s"(l.max_sweeps));
    asm volatile("" : "+s"(l.Gc), "+s"(l.gc0), "+s"(l.S), "+s"(l.lo), "+s"(l.hi));
    asm volatile("" : "+s"(l.Gu), "+s"(l.gu0), "+s"(l.Tu), "+s"(l.Gy), "+s"(l.gy0), "+s"(l.Ty));
}

// More bounded rows than a wavefront has lanes (controller.py:180-207 bounds every output over the whole horizon: P x No rows):
// the dual is solved over a WORKING SET of at most 64 rows, one per lane, grown one row at a time --
//   1. with the multipliers of the working set, every row's value c_i = c0_i - sum_j S[i][W_j] nu_j is formed (each lane takes
//      every 64th row; c0_i = gc0_i + Gc_i theta) and the row that violates its bounds the most is found;
//   2. none does (beyond the stopping tolerance): the multipliers are optimal -- rows outside the set are feasible with zero
//      multipliers, rows inside satisfy the KKT conditions by 3 -- done;
//   3. otherwise that row joins the set (lane |W|) and the dual coordinate sweeps (Hildreth, as below) run over the set from
//      the multipliers they had, until the stopping rule of the small solver holds;
// and so on: every pass adds a row the previous optimum violated, so the dual objective rises strictly and no set repeats.
// Every pass first sheds the rows whose multipliers have returned to zero (round 5; ADVICE round 4: they used to keep their
// lanes, so a QP with fewer than 64 rows active at the optimum could still fill the set on the way there and fall to the
// fallback input).
// Not converged (status 1, the caller's fallback): more than 64 rows active together, or sweeps / passes exhausted.  Nothing is kept
// in LDS (the callers have none to spare): the rows are re-formed from the law's matrices in every pass -- a few thousand
// independent loads per rollout and pass, served by the L2.
__device__ __noinline__ int mpc_solve_large(const MpcLawDev& law, double theta_lane, int lane, double& out) {
    const int m = law.n_rows, nt = law.n_theta;
    int nW = 0;
    int myrow = -1;                                   // lane j < nW: the row it holds
    double nu = 0.0, c = 0.0, c0 = 0.0, lo = -INFINITY, hi = INFINITY, sii = 1.0;
    int converged = 0;
    double th[GSE_MPC_MAX_THETA];
#pragma unroll
    for (int k = 0; k < GSE_MPC_MAX_THETA; ++k) th[k] = __shfl(theta_lane, k, 64);
    for (int pass = 0; pass <= 4 * 64; ++pass) {          // (every pass adds a row the last optimum violated; evictions make room)
        // 1. the most violated row outside the working set
        double worst = 0.0, worst_c = 0.0, worst_c0 = 0.0;
        int worst_row = -1;
        for (int base = 0; base < m; base += 64) {        // (every lane runs every trip: the shuffles below need them all)
            const int i = min(base + lane, m - 1);
            const bool valid = base + lane < m;
            double ci0 = law.gc0[i];
#pragma unroll
            for (int k = 0; k < GSE_MPC_MAX_THETA; ++k)
                if (k < nt) ci0 += law.Gc[i * nt + k] * th[k];
            double ci = ci0;
            bool inside = !valid;
            for (int j = 0; j < nW; ++j) {
                const int wj = __shfl(myrow, j, 64);
                const double nj = __shfl(nu, j, 64);
                inside = inside || (wj == i);
                if (nj != 0.0) ci -= law.S[(int64_t)i * m + wj] * nj;
            }
            if (inside) continue;
            const double l = law.lo[i], h = law.hi[i];
            const double over = ci - h, under = l - ci;
            const double viol = fmax(over, under);
            const double tol = law.eps_abs + law.eps_rel * (fabs(ci0) + (over > under ? fabs(h) : fabs(l)));
            if (viol > tol && viol - tol > worst) { worst = viol - tol; worst_row = i; worst_c = ci; worst_c0 = ci0; }
        }
        double best = worst;
        int best_lane = lane;
#pragma unroll
        for (int s2 = 32; s2 >= 1; s2 >>= 1) {
            const double ob = __shfl_xor(best, s2, 64);
            const int ol = __shfl_xor(best_lane, s2, 64);
            if (ob > best || (ob == best && ol < best_lane)) { best = ob; best_lane = ol; }
        }
        if (!(best > 0.0)) { converged = 1; break; }
        {
            // Before a row joins (round 5): rows whose multiplier has returned to zero and that sit strictly inside their bounds
            // only occupy a lane -- they joined on the way and the optimum over the set has left them.  They leave (rows
            // outside the set again, with zero multipliers: feasible, and step 1 watches them) and the ranks close.  Shedding
            // them in EVERY pass, not only when the set is full, also keeps the set near the number of active rows -- a set
            // with many slack rows more than the problem has dimensions is rank-deficient, and the coordinate sweeps crawl
            // on it (a 40-dimensional projection with 52 rows in the set: 20 000 sweeps did not finish one pass).
            const bool keep = lane < nW && !(nu == 0.0 && c > lo && c < hi);
            const unsigned long long km = __ballot(keep);
            const int kept = __popcll(km);
            if (kept < nW) {
                unsigned long long left = km;
                int src = -1;
                for (int t = 0; t <= lane; ++t) {             // the lane-th kept lane (none for lane >= kept)
                    src = __ffsll((long long)left) - 1;
                    left &= left - 1;
                }
                const int from = src < 0 ? lane : src;
                myrow = __shfl(myrow, from, 64);
                nu = __shfl(nu, from, 64); c = __shfl(c, from, 64); c0 = __shfl(c0, from, 64);
                lo = __shfl(lo, from, 64); hi = __shfl(hi, from, 64); sii = __shfl(sii, from, 64);
                if (lane >= kept) { myrow = -1; nu = 0.0; c = 0.0; c0 = 0.0; lo = -INFINITY; hi = INFINITY; sii = 1.0; }
                nW = kept;
            }
        }
        if (nW == 64) break;                              // 64 rows, every one active or on a bound: more than this solver holds
        // 2. that row joins the working set on lane nW
        const int r = __shfl(worst_row, best_lane, 64);
        const double rc = __shfl(worst_c, best_lane, 64), rc0 = __shfl(worst_c0, best_lane, 64);
        if (lane == nW) {
            myrow = r; nu = 0.0; c = rc; c0 = rc0;
            lo = law.lo[r]; hi = law.hi[r]; sii = law.S[(int64_t)r * m + r];
        }
        ++nW;
        // 3. dual coordinate sweeps over the set (the small solver's, with S gathered through the set's row numbers)
        const bool row = lane < nW;
        int inner = 0;
        for (int sweep = 0; sweep < law.max_sweeps && !inner; ++sweep) {
            for (int i = 0; i < nW; ++i) {
                const int wi = __shfl(myrow, i, 64);
                double delta = 0.0;
                if (lane == i && sii > 0.0) {
                    const double rr = c + sii * nu;
                    const double want = (rr > hi) ? (rr - hi) / sii : ((rr < lo) ? (rr - lo) / sii : 0.0);
                    delta = want - nu;
                    nu = want;
                }
                delta = __shfl(delta, i, 64);
                if (row && delta != 0.0) c -= law.S[(int64_t)myrow * m + wi] * delta;
            }
            double res = 0.0, scale = fabs(c0);
            if (row) {
                if (nu > 0.0) { res = fabs(c - hi); scale += fabs(hi); }
                else if (nu < 0.0) { res = fabs(c - lo); scale += fabs(lo); }
                else {
                    res = fmax(fmax(c - hi, lo - c), 0.0);
                    if (c > hi) scale += fabs(hi);
                    if (c < lo) scale += fabs(lo);
                }
            }
            const bool ok = !row || (res <= law.eps_abs + law.eps_rel * scale);
            inner = __all(ok) ? 1 : 0;
        }
        if (!inner) break;                            // sweeps exhausted
    }
    // outputs: lane k < n_u forms du0[k], lane 32 + k forms y1[k]
    const bool for_u = lane < law.n_u, for_y = lane >= 32 && lane - 32 < law.n_y;
    const bool mine = for_u || for_y;
    const int k = for_u ? lane : lane - 32;
    const double* G = for_u ? law.Gu : law.Gy;
    double acc = 0.0;
    if (mine) {
        acc = for_u ? law.gu0[k] : law.gy0[k];
        for (int q = 0; q < nt; ++q) acc += G[k * nt + q] * th[q];
    }
    for (int j = 0; j < nW; ++j) {
        const int wj = __shfl(myrow, j, 64);
        const double nj = __shfl(nu, j, 64);
        if (mine && nj != 0.0) acc -= (for_u ? law.Tu : law.Ty)[(int64_t)k * m + wj] * nj;
    }
    out = acc;
    return law.report_unsolved ? 0 : converged;
}

// The law for one rollout, by one wavefront.  theta_lane: lane k < n_theta holds theta[k].  On return lane k < n_u holds
// du0[k] in out, lane 32 + k (k < n_y) holds y1[k]; the return value (uniform) is 1 if the sweeps converged.
template <bool kManyRows = false>     // kManyRows: the caller takes laws with more than 64 bounded rows (mpc_solve_large)
__device__ __forceinline__ int mpc_solve(const MpcLawDev& law, double theta_lane, int lane, double& out) {
    if constexpr (kManyRows) {
        if (law.n_rows > 64) return mpc_solve_large(law, theta_lane, lane, out);
    }
    const int m = law.n_rows, nt = law.n_theta;
    // this lane's row at the unconstrained optimum
    const bool row = lane < m;
    double c = 0.0, lo = -INFINITY, hi = INFINITY, sii = 1.0;
    if (row) {
        c = law.gc0[lane];
        lo = law.lo[lane];
        hi = law.hi[lane];
        sii = law.S[lane * m + lane];
    }
    {   // (all coefficient loads first, in flight together: this kernel is nothing but dependent latencies)
        double gk[GSE_MPC_MAX_THETA];
#pragma unroll
        for (int k = 0; k < GSE_MPC_MAX_THETA; ++k) gk[k] = (row && k < nt) ? law.Gc[lane * nt + k] : 0.0;
#pragma unroll
        for (int k = 0; k < GSE_MPC_MAX_THETA; ++k) {
            const double tk = __shfl(theta_lane, k, 64);
            if (k < nt) c += gk[k] * tk;
        }
    }
    const double c0 = c;
    double nu = 0.0;
    int converged = (m == 0) ? 1 : 0;
    // a few bounded rows (the usual case: two): this lane's row of S stays in registers, no load inside the sweeps
    constexpr int kFewRows = 8;
    const bool few = m <= kFewRows;
    double srow[kFewRows];
#pragma unroll
    for (int i = 0; i < kFewRows; ++i) srow[i] = (few && row && i < m) ? law.S[lane * m + i] : 0.0;
    auto coordinate = [&](int i, double s_li) {
        // lane i decides its multiplier; everybody applies the change
        double delta = 0.0;
        if (lane == i && sii > 0.0) {
            const double r = c + sii * nu;
            const double want = (r > hi) ? (r - hi) / sii : ((r < lo) ? (r - lo) / sii : 0.0);
            delta = want - nu;
            nu = want;
        }
        delta = __shfl(delta, i, 64);
        if (row && delta != 0.0) c -= s_li * delta;                           // S symmetric: column i = row i
    };
    // One or two bounded rows (every wiring of sim_base.get_parts: u_0 >= 0 for both inputs): the dual has at most nine
    // candidate active sets -- each row free, on its lower or on its upper bound -- and each is a closed form; lane L tries
    // candidate (L % 3, L / 3), the first that satisfies the KKT conditions (feasible free rows, multipliers of the right
    // sign) IS the optimum (convex QP).  No sweeps: the coordinate ascent below needs ~40 of them to reach 1e-12 when the two
    // rows are coupled, 12 of the controller kernel's 17 us.  Nothing valid (contradictory bounds, a row without curvature)
    // falls through to the sweeps, which then report what they always did.
    if (m >= 1 && m <= 2) {
        const double c0_0 = __shfl(c0, 0, 64), c0_1 = (m > 1) ? __shfl(c0, 1, 64) : 0.0;
        const double lo0 = __shfl(lo, 0, 64), hi0 = __shfl(hi, 0, 64);
        const double lo1 = (m > 1) ? __shfl(lo, 1, 64) : -INFINITY, hi1 = (m > 1) ? __shfl(hi, 1, 64) : INFINITY;
        const double s00 = __shfl(srow[0], 0, 64), s01 = (m > 1) ? __shfl(srow[1], 0, 64) : 0.0;
        const double s10 = (m > 1) ? __shfl(srow[0], 1, 64) : 0.0, s11 = (m > 1) ? __shfl(srow[1], 1, 64) : 1.0;
        const int a0 = lane % 3, a1 = (lane / 3) % 3;                  // 0 free, 1 on lo, 2 on hi
        const double b0 = (a0 == 1) ? lo0 : hi0, b1 = (a1 == 1) ? lo1 : hi1;
        double n0 = 0.0, n1 = 0.0;
        if (a0 && !a1) n0 = (c0_0 - b0) / s00;
        if (!a0 && a1) n1 = (c0_1 - b1) / s11;
        if (a0 && a1) {
            const double det = s00 * s11 - s01 * s10, r0 = c0_0 - b0, r1 = c0_1 - b1;
            n0 = (s11 * r0 - s01 * r1) / det;
            n1 = (s00 * r1 - s10 * r0) / det;
        }
        const double c_0 = a0 ? b0 : c0_0 - s01 * n1, c_1 = a1 ? b1 : c0_1 - s10 * n0;
        const double t0 = law.eps_abs + law.eps_rel * (fabs(c0_0) + (c_0 > hi0 ? fabs(hi0) : (c_0 < lo0 ? fabs(lo0) : 0.0)));
        const double t1 = law.eps_abs + law.eps_rel * (fabs(c0_1) + (c_1 > hi1 ? fabs(hi1) : (c_1 < lo1 ? fabs(lo1) : 0.0)));
        bool valid = lane < (m > 1 ? 9 : 3) && (s00 > 0.0) && (m < 2 || s11 > 0.0);
        valid = valid && ((a0 == 0) ? (c_0 >= lo0 - t0 && c_0 <= hi0 + t0) : ((a0 == 1) ? (n0 < 0.0) : (n0 > 0.0)));
        valid = valid && ((a1 == 0) ? (c_1 >= lo1 - t1 && c_1 <= hi1 + t1) : ((a1 == 1) ? (n1 < 0.0) : (n1 > 0.0)));
        const unsigned long long found = __ballot(valid);
        if (found) {
            const int src = __ffsll((long long)found) - 1;
            const double v0 = __shfl(n0, src, 64), v1 = __shfl(n1, src, 64);
            const double w0 = __shfl(c_0, src, 64), w1 = __shfl(c_1, src, 64);
            nu = (lane == 0) ? v0 : ((lane == 1 && m > 1) ? v1 : 0.0);
            if (row) c = (lane == 0) ? w0 : w1;
            converged = 1;
        }
    }
    for (int sweep = 0; sweep < law.max_sweeps && !converged; ++sweep) {
        if (few) {
#pragma unroll
            for (int i = 0; i < kFewRows; ++i)
                if (i < m) coordinate(i, srow[i]);
        } else {
            for (int i = 0; i < m; ++i) coordinate(i, row ? law.S[lane * m + i] : 0.0);
        }
        // residual of this lane's row: distance to the bound it is supposed to sit on, or its violation
        double res = 0.0, scale = fabs(c0);
        if (row) {
            if (nu > 0.0) { res = fabs(c - hi); scale += fabs(hi); }
            else if (nu < 0.0) { res = fabs(c - lo); scale += fabs(lo); }
            else {
                res = fmax(fmax(c - hi, lo - c), 0.0);
                if (c > hi) scale += fabs(hi);
                if (c < lo) scale += fabs(lo);
            }
        }
        const bool ok = !row || (res <= law.eps_abs + law.eps_rel * scale);
        converged = __all(ok) ? 1 : 0;
    }
    // outputs: lane k < n_u forms du0[k], lane 32 + k forms y1[k]; theta and the multipliers come round by shuffles
    const bool for_u = lane < law.n_u, for_y = lane >= 32 && lane - 32 < law.n_y;
    const bool mine = for_u || for_y;
    const int k = for_u ? lane : lane - 32;
    const double* G = for_u ? law.Gu : law.Gy;
    double acc = 0.0;
    if (mine) acc = for_u ? law.gu0[k] : law.gy0[k];
    {
        double gq[GSE_MPC_MAX_THETA];
#pragma unroll
        for (int q = 0; q < GSE_MPC_MAX_THETA; ++q) gq[q] = (mine && q < nt) ? G[k * nt + q] : 0.0;
#pragma unroll
        for (int q = 0; q < GSE_MPC_MAX_THETA; ++q) {
            const double tq = __shfl(theta_lane, q, 64);
            if (q < nt) acc += gq[q] * tq;
        }
    }
    for (int i = 0; i < m; ++i) {
        const double nui = __shfl(nu, i, 64);
        if (mine && nui != 0.0) acc -= (for_u ? law.Tu : law.Ty)[k * m + i] * nui;
    }
    out = acc;
    return law.report_unsolved ? 0 : converged;     // (fault injection: the caller's fallback branch, include/gse.h)
}

struct LoopGlue {
    int32_t n_state_full, n_input_full, n_output_full, pad;
    int32_t states[GSE_MPC_MAX_OUT], inputs[GSE_MPC_MAX_OUT], outputs[GSE_MPC_MAX_OUT];
    double x_bar[GSE_MPC_MAX_OUT], u_bar[GSE_MPC_MAX_OUT], y_bar[GSE_MPC_MAX_OUT], u_fallback[GSE_MPC_MAX_OUT];
};

// The controller call of the closed loop for rollout b, by ONE wavefront (lane = its lane): the filter's moments -> point
// estimate -> deviation variables -> MPC -> next input, bias bookkeeping included (sim_base.py:253-274).
template <bool kManyRows = false>
__device__ __forceinline__ void loop_control_wave(const MpcLawDev& law, const LoopGlue& g, const double* __restrict__ moments,
                                                  const double* __restrict__ y_meas, double* __restrict__ u,
                                                  double* __restrict__ y_pred, int32_t* __restrict__ known,
                                                  int32_t* __restrict__ status, double* __restrict__ theta_out, int64_t b,
                                                  int lane) {
    const int nx = law.n_theta - law.n_u - law.n_y, ni = law.n_u, no = law.n_y;
    // theta = (x0, u_-1, bias) in deviation variables, clipped as controller.py:238-241
    double th = 0.0, bias = 0.0, um1 = 0.0;
    if (lane < nx) {
        // point estimate = weights @ particles, NOT normalised (particle.py:318-320): S1 + pivot * S0
        const double* m = moments + b * 26;
        const int sidx = g.states[lane];
        th = (m[1 + sidx] + m[21 + sidx] * m[0]) - g.x_bar[lane];
    } else if (lane < nx + ni) {
        const int i = lane - nx;
        um1 = u[b * g.n_input_full + g.inputs[i]] - g.u_bar[i];
        th = um1;
    } else if (lane < nx + ni + no) {
        const int o = lane - nx - ni;
        const double ydev = y_meas[b * g.n_output_full + g.outputs[o]] - g.y_bar[o];
        const double yd = fmax(fmin(ydev, 1e10), -1e10);
        bias = known[b] ? yd - y_pred[b * no + o] : 0.0;          // no bias before the first solved step (:248-251)
        th = bias;
    }
    if (lane < nx + ni) th = fmax(fmin(th, 1e10), -1e10);
    if (theta_out && lane < law.n_theta) theta_out[b * law.n_theta + lane] = th;
    double out;
    const int converged = mpc_solve<kManyRows>(law, th, lane, out);
    // next input: the first move on top of u_-1 (controller.py:272-275); when the QP is not solved the reference puts
    // [0.06, 0.2] where the DEVIATION input goes (sim_base.py:271-274), the operating point is added either way
    const double um1_i = __shfl(um1, nx + (lane < ni ? lane : 0), 64);
    if (lane < ni) u[b * g.n_input_full + g.inputs[lane]] = (converged ? out + um1_i : g.u_fallback[lane]) + g.u_bar[lane];
    const double bias_o = __shfl(bias, nx + ni + (lane >= 32 && lane - 32 < no ? lane - 32 : 0), 64);
    if (converged && lane >= 32 && lane - 32 < no) y_pred[b * no + lane - 32] = out - bias_o;
    if (lane == 0) {
        if (converged) known[b] = 1;
        status[b] = converged ? 0 : 1;
    }
}

struct PlantArgs {
    DrawParams state_noise, meas_noise;
    RngKey rng_state, rng_meas;
    double dt;
};

// The plant of the closed loop for rollout b, by ONE thread (sim_base.py:278-287, BioreactorModel.py:95-122): Euler step of the
// fp64 model, clamp, state noise, outputs, measurement noise.  dp: the two draws' parameters in LDS (state, measurement).
// b indexes the arrays, elem is the rollout's random-number element (the same unless the caller keeps its rollout's rows at 0).
__device__ __forceinline__ void plant_step_one(double* __restrict__ X, const double* __restrict__ u, int64_t b, int64_t elem,
                                               const DrawParams* dp, const RngKey& rng_state, const RngKey& rng_meas, double dt,
                                               double* __restrict__ y, double* __restrict__ y_meas, double* __restrict__ z,
                                               double* __restrict__ log_x, double* __restrict__ log_u,
                                               const float* __restrict__ inj_state, const float* __restrict__ inj_meas) {
    double x[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) x[k] = X[b * 5 + k];
    const double fg = u[b * 2], fm = u[b * 2 + 1];
    double dx[5];
    bioreactor_increment(x, fg, fm, 1.0, dx);              // DEs(): the derivative (homeostatic_DEs with dt = 1)
#pragma unroll
    for (int k = 0; k < 5; ++k) x[k] = x[k] + dx[k] * dt;
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = fmax(x[k], 0.0);    // BioreactorModel.py:108
    uint32_t wd[8];
    float e[5];
    if (inj_state) {                                       // parity mode: the draw is an input (gse_plant_step_injected)
#pragma unroll
        for (int k = 0; k < 5; ++k) e[k] = inj_state[b * 5 + k];
    } else {
        rng_words8(rng_state, (uint64_t)elem, 0u, wd);
        mixture_draw5(dp[0], wd, e);                       // state_pdf.draw(): fp32 values added to the fp64 state
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) x[k] += (double)e[k];
    const double mm[5] = {180.0, 24.6, 116.0, 46.0, 1.0};   // BioreactorModel.py:118-120
    double out[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) out[k] = x[k] * mm[k];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        X[b * 5 + k] = x[k];
        if (log_x) log_x[b * 5 + k] = x[k];
        y[b * 5 + k] = out[k];
    }
    if (inj_meas) {
        e[0] = inj_meas[b * 2];
        e[1] = inj_meas[b * 2 + 1];
    } else {
        rng_words8(rng_meas, (uint64_t)elem, 0u, wd);
        mixture_draw5(dp[1], wd, e);                       // measurement noise on the controlled outputs (Cg, Cfa)
    }
    out[0] += (double)e[0];
    out[2] += (double)e[1];
#pragma unroll
    for (int k = 0; k < 5; ++k) y_meas[b * 5 + k] = out[k];
    z[b * 2] = out[0];
    z[b * 2 + 1] = out[2];
    if (log_u) { log_u[b * 2] = fg; log_u[b * 2 + 1] = fm; }
}

#ifndef __HIPCC_RTC__
int pack_mpc_law(const char* fn, const gse_mpc_law* law, MpcLawDev& d, int max_rows = GSE_MPC_MAX_ROWS);   // (gse_mpc.hip)
int pack_loop_glue(const char* fn, const gse_loop_glue* glue, const MpcLawDev& d, LoopGlue& g);
#endif

}  // namespace gse
