// gse_mpc.hip -- the controller step of the closed loop (controller.py:236-279) as an explicit law, B rollouts per launch.
//
// The reference re-solves a sparse QP with OSQP on every control tick.  Of that QP only theta = (x0, u_{-1}, bias)
// changes between ticks, so gpu_se_amd/controller.py condenses it once on the host: the equality constraints (the
// linear model) are eliminated, the dense Hessian in the input moves is factorised, and what is left per tick is
//
//     c0 = Gc theta + gc0            value of each BOUNDED row at the unconstrained optimum
//     dual problem in nu (one multiplier per bounded row):   c(nu) = c0 - S nu,   lo <= c <= hi,
//                                    nu_i > 0 only with c_i = hi_i,  nu_i < 0 only with c_i = lo_i
//     du0 = Gu theta + gu0 - Tu nu   y1 = Gy theta + gy0 - Ty nu
//
// The dual is solved by Hildreth's procedure (exact coordinate ascent, Gauss-Seidel order): for row i, with the other
// multipliers fixed, r = c_i + S_ii nu_i is the row's value without its own multiplier; nu_i becomes (r - hi_i) / S_ii
// if r > hi_i, (r - lo_i) / S_ii if r < lo_i, else 0.  One wavefront per rollout, lane i owns row i: c is kept up to
// date incrementally (a changed nu_i moves every c_j by -S_ji delta: one broadcast and one FMA per lane), so a
// coordinate step costs a handful of instructions and no reduction.  With the bioreactor's wiring there are two
// bounded rows (u0 >= 0 for both inputs, sim_base.py:82-85) and the sweeps end after two or three passes.
//
// This is latency-bound scalar-ish work (a few hundred bytes per rollout); what matters is that it runs on the device,
// between the moments kernel and the next predict, so that the closed loop never waits for the host.
#include "gse_host.h"

namespace gse {

struct MpcLawDev {
    int32_t n_theta, n_u, n_y, n_rows, max_sweeps;
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

__global__ __launch_bounds__(64 * kMpcWavesPerBlock) void mpc_step_kernel(const MpcLawDev law,
                                                                           const double* __restrict__ theta, int64_t B,
                                                                           double* __restrict__ du0,
                                                                           double* __restrict__ y1,
                                                                           int32_t* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * kMpcWavesPerBlock + (threadIdx.x >> 6);
    if (b >= B) return;                                   // whole wavefronts leave; no barrier below
    const int m = law.n_rows, nt = law.n_theta;
    const double* th = theta + b * nt;

    // this lane's row at the unconstrained optimum
    const bool row = lane < m;
    double c = 0.0, lo = -INFINITY, hi = INFINITY, sii = 1.0;
    if (row) {
        c = law.gc0[lane];
        for (int k = 0; k < nt; ++k) c += law.Gc[lane * nt + k] * th[k];
        lo = law.lo[lane];
        hi = law.hi[lane];
        sii = law.S[lane * m + lane];
    }
    const double c0 = c;
    double nu = 0.0;
    int converged = (m == 0) ? 1 : 0;
    for (int sweep = 0; sweep < law.max_sweeps && !converged; ++sweep) {
        for (int i = 0; i < m; ++i) {
            // lane i decides its multiplier; everybody applies the change
            double delta = 0.0;
            if (lane == i && sii > 0.0) {
                const double r = c + sii * nu;
                const double want = (r > hi) ? (r - hi) / sii : ((r < lo) ? (r - lo) / sii : 0.0);
                delta = want - nu;
                nu = want;
            }
            delta = __shfl(delta, i, 64);
            if (row && delta != 0.0) c -= law.S[lane * m + i] * delta;        // S symmetric: column i = row i
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
    // outputs: lane k < n_u forms du0[k], lane 32 + k forms y1[k]; the multipliers come round by shuffles
    const bool for_u = lane < law.n_u, for_y = lane >= 32 && lane - 32 < law.n_y;
    const int k = for_u ? lane : lane - 32;
    double acc = 0.0;
    if (for_u || for_y) {
        const double* G = for_u ? law.Gu : law.Gy;
        acc = for_u ? law.gu0[k] : law.gy0[k];
        for (int q = 0; q < nt; ++q) acc += G[k * nt + q] * th[q];
    }
    for (int i = 0; i < m; ++i) {
        const double nui = __shfl(nu, i, 64);
        if ((for_u || for_y) && nui != 0.0) acc -= (for_u ? law.Tu : law.Ty)[k * m + i] * nui;
    }
    if (for_u) du0[b * law.n_u + k] = acc;
    if (for_y) y1[b * law.n_y + k] = acc;
    if (lane == 0) status[b] = converged ? 0 : 1;
}

}  // namespace gse

using namespace gse;

extern "C" int gse_mpc_step(const gse_mpc_law* law, const double* theta, int64_t B, double* du0, double* y1,
                            int32_t* status, void* stream) {
    GSE_REQUIRE(law && theta && du0 && y1 && status && B >= 1, "gse_mpc_step: bad argument");
    GSE_REQUIRE(law->n_rows >= 0 && law->n_rows <= GSE_MPC_MAX_ROWS,
                "gse_mpc_step: %d bounded rows; the device solver takes at most %d", law->n_rows, GSE_MPC_MAX_ROWS);
    GSE_REQUIRE(law->n_theta >= 1 && law->n_theta <= GSE_MPC_MAX_THETA && law->n_u >= 1 && law->n_u <= GSE_MPC_MAX_OUT &&
                    law->n_y >= 1 && law->n_y <= GSE_MPC_MAX_OUT,
                "gse_mpc_step: law dimensions (n_theta %d, n_u %d, n_y %d) out of range", law->n_theta, law->n_u, law->n_y);
    GSE_REQUIRE(law->max_sweeps >= 1 && law->eps_abs >= 0.0 && law->eps_rel >= 0.0, "gse_mpc_step: bad stopping rule");
    GSE_REQUIRE(law->Gu && law->gu0 && law->Gy && law->gy0, "gse_mpc_step: null law matrix");
    GSE_REQUIRE(law->n_rows == 0 || (law->Gc && law->gc0 && law->S && law->lo && law->hi && law->Tu && law->Ty),
                "gse_mpc_step: null law matrix");
    MpcLawDev d;
    d.n_theta = law->n_theta; d.n_u = law->n_u; d.n_y = law->n_y; d.n_rows = law->n_rows; d.max_sweeps = law->max_sweeps;
    d.eps_abs = law->eps_abs; d.eps_rel = law->eps_rel;
    d.Gc = law->Gc; d.gc0 = law->gc0; d.S = law->S; d.lo = law->lo; d.hi = law->hi;
    d.Gu = law->Gu; d.gu0 = law->gu0; d.Tu = law->Tu; d.Gy = law->Gy; d.gy0 = law->gy0; d.Ty = law->Ty;
    const unsigned blocks = (unsigned)((B + kMpcWavesPerBlock - 1) / kMpcWavesPerBlock);
    hipLaunchKernelGGL(mpc_step_kernel, dim3(blocks), dim3(64 * kMpcWavesPerBlock), 0, (hipStream_t)stream, d, theta, B, du0,
                       y1, status);
    GSE_LAUNCH_CHECK("mpc_step_kernel");
    return GSE_OK;
}
