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
// With one or two bounded rows the dual is solved in closed form (mpc_solve: nine candidate active sets, one per lane).
// Otherwise by Hildreth's procedure (exact coordinate ascent, Gauss-Seidel order): for row i, with the other
// multipliers fixed, r = c_i + S_ii nu_i is the row's value without its own multiplier; nu_i becomes (r - hi_i) / S_ii
// if r > hi_i, (r - lo_i) / S_ii if r < lo_i, else 0.  One wavefront per rollout, lane i owns row i: c is kept up to
// date incrementally (a changed nu_i moves every c_j by -S_ji delta: one broadcast and one FMA per lane), so a
// coordinate step costs a handful of instructions and no reduction.  With the bioreactor's wiring there are two
// bounded rows (u0 >= 0 for both inputs, sim_base.py:82-85) and the sweeps end after two or three passes.
//
// This is latency-bound scalar-ish work (a few hundred bytes per rollout); what matters is that it runs on the device,
// between the moments kernel and the next predict, so that the closed loop never waits for the host.
#include "gse_host.h"
#include "gse_loop_device.h"
#include "gse_merge_device.h"

namespace gse {

__global__ __launch_bounds__(64 * kMpcWavesPerBlock) void mpc_step_kernel(const MpcLawDev law,
                                                                           const double* __restrict__ theta, int64_t B,
                                                                           double* __restrict__ du0,
                                                                           double* __restrict__ y1,
                                                                           int32_t* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * kMpcWavesPerBlock + (threadIdx.x >> 6);
    if (b >= B) return;                                   // whole wavefronts leave; no barrier below
    const double th = lane < law.n_theta ? theta[b * law.n_theta + lane] : 0.0;
    double out;
    const int converged = mpc_solve<true>(law, th, lane, out);
    if (lane < law.n_u) du0[b * law.n_u + lane] = out;
    if (lane >= 32 && lane - 32 < law.n_y) y1[b * law.n_y + lane - 32] = out;
    if (lane == 0) status[b] = converged ? 0 : 1;
}

// ------------------------------------------------------------------------------------------------
// The controller call of the closed loop (sim_base.py:253-274) for B rollouts without leaving the device:
// the filter's moments -> point estimate -> deviation variables -> MPC -> next input, bias bookkeeping included.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * kMpcWavesPerBlock) void loop_control_kernel(
    const MpcLawDev law, const LoopGlue g, const double* __restrict__ moments, const double* __restrict__ y_meas,
    double* __restrict__ u, double* __restrict__ y_pred, int32_t* __restrict__ known, int32_t* __restrict__ status,
    double* __restrict__ theta_out, int64_t B) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * kMpcWavesPerBlock + (threadIdx.x >> 6);
    if (b >= B) return;
    loop_control_wave<true>(law, g, moments, y_meas, u, y_pred, known, status, theta_out, b, lane);
}

// ------------------------------------------------------------------------------------------------
// The plant of the closed loop (sim_base.py:278-287, BioreactorModel.py:95-122): Euler step of the fp64 model, clamp,
// state noise, outputs, measurement noise -- one thread per rollout.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void plant_step_kernel(double* __restrict__ X, const double* __restrict__ u, int64_t B,
                                                         const PlantArgs a, double* __restrict__ y,
                                                         double* __restrict__ y_meas, double* __restrict__ z,
                                                         double* __restrict__ log_x, double* __restrict__ log_u,
                                                         const float* __restrict__ inj_state,
                                                         const float* __restrict__ inj_meas) {
    __shared__ DrawParams dp[2];
    {
        const uint32_t* s0 = reinterpret_cast<const uint32_t*>(&a.state_noise);
        const uint32_t* s1 = reinterpret_cast<const uint32_t*>(&a.meas_noise);
        uint32_t* d0 = reinterpret_cast<uint32_t*>(&dp[0]);
        uint32_t* d1 = reinterpret_cast<uint32_t*>(&dp[1]);
        for (int i = threadIdx.x; i < (int)(sizeof(DrawParams) / 4); i += 256) { d0[i] = s0[i]; d1[i] = s1[i]; }
        __syncthreads();
    }
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    plant_step_one(X, u, b, b, dp, a.rng_state, a.rng_meas, a.dt, y, y_meas, z, log_x, log_u, inj_state, inj_meas);
}

// ------------------------------------------------------------------------------------------------
// The rest of a control tick of the resident closed loop in ONE launch (round 3): the reduction that finishes the estimator's
// moment partials, the controller call and the plant -- three launches of one workgroup / wavefront / thread per rollout each
// (4.7 + 8.0 + 4.6 us of a 192 us tick at 32 x 2^18 particles, mostly launch and drain), chained inside one workgroup per
// rollout.  The device functions of the three kernels, in their order: the same records bit for bit.
// ------------------------------------------------------------------------------------------------
struct LoopTailArgs {
    const double* partials;         // [B][kMomentBlocks][kMomentWidth]
    int nblk, width, pivot_at, pivot_count;
    const float* pivot_src;
    int64_t pivot_filter_stride, pivot_elem_stride;
    double* moments;                // [B][26]
    MpcLawDev law;
    LoopGlue glue;
    const double* y_meas;
    double *u, *y_pred;
    int32_t *known, *status;
    PlantArgs plant;
    double *X, *y, *y_meas_out, *z, *log_x, *log_u;
    const float *inj_state, *inj_meas;
};

__global__ __launch_bounds__(64 * kFinishGroups) void loop_tail_kernel(const LoopTailArgs a) {
    __shared__ double part[kFinishGroups][64];
    __shared__ DrawParams dp[2];
    const int b = blockIdx.x, tid = threadIdx.x;
    {
        const uint32_t* s0 = reinterpret_cast<const uint32_t*>(&a.plant.state_noise);
        const uint32_t* s1 = reinterpret_cast<const uint32_t*>(&a.plant.meas_noise);
        uint32_t* d0 = reinterpret_cast<uint32_t*>(&dp[0]);
        uint32_t* d1 = reinterpret_cast<uint32_t*>(&dp[1]);
        for (int i = tid; i < (int)(sizeof(DrawParams) / 4); i += 64 * kFinishGroups) { d0[i] = s0[i]; d1[i] = s1[i]; }
    }
    moments_finish_block(a.partials + (int64_t)b * kMomentBlocks * kMomentWidth, kMomentWidth, a.nblk, a.width, a.moments, 26,
                         a.pivot_src, a.pivot_filter_stride, a.pivot_elem_stride, a.pivot_at, a.pivot_count, b, part);
    __syncthreads();                                   // (the row, written to global memory by this workgroup)
    if (tid < 64) loop_control_wave(a.law, a.glue, a.moments, a.y_meas, a.u, a.y_pred, a.known, a.status, nullptr, b, tid);
    __syncthreads();                                   // (the new input)
    if (tid == 0)
        plant_step_one(a.X, a.u, b, b, dp, a.plant.rng_state, a.plant.rng_meas, a.plant.dt, a.y, a.y_meas_out, a.z, a.log_x,
                       a.log_u, a.inj_state, a.inj_meas);
}

// ------------------------------------------------------------------------------------------------
// The per-tick records of a closed loop, from the estimator's moments (rows of 26 doubles: S0, S1[5], S2 upper[15], pivot[5];
// gse_pf_moments / gse_gsukf_moments26): the point estimate weights @ particles = S1 + pivot S0 (filter/particle.py:318-320)
// and the largest singular value of the covariance S2 - S1 e^T - e S1^T + S0 e e^T, e = mean - pivot (:322-327) -- for a
// symmetric matrix the largest |eigenvalue|, by cyclic Jacobi rotations in fp64 (5 x 5: ten rotations a sweep, eight sweeps
// take the off-diagonal below 1e-30 of the norm).  One thread per row; a 499-tick loop of 32 rollouts is 16 000 rows, which
// the host used to work off with NumPy for longer than the loop itself ran on the device.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void moments_estimates_kernel(const double* __restrict__ moments, int64_t R,
                                                                double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= R) return;
    const double* m = moments + i * 26;
    const double s0 = m[0];
    double s1[5], pv[5], mean[5], e[5], A[5][5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        s1[k] = m[1 + k];
        pv[k] = m[21 + k];
        mean[k] = s1[k] + pv[k] * s0;
        e[k] = mean[k] - pv[k];
    }
    int c = 6;
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int q = r; q < 5; ++q) {
            const double v = m[c++] - s1[r] * e[q] - e[r] * s1[q] + s0 * e[r] * e[q];
            A[r][q] = v;
            A[q][r] = v;
        }
    for (int sweep = 0; sweep < 8; ++sweep) {
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = p + 1; q < 5; ++q) {
                const double apq = A[p][q];
                if (apq != 0.0) {
                    const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
#pragma unroll
                    for (int k = 0; k < 5; ++k) {                       // A <- J^T A J, J the rotation in the (p, q) plane
                        const double akp = A[k][p], akq = A[k][q];
                        A[k][p] = cs * akp - sn * akq;
                        A[k][q] = sn * akp + cs * akq;
                    }
#pragma unroll
                    for (int k = 0; k < 5; ++k) {
                        const double apk = A[p][k], aqk = A[q][k];
                        A[p][k] = cs * apk - sn * aqk;
                        A[q][k] = sn * apk + cs * aqk;
                    }
                }
            }
    }
    double big = 0.0;
#pragma unroll
    for (int k = 0; k < 5; ++k) big = fmax(big, fabs(A[k][k]));
#pragma unroll
    for (int k = 0; k < 5; ++k) out[i * 6 + k] = mean[k];
    out[i * 6 + 5] = big;
}

// ------------------------------------------------------------------------------------------------
// Posterior merge of a SHARDED ensemble on the device (gpu_se_amd/distributed.py).  merge_row_kernel: one shard's
// moments (26 doubles about its own pivot) -> its row of the merge buffer { log A, S0, sum w (x - p0) (5), upper triangle of
// sum w (x - p0)(x - p0)^T (15) } about the pivot p0 all ranks share (distributed.pack_merge_row).  merge_rows_kernel: the
// all-reduced rows of all shards -> the moments of the WHOLE ensemble in the 26-double layout every consumer here takes
// (S0 = 1, first and second moments about p0, p0): shard p enters with a_p = A_p S0_p / sum_q A_q S0_q, formed in the log
// domain (distributed.merge_rows / shard_shares).  One thread each: a few hundred flops between a moments kernel and the
// controller, so that a sharded closed loop never brings an estimate to the host.
// ------------------------------------------------------------------------------------------------
__global__ void merge_row_kernel(const double* __restrict__ m, const double* __restrict__ log_island,
                                 const double* __restrict__ wscale, const Pivot5 common, double* __restrict__ row) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    // (GSE_OP_WEIGHT_SCALE: the sums are those of the stored weights)
    merge_row_thread(m, log_island[0], wscale ? (int)wscale[0] : 0, common, row);
}

__global__ void merge_rows_kernel(const double* __restrict__ rows, int P, const Pivot5 common, double* __restrict__ out,
                                  int32_t* __restrict__ status) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    merge_rows_thread(rows, P, common, out, status);
}

// The tail of gse_pf_merge_begin for the LIVE particles and weights (no resample pending): ONE workgroup finishes the shard's
// moment partials (moments_finish_block's arithmetic, all loads requested at once), its first wavefront runs merge_tail_wave.
// (With a resample pending the resampled-moments kernel's last workgroup does all this itself: gse_pf_kernels.h.)
struct MergeFinishArgs {
    const double* partials;          // [kMomentBlocks][kMomentWidth] of filter 0
    int nblk;
    const float* pivot_src;
    int64_t pivot_elem_stride;
    MergeTail tail;
};

__global__ __launch_bounds__(64 * kFinishGroups) void merge_finish_kernel(const MergeFinishArgs a) {
    __shared__ double part[kFinishGroups][64];
    __shared__ double sm[32];                          // the shard's moments (26)
    __shared__ double sx[32];                          // its row (22)
    const int tid = threadIdx.x;
    {   // moments_finish_block (same order of additions), the <= 32 partials of a thread in flight together
        const int c = tid & 63, g = tid >> 6;
        constexpr int kPer = kMomentBlocks / kFinishGroups;       // 32
        double v[kPer];
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int q = g + j * kFinishGroups;
            v[j] = (c < 21 && q < a.nblk) ? a.partials[(int64_t)q * kMomentWidth + c] : 0.0;
        }
        double acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[j] = 0.0 + v[j];
#pragma unroll
            for (int i = 1; i < kPer / 8; ++i) acc[j] += v[j + 8 * i];
        }
        part[g][c] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
        __syncthreads();
        if (g == 0 && c < 21) {
            double t[kFinishGroups];
#pragma unroll
            for (int k = 0; k < kFinishGroups; ++k) t[k] = part[k][c];
#pragma unroll
            for (int span = 1; span < kFinishGroups; span *= 2)
#pragma unroll
                for (int k = 0; k < kFinishGroups; k += 2 * span) t[k] += t[k + span];
            sm[c] = t[0];
            a.tail.moments26[c] = t[0];
        }
        if (g == 1 && c < GSE_NX) {
            const double pvt = (double)a.pivot_src[c * a.pivot_elem_stride];
            sm[21 + c] = pvt;
            a.tail.moments26[21 + c] = pvt;
        }
    }
    for (int i = tid; i < a.tail.P * 22; i += 64 * kFinishGroups)
        if (i / 22 != a.tail.rank) a.tail.rows[i] = 0.0;
    __syncthreads();
    if (tid < 64) merge_tail_wave(a.tail, sm, sx, 0.0, tid);      // (never pending here: no weight sum to absorb)
}

// A shard's accumulated log-weight absorbs the sum of its live weights (what the particle filter's gather stage does inside
// its launch, pf_step_kernel: island[b] += log(W)) -- as a call of its own for estimators whose resample does not carry the
// bookkeeping (the sharded Gaussian-sum filter): ONE workgroup adds the N fp32 weights in fp64, in a fixed order.
__global__ __launch_bounds__(1024) void island_absorb_kernel(const float* __restrict__ w, int64_t N, double* __restrict__ log_island,
                                                             double* __restrict__ sum_out) {
    __shared__ double part[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (int64_t i0 = tid; i0 < N; i0 += 4096) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t i = i0 + (int64_t)j * 1024;
            if (i < N) a[j] += (double)w[i];
        }
    }
    const double v = wave_sum((a[0] + a[1]) + (a[2] + a[3]));
    if (lane == 0) part[wv] = v;
    __syncthreads();
    if (tid == 0) {
        double t[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) t[k] = part[k];
#pragma unroll
        for (int span = 1; span < 16; span *= 2)
#pragma unroll
            for (int k = 0; k < 16; k += 2 * span) t[k] += t[k + span];
        log_island[0] += log(t[0]);
        if (sum_out) sum_out[0] = t[0];
    }
}

__global__ __launch_bounds__(64) void merge_end_kernel(const double* __restrict__ rows, int P, const Pivot5 common, int flags,
                                                       double* __restrict__ merged26, double* __restrict__ estimate6,
                                                       int32_t* __restrict__ status) {
    __shared__ double sm[32];
    const int lane = threadIdx.x;
    double mine;
    const int st = merge_rows_wave(rows, P, common, lane, mine);
    if (lane == 0) status[0] = st;
    if (lane < 26) merged26[lane] = mine;
    if (!estimate6) return;
    if (lane < 32) sm[lane] = (lane < 26) ? mine : 0.0;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (flags & GSE_MERGE_COV_SIZE) estimates_wave(sm, estimate6, lane);
    else if (lane < GSE_NX) estimate6[lane] = sm[1 + lane] + sm[21 + lane] * sm[0];
}

}  // namespace gse

using namespace gse;

namespace gse {

int pack_mpc_law(const char* fn, const gse_mpc_law* law, MpcLawDev& d, int max_rows) {
    GSE_REQUIRE(law, "%s: null law", fn);
    GSE_REQUIRE(law->n_rows >= 0 && law->n_rows <= max_rows,
                "%s: %d bounded rows; this entry point takes at most %d", fn, law->n_rows, max_rows);
    GSE_REQUIRE(law->n_theta >= 1 && law->n_theta <= GSE_MPC_MAX_THETA && law->n_u >= 1 && law->n_u <= GSE_MPC_MAX_OUT &&
                    law->n_y >= 1 && law->n_y <= GSE_MPC_MAX_OUT,
                "%s: law dimensions (n_theta %d, n_u %d, n_y %d) out of range", fn, law->n_theta, law->n_u, law->n_y);
    GSE_REQUIRE(law->max_sweeps >= 1 && law->eps_abs >= 0.0 && law->eps_rel >= 0.0, "%s: bad stopping rule", fn);
    GSE_REQUIRE(law->Gu && law->gu0 && law->Gy && law->gy0, "%s: null law matrix", fn);
    GSE_REQUIRE(law->n_rows == 0 || (law->Gc && law->gc0 && law->S && law->lo && law->hi && law->Tu && law->Ty),
                "%s: null law matrix", fn);
    d.n_theta = law->n_theta; d.n_u = law->n_u; d.n_y = law->n_y; d.n_rows = law->n_rows; d.max_sweeps = law->max_sweeps;
    d.report_unsolved = law->report_unsolved;
    d.eps_abs = law->eps_abs; d.eps_rel = law->eps_rel;
    d.Gc = law->Gc; d.gc0 = law->gc0; d.S = law->S; d.lo = law->lo; d.hi = law->hi;
    d.Gu = law->Gu; d.gu0 = law->gu0; d.Tu = law->Tu; d.Gy = law->Gy; d.gy0 = law->gy0; d.Ty = law->Ty;
    return GSE_OK;
}

int pack_loop_glue(const char* fn, const gse_loop_glue* glue, const MpcLawDev& d, LoopGlue& g) {
    GSE_REQUIRE(glue, "%s: null glue", fn);
    const int nx = d.n_theta - d.n_u - d.n_y;
    GSE_REQUIRE(nx >= 1 && nx <= GSE_MPC_MAX_OUT, "%s: n_theta must be Nx + Ni + No with 1 <= Nx <= %d", fn, GSE_MPC_MAX_OUT);
    GSE_REQUIRE(glue->n_state_full == GSE_NX, "%s: the estimator's moments are those of a %d-state filter", fn, GSE_NX);
    GSE_REQUIRE(glue->n_input_full >= d.n_u && glue->n_output_full >= d.n_y && glue->n_input_full <= 64 &&
                    glue->n_output_full <= 64, "%s: bad plant dimensions", fn);
    g.n_state_full = glue->n_state_full; g.n_input_full = glue->n_input_full; g.n_output_full = glue->n_output_full;
    g.pad = 0;
    for (int k = 0; k < GSE_MPC_MAX_OUT; ++k) {
        g.states[k] = glue->states[k]; g.inputs[k] = glue->inputs[k]; g.outputs[k] = glue->outputs[k];
        g.x_bar[k] = glue->x_bar[k]; g.u_bar[k] = glue->u_bar[k]; g.y_bar[k] = glue->y_bar[k];
        g.u_fallback[k] = glue->u_fallback[k];
    }
    for (int k = 0; k < nx; ++k)
        GSE_REQUIRE(g.states[k] >= 0 && g.states[k] < g.n_state_full, "%s: states[%d] out of range", fn, k);
    for (int k = 0; k < d.n_u; ++k)
        GSE_REQUIRE(g.inputs[k] >= 0 && g.inputs[k] < g.n_input_full, "%s: inputs[%d] out of range", fn, k);
    for (int k = 0; k < d.n_y; ++k)
        GSE_REQUIRE(g.outputs[k] >= 0 && g.outputs[k] < g.n_output_full, "%s: outputs[%d] out of range", fn, k);
    return GSE_OK;
}

}  // namespace gse

extern "C" {

int gse_mpc_step(const gse_mpc_law* law, const double* theta, int64_t B, double* du0, double* y1, int32_t* status,
                 void* stream) {
    GSE_REQUIRE(theta && du0 && y1 && status && B >= 1, "gse_mpc_step: bad argument");
    MpcLawDev d;
    if (int rc = pack_mpc_law("gse_mpc_step", law, d, GSE_MPC_MAX_ROWS_TOTAL)) return rc;
    const unsigned blocks = (unsigned)((B + kMpcWavesPerBlock - 1) / kMpcWavesPerBlock);
    hipLaunchKernelGGL(mpc_step_kernel, dim3(blocks), dim3(64 * kMpcWavesPerBlock), 0, (hipStream_t)stream, d, theta, B, du0,
                       y1, status);
    GSE_LAUNCH_CHECK("mpc_step_kernel");
    return GSE_OK;
}

int gse_loop_control(const gse_mpc_law* law, const gse_loop_glue* glue, const double* moments, const double* y_meas,
                     double* u, double* y_pred, int32_t* known, int32_t* status, double* theta_out, int64_t B,
                     void* stream) {
    GSE_REQUIRE(glue && moments && y_meas && u && y_pred && known && status && B >= 1, "gse_loop_control: bad argument");
    MpcLawDev d;
    if (int rc = pack_mpc_law("gse_loop_control", law, d, GSE_MPC_MAX_ROWS_TOTAL)) return rc;
    LoopGlue g;
    if (int rc = pack_loop_glue("gse_loop_control", glue, d, g)) return rc;
    const unsigned blocks = (unsigned)((B + kMpcWavesPerBlock - 1) / kMpcWavesPerBlock);
    hipLaunchKernelGGL(loop_control_kernel, dim3(blocks), dim3(64 * kMpcWavesPerBlock), 0, (hipStream_t)stream, d, g, moments,
                       y_meas, u, y_pred, known, status, theta_out, B);
    GSE_LAUNCH_CHECK("loop_control_kernel");
    return GSE_OK;
}

int gse_plant_step(double* X, const double* u, double dt, const gse_mixture* state_noise, const gse_mixture* meas_noise,
                   const gse_rng* rng, double* y, double* y_meas, double* z, double* log_x, double* log_u, int64_t B,
                   void* stream) {
    GSE_REQUIRE(X && u && state_noise && meas_noise && rng && y && y_meas && z && B >= 1, "gse_plant_step: bad argument");
    if (int rc = check_mixture("gse_plant_step", state_noise, GSE_NX)) return rc;
    if (int rc = check_mixture("gse_plant_step", meas_noise, GSE_NY)) return rc;
    PlantArgs a;
    pack_draw(a.state_noise, *state_noise);
    pack_draw(a.meas_noise, *meas_noise);
    a.rng_state = make_rng_key(*rng, kPurposePlantState);
    a.rng_meas = make_rng_key(*rng, kPurposePlantMeas);
    a.dt = dt;
    hipLaunchKernelGGL(plant_step_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X, u, B, a, y,
                       y_meas, z, log_x, log_u, (const float*)nullptr, (const float*)nullptr);
    GSE_LAUNCH_CHECK("plant_step_kernel");
    return GSE_OK;
}

int gse_pf_moments_resampled_tail(const float* x, const float* w, int64_t ld, int B, int64_t N, const double* r, int slot,
                                  double* out, void* workspace, const gse_loop_tail* t, void* stream) {
    GSE_REQUIRE(out && t, "gse_pf_moments_resampled_tail: null argument");
    GSE_REQUIRE(t->glue && t->y_meas && t->u && t->y_pred && t->known && t->status && t->X && t->y && t->y_meas_out && t->z,
                "gse_pf_moments_resampled_tail: null argument in the tail");
    GSE_REQUIRE((t->state_draw != nullptr) == (t->meas_draw != nullptr),
                "gse_pf_moments_resampled_tail: the plant's two draws are injected together or not at all");
    LoopTailArgs a = {};
    if (int rc = pack_mpc_law("gse_pf_moments_resampled_tail", t->law, a.law)) return rc;
    if (int rc = pack_loop_glue("gse_pf_moments_resampled_tail", t->glue, a.law, a.glue)) return rc;
    if (!t->state_draw) {
        GSE_REQUIRE(t->state_noise && t->meas_noise && t->rng, "gse_pf_moments_resampled_tail: null noise argument");
        if (int rc = check_mixture("gse_pf_moments_resampled_tail", t->state_noise, GSE_NX)) return rc;
        if (int rc = check_mixture("gse_pf_moments_resampled_tail", t->meas_noise, GSE_NY)) return rc;
        pack_draw(a.plant.state_noise, *t->state_noise);
        pack_draw(a.plant.meas_noise, *t->meas_noise);
        a.plant.rng_state = make_rng_key(*t->rng, kPurposePlantState);
        a.plant.rng_meas = make_rng_key(*t->rng, kPurposePlantMeas);
    }
    a.plant.dt = t->dt;
    hipStream_t st = (hipStream_t)stream;
    const int nblk = launch_moments_resampled_partials("gse_pf_moments_resampled_tail", x, w, ld, B, N, r, slot, workspace, st);
    if (nblk < 0) return nblk;
    a.partials = carve_workspace(workspace, B, N).partials;
    a.nblk = nblk; a.width = 21; a.pivot_at = 21; a.pivot_count = GSE_NX;
    a.pivot_src = x; a.pivot_filter_stride = GSE_NX * ld; a.pivot_elem_stride = ld;
    a.moments = out;
    a.y_meas = t->y_meas; a.u = t->u; a.y_pred = t->y_pred; a.known = t->known; a.status = t->status;
    a.X = t->X; a.y = t->y; a.y_meas_out = t->y_meas_out; a.z = t->z; a.log_x = t->log_x; a.log_u = t->log_u;
    a.inj_state = t->state_draw; a.inj_meas = t->meas_draw;
    hipLaunchKernelGGL(loop_tail_kernel, dim3((unsigned)B), dim3(64 * kFinishGroups), 0, st, a);
    GSE_LAUNCH_CHECK("loop_tail_kernel");
    return GSE_OK;
}

int gse_merge_row(const double* moments26, const double* log_island, const double* wscale, const double* common_pivot,
                  double* row, void* stream) {
    GSE_REQUIRE(moments26 && log_island && common_pivot && row, "gse_merge_row: null argument");
    Pivot5 pv;
    for (int k = 0; k < GSE_NX; ++k) pv.p[k] = common_pivot[k];
    hipLaunchKernelGGL(merge_row_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, moments26, log_island, wscale, pv, row);
    GSE_LAUNCH_CHECK("merge_row_kernel");
    return GSE_OK;
}

int gse_merge_rows(const double* rows, int P, const double* common_pivot, double* moments26_out, int32_t* status, void* stream) {
    GSE_REQUIRE(rows && common_pivot && moments26_out && P >= 1 && P <= 4096, "gse_merge_rows: bad argument");
    Pivot5 pv;
    for (int k = 0; k < GSE_NX; ++k) pv.p[k] = common_pivot[k];
    hipLaunchKernelGGL(merge_rows_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, rows, P, pv, moments26_out, status);
    GSE_LAUNCH_CHECK("merge_rows_kernel");
    return GSE_OK;
}

int gse_pf_merge_begin(const float* x, const float* w, int64_t ld, int64_t N, const gse_merge_spec* spec, void* workspace,
                       void* stream) {
    GSE_REQUIRE(spec, "gse_pf_merge_begin: null spec");
    GSE_REQUIRE(spec->P >= 1 && spec->P <= 4096 && spec->rank >= 0 && spec->rank < spec->P, "gse_pf_merge_begin: bad P / rank");
    GSE_REQUIRE(spec->rows && spec->moments26 && spec->merged26 && spec->status, "gse_pf_merge_begin: null output");
    GSE_REQUIRE((spec->flags & ~(GSE_MERGE_PENDING | GSE_MERGE_COV_SIZE)) == 0, "gse_pf_merge_begin: bad flags 0x%x", spec->flags);
    hipStream_t st = (hipStream_t)stream;
    if (spec->flags & GSE_MERGE_PENDING) {
        // ONE launch: the resampled-moments kernel, whose last workgroup to finish runs the tail (gse_pf_kernels.h)
        const double r = spec->r;
        const int nblk = launch_moments_resampled_partials("gse_pf_merge_begin", x, w, ld, 1, N, &r, spec->slot, workspace, st, spec);
        return nblk < 0 ? nblk : GSE_OK;
    }
    const int nblk = launch_moments_partials("gse_pf_merge_begin", x, w, ld, 1, N, workspace, st);
    if (nblk < 0) return nblk;
    MergeFinishArgs a;
    a.partials = carve_workspace(workspace, 1, N).partials;
    a.nblk = nblk;
    a.pivot_src = x; a.pivot_elem_stride = ld;
    pack_merge_tail(spec, a.tail);
    hipLaunchKernelGGL(merge_finish_kernel, dim3(1), dim3(64 * kFinishGroups), 0, st, a);
    GSE_LAUNCH_CHECK("merge_finish_kernel");
    return GSE_OK;
}

int gse_island_absorb(const float* w, int64_t N, double* log_island, double* sum_out, void* stream) {
    GSE_REQUIRE(w && log_island && N >= 1, "gse_island_absorb: bad argument");
    hipLaunchKernelGGL(island_absorb_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, w, N, log_island, sum_out);
    GSE_LAUNCH_CHECK("island_absorb_kernel");
    return GSE_OK;
}

int gse_merge_end(const gse_merge_spec* spec, void* stream) {
    GSE_REQUIRE(spec && spec->rows && spec->merged26 && spec->status && spec->P >= 1 && spec->P <= 4096, "gse_merge_end: bad argument");
    Pivot5 pv;
    for (int k = 0; k < GSE_NX; ++k) pv.p[k] = spec->common_pivot[k];
    hipLaunchKernelGGL(merge_end_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)spec->rows, spec->P, pv,
                       (int)spec->flags, spec->merged26, spec->estimate6, spec->status);
    GSE_LAUNCH_CHECK("merge_end_kernel");
    return GSE_OK;
}

int gse_moments_estimates(const double* moments, int64_t R, double* out, void* stream) {
    GSE_REQUIRE(moments && out && R >= 1, "gse_moments_estimates: bad argument");
    hipLaunchKernelGGL(moments_estimates_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, (hipStream_t)stream, moments, R,
                       out);
    GSE_LAUNCH_CHECK("moments_estimates_kernel");
    return GSE_OK;
}

int gse_plant_step_injected(double* X, const double* u, double dt, const float* state_draw, const float* meas_draw, double* y,
                            double* y_meas, double* z, double* log_x, double* log_u, int64_t B, void* stream) {
    GSE_REQUIRE(X && u && state_draw && meas_draw && y && y_meas && z && B >= 1, "gse_plant_step_injected: bad argument");
    PlantArgs a = {};
    a.dt = dt;
    hipLaunchKernelGGL(plant_step_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, X, u, B, a, y,
                       y_meas, z, log_x, log_u, state_draw, meas_draw);
    GSE_LAUNCH_CHECK("plant_step_kernel");
    return GSE_OK;
}

}  // extern "C"
