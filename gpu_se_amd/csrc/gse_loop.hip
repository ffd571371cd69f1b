// gse_loop.hip -- a whole closed loop in ONE launch (round 3): B rollouts, each a Gaussian-sum filter of at most 64 Gaussians +
// the MPC law + the plant, tick after tick inside a persistent kernel -- one workgroup of 1024 threads per rollout, the bank
// in LDS from the first tick to the last.
//
// Why: a tick of such a loop is five small launches (the one-launch filter cycle, two for the moments, the controller, the
// plant: sim_base.py:243-309 for B rollouts) and the device idles between them -- 499 ticks took 0.021 s, which was the rate at
// which the host can enqueue (42 us per tick), not device work.  Rollouts do not talk to each other, so nothing forces a
// launch boundary: here a tick is ~15 us of dependent work and the loop 7 ms.  Every piece is the device function the
// separate launches run (gsukf_predict_block / gsukf_update_block / small_bank_ancestors of the cycle kernel, the wavefront
// reduction of gsukf_moments_kernel, loop_control_wave, plant_step_one), same Philox elements and steps, same order of
// operations within a tick -- so the records are those of the launch-per-piece loop bit for bit
// (tests/test_gpu_closedloop.py::test_persistent_loop_equals_the_launch_per_piece_loop).
#include "gse_gsukf_kernels.h"
#include "gse_loop_device.h"

namespace gse {

struct GsukfLoopArgs {
    GsukfPredictArgs pred;          // sigma weights, dt, the filter's draw and Philox key (step = its first predict's); io unused
    Pdf2 pdf;                       // measurement mixture
    PlantArgs plant;                // the plant's two draws; Philox keys with step 0 (tick t uses step t)
    MpcLawDev law;
    LoopGlue glue;
    const unsigned char* schedule;  // [ticks]: bit 0 predict due, bit 1 control due (entry 0 unused)
    const double* r;                // [updates][B] uniform numbers of the resamples
    float *mu, *P, *w;              // the banks [B][N][5], [B][N][5][5], [B][N]: in / out
    double *X, *u, *z;              // plant state [B][5], current input [B][2], current measurement [B][2]: in / out
    double *y_pred;                 // [B][n_y] in / out
    int32_t* known;                 // [B] in / out
    double *us, *xs, *ys, *ys_meas, *moments;   // records [ticks][B][2 | 5 | 5 | 5 | 26]; row 0 is the caller's
    int32_t* status;                // [ticks][B]
    int32_t* chol_status;           // Gaussians whose covariance could not be factorised
    int ticks, N, B;
};

// moments of the bank in the particle filter's 26-double row (gse_gsukf_moments26): the arithmetic of gsukf_moments_kernel<1> +
// moments_finish_kernel for a bank of at most 64 -- every Gaussian is one lane of wavefront 0, the wavefront sum is the total
__device__ __forceinline__ void bank_moments26(const float (*mu_s)[kNx], const float (*P_s)[kNxx], const float* w_s, int N,
                                               double* out26, int tid) {
    if (tid >= 64) return;
    double acc[1 + kNx + 2 * kTri];
#pragma unroll
    for (int c = 0; c < 1 + kNx + 2 * kTri; ++c) acc[c] = 0.0;
    double pv[kNx];
#pragma unroll
    for (int k = 0; k < kNx; ++k) pv[k] = (double)mu_s[0][k];
    if (tid < N) {
        const double wp = (double)w_s[tid];
        double d[kNx];
#pragma unroll
        for (int k = 0; k < kNx; ++k) d[k] = (double)mu_s[tid][k] - pv[k];
        acc[0] += wp;
        int c = 1 + kNx;
#pragma unroll
        for (int k = 0; k < kNx; ++k) {
            const double wd = wp * d[k];
            acc[1 + k] += wd;
#pragma unroll
            for (int l = k; l < kNx; ++l) acc[c++] += wd * d[l];
        }
#pragma unroll
        for (int k = 0; k < kNx; ++k)
#pragma unroll
            for (int l = k; l < kNx; ++l) acc[c++] += wp * (double)P_s[tid][k * kNx + l];
    }
#pragma unroll
    for (int i = 0; i < kTri; ++i) acc[1 + kNx + i] += acc[1 + kNx + kTri + i];
#pragma unroll
    for (int c = 0; c < 1 + kNx + kTri; ++c) {
        const double total = wave_sum(acc[c]);
        if (tid == c) out26[c] = (total + 0.0) + (0.0 + 0.0);      // (the four wavefront totals of a 256-thread block: three zeros)
    }
    if (tid < kNx) out26[1 + kNx + kTri + tid] = pv[tid];
}

__global__ __launch_bounds__(kSmallBlock) void gsukf_closed_loop_kernel(const GsukfLoopArgs a) {
    static_assert(kNx == GSE_NX, "the closed loop is the bioreactor's");
    __shared__ DrawParams dp;
    __shared__ DrawParams dpp[2];
    __shared__ float fac[kBank][kFacStride];
    __shared__ float tile[kBank][kTileStride];
    __shared__ double eta[kBank][kEtaStride];
    __shared__ float mu_s[kSmallBank][kNx];
    __shared__ float P_s[kSmallBank][kNxx];
    __shared__ float w_s[kSmallBank];
    __shared__ double c_s[kSmallBank];
    __shared__ int anc_s[kSmallBank];
    __shared__ double total_s;
    const int tid = threadIdx.x, b = blockIdx.x, N = a.N, B = a.B;
    {
        const uint32_t* s0 = reinterpret_cast<const uint32_t*>(&a.pred.draw);
        const uint32_t* s1 = reinterpret_cast<const uint32_t*>(&a.plant.state_noise);
        const uint32_t* s2 = reinterpret_cast<const uint32_t*>(&a.plant.meas_noise);
        uint32_t* d0 = reinterpret_cast<uint32_t*>(&dp);
        uint32_t* d1 = reinterpret_cast<uint32_t*>(&dpp[0]);
        uint32_t* d2 = reinterpret_cast<uint32_t*>(&dpp[1]);
        for (int i = tid; i < (int)(sizeof(DrawParams) / 4); i += kSmallBlock) { d0[i] = s0[i]; d1[i] = s1[i]; d2[i] = s2[i]; }
    }
    // the bank: global -> LDS
    for (int i = tid; i < N * kNx; i += kSmallBlock) mu_s[i / kNx][i % kNx] = a.mu[(int64_t)b * N * kNx + i];
    for (int i = tid; i < N * kNxx; i += kSmallBlock) P_s[i / kNxx][i % kNxx] = a.P[(int64_t)b * N * kNxx + i];
    for (int i = tid; i < N; i += kSmallBlock) w_s[i] = a.w[(int64_t)b * N + i];
    __syncthreads();
    auto from_lds = [&](int t, const float*& m_, const float*& p_) { m_ = mu_s[t]; p_ = P_s[t]; };
    auto to_lds = [&](int t, float*& m_, float*& p_) { m_ = mu_s[t]; p_ = P_s[t]; };
    const int j = tid & (kGroup - 1), gl = tid / kGroup;
    int n_predict = 0, n_update = 0;
    for (int t = 1; t < a.ticks; ++t) {
        const int due = a.schedule[t];
        FilterIO bio;
        bio.u0 = a.u[2 * b]; bio.u1 = a.u[2 * b + 1]; bio.z0 = a.z[2 * b]; bio.z1 = a.z[2 * b + 1];
        if (due & 1) {
            RngKey key = a.pred.rng;
            key.step_lo += (uint32_t)n_predict;          // (the class counts a step per predict; 32 bits last a billion ticks)
            ++n_predict;
            gsukf_predict_block<false>(from_lds, to_lds, fac, tile, N, b, 0, nullptr, dp, a.chol_status, a.pred, bio, key);
            __syncthreads();
        }
        double* mom_t = a.moments + ((int64_t)t * B + b) * 26;
        if (due & 2) {
            gsukf_update_block(from_lds, to_lds, [&](int g) { return (const float*)&w_s[g]; }, [&](int g) { return &w_s[g]; }, fac,
                               tile, eta, N, b, a.chol_status, bio, a.pdf, a.pred.sw);
            __syncthreads();
            small_bank_ancestors(w_s, c_s, anc_s, &total_s, N, a.r[(int64_t)n_update * B + b], tid);
            ++n_update;
            // the rows of the ancestors, in place: through registers
            float keep[2];
            const bool live = gl < N;
            const int src = live ? anc_s[gl] : 0;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = j + q * kGroup;
                keep[q] = (live && e < kRow) ? (e < kNx ? mu_s[src][e] : P_s[src][e - kNx]) : 0.0f;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = j + q * kGroup;
                if (live && e < kRow) {
                    if (e < kNx) mu_s[gl][e] = keep[q]; else P_s[gl][e - kNx] = keep[q];
                }
            }
            if (tid < N) w_s[tid] = (float)(1.0 / (double)N);
            __syncthreads();
            bank_moments26(mu_s, P_s, w_s, N, mom_t, tid);
            __syncthreads();
            if (tid < 64)
                loop_control_wave(a.law, a.glue, a.moments + (int64_t)t * B * 26, a.ys_meas + (int64_t)(t - 1) * B * 5, a.u, a.y_pred,
                                  a.known, a.status + (int64_t)t * B, nullptr, b, tid);
            __syncthreads();
        }
        if (tid == 0) {
            RngKey ks = a.plant.rng_state, km = a.plant.rng_meas;
            ks.step_lo = (uint32_t)t;
            km.step_lo = (uint32_t)t;
            plant_step_one(a.X, a.u, b, dpp, ks, km, a.plant.dt, a.ys + (int64_t)t * B * 5, a.ys_meas + (int64_t)t * B * 5, a.z,
                           a.xs + (int64_t)t * B * 5, a.us + (int64_t)t * B * 2, nullptr, nullptr);
        }
        if (!(due & 2)) bank_moments26(mu_s, P_s, w_s, N, mom_t, tid);
        __syncthreads();
    }
    // the bank: LDS -> global
    for (int i = tid; i < N * kNx; i += kSmallBlock) a.mu[(int64_t)b * N * kNx + i] = mu_s[i / kNx][i % kNx];
    for (int i = tid; i < N * kNxx; i += kSmallBlock) a.P[(int64_t)b * N * kNxx + i] = P_s[i / kNxx][i % kNxx];
    for (int i = tid; i < N; i += kSmallBlock) a.w[(int64_t)b * N + i] = w_s[i];
}

}  // namespace gse

using namespace gse;

extern "C" int gse_gsukf_closed_loop(const gse_loop_spec* s, void* stream) {
    GSE_REQUIRE(s, "gse_gsukf_closed_loop: null spec");
    GSE_REQUIRE(s->B >= 1 && s->B <= GSE_MAX_B && s->N >= 1 && s->N <= kSmallBank && s->ticks >= 2,
                "gse_gsukf_closed_loop: B = %d (1..%d), N = %d (1..%d Gaussians: the one-workgroup cycle), ticks = %d", s->B,
                GSE_MAX_B, s->N, kSmallBank, s->ticks);
    GSE_REQUIRE(s->mu && s->P && s->w && s->X && s->u && s->z && s->y_pred && s->known && s->us && s->xs && s->ys && s->ys_meas &&
                    s->moments && s->status && s->chol_status && s->schedule && s->r && s->w_sigma && s->state_pdf && s->rng &&
                    s->measurement_pdf && s->plant_state_noise && s->plant_meas_noise && s->plant_rng && s->law && s->glue,
                "gse_gsukf_closed_loop: null argument");
    if (int rc = check_mixture("gse_gsukf_closed_loop", s->state_pdf, GSE_NX)) return rc;
    if (int rc = check_mixture("gse_gsukf_closed_loop", s->measurement_pdf, GSE_NY)) return rc;
    if (int rc = check_mixture("gse_gsukf_closed_loop", s->plant_state_noise, GSE_NX)) return rc;
    if (int rc = check_mixture("gse_gsukf_closed_loop", s->plant_meas_noise, GSE_NY)) return rc;
    GsukfLoopArgs a = {};
    pack_sigma_weights(a.pred.sw, s->w_sigma, GSE_NSIGMA);
    a.pred.dt = s->dt;
    pack_draw(a.pred.draw, *s->state_pdf);
    a.pred.rng = make_rng_key(*s->rng, kPurposeGsukf);
    pack_pdf2(a.pdf, *s->measurement_pdf);
    pack_draw(a.plant.state_noise, *s->plant_state_noise);
    pack_draw(a.plant.meas_noise, *s->plant_meas_noise);
    a.plant.rng_state = make_rng_key(*s->plant_rng, kPurposePlantState);
    a.plant.rng_meas = make_rng_key(*s->plant_rng, kPurposePlantMeas);
    a.plant.dt = s->dt;
    if (int rc = pack_mpc_law("gse_gsukf_closed_loop", s->law, a.law)) return rc;
    if (int rc = pack_loop_glue("gse_gsukf_closed_loop", s->glue, a.law, a.glue)) return rc;
    a.schedule = s->schedule; a.r = s->r;
    a.mu = s->mu; a.P = s->P; a.w = s->w; a.X = s->X; a.u = s->u; a.z = s->z; a.y_pred = s->y_pred; a.known = s->known;
    a.us = s->us; a.xs = s->xs; a.ys = s->ys; a.ys_meas = s->ys_meas; a.moments = s->moments; a.status = s->status;
    a.chol_status = s->chol_status;
    a.ticks = s->ticks; a.N = s->N; a.B = s->B;
    hipLaunchKernelGGL(gsukf_closed_loop_kernel, dim3((unsigned)s->B), dim3(kSmallBlock), 0, (hipStream_t)stream, a);
    GSE_LAUNCH_CHECK("gsukf_closed_loop_kernel");
    return GSE_OK;
}
