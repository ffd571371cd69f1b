// gse_loop.hip -- a whole closed loop in ONE launch (round 3): B rollouts, each a Gaussian-sum filter of at most 64 Gaussians +
// the MPC law + the plant, tick after tick inside a persistent kernel -- one workgroup of 1024 threads per rollout, the bank
// in LDS from the first tick to the last.
//
// Why: a tick of such a loop is five small launches (the one-launch filter cycle, two for the moments, the controller, the
// plant: sim_base.py:243-309 for B rollouts) and the device idles between them -- 499 ticks took 0.021 s, which was the rate at
// which the host can enqueue (42 us per tick), not device work.  Rollouts do not talk to each other, so nothing forces a
// launch boundary: here a tick is 19 us of dependent work and the loop 9.7 ms (DESIGN.md section 6 has the phases).  Every piece is the device function the
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
// moments_finish_kernel for a bank of at most 64.  There every Gaussian is one lane of wavefront 0 and each of the 21 sums is
// that wavefront's butterfly; here wavefront c forms sum c (its lane g holds Gaussian g's term, the same expression), so
// the 21 butterflies run side by side on the 16 wavefronts instead of one after the other on one: 4.8 us -> about 1.
__device__ __forceinline__ void bank_moments26(const float (*mu_s)[kNx], const float (*P_s)[kNxx], const float* w_s, int N,
                                               double* out26, int tid) {
    // (no contraction: the kernel this restates rounds each product before it adds)
#pragma clang fp contract(off)
    constexpr int kSums = 1 + kNx + kTri;
    const int g = tid & 63, wv = tid >> 6;
    for (int c = wv; c < kSums; c += kSmallBlock / 64) {
        // sum c: 0 the weight, 1..5 w d_k, then the upper triangle (k, l >= k) row by row of w d_k d_l + w P_kl.
        // (gsukf_moments_kernel keeps sum w d d^T and sum w P apart and folds them at the end; with one Gaussian per lane each
        //  holds a single term, 0 + x = x exactly, so the fold is formed directly)
        int k = 0, l = 0;
        if (c > kNx) {
            int e = c - 1 - kNx;
            while (e >= kNx - k) { e -= kNx - k; ++k; }
            l = k + e;
        } else if (c >= 1) {
            k = c - 1;
        }
        double term = 0.0;
        if (g < N) {
            const double wp = (double)w_s[g];
            const double wd = wp * ((double)mu_s[g][k] - (double)mu_s[0][k]);
            term = c == 0 ? wp : c <= kNx ? wd : wd * ((double)mu_s[g][l] - (double)mu_s[0][l]) + wp * (double)P_s[g][k * kNx + l];
        }
        const double total = wave_sum(term);
        if (g == 0) out26[c] = (total + 0.0) + (0.0 + 0.0);        // (the four wavefront totals of a 256-thread block: three zeros)
    }
    if (tid < kNx) out26[kSums + tid] = (double)mu_s[0][tid];
}

// (The arguments come through device memory, not the kernel-argument segment: the tick loop reads them through a pointer it
//  launders every tick -- see there -- and taking the address of a by-value kernel argument makes a private copy of it.)
__global__ __launch_bounds__(kSmallBlock) void gsukf_closed_loop_kernel(const GsukfLoopArgs* __restrict__ args) {
    const GsukfLoopArgs& a = *args;
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
    // the rollout's scalars live in LDS for the whole loop (global memory sees them once, at the end) and so do the law's
    // matrices when there are few bounded rows: the controller call is a chain of dependent reads, now LDS reads
    __shared__ double st_u[2], st_z[2], st_X[5], st_ypred[GSE_MPC_MAX_OUT], st_ymeas[5], mom_s[26];
    __shared__ int32_t st_known;
    constexpr int kLawRows = 8;
    __shared__ double law_s[2 * kLawRows * GSE_MPC_MAX_THETA + kLawRows * kLawRows + 3 * kLawRows +
                            2 * GSE_MPC_MAX_OUT * (GSE_MPC_MAX_THETA + 1 + kLawRows) + GSE_MPC_MAX_OUT * GSE_MPC_MAX_THETA];
    const int tid0 = threadIdx.x, b = blockIdx.x, N0 = a.N, B = a.B;
    int tid = tid0, N = N0;
    MpcLawDev law0 = a.law;
    MpcLawDev& law = law0;
    if (law.n_rows <= kLawRows) {
        const int m = law.n_rows, nt = law.n_theta, nu = law.n_u, ny = law.n_y;
        double* at = law_s;
        auto stage = [&](const double* __restrict__& field, int count) {
            for (int i = tid; i < count; i += kSmallBlock) at[i] = field[i];
            field = at;
            at += count;
        };
        if (m) { stage(law.Gc, m * nt); stage(law.gc0, m); stage(law.S, m * m); stage(law.lo, m); stage(law.hi, m);
                 stage(law.Tu, nu * m); stage(law.Ty, ny * m); }
        stage(law.Gu, nu * nt); stage(law.gu0, nu); stage(law.Gy, ny * nt); stage(law.gy0, ny);
    }
    if (tid < 2) { st_u[tid] = a.u[2 * b + tid]; st_z[tid] = a.z[2 * b + tid]; }
    if (tid < 5) { st_X[tid] = a.X[5 * b + tid]; st_ymeas[tid] = a.ys_meas[(int64_t)b * 5 + tid]; }      // (row 0 of the records)
    if (tid < law.n_y) st_ypred[tid] = a.y_pred[(int64_t)b * law.n_y + tid];
    if (tid == 0) st_known = a.known[b];
    __shared__ LoopGlue glue_s;                  // (indexed by lane in the controller call: from LDS, not from the argument block)
    {
        const uint32_t* sg = reinterpret_cast<const uint32_t*>(&a.glue);
        uint32_t* dg = reinterpret_cast<uint32_t*>(&glue_s);
        for (int i = tid; i < (int)(sizeof(LoopGlue) / 4); i += kSmallBlock) dg[i] = sg[i];
    }
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
    int n_predict = 0, n_update = 0;
    // The plant of tick t is one thread's work (an fp64 model step, two draws, the records: 2.2 us): it runs on the LAST thread of
    // the workgroup during tick t + 1's predict -- which needs the input the controller left, not the measurement the plant is
    // about to produce -- beside the phase that forms the sigma points' mean and covariance on the first wavefront (2.5 us).
    auto plant = [&](const GsukfLoopArgs& aa, int t) {
        RngKey ks = aa.plant.rng_state, km = aa.plant.rng_meas;
        ks.step_lo = (uint32_t)t;
        km.step_lo = (uint32_t)t;
        plant_step_one(st_X, st_u, 0, b, dpp, ks, km, aa.plant.dt, aa.ys + ((int64_t)t * B + b) * 5, st_ymeas, st_z,
                       aa.xs + ((int64_t)t * B + b) * 5, aa.us + ((int64_t)t * B + b) * 2, nullptr, nullptr);
#pragma unroll
        for (int k = 0; k < 5; ++k) aa.ys_meas[((int64_t)t * B + b) * 5 + k] = st_ymeas[k];
    };
    const int ticks = a.ticks;
    for (int t = 1; t < ticks; ++t) {
        // (the arguments through a pointer the compiler cannot see through: otherwise everything in the tick that depends on
        //  them alone -- sigma weights as doubles, mixture constants, the law's pointers -- is hoisted out of the tick loop,
        //  kept live across all phases and spilled: 576 bytes of scratch per lane and ten of a tick's 26 us)
        const GsukfLoopArgs* ap = args;
        asm volatile("" : "+s"(ap));
        const GsukfLoopArgs& aa = *ap;
        const int tid = per_iteration(tid0), N = per_iteration_uniform(N0);
        const int j = tid & (kGroup - 1), gl = tid / kGroup;
        MpcLawDev law = law0;
        per_iteration_uniform(law);
        const int due = aa.schedule[t];
#ifdef GSE_LOOP_STAMPS
        double stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define STAMP(i) stamp[i] = (double)wall_clock64()
#else
#define STAMP(i)
#endif
        STAMP(0);
        // (beside phase C of the predict -- one wavefront's work, 2.5 us -- rather than beside the first factorisation, 0.8 us,
        //  which it kept waiting; a tick without a predict runs it here)
        const bool plant_due = t > 1 && tid == kSmallBlock - 1;
        if (plant_due && !(due & 1)) plant(aa, t - 1);
        FilterIO bio;
        bio.u0 = st_u[0]; bio.u1 = st_u[1]; bio.z0 = 0.0; bio.z1 = 0.0;
        if (due & 1) {
            RngKey key = aa.pred.rng;
            key.step_lo += (uint32_t)n_predict;          // (the class counts a step per predict; 32 bits last a billion ticks)
            ++n_predict;
            gsukf_predict_block<false>(from_lds, to_lds, fac, tile, N, b, 0, nullptr, dp, aa.chol_status, aa.pred, bio, key,
                                       [&]() { if (plant_due) plant(aa, t - 1); });
            __syncthreads();
        }
        STAMP(1);
        double* mom_t = aa.moments + ((int64_t)t * B + b) * 26;
        if (due & 2) {
            __syncthreads();                              // (the plant of the tick before has left its measurement)
            bio.z0 = st_z[0]; bio.z1 = st_z[1];
            gsukf_update_block(from_lds, to_lds, [&](int g) { return (const float*)&w_s[g]; }, [&](int g) { return &w_s[g]; }, fac,
                               tile, eta, N, b, aa.chol_status, bio, aa.pdf, aa.pred.sw);
            __syncthreads();
            STAMP(2);
            small_bank_ancestors(w_s, c_s, anc_s, &total_s, N, aa.r[(int64_t)n_update * B + b], tid);
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
            STAMP(3);
            bank_moments26(mu_s, P_s, w_s, N, mom_s, tid);
            __syncthreads();
            STAMP(4);
            if (tid < 26) mom_t[tid] = mom_s[tid];
            if (tid < 64)
                loop_control_wave(law, glue_s, mom_s, st_ymeas, st_u, st_ypred, &st_known, aa.status + (int64_t)t * B + b, nullptr, 0,
                                  tid);
            __syncthreads();
        }
        if (!(due & 2)) {
            bank_moments26(mu_s, P_s, w_s, N, mom_s, tid);
            __syncthreads();
            if (tid < 26) mom_t[tid] = mom_s[tid];
        }
        __syncthreads();
#ifdef GSE_LOOP_STAMPS
        STAMP(5);
        if (tid == 0 && b == 0)
            for (int i = 0; i < 6; ++i) mom_t[6 + i] = stamp[i];
#endif
    }
    if (tid == kSmallBlock - 1) plant(a, a.ticks - 1);
    __syncthreads();
    if (tid < 2) { a.u[2 * b + tid] = st_u[tid]; a.z[2 * b + tid] = st_z[tid]; }
    if (tid < 5) a.X[5 * b + tid] = st_X[tid];
    if (tid < law.n_y) a.y_pred[(int64_t)b * law.n_y + tid] = st_ypred[tid];
    if (tid == 0) a.known[b] = st_known;
    // the bank: LDS -> global
    for (int i = tid; i < N * kNx; i += kSmallBlock) a.mu[(int64_t)b * N * kNx + i] = mu_s[i / kNx][i % kNx];
    for (int i = tid; i < N * kNxx; i += kSmallBlock) a.P[(int64_t)b * N * kNxx + i] = P_s[i / kNxx][i % kNxx];
    for (int i = tid; i < N; i += kSmallBlock) a.w[(int64_t)b * N + i] = w_s[i];
}

// The loop kernel reads its argument block through a pointer (see above); this launch puts the block there.  A by-value
// kernel argument travels in the launch's own kernarg segment: nothing is read from host memory after the call returns, the
// call stays enqueue-only and capturable (a hipMemcpyAsync from the pageable stack frame was neither: the copy blocks the
// host until the stream reaches it, and under capture the node would keep a dangling host pointer).
__global__ __launch_bounds__(256) void store_loop_args_kernel(const GsukfLoopArgs a, GsukfLoopArgs* __restrict__ dst) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&a);
    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
    for (int i = threadIdx.x; i < (int)(sizeof(GsukfLoopArgs) / 4); i += 256) d[i] = src[i];
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
    GSE_REQUIRE(s->args_buffer && aligned16(s->args_buffer),
                "gse_gsukf_closed_loop: args_buffer must be a 16-byte aligned device buffer of GSE_LOOP_ARGS_BYTES bytes");
    static_assert(sizeof(GsukfLoopArgs) <= GSE_LOOP_ARGS_BYTES, "GSE_LOOP_ARGS_BYTES is too small");
    static_assert(sizeof(GsukfLoopArgs) % 4 == 0, "argument block copied word by word");
    hipLaunchKernelGGL(store_loop_args_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a, (GsukfLoopArgs*)s->args_buffer);
    GSE_LAUNCH_CHECK("store_loop_args_kernel");
    hipLaunchKernelGGL(gsukf_closed_loop_kernel, dim3((unsigned)s->B), dim3(kSmallBlock), 0, (hipStream_t)stream,
                       (const GsukfLoopArgs*)s->args_buffer);
    GSE_LAUNCH_CHECK("gsukf_closed_loop_kernel");
    return GSE_OK;
}
