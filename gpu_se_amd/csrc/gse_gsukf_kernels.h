// gse_gsukf_kernels.h -- the Gaussian-sum unscented Kalman filter kernels (device code only), in a header of their own
// for the same reason as gse_pf_kernels.h: gse_gsukf.hip compiles them for the built-in bioreactor model, gse_rtc.hip
// compiles them at run time for a user model (what Numba's JIT does for the reference's GS-UKF, filter/gs_ukf.py:243-275).
// Every kernel is a template so that a run-time compilation instantiates the one it was asked for and nothing else.
//
// A bank of N Gaussians (mu_i, P_i, w_i); each is a UKF with 2*kNx+1 unscaled sigma points (11 for the bioreactor's 5
// states; filter/gs_ukf.py).
// A workgroup of 1024 threads takes 64 Gaussians and every phase alternates between two mappings:
//   * per SIGMA POINT, 16 lanes per Gaussian (lane j < 11 owns sigma point j): the model step with its noise draw
//     (predict) and the output function (update) -- the part with 11-fold parallelism per Gaussian;
//   * per GAUSSIAN, one lane each (the 64 lanes of wavefront 0): the 5x5 Cholesky, the 11-term contractions (mean, P,
//     P_xy, P_yy), the 2x2 pseudo-inverse, the gain and the covariance update.  A wavefront issues an instruction once
//     whatever the number of lanes that need it, so doing this per-Gaussian algebra in all 16 lanes of every Gaussian
//     (the first version) cost 16 wavefronts' worth of issue slots per 64 Gaussians instead of one.
// Large banks take the one-lane-per-Gaussian kernels further down; banks of at most 64 the one-launch cycle kernel.
// Hand-offs go through LDS (rows padded to odd strides: a lane per Gaussian walks rows).  Sums run in the reference's
// order (j = 0..10) so that results track the NumPy class closely.  These contractions are 5x11x5 at most: far too small
// to feed the matrix cores (a 16x16x4 fp32 MFMA tile would be >90 % padding) and nowhere near a bound.
#pragma once

#include "gse_host.h"

namespace gse {

// States of the model the kernels are compiled for: 5 for the built-in bioreactor; a user model brings its own count
// (GSE_USER_NX, 1..5) and every shape below follows -- the storage is (mu[kNx], P[kNx][kNx]) per Gaussian, row-major.
constexpr int kNx = kModelNx;
constexpr int kNs = 2 * kNx + 1;                 // sigma points per Gaussian
constexpr int kTri = kNx * (kNx + 1) / 2;        // packed triangle of a kNx x kNx matrix
constexpr int kNxx = kNx * kNx;
constexpr int kRow = kNx + kNxx;                 // floats per Gaussian (30 for the bioreactor)
static_assert(kNs <= GSE_NSIGMA, "at most 11 sigma points (5 states)");
constexpr int kGroup = 16;                       // lanes per Gaussian in the per-sigma-point parts
constexpr int kGaussPerBlock = kBlock / kGroup;  // 16 (gsukf_sigma_kernel only)
#ifndef GSE_GSUKF_WIDE_BANK
#define GSE_GSUKF_WIDE_BANK 64   // (16 / 32 / 64 Gaussians per workgroup, 2^18 Gaussians: 64 is 15 % faster than 16 or 32)
#endif
constexpr int kWideBank = GSE_GSUKF_WIDE_BANK;   // Gaussians per workgroup of the separate predict / update kernels
constexpr int kBank = 64;                        // Gaussians per workgroup of the one-launch cycle kernel
constexpr int kFacStride = (kNx + kTri) | 1;     // LDS row: m[kNx], L[kTri] (odd stride; 21 for 5 states)
constexpr int kTileStride = (kNs * kNx + 2) | 1; // odd; 57 floats for 5 states
constexpr int kEtaStride = kNs * 2 + 1;          // 23 doubles for 5 states

// Packed lower-triangular index helpers: entry (i, j), j <= i, lives at i*(i+1)/2 + j.
__device__ __forceinline__ constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }

// Cholesky factor of a symmetric 5x5 matrix given by its packed lower triangle (fp32, like LAPACK spotrf on
// the fp32 covariances).  Returns false when a pivot is not positive.
// Round 4: the factorisation starts from the hardware's square root and reciprocal (v_sqrt_f32, v_rcp_f32) and CORRECTS them
// -- the root by one residual step, the reciprocal of L_jj by one Newton step, every quotient s / L_jj by one residual step
// (Markstein) -- five reciprocals for ten divisions.  The general sqrtf and division are long sequences on this hardware
// (range scaling, Newton steps, fix-up, with their branches: ~35 instructions each), 685 of the 3 937 vector instructions of
// the one-lane-per-Gaussian cycle kernel, which is bound by exactly that count; the corrected estimates keep 550 of the
// saving and give the correctly rounded results but for rare last-bit cases (GSE_CHOL_FAST = 1, the raw estimates, moves
// one weight in a thousand by 1e-4: the update of the mean amplifies an ulp of L).  One definition for every kernel, so
// they agree with each other to the bit whatever the setting.
#ifndef GSE_CHOL_FAST
#define GSE_CHOL_FAST 2          // 0: sqrtf and division (rounds 1-3); 1: the hardware's estimates as they are; 2: corrected
#endif
__device__ __forceinline__ bool chol5(const float (&A)[kTri], float (&L)[kTri]) {
    // (no contraction into FMAs left to the compiler: the same source inlined into different kernels must give the same
    //  bits, and whether `s - a * b` is fused depends on the surrounding code)
#pragma clang fp contract(off)
    bool ok = true;
#if GSE_CHOL_FAST
    float inv[kNx];                                  // 1 / L_jj, once per column (five reciprocals for ten divisions)
#endif
#pragma unroll
    for (int i = 0; i < kNx; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            float s = A[tri(i, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= L[tri(i, k)] * L[tri(j, k)];
            if (i == j) {
                ok = ok && (s > 0.0f);
#if GSE_CHOL_FAST == 2
                // the hardware's estimates, corrected: sqrt by one residual step (g + (s - g g) / (2 g)), the reciprocal by one
                // Newton step -- correctly rounded but for rare last-bit cases, without the general routines' range handling
                {
                    const float g0 = __builtin_amdgcn_sqrtf(s);
                    const float r0 = __builtin_amdgcn_rcpf(g0);
                    const float g1 = fmaf(fmaf(-g0, g0, s), 0.5f * r0, g0);
                    L[tri(i, i)] = (s > 0.0f) ? g1 : g0;
                    const float rr = __builtin_amdgcn_rcpf(L[tri(i, i)]);
                    inv[i] = fmaf(rr, fmaf(-L[tri(i, i)], rr, 1.0f), rr);
                }
#elif GSE_CHOL_FAST
                L[tri(i, i)] = __builtin_amdgcn_sqrtf(s);
                inv[i] = __builtin_amdgcn_rcpf(L[tri(i, i)]);
#else
                L[tri(i, i)] = sqrtf(s);
#endif
            } else {
#if GSE_CHOL_FAST == 2
                {                                    // q = s / d from the refined reciprocal, one residual correction (Markstein)
                    const float q0 = s * inv[j];
                    L[tri(i, j)] = fmaf(fmaf(-L[tri(j, j)], q0, s), inv[j], q0);
                }
#elif GSE_CHOL_FAST
                L[tri(i, j)] = s * inv[j];
#else
                L[tri(i, j)] = s / L[tri(j, j)];
#endif
            }
        }
    }
    return ok;
}

// Loads (mu, lower triangle of P) of one Gaussian and factorises P, retrying once with 1e-10 on the diagonal
// (filter/gs_ukf.py:72-75).  ok = false if the matrix is still not positive definite.
__device__ __forceinline__ void load_and_factor(const float* __restrict__ mu_g, const float* __restrict__ P_g,
                                                float (&mu)[kNx], float (&L)[kTri], bool& ok) {
    float A[kTri];
#pragma unroll
    for (int k = 0; k < kNx; ++k) mu[k] = mu_g[k];
#pragma unroll
    for (int i = 0; i < kNx; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) A[tri(i, j)] = P_g[i * kNx + j];
    ok = chol5(A, L);
    if (!ok) {
#pragma unroll
        for (int i = 0; i < kNx; ++i) A[tri(i, i)] += 1e-10f;
        ok = chol5(A, L);
    }
}

// Sigma point j (0..2 kNx) of N(mu, L L^T): mu, mu + L[:, c], mu - L[:, c]   (filter/gs_ukf.py:76-80).  s is a row as the
// model functions take it: GSE_NX wide, states the model does not have stay zero.
__device__ __forceinline__ void sigma_point(const float (&mu)[kNx], const float (&L)[kTri], int j, float (&s)[GSE_NX]) {
    const int c = (j == 0) ? -1 : (j - 1) % kNx;
    const float sign = (j <= kNx) ? 1.0f : -1.0f;
#pragma unroll
    for (int k = kNx; k < GSE_NX; ++k) s[k] = 0.0f;
#pragma unroll
    for (int k = 0; k < kNx; ++k) {
        float col = 0.0f;
#pragma unroll
        for (int cc = 0; cc <= k; ++cc) col = (c == cc) ? L[tri(k, cc)] : col;
        // Column c of the LOWER-triangular factor is zero above row c: those states of the sigma point ARE the mean's.  Said so
        // (rather than adding a signed zero the compiler may not drop), the one-lane-per-Gaussian kernels -- j is a constant
        // in their unrolled loops -- evaluate whatever part of the model depends on such states once per Gaussian instead of
        // once per sigma point: for the bioreactor the Monod quotient and the Cx-only rate limits in 6 of the 11 points.
        // (Bits: mu + (+-0) = mu except that -0 + 0 = +0; a mean component that is exactly -0 does not occur.)
        s[k] = (c < 0 || c > k) ? mu[k] : mu[k] + sign * col;
    }
}

struct SigmaWeights {
    float w[GSE_NSIGMA];
    float scl32;          // sum(w) in fp32  -- the divisor numpy.average uses for fp32 data
    double scl64;         // sum(w) accumulated in fp64 -- its divisor for fp64 data
};

#ifndef __HIPCC_RTC__            // ---- host only
// numpy.average divides by sum(w): pairwise_sum's order for n values (n < 8: left to right; 8 <= n < 128: eight interleaved
// accumulators folded as ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)), then the n % 8 remaining values in order).
template <class T>
inline T numpy_sum(const float* a, int n) {
    if (n < 8) {
        T res = (T)0;
        for (int i = 0; i < n; ++i) res += (T)a[i];
        return res;
    }
    T r[8];
    for (int k = 0; k < 8; ++k) r[k] = (T)a[k];
    int i = 8;
    for (; i < n - (n % 8); i += 8)
        for (int k = 0; k < 8; ++k) r[k] += (T)a[i + k];
    T res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += (T)a[i];
    return res;
}

inline void pack_sigma_weights(SigmaWeights& sw, const float* w_sigma, int ns) {
    for (int i = 0; i < GSE_NSIGMA; ++i) sw.w[i] = i < ns ? w_sigma[i] : 0.0f;
    sw.scl32 = numpy_sum<float>(w_sigma, ns);
    sw.scl64 = numpy_sum<double>(w_sigma, ns);
}

#endif

// ------------------------------------------------------------------------------------------------
template <int kUnused>
__global__ __launch_bounds__(kBlock) void gsukf_sigma_kernel(const float* __restrict__ mu, const float* __restrict__ P,
                                                             float* __restrict__ sigmas, int64_t N,
                                                             int32_t* __restrict__ status) {
    const int b = blockIdx.y;
    const int j = threadIdx.x & (kGroup - 1);
    const int64_t g = (int64_t)blockIdx.x * kGaussPerBlock + (threadIdx.x / kGroup);
    if (g >= N) return;
    const int64_t gi = (int64_t)b * N + g;
    float m[kNx], L[kTri], s[GSE_NX];
    bool ok;
    load_and_factor(mu + gi * kNx, P + gi * kNxx, m, L, ok);
    if (j >= kNs) return;
    sigma_point(m, L, j, s);
    if (!ok && j == 0 && status) atomicAdd(status, 1);
    float* out = sigmas + (gi * kNs + j) * kNx;
#pragma unroll
    for (int k = 0; k < kNx; ++k) out[k] = s[k];
}

// ------------------------------------------------------------------------------------------------
// Inputs and measurements of the banks: host values carried in the kernel arguments (io), or -- for a closed loop whose
// controller and plant run on the device -- [B][2] fp64 arrays in device memory, read when the kernel runs.
__device__ __forceinline__ FilterIO bank_inputs(const FilterIOs& io, const double* u_dev, const double* z_dev, int b) {
    FilterIO f = io.f[b];
    if (u_dev) { f.u0 = u_dev[2 * b]; f.u1 = u_dev[2 * b + 1]; }
    if (z_dev) { f.z0 = z_dev[2 * b]; f.z1 = z_dev[2 * b + 1]; }
    return f;
}

struct GsukfPredictArgs {
    FilterIOs io;
    const double* u_dev;        // null: io
    const double* z_dev;        // (kernels that also update)
    DrawParams draw;
    RngKey rng;
    SigmaWeights sw;
    double dt;
};

// Mean and covariance of the 11 propagated sigma points of one Gaussian (filter/gs_ukf.py:96-103), shared by every
// kernel that predicts: mean = (sum_j sigma_j w_j) / sum(w) in fp32, j in order, no FMA contraction (numpy.average);
// P = sum_j d_j (d_j w_j)^T entry by entry of the upper triangle, mirrored.  P_o: kNx x kNx floats, row-major.
__device__ __forceinline__ void sigma_mean_cov(const float (&sg)[kNs * kNx], const SigmaWeights& sw, float (&mean)[kNx],
                                               float* P_o) {
    {
#pragma clang fp contract(off)
#pragma unroll
        for (int k = 0; k < kNx; ++k) {
            float acc = sg[k] * sw.w[0];
#pragma unroll
            for (int q = 1; q < kNs; ++q) acc = acc + sg[q * kNx + k] * sw.w[q];
            mean[k] = acc;
        }
        // (the usual sigma weights sum to exactly 1 in fp32, and x / 1 = x: one scalar branch around five IEEE divisions)
        if (sw.scl32 != 1.0f) {
#pragma unroll
            for (int k = 0; k < kNx; ++k) mean[k] = mean[k] / sw.scl32;
        }
    }
#pragma unroll
    for (int r = 0; r < kNx; ++r)
#pragma unroll
        for (int c = r; c < kNx; ++c) {
            float acc = 0.0f;
#pragma unroll
            for (int q = 0; q < kNs; ++q) {
                const float dr = sg[q * kNx + r] - mean[r];
                const float dc = sg[q * kNx + c] - mean[c];
                acc = fmaf(dr, dc * sw.w[q], acc);
            }
            P_o[r * kNx + c] = acc;
            P_o[c * kNx + r] = acc;
        }
}

// ------------------------------------------------------------------------------------------------
// The state noise of the sigma points of Gaussian g (filter/gs_ukf.py:87-91 draws once per sigma point).  Round 3: the points
// come in PAIRS (2i, 2i + 1) that share THREE Philox blocks -- 12 words for ten normals and two selectors -- where every point
// used to take two blocks of its own and leave a word and a normal unused: 17 blocks and 28 Box-Muller pairs per Gaussian
// instead of 22 and 33, in kernels that are bound by exactly such instructions.  Counter element g * 11 + 2i, sub-blocks
//   A (sub 0): two Box-Muller pairs = normals 0..3 of point 2i,        C (sub 2): the same for point 2i + 1,
//   B (sub 1): words 0, 1 one Box-Muller pair whose cosine half is normal 4 of point 2i and whose sine half normal 4 of
//              point 2i + 1;  word 3 selects the component of point 2i, word 2 that of point 2i + 1 (top 24 bits);
// the last point (no partner) is the even half of a pair.  So point j reads its own block (sub 2 (j & 1)) and block B of
// element g * 11 + (j & ~1): two blocks for a lane that holds one sigma point, as before.  (oracle: gsukf_predict_noise.)
// ------------------------------------------------------------------------------------------------
template <int kDiagonal = -1>
__device__ __forceinline__ void gsukf_point_noise(const DrawParams& dp, const RngKey& rng, int64_t g, int j, uint32_t fb,
                                                  float (&e)[GSE_NX]) {
    const uint32_t odd = (uint32_t)j & 1u;
    const uint64_t elem = (uint64_t)(g * kNs + (j & ~1));
    uint32_t own[4], shared[4];
    rng_block_of_elem(rng, elem, fb, 2u * odd, own);
    rng_block_of_elem(rng, elem, fb, 1u, shared);
    float n[6], c4, s4;
    box_muller(own[0], own[1], n[0], n[1]);
    box_muller(own[2], own[3], n[2], n[3]);
    box_muller(shared[0], shared[1], c4, s4);
    n[4] = odd ? s4 : c4;
    mixture_transform5<kDiagonal>(dp, odd ? shared[2] : shared[3], n, e);
}
// both points of pair i at once (a lane that holds the whole Gaussian): three blocks, five Box-Muller pairs
template <int kDiagonal>
__device__ __forceinline__ void gsukf_pair_noise(const DrawParams& dp, const RngKey& rng, int64_t g, int i, uint32_t fb,
                                                 float (&e0)[GSE_NX], float (&e1)[GSE_NX]) {
    const uint64_t elem = (uint64_t)(g * kNs + 2 * i);
    uint32_t A[4], B[4], C[4];
    rng_block_of_elem(rng, elem, fb, 0u, A);
    rng_block_of_elem(rng, elem, fb, 1u, B);
    rng_block_of_elem(rng, elem, fb, 2u, C);
    float n0[5], n1[5];
    box_muller(A[0], A[1], n0[0], n0[1]);
    box_muller(A[2], A[3], n0[2], n0[3]);
    box_muller(C[0], C[1], n1[0], n1[1]);
    box_muller(C[2], C[3], n1[2], n1[3]);
    box_muller(B[0], B[1], n0[4], n1[4]);
    mixture_transform5<kDiagonal>(dp, B[3], n0, e0);
    mixture_transform5<kDiagonal>(dp, B[2], n1, e1);
}

// Phase A of predict and update: lane t < nloc of the workgroup factorises local Gaussian t and leaves (m, L) in fac[t].
// Src: t -> (mu pointer, P pointer); global memory or LDS.  Ends with a workgroup barrier.
template <class Src>
__device__ __forceinline__ void gsukf_factor_block(Src src, float (*fac)[kFacStride], int nloc, int32_t* status) {
    const int t = threadIdx.x;
    if (t < nloc) {
        float m[kNx], L[kTri];
        bool ok;
        const float *mu_g, *P_g;
        src(t, mu_g, P_g);
        load_and_factor(mu_g, P_g, m, L, ok);
        if (!ok && status) atomicAdd(status, 1);
#pragma unroll
        for (int k = 0; k < kNx; ++k) fac[t][k] = m[k];
#pragma unroll
        for (int k = 0; k < kTri; ++k) fac[t][kNx + k] = L[k];
    }
    __syncthreads();
}

__device__ __forceinline__ void read_factor(const float* fac_g, float (&m)[kNx], float (&L)[kTri]) {
#pragma unroll
    for (int k = 0; k < kNx; ++k) m[k] = fac_g[k];
#pragma unroll
    for (int k = 0; k < kTri; ++k) L[k] = fac_g[kNx + k];
}

// Predict of the workgroup's nloc Gaussians (local index gl = tid / 16, lane j = tid % 16).  src / dst: local index ->
// where (mu, P) are read / written (may be the same memory: every read precedes a barrier that every write follows).
// g0: filter index of local Gaussian 0 (RNG element), noise: injected rows [N][11][5] of this filter or null.
// Must be called by every thread of the workgroup.
// meanwhile: called by every thread between phases B and C (after the barrier): what it does on lanes other than the first 64
// runs beside phase C, which occupies one wavefront (the persistent loop puts the plant's step there).
struct NothingMeanwhile {
    __device__ __forceinline__ void operator()() const {}
};
template <bool kInjected, class Src, class Dst, class Meanwhile = NothingMeanwhile>
__device__ __forceinline__ void gsukf_predict_block(Src src, Dst dst, float (*fac)[kFacStride], float (*tile)[kTileStride],
                                                    int nloc, int b, int64_t g0, const float* noise, const DrawParams& dp,
                                                    int32_t* status, const GsukfPredictArgs& a, const FilterIO& bio,
                                                    const RngKey& rng, Meanwhile meanwhile = Meanwhile()) {
    // rng: a.rng for a launch that predicts once; a kernel that predicts tick after tick (gse_loop.hip) hands in the key of
    // the tick
    gsukf_factor_block(src, fac, nloc, status);
    // B. one sigma point per lane, packed densely (11 per Gaussian: 704 points keep 11 of the 16 wavefronts busy, the
    //    other five skip the model step and the noise draw altogether)
    const int gl = (int)threadIdx.x / kNs, j = (int)threadIdx.x - gl * kNs;
    if (gl < nloc) {
        float m[kNx], L[kTri], s[GSE_NX];
        read_factor(fac[gl], m, L);
        sigma_point(m, L, j, s);
        model_step(s, bio.u0, bio.u1, a.dt);
        float e[GSE_NX];
        if (kInjected) {
            const float* nz = noise + (g0 + gl) * kNs * kNx;
#pragma unroll
            for (int k = 0; k < kNx; ++k) e[k] = nz[j * kNx + k];
        } else {
            gsukf_point_noise(dp, rng, g0 + gl, j, (uint32_t)b, e);
        }
#pragma unroll
        for (int k = 0; k < kNx; ++k) tile[gl][j * kNx + k] = s[k] + e[k];
    }
    __syncthreads();
    meanwhile();
    // C. one Gaussian per lane: mean and covariance of its sigma points
    const int t = threadIdx.x;
    if (t < nloc) {
        float sg[kNs * kNx];
#pragma unroll
        for (int i = 0; i < kNs * kNx; ++i) sg[i] = tile[t][i];
        float mean[kNx];
        float *mu_o, *P_o;
        dst(t, mu_o, P_o);
        sigma_mean_cov(sg, a.sw, mean, P_o);
#pragma unroll
        for (int k = 0; k < kNx; ++k) mu_o[k] = mean[k];
    }
}

template <bool kInjected, int kBankT>
__global__ __launch_bounds__(kBankT * kGroup) void gsukf_predict_kernel(float* __restrict__ mu, float* __restrict__ P,
                                                                        const float* __restrict__ noise, int64_t N,
                                                                        int32_t* __restrict__ status,
                                                                        const GsukfPredictArgs a) {
    __shared__ DrawParams dp;
    __shared__ float fac[kBankT][kFacStride];
    __shared__ float tile[kBankT][kTileStride];
    if (!kInjected) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&a.draw);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&dp);
        for (int i = threadIdx.x; i < (int)(sizeof(DrawParams) / 4); i += kBankT * kGroup) dst[i] = src[i];
        __syncthreads();
    }
    const int b = blockIdx.y;
    const int64_t g0 = (int64_t)blockIdx.x * kBankT;
    const int nloc = (int)min((int64_t)kBankT, N - g0);
    float* mu_b = mu + ((int64_t)b * N + g0) * kNx;
    float* P_b = P + ((int64_t)b * N + g0) * kNxx;
    auto src = [&](int t, const float*& m_, const float*& p_) { m_ = mu_b + t * kNx; p_ = P_b + t * kNxx; };
    auto dst = [&](int t, float*& m_, float*& p_) { m_ = mu_b + t * kNx; p_ = P_b + t * kNxx; };
    gsukf_predict_block<kInjected>(src, dst, fac, tile, nloc, b, g0, kInjected ? noise + (int64_t)b * N * kNs * kNx : nullptr,
                                   dp, status, a, bank_inputs(a.io, a.u_dev, a.z_dev, b), a.rng);
}

// ------------------------------------------------------------------------------------------------
struct GsukfUpdateArgs {
    FilterIOs io;
    const double* u_dev;
    const double* z_dev;
    Pdf2 pdf;
    SigmaWeights sw;
};

// Pseudo-inverse of a symmetric 2x2 matrix with numpy.linalg.pinv's default cut-off (singular values
// <= 1e-15 * largest are dropped; filter/gs_ukf.py:132).
__device__ __forceinline__ void pinv_sym2(double a, double s, double d, double (&X)[2][2]) {
#pragma clang fp contract(off)
    const double half_tr = 0.5 * (a + d);
    const double disc = sqrt(0.25 * (a - d) * (a - d) + s * s);
    const double big = (fabs(half_tr + disc) >= fabs(half_tr - disc)) ? half_tr + disc : half_tr - disc;
    const double det = a * d - s * s;
    const double small = (big != 0.0) ? det / big : 0.0;
    const double cut = 1e-15 * fabs(big);
    if (!(fabs(big) > 0.0)) {
        X[0][0] = X[0][1] = X[1][0] = X[1][1] = 0.0;
    } else if (fabs(small) > cut) {
        const double inv = 1.0 / det;
        X[0][0] = d * inv; X[0][1] = -s * inv; X[1][0] = -s * inv; X[1][1] = a * inv;
    } else {                                   // rank one: v v^T / big
        double v0 = s, v1 = big - a;
        const double w0 = big - d, w1 = s;
        if (w0 * w0 + w1 * w1 > v0 * v0 + v1 * v1) { v0 = w0; v1 = w1; }
        const double n2 = v0 * v0 + v1 * v1;
        const double sc = (n2 > 0.0) ? 1.0 / (n2 * big) : 0.0;
        X[0][0] = v0 * v0 * sc; X[0][1] = v0 * v1 * sc; X[1][0] = X[0][1]; X[1][1] = v1 * v1 * sc;
    }
}

// The per-Gaussian algebra of the update (filter/gs_ukf.py:116-149): eta_mean, P_yy, P_xy in fp64 with j in order (the CPU
// class keeps the etas in fp64), K = P_xy pinv(P_yy), mu += K (z - eta_mean), P -= (K P_yy) K^T, w *= pdf(z - g(mu_new)).
// eta(q, c): output c of sigma point q (fp64); dev(q, k): sigma_q[k] - mu[k] (fp32); m: the prior mean; P_in / P_o: kNx x kNx
// floats row-major (may be the same memory: every entry is read before it is written).  kUnrolled: the 11-term loop is unrolled
// (operands in registers) or kept as a loop (operands in LDS: unrolled, the compiler hoists all 77 reads and spills).
// One definition for every kernel that updates, so they agree to the bit.
template <bool kUnrolled, class Eta, class Dev>
__device__ __forceinline__ void gsukf_update_gaussian(Eta eta, Dev dev, const float* m, const float* P_in, float* mu_o,
                                                      float* P_o, const float* w_in, float* w_out, double u0, double u1,
                                                      double z0, double z1, const Pdf2& pdf, const SigmaWeights& sw) {
    // (only the fma() calls written out below are fused: see chol5)
#pragma clang fp contract(off)
    double em[2];
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) {
#pragma clang fp contract(off)
        double acc = eta(0, q2) * (double)sw.w[0];
#pragma unroll
        for (int q = 1; q < kNs; ++q) acc = acc + eta(q, q2) * (double)sw.w[q];
        em[q2] = acc / sw.scl64;
    }
    double pyy[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    double pxy[kNx][2];
#pragma unroll
    for (int k = 0; k < kNx; ++k) pxy[k][0] = pxy[k][1] = 0.0;
    auto term = [&](int q) {
        const double wq = (double)sw.w[q];
        const double e0 = eta(q, 0) - em[0], e1 = eta(q, 1) - em[1];
        const double ew0 = e0 * wq, ew1 = e1 * wq;
        pyy[0][0] = fma(e0, ew0, pyy[0][0]);
        pyy[0][1] = fma(e0, ew1, pyy[0][1]);
        pyy[1][0] = fma(e1, ew0, pyy[1][0]);
        pyy[1][1] = fma(e1, ew1, pyy[1][1]);
#pragma unroll
        for (int k = 0; k < kNx; ++k) {
            // Sigma point q is the mean in the states above its column of the lower-triangular factor (sigma_point): there
            // the deviation is m - m = +0, the product a signed zero, and pxy -- which starts at +0 and is never -0 -- keeps
            // its bits.  With q a constant (the unrolled form of the one-lane-per-Gaussian kernels) 25 of the 55 pairs
            // drop out, a conversion and two fp64 FMAs each; the rolled form multiplies by the zero it read.
            if (kUnrolled && (q == 0 || (q - 1) % kNx > k)) continue;
            const double dk = (double)dev(q, k);
            pxy[k][0] = fma(dk, ew0, pxy[k][0]);
            pxy[k][1] = fma(dk, ew1, pxy[k][1]);
        }
    };
    if constexpr (kUnrolled) {
#pragma unroll
        for (int q = 0; q < kNs; ++q) term(q);
    } else {
#pragma unroll 1
        for (int q = 0; q < kNs; ++q) term(q);
    }
    double X[2][2];
    pinv_sym2(pyy[0][0], 0.5 * (pyy[0][1] + pyy[1][0]), pyy[1][1], X);
    double K[kNx][2];
#pragma unroll
    for (int k = 0; k < kNx; ++k) {
        K[k][0] = pxy[k][0] * X[0][0] + pxy[k][1] * X[1][0];
        K[k][1] = pxy[k][0] * X[0][1] + pxy[k][1] * X[1][1];
    }
    const double i0 = z0 - em[0], i1 = z1 - em[1];
    float mnew[GSE_NX];                       // (a row as the model's g takes it: states it does not have stay zero)
#pragma unroll
    for (int k = kNx; k < GSE_NX; ++k) mnew[k] = 0.0f;
#pragma unroll
    for (int k = 0; k < kNx; ++k) mnew[k] = (float)((double)m[k] + (K[k][0] * i0 + K[k][1] * i1));
    // P -= (K P_yy) K^T, entry (r, c) and its mirror (which uses (K P_yy)[c] . K[r]; identical when P_yy is symmetric to
    // rounding).  Every entry is read before it is written, and by this lane only.
#pragma unroll
    for (int r = 0; r < kNx; ++r) {
        const double t0 = K[r][0] * pyy[0][0] + K[r][1] * pyy[1][0];     // row r of K P_yy
        const double t1 = K[r][0] * pyy[0][1] + K[r][1] * pyy[1][1];
#pragma unroll
        for (int c = r; c < kNx; ++c) {
            const float prc = (float)((double)P_in[r * kNx + c] - (t0 * K[c][0] + t1 * K[c][1]));
            float pcr = prc;
            if (r != c) {
                const double v0 = K[c][0] * pyy[0][0] + K[c][1] * pyy[1][0];
                const double v1 = K[c][0] * pyy[0][1] + K[c][1] * pyy[1][1];
                pcr = (float)((double)P_in[c * kNx + r] - (v0 * K[r][0] + v1 * K[r][1]));
            }
            P_o[r * kNx + c] = prc;
            if (r != c) P_o[c * kNx + r] = pcr;
        }
    }
#pragma unroll
    for (int k = 0; k < kNx; ++k) mu_o[k] = mnew[k];
    // global update: w *= pdf(z - g(mu_new))
    float y0, y1;
    model_output(mnew, u0, u1, y0, y1);
    const double lik = mixture_pdf2<0>(pdf, z0 - (double)y0, z1 - (double)y1);
    *w_out = (float)((double)(*w_in) * lik);
}

// Update of the workgroup's nloc Gaussians, in place where src / dst point (global memory or LDS); weights are read
// through w_in(t) and written through w_out(t).  Must be called by every thread of the workgroup.
template <class Src, class Dst, class Win, class Wout>
__device__ __forceinline__ void gsukf_update_block(Src src, Dst dst, Win w_in, Wout w_out, float (*fac)[kFacStride],
                                                   float (*dev)[kTileStride], double (*eta)[kEtaStride], int nloc, int b,
                                                   int32_t* status, const FilterIO& bio, const Pdf2& pdf,
                                                   const SigmaWeights& sw) {
    gsukf_factor_block(src, fac, nloc, status);
    // B. one sigma point per lane: outputs and deviations
    const int j = threadIdx.x & (kGroup - 1), gl = threadIdx.x / kGroup;
    if (gl < nloc && j < kNs) {
        float m[kNx], L[kTri], s[GSE_NX];
        read_factor(fac[gl], m, L);
        sigma_point(m, L, j, s);
        float y0, y1;
        model_output(s, bio.u0, bio.u1, y0, y1);
        eta[gl][j * 2 + 0] = (double)y0;
        eta[gl][j * 2 + 1] = (double)y1;
#pragma unroll
        for (int k = 0; k < kNx; ++k) dev[gl][j * kNx + k] = s[k] - m[k];
    }
    __syncthreads();
    // C. one Gaussian per lane
    const int t = threadIdx.x;
    if (t >= nloc) return;
    const double* eta_g = eta[t];
    const float* dev_g = dev[t];
    const float *mu_g, *P_g;
    src(t, mu_g, P_g);
    float *mu_o, *P_o;
    dst(t, mu_o, P_o);
    gsukf_update_gaussian<false>([&](int q, int c) { return eta_g[q * 2 + c]; }, [&](int q, int k) { return dev_g[q * kNx + k]; },
                                 fac[t], P_g, mu_o, P_o, w_in(t), w_out(t), bio.u0, bio.u1, bio.z0, bio.z1, pdf,
                                 sw);
}

template <int kBankT>
__global__ __launch_bounds__(kBankT * kGroup) void gsukf_update_kernel(float* __restrict__ mu, float* __restrict__ P,
                                                                       float* __restrict__ w, int64_t N,
                                                                       int32_t* __restrict__ status,
                                                                       const GsukfUpdateArgs a) {
    __shared__ float fac[kBankT][kFacStride];
    __shared__ float dev[kBankT][kTileStride];       // sigma_j - mu
    __shared__ double eta[kBankT][kEtaStride];       // g(sigma_j)
    const int b = blockIdx.y;
    const int64_t g0 = (int64_t)blockIdx.x * kBankT;
    const int nloc = (int)min((int64_t)kBankT, N - g0);
    float* mu_b = mu + ((int64_t)b * N + g0) * kNx;
    float* P_b = P + ((int64_t)b * N + g0) * kNxx;
    float* w_b = w + (int64_t)b * N + g0;
    auto src = [&](int t, const float*& m_, const float*& p_) { m_ = mu_b + t * kNx; p_ = P_b + t * kNxx; };
    auto dst = [&](int t, float*& m_, float*& p_) { m_ = mu_b + t * kNx; p_ = P_b + t * kNxx; };
    gsukf_update_block(src, dst, [&](int t) { return (const float*)(w_b + t); }, [&](int t) { return w_b + t; }, fac, dev, eta,
                       nloc, b, status, bank_inputs(a.io, a.u_dev, a.z_dev, b), a.pdf, a.sw);
}

// ------------------------------------------------------------------------------------------------
// Large banks: ONE LANE PER GAUSSIAN.  The kernels above give a Gaussian 16 lanes for its 11 sigma points and one lane for
// its algebra, handing over through LDS -- right for a few hundred Gaussians (parallelism is scarce, latency rules), wrong
// for 2^18: the single-lane phases run in one wavefront of sixteen while the workgroup holds a whole CU (round 1: 143 us
// per cycle, 0.17 of the HBM roofline).  Here a workgroup of 256 threads takes 256 Gaussians and a lane does everything for
// its own: factorisation, the 11 sigma points one after the other (model step + noise draw, or output function), the
// contractions -- all in registers, no hand-off, no barrier inside, every lane busy in every phase.  A lane's row (mu, P: 30
// floats) lives in LDS between the phases, stride 31 floats (conflict-free when a lane walks its row), and the lane itself
// moves it from and to global memory (lane_row_in / lane_row_out below; rounds 1-2 staged the workgroup's rows cooperatively,
// stage_bank_*: kept for the row-copy kernel and as GSE_GSUKF_DIRECT_ROWS=0).  Same device functions, same sigma-point
// order and the same random-number elements as the kernels above: results agree bit for bit.
constexpr int kLaneBank = 256;
constexpr int kStageStride = kRow | 1;           // mu + P of a Gaussian, padded to an odd stride (31 floats for 5 states)

__device__ __forceinline__ void stage_bank_in(float* stage, const float* mu_b, const float* P_b, int nloc) {
    for (int i = threadIdx.x; i < nloc * kNx; i += kLaneBank) {
        const int g = i / kNx;
        stage[g * kStageStride + (i - g * kNx)] = mu_b[i];
    }
    for (int i = threadIdx.x; i < nloc * kNxx; i += kLaneBank) {
        const int g = i / kNxx;
        stage[g * kStageStride + kNx + (i - g * kNxx)] = P_b[i];
    }
    __syncthreads();
}
// The same with the rows of the bank's ANCESTORS (a resample whose copy was deferred into this launch): local Gaussian g is
// the row anc[g] of the filter's arrays mu_f / P_f; a row's 30 floats are still read by consecutive threads.
__device__ __forceinline__ void stage_bank_in_gathered(float* stage, int* anc_s, const float* mu_f, const float* P_f,
                                                       const int32_t* anc, int nloc) {
    for (int g = threadIdx.x; g < nloc; g += kLaneBank) anc_s[g] = anc[g];
    __syncthreads();
    for (int i = threadIdx.x; i < nloc * kNx; i += kLaneBank) {
        const int g = i / kNx;
        stage[g * kStageStride + (i - g * kNx)] = mu_f[(int64_t)anc_s[g] * kNx + (i - g * kNx)];
    }
    for (int i = threadIdx.x; i < nloc * kNxx; i += kLaneBank) {
        const int g = i / kNxx;
        stage[g * kStageStride + kNx + (i - g * kNxx)] = P_f[(int64_t)anc_s[g] * kNxx + (i - g * kNxx)];
    }
    __syncthreads();
}
__device__ __forceinline__ void stage_bank_out(const float* stage, float* mu_b, float* P_b, int nloc) {
    __syncthreads();
    for (int i = threadIdx.x; i < nloc * kNx; i += kLaneBank) {
        const int g = i / kNx;
        mu_b[i] = stage[g * kStageStride + (i - g * kNx)];
    }
    for (int i = threadIdx.x; i < nloc * kNxx; i += kLaneBank) {
        const int g = i / kNxx;
        P_b[i] = stage[g * kStageStride + kNx + (i - g * kNxx)];
    }
}

// A lane's OWN row between global memory and its LDS row, without the workgroup: a Gaussian's 5 + 25 floats are contiguous
// in the filter's arrays, dword-aligned, and global memory takes 16-byte accesses at dword alignment -- eight loads (stores)
// per lane and no index arithmetic, where the staged copy above spends 10 vector instructions per element on i / 5, i / 25 and
// the padded LDS address (615 of the 4 467 a wavefront of the fused predict + update issues, a kernel bound by exactly that).
// The accesses of a wavefront are rows 20 / 100 bytes apart: every cache line is used whole, by several instructions, which
// the texture path -- idle in these kernels -- absorbs.
#ifndef GSE_GSUKF_DIRECT_ROWS
#define GSE_GSUKF_DIRECT_ROWS 1
#endif
__device__ __forceinline__ void lane_row_in(float* mine, const float* __restrict__ mu_row, const float* __restrict__ P_row) {
    float r[kRow];
    __builtin_memcpy(r, mu_row, kNx * sizeof(float));
    __builtin_memcpy(r + kNx, P_row, kNxx * sizeof(float));
#pragma unroll
    for (int k = 0; k < kRow; ++k) mine[k] = r[k];
}
__device__ __forceinline__ void lane_row_out(const float* mine, float* __restrict__ mu_row, float* __restrict__ P_row) {
    float r[kRow];
#pragma unroll
    for (int k = 0; k < kRow; ++k) r[k] = mine[k];
    __builtin_memcpy(mu_row, r, kNx * sizeof(float));
    __builtin_memcpy(P_row, r + kNx, kNxx * sizeof(float));
}

#ifndef GSE_GSUKF_PRED_WAVES
#define GSE_GSUKF_PRED_WAVES 1   // waves per SIMD the register allocator must leave room for (tuning knob)
#endif
#ifndef GSE_GSUKF_PRED_SERIAL
#define GSE_GSUKF_PRED_SERIAL 0  // 1: a scheduling barrier between the sigma points of a lane (tuning knob)
#endif
// predict of ONE Gaussian: read from (mu_src, P_src) -- global memory or the lane's LDS row: the mean and the lower triangle are
// all it needs -- and left in `mine` (mu[kNx], P[kNx][kNx]); g: its index in the filter (random-number element)
// kDiagonal: what the launch knows about the noise mixture's factors (1 all diagonal, 0 not, -1 look at the flag in each draw)
template <bool kInjected, int kDiagonal = -1>
__device__ __forceinline__ void gsukf_predict_lane(const float* __restrict__ mu_src, const float* __restrict__ P_src, float* mine,
                                                   int b, int64_t g, int64_t N, const float* noise, const DrawParams& dp,
                                                   int32_t* status, const GsukfPredictArgs& a, const FilterIO& bio) {
    float m[kNx], L[kTri];
    bool ok;
    load_and_factor(mu_src, P_src, m, L, ok);
    if (!ok && status) atomicAdd(status, 1);
    float sg[kNs * kNx];
    // the sigma points two at a time: a pair shares its Philox blocks (gsukf_pair_noise); the last one is half a pair
#pragma unroll
    for (int i = 0; i < (kNs + 1) / 2; ++i) {
        const int j0 = 2 * i, j1 = 2 * i + 1;
        const bool both = j1 < kNs;
        float s0[GSE_NX], s1[GSE_NX], e0[GSE_NX], e1[GSE_NX];
        sigma_point(m, L, j0, s0);
        model_step(s0, bio.u0, bio.u1, a.dt);
        if (both) {
            sigma_point(m, L, j1, s1);
            model_step(s1, bio.u0, bio.u1, a.dt);
        }
        if (kInjected) {
            const float* nz = noise + (((int64_t)b * N + g) * kNs + j0) * kNx;
#pragma unroll
            for (int k = 0; k < kNx; ++k) e0[k] = nz[k];
            if (both) {
#pragma unroll
                for (int k = 0; k < kNx; ++k) e1[k] = nz[kNx + k];
            }
        } else if (both) {
            gsukf_pair_noise<kDiagonal>(dp, a.rng, g, i, (uint32_t)b, e0, e1);
        } else {
            gsukf_point_noise<kDiagonal>(dp, a.rng, g, j0, (uint32_t)b, e0);
        }
#pragma unroll
        for (int k = 0; k < kNx; ++k) sg[j0 * kNx + k] = s0[k] + e0[k];
        if (both) {
#pragma unroll
            for (int k = 0; k < kNx; ++k) sg[j1 * kNx + k] = s1[k] + e1[k];
        }
#if GSE_GSUKF_PRED_SERIAL
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
    float mean[kNx];
    sigma_mean_cov(sg, a.sw, mean, mine + kNx);
#pragma unroll
    for (int k = 0; k < kNx; ++k) mine[k] = mean[k];
}

// update of ONE Gaussian held in `mine`; its weight at *w
__device__ __forceinline__ void gsukf_update_lane(float* mine, float* w, double u0, double u1, double z0, double z1,
                                                  int32_t* status, const Pdf2& pdf, const SigmaWeights& sw) {
    float m[kNx], L[kTri];
    bool ok;
    load_and_factor(mine, mine + kNx, m, L, ok);
    if (!ok && status) atomicAdd(status, 1);
    // outputs of the 11 sigma points (fp64, as the kernels above keep them); the deviations sigma_j - mu are re-formed
    // from the factor where the contraction needs them (a select and a subtraction) instead of being kept
    double eta[kNs][2];
#pragma unroll
    for (int j = 0; j < kNs; ++j) {
        float s[GSE_NX], y0, y1;
        sigma_point(m, L, j, s);
        model_output(s, u0, u1, y0, y1);
        eta[j][0] = (double)y0;
        eta[j][1] = (double)y1;
    }
    float prior[kNx];
#pragma unroll
    for (int k = 0; k < kNx; ++k) prior[k] = m[k];
    gsukf_update_gaussian<true>([&](int q, int c) { return eta[q][c]; },
                                [&](int q, int k) {
                                    float s[GSE_NX];
                                    sigma_point(m, L, q, s);
                                    return s[k] - m[k];
                                },
                                prior, mine + kNx, mine, mine + kNx, w, w, u0, u1, z0, z1, pdf, sw);
}

template <bool kInjected>
__global__ __launch_bounds__(kLaneBank, GSE_GSUKF_PRED_WAVES) void gsukf_predict_lanes_kernel(
    float* __restrict__ mu, float* __restrict__ P, const float* __restrict__ noise, int64_t N, int32_t* __restrict__ status,
    const GsukfPredictArgs a) {
    __shared__ DrawParams dp;
    __shared__ float stage[kLaneBank * kStageStride];
    const int t = threadIdx.x, b = blockIdx.y;
    const int64_t g0 = (int64_t)blockIdx.x * kLaneBank;
    const int nloc = (int)min((int64_t)kLaneBank, N - g0);
    float* mu_b = mu + ((int64_t)b * N + g0) * kNx;
    float* P_b = P + ((int64_t)b * N + g0) * kNxx;
    if (!kInjected) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&a.draw);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&dp);
        for (int i = t; i < (int)(sizeof(DrawParams) / 4); i += kLaneBank) dst[i] = src[i];
    }
    const FilterIO bio = bank_inputs(a.io, a.u_dev, a.z_dev, b);
#if GSE_GSUKF_DIRECT_ROWS
    if (!kInjected) __syncthreads();                                   // (the draw's parameters)
    if (t < nloc) {
        float* mine = stage + t * kStageStride;
        gsukf_predict_lane<kInjected>(mu_b + t * kNx, P_b + t * kNxx, mine, b, g0 + t, N, noise, dp, status, a, bio);
        lane_row_out(mine, mu_b + t * kNx, P_b + t * kNxx);
    }
#else
    stage_bank_in(stage, mu_b, P_b, nloc);
    if (t < nloc) {
        float* mine = stage + t * kStageStride;
        gsukf_predict_lane<kInjected>(mine, mine + kNx, mine, b, g0 + t, N, noise, dp, status, a, bio);
    }
    stage_bank_out(stage, mu_b, P_b, nloc);
#endif
}

template <int kUnused>
__global__ __launch_bounds__(kLaneBank) void gsukf_update_lanes_kernel(float* __restrict__ mu, float* __restrict__ P,
                                                                       float* __restrict__ w, int64_t N,
                                                                       int32_t* __restrict__ status, const GsukfUpdateArgs a) {
    __shared__ float stage[kLaneBank * kStageStride];
    const int t = threadIdx.x, b = blockIdx.y;
    const int64_t g0 = (int64_t)blockIdx.x * kLaneBank;
    const int nloc = (int)min((int64_t)kLaneBank, N - g0);
    float* mu_b = mu + ((int64_t)b * N + g0) * kNx;
    float* P_b = P + ((int64_t)b * N + g0) * kNxx;
    const FilterIO bio = bank_inputs(a.io, a.u_dev, a.z_dev, b);
#if GSE_GSUKF_DIRECT_ROWS
    if (t < nloc) {
        float* mine = stage + t * kStageStride;
        lane_row_in(mine, mu_b + t * kNx, P_b + t * kNxx);
        gsukf_update_lane(mine, w + (int64_t)b * N + g0 + t, bio.u0, bio.u1, bio.z0, bio.z1, status, a.pdf, a.sw);
        lane_row_out(mine, mu_b + t * kNx, P_b + t * kNxx);
    }
#else
    stage_bank_in(stage, mu_b, P_b, nloc);
    if (t < nloc)
        gsukf_update_lane(stage + t * kStageStride, w + (int64_t)b * N + g0 + t, bio.u0, bio.u1, bio.z0, bio.z1, status, a.pdf,
                          a.sw);
    stage_bank_out(stage, mu_b, P_b, nloc);
#endif
}

// predict and update of a large bank in ONE launch (the class defers predict() until update()): what predict leaves in
// the staged rows is what update picks up -- the same fp32 values the separate launches pass through memory, so the same
// bits, without writing 120 bytes per Gaussian and reading them back, and one launch less.
// What a workgroup of the one-lane-per-Gaussian kernels leaves of its 256 new weights (wq, in LDS) for the resample that
// follows: they are one QUARTER of a tile of 1024, and wavefront 0 forms of them what wavefront `quarter` of the tile-sum /
// cumulative-sum kernels forms -- 64 lanes x 4 consecutive weights, masked_sum4, the wavefront scan: the quarter's sum
// ([B][T][4]) and, if asked for, every quad's exclusive prefix inside the quarter ([B][T][256]; the scan slot format of the
// particle filter's cluster kernels, gse_pf_cluster_kernels.h).  Called by the 64 lanes of wavefront 0; q: the quarter's index
// among the bank's quarters (4 tile + quarter).
__device__ __forceinline__ void lanes_quarter_scan(const float* wq, int64_t g0, int64_t N, int b, int q, int lane,
                                                   double* __restrict__ quarter_sums, double* __restrict__ quad_prefix) {
    const int64_t T = (N + kTile - 1) / kTile;
    const float4 q4 = *reinterpret_cast<const float4*>(&wq[4 * lane]);
    const double mine4 = masked_sum4(q4.x, q4.y, q4.z, q4.w, g0 + 4 * lane, N);
    const double inc = wave_inclusive_scan(mine4, lane);
    if (quad_prefix) {
        const double ex = wave_shift_up1(inc);
        quad_prefix[((int64_t)b * T + (q >> 2)) * kBlock + (q & 3) * 64 + lane] = ex;
    }
    if (lane == 63) quarter_sums[(int64_t)b * (4 * T) + q] = inc;
}

struct GsukfPredictUpdateArgs {
    GsukfPredictArgs pred;      // (its io carries u and z)
    Pdf2 pdf;
};
// anc != null: the bank's rows are first gathered from mu_in / P_in (the ancestors of a resample that left its copy to this
// launch -- gse_gsukf_resample_begin) and the results go to mu / P, which must then be other arrays; anc == null: in place.
// kDiagonal: see gsukf_predict_lane (the host knows the mixture when it launches; a branch per draw inside the kernel costs
// copies of what is alive across it, and both variants side by side in ONE kernel overflow the instruction cache: 53 us).
template <bool kInjected, int kDiagonal = -1>
__global__ __launch_bounds__(kLaneBank, GSE_GSUKF_PRED_WAVES) void gsukf_predict_update_lanes_kernel(
    float* __restrict__ mu, float* __restrict__ P, float* __restrict__ w, const float* __restrict__ noise, int64_t N,
    int32_t* __restrict__ status, const float* __restrict__ mu_in, const float* __restrict__ P_in,
    const int32_t* __restrict__ anc, double* __restrict__ quarter_sums, double* __restrict__ quad_prefix,
    const GsukfPredictUpdateArgs a) {
    __shared__ DrawParams dp;
    __shared__ float stage[kLaneBank * kStageStride];
#if !GSE_GSUKF_DIRECT_ROWS
    __shared__ int anc_s[kLaneBank];
#endif
    __shared__ __attribute__((aligned(16))) float wq[kLaneBank];
    const int t = threadIdx.x, b = blockIdx.y;
    const int64_t g0 = (int64_t)blockIdx.x * kLaneBank;
    const int nloc = (int)min((int64_t)kLaneBank, N - g0);
    float* mu_b = mu + ((int64_t)b * N + g0) * kNx;
    float* P_b = P + ((int64_t)b * N + g0) * kNxx;
    if (!kInjected) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&a.pred.draw);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&dp);
        for (int i = t; i < (int)(sizeof(DrawParams) / 4); i += kLaneBank) dst[i] = src[i];
    }
#if GSE_GSUKF_DIRECT_ROWS
    if (!kInjected) __syncthreads();                                   // (the draw's parameters)
#else
    if (anc)
        stage_bank_in_gathered(stage, anc_s, mu_in + (int64_t)b * N * kNx, P_in + (int64_t)b * N * kNxx,
                               anc + (int64_t)b * N + g0, nloc);
    else
        stage_bank_in(stage, mu_b, P_b, nloc);
#endif
    const FilterIO bio = bank_inputs(a.pred.io, a.pred.u_dev, a.pred.z_dev, b);
    float w_new = 0.0f;
    if (t < nloc) {
        float* mine = stage + t * kStageStride;
#if GSE_GSUKF_DIRECT_ROWS
        // (predict reads the mean and the lower triangle straight from the row in global memory -- the ancestor's, when a
        //  resample left its copy to this launch -- and leaves its result in the lane's LDS row for update)
        const int64_t row = anc ? (int64_t)b * N + anc[(int64_t)b * N + g0 + t] : (int64_t)b * N + g0 + t;
        const float* mu_row = (anc ? mu_in : mu) + row * kNx;
        const float* P_row = (anc ? P_in : P) + row * kNxx;
#else
        const float *mu_row = mine, *P_row = mine + kNx;
#endif
        gsukf_predict_lane<kInjected, kDiagonal>(mu_row, P_row, mine, b, g0 + t, N, noise, dp, status, a.pred, bio);
        float* wg = w + (int64_t)b * N + g0 + t;
        w_new = *wg;
        gsukf_update_lane(mine, &w_new, bio.u0, bio.u1, bio.z0, bio.z1, status, a.pdf, a.pred.sw);
        *wg = w_new;
    }
    if (quarter_sums) {
        // The workgroup's 256 new weights are one QUARTER of a tile of 1024: the sum tile_sum_kernel's wavefront `quarter`
        // forms of them -- 64 lanes x 4 consecutive weights, masked_sum4, the wavefront scan -- left for the ancestors kernel,
        // which adds a tile's four as block_exclusive_scan adds its four wavefront totals: the same tile sums, bit for bit,
        // without the launch of their own (4.7 us of a 58 us cycle at 2^18 Gaussians).  Nobody waits for anybody: the
        // consumer is the next kernel on the stream.
        wq[t] = w_new;
        __syncthreads();
        if (t < 64) lanes_quarter_scan(wq, g0, N, b, (int)blockIdx.x, t, quarter_sums, quad_prefix);
    }
#if GSE_GSUKF_DIRECT_ROWS
    if (t < nloc) lane_row_out(stage + t * kStageStride, mu_b + t * kNx, P_b + t * kNxx);
#else
    stage_bank_out(stage, mu_b, P_b, nloc);
#endif
}

// The copy of a deferred resample on its own (an accessor looks at the bank before the next predict + update): rows by index.
template <int kUnused>
__global__ __launch_bounds__(kLaneBank) void gsukf_gather_rows_kernel(const float* __restrict__ mu_in, const float* __restrict__ P_in,
                                                                      const int32_t* __restrict__ anc, float* __restrict__ mu,
                                                                      float* __restrict__ P, int64_t N) {
    __shared__ float stage[kLaneBank * kStageStride];
    __shared__ int anc_s[kLaneBank];
    const int b = blockIdx.y;
    const int64_t g0 = (int64_t)blockIdx.x * kLaneBank;
    const int nloc = (int)min((int64_t)kLaneBank, N - g0);
    stage_bank_in_gathered(stage, anc_s, mu_in + (int64_t)b * N * kNx, P_in + (int64_t)b * N * kNxx, anc + (int64_t)b * N + g0,
                           nloc);
    stage_bank_out(stage, mu + ((int64_t)b * N + g0) * kNx, P + ((int64_t)b * N + g0) * kNxx, nloc);
}

// ------------------------------------------------------------------------------------------------
// The whole cycle predict -> update -> resample of a bank of at most 64 Gaussians in ONE workgroup of 1024 threads
// (64 groups of 16 lanes): BASELINE config 3 is launch-latency-bound, so one launch instead of five is the lever.
// mu, P live in LDS between the phases; the resample is the arithmetic of cumsum_kernel + the gather for a single
// tile (all per-thread sums sit in wavefront 0), so the result is bit-identical to the separate launches.
constexpr int kSmallBank = 64;
constexpr int kSmallBlock = kSmallBank * kGroup;      // 1024

// Ancestors of a bank of at most 64 Gaussians held by ONE workgroup (weights w_s in LDS): the arithmetic of cumsum_kernel +
// the gather's comparison for a single tile -- all per-thread sums sit in wavefront 0 -- so the result is bit-identical to the
// separate launches.  Leaves anc_s[0 .. N) and *total_s; ends with a workgroup barrier.  Every thread must call it.
__device__ __forceinline__ void small_bank_ancestors(const float* w_s, double* c_s, int* anc_s, double* total_s, int N, double r,
                                                     int tid) {
    if (tid < 64) {
        const int i0 = tid * 4;
        double sc[4];
        sc[0] = (i0 + 0 < N) ? (double)w_s[min(i0 + 0, kSmallBank - 1)] : 0.0;
        sc[1] = sc[0] + ((i0 + 1 < N) ? (double)w_s[min(i0 + 1, kSmallBank - 1)] : 0.0);
        sc[2] = sc[1] + ((i0 + 2 < N) ? (double)w_s[min(i0 + 2, kSmallBank - 1)] : 0.0);
        sc[3] = sc[2] + ((i0 + 3 < N) ? (double)w_s[min(i0 + 3, kSmallBank - 1)] : 0.0);
        const double inc = wave_inclusive_scan(sc[3], tid);
        const double W = __shfl(inc, 63, 64);                       // workgroup total: wavefronts 1.. hold zeros
        double excl = wave_shift_up1(inc);
        if (tid == 0) excl = 0.0;
        const double top = W / W;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (i0 + p < N) c_s[i0 + p] = (i0 + p == N - 1) ? top : fmin((0.0 + (excl + sc[p])) / W, top);
        if (tid == 0) *total_s = W;
    }
    __syncthreads();
    if (tid < N) {                                                   // ancestor of output tid
        const double W = *total_s;
        int k = 0;
        if ((W > 0.0) && (W < INFINITY)) {
            const double u = ((double)tid + r) / (double)N;
            int lo = 0, hi = N - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (c_s[mid] < u) lo = mid + 1; else hi = mid;
            }
            k = lo;
        }
        anc_s[tid] = k;
    }
    __syncthreads();
}

struct GsukfCycleArgs {
    GsukfPredictArgs pred;
    Pdf2 pdf;
    double r[GSE_MAX_B];
};

template <bool kInjected>
__global__ __launch_bounds__(kSmallBlock) void gsukf_cycle_kernel(float* __restrict__ mu, float* __restrict__ P,
                                                                  float* __restrict__ w, const float* __restrict__ noise,
                                                                  int32_t* __restrict__ idx_out, int N,
                                                                  int32_t* __restrict__ status, const GsukfCycleArgs a) {
    __shared__ DrawParams dp;
    __shared__ float fac[kBank][kFacStride];
    __shared__ float tile[kBank][kTileStride];                   // predict: sigma points; update: deviations
    __shared__ double eta[kBank][kEtaStride];
    __shared__ float mu_s[kSmallBank][kNx];
    __shared__ float P_s[kSmallBank][kNxx];
    __shared__ float w_s[kSmallBank];
    __shared__ double c_s[kSmallBank];
    __shared__ int anc_s[kSmallBank];
    static_assert(kSmallBank == kBank, "the cycle kernel is one bank");
    const int tid = threadIdx.x, b = blockIdx.y;
    const int j = tid & (kGroup - 1), gl = tid / kGroup;
    const bool live = gl < N;
    const int64_t gi = (int64_t)b * N + (live ? gl : 0);
    if (!kInjected) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&a.pred.draw);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&dp);
        for (int i = tid; i < (int)(sizeof(DrawParams) / 4); i += kSmallBlock) dst[i] = src[i];
        __syncthreads();
    }
    float* mu_b = mu + (int64_t)b * N * kNx;
    float* P_b = P + (int64_t)b * N * kNxx;
    float* w_b = w + (int64_t)b * N;
    auto from_global = [&](int t, const float*& m_, const float*& p_) { m_ = mu_b + t * kNx; p_ = P_b + t * kNxx; };
    auto from_lds = [&](int t, const float*& m_, const float*& p_) { m_ = mu_s[t]; p_ = P_s[t]; };
    auto to_lds = [&](int t, float*& m_, float*& p_) { m_ = mu_s[t]; p_ = P_s[t]; };
    // 1. predict: global -> LDS
    const FilterIO bio = bank_inputs(a.pred.io, a.pred.u_dev, a.pred.z_dev, b);
    gsukf_predict_block<kInjected>(from_global, to_lds, fac, tile, N, b, 0,
                                   kInjected ? noise + (int64_t)b * N * kNs * kNx : nullptr, dp, status, a.pred, bio, a.pred.rng);
    __syncthreads();
    // 2. update in LDS; the weight goes global -> LDS
    gsukf_update_block(from_lds, to_lds, [&](int t) { return (const float*)(w_b + t); }, [&](int t) { return &w_s[t]; }, fac,
                       tile, eta, N, b, status, bio, a.pdf, a.pred.sw);
    __syncthreads();
    // 3. resample, one tile: thread t < 16 of wavefront 0 owns weights 4t..4t+3, exactly as cumsum_kernel does
    __shared__ double total_s;
    small_bank_ancestors(w_s, c_s, anc_s, &total_s, N, a.r[b], tid);
    if (tid < N) {
        if (idx_out) idx_out[(int64_t)b * N + tid] = anc_s[tid];
        w[(int64_t)b * N + tid] = (float)(1.0 / (double)N);
    }
    __syncthreads();
    // 4. gather LDS -> global (in place: the source is LDS)
    if (live) {
        const int src = anc_s[gl];
        for (int e = j; e < kRow; e += kGroup) {
            if (e < kNx) mu[gi * kNx + e] = mu_s[src][e];
            else P[gi * kNxx + (e - kNx)] = P_s[src][e - kNx];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// moments: one thread per Gaussian, blocks stride; 1 + kNx + 2 kTri fp64 sums (36 for 5 states): sum w, sum w d,
// sum w d d^T (upper triangle) with d = mu - pivot, sum w P (upper triangle)
constexpr int kGsukfMoments = 1 + kNx + 2 * kTri;
// kFold: sum w P is added into sum w d d^T, which gives the first 1 + kNx + kTri sums the meaning they have for a particle
// filter (gse_pf_moments): whoever consumes those -- the closed loop's controller kernel -- takes either estimator.
template <int kFold>
__global__ __launch_bounds__(kBlock) void gsukf_moments_kernel(const float* __restrict__ mu, const float* __restrict__ P,
                                                               const float* __restrict__ w,
                                                               double* __restrict__ partials, int64_t N) {
    __shared__ double red[4 * kGsukfMoments];
    const int b = blockIdx.y;
    const float* mb = mu + (int64_t)b * N * kNx;
    const float* Pb = P + (int64_t)b * N * kNxx;
    const float* wb = w + (int64_t)b * N;
    double pv[kNx];
#pragma unroll
    for (int k = 0; k < kNx; ++k) pv[k] = (double)mb[k];
    double acc[kGsukfMoments];
#pragma unroll
    for (int c = 0; c < kGsukfMoments; ++c) acc[c] = 0.0;
    for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < N; g += (int64_t)gridDim.x * kBlock) {
        const double wp = (double)wb[g];
        double d[kNx];
#pragma unroll
        for (int k = 0; k < kNx; ++k) d[k] = (double)mb[g * kNx + k] - pv[k];
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
            for (int l = k; l < kNx; ++l) acc[c++] += wp * (double)Pb[g * kNxx + k * kNx + l];
    }
    if (kFold) {
#pragma unroll
        for (int i = 0; i < kTri; ++i) acc[1 + kNx + i] += acc[1 + kNx + kTri + i];
    }
    double* out = partials + ((int64_t)b * kMomentBlocks + blockIdx.x) * kMomentWidth;
    block_sums<kGsukfMoments>(acc, red, [&](int c, double total) { out[c] = total; });
}

}  // namespace gse
