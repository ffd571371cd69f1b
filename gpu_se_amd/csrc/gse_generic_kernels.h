// gse_generic_kernels.h -- the two filters for user models BEYOND the fast path's bounds (round 5).
//
// The reference hands any Python f / g of any dimension to Numba (filter/particle.py:176-208, filter/gs_ukf.py:243-275).  The
// kernels of gse_pf_kernels.h / gse_gsukf_kernels.h are laid out for at most 5 states, 2 inputs, 2 outputs (state-major rows of
// five, a 2 x 2 measurement density, sixteen-byte tricks everywhere): models inside those bounds run them, compiled for the model
// at run time.  A model with up to GSE_G_MAX_NX = 8 states, GSE_G_MAX_NU = 4 inputs, GSE_G_MAX_NY = 4 outputs runs THESE kernels
// instead: the same algorithms written for any dimension -- one thread per particle / per Gaussian, rows of NX floats,
// arithmetic spelled out operation by operation so that a NumPy restatement reproduces every bit (tests/test_gpu_usermodel.py)
// -- compiled by hiprtc with the model's dimensions as constants (gse_rtc.hip) and launched by name from the Python class
// (gpu_se_amd/filter/generic.py through gse_model_launch), as the reference launches its Numba kernels from Python.  They are
// correct and parallel, not tuned: a particle moves 8 NX bytes per predict where the fast path's fused launch moves 11 in all.
//
// Arithmetic (the contract the restatement follows):
//   * f and g are evaluated in fp64 on the fp32-stored state, without FMA contraction;
//   * predict:  x <- fp32(x + f(x, u, dt)), then x <- x + e in fp32, e the state noise (fp32);
//   * update:   y = fp32(g(x, u));  lik = sum_d cw_d exp(-1/2 a^T Sinv_d a), a = z - y in fp64, t_j = sum_i a_i Sinv[i][j] with i
//               ascending, q = sum_j t_j a_j with j ascending;  w <- fp32(w lik);
//   * GS-UKF:   Cholesky of P in fp32, column by column (sqrtf, fp32 division); sigma points mu +- L[:, j] in fp32; sums over
//               the sigma points in fp64 with j ascending; the gain from a Cholesky solve of P_yy in fp64; stores round to fp32.
#pragma once
#include "gse_device.h"

#ifdef GSE_GENERIC_MODEL
#include GSE_GENERIC_MODEL          // GSE_G_NX, GSE_G_NU, GSE_G_NY, generic_increment(), generic_output()
#endif

namespace gse {

constexpr int kGMaxNx = 8, kGMaxNu = 4, kGMaxNy = 4, kGMaxNd = 4;

// A Gaussian mixture of up to 4 components in up to 8 dimensions (include/gse.h: gse_mixture_n), in DEVICE memory for the
// model kernels (built by the host: gpu_se_amd/filter/generic.py), a kernel argument for the mixture class's own two kernels.
using GenericMixture = gse_mixture_n;
static_assert(GSE_N_MAX_NX == kGMaxNx && GSE_MAX_ND == kGMaxNd, "gse_mixture_n is the generic kernels' mixture");

// The draw and the density for a dimension known at run time (MultivariateGaussianSum of more than GSE_NX dimensions: gse_rows.hip).
// The SAME words, pairs, selector and order of additions as generic_draw / generic_pdf below, which know it at compile time.
__device__ __forceinline__ void generic_draw_n(const GenericMixture& m, const RngKey& key, uint64_t elem, float (&e)[kGMaxNx]) {
#pragma clang fp contract(off)
    const int n = m.n, pairs = (n + 1) / 2;
    uint32_t wd[12];                                      // 2 * 4 + 1 words at most: three blocks
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        uint32_t c[4];
        rng_block_of_elem(key, elem, 0u, (uint32_t)s, c);
        wd[4 * s] = c[0]; wd[4 * s + 1] = c[1]; wd[4 * s + 2] = c[2]; wd[4 * s + 3] = c[3];
    }
    float nrm[kGMaxNx];
#pragma unroll
    for (int p = 0; p < kGMaxNx / 2; ++p) box_muller(wd[2 * p], wd[2 * p + 1], nrm[2 * p], nrm[2 * p + 1]);
    const float v = (float)(wd[2 * pairs] >> 8) * 5.9604644775390625e-08f;
    int d = m.nd - 1;
    for (int k = m.nd - 2; k >= 0; --k)
        if (v < m.cumw[k]) d = k;
    for (int i = 0; i < n; ++i) {
        float acc = m.mean[d][i];
        for (int j = 0; j <= i; ++j) acc = acc + m.chol[d][i][j] * nrm[j];
        e[i] = acc;
    }
}

__device__ __forceinline__ double generic_pdf_n(const GenericMixture& m, const double (&a)[kGMaxNx]) {
#pragma clang fp contract(off)
    const int n = m.n;
    double lik = 0.0;
    for (int d = 0; d < m.nd; ++d) {
        double q = 0.0;
        for (int j = 0; j < n; ++j) {
            double t = 0.0;
            for (int i = 0; i < n; ++i) t = t + a[i] * m.inv[d][i][j];
            q = q + t * a[j];
        }
        lik = lik + m.cw[d] * exp(-0.5 * q);
    }
    return lik;
}

struct GenericIO {
    double u[kGMaxNu], z[kGMaxNy], dt;
};

#ifdef GSE_GENERIC_MODEL
constexpr int kGNx = GSE_G_NX, kGNu = GSE_G_NU, kGNy = GSE_G_NY, kGNs = 2 * GSE_G_NX + 1;
static_assert(kGNx >= 1 && kGNx <= kGMaxNx && kGNu >= 1 && kGNu <= kGMaxNu && kGNy >= 1 && kGNy <= kGMaxNy, "model dimensions");

// ---- the draw: Philox4x32-10 blocks sub = 0, 1, ... of element `elem` give words 0 .. ; pair p = words (2p, 2p + 1) is one
//      Box-Muller pair = normals 2p, 2p + 1; the word behind the pairs selects the component (top 24 bits against the
//      normalised cumulative weights); e = mean + L n row by row, j ascending, one multiply and one add per term (fp32).
__device__ __forceinline__ void generic_draw(const GenericMixture& m, const RngKey& key, uint64_t elem, float (&e)[kGNx]) {
#pragma clang fp contract(off)
    constexpr int kPairs = (kGNx + 1) / 2, kWords = 2 * kPairs + 1, kBlocks = (kWords + 3) / 4;
    uint32_t wd[4 * kBlocks];
#pragma unroll
    for (int s = 0; s < kBlocks; ++s) {
        uint32_t c[4];
        rng_block_of_elem(key, elem, 0u, (uint32_t)s, c);
        wd[4 * s] = c[0]; wd[4 * s + 1] = c[1]; wd[4 * s + 2] = c[2]; wd[4 * s + 3] = c[3];
    }
    float n[2 * kPairs];
#pragma unroll
    for (int p = 0; p < kPairs; ++p) box_muller(wd[2 * p], wd[2 * p + 1], n[2 * p], n[2 * p + 1]);
    const float v = (float)(wd[2 * kPairs] >> 8) * 5.9604644775390625e-08f;       // [0, 1)
    int d = m.nd - 1;
    for (int k = m.nd - 2; k >= 0; --k)
        if (v < m.cumw[k]) d = k;
#pragma unroll
    for (int i = 0; i < kGNx; ++i) {
        float acc = m.mean[d][i];
#pragma unroll
        for (int j = 0; j <= i; ++j) acc = acc + m.chol[d][i][j] * n[j];
        e[i] = acc;
    }
}

// sum_d cw_d exp(-1/2 a^T Sinv_d a), a[n] fp64 (MultivariateGaussianSum.py:39-63)
template <int kN>
__device__ __forceinline__ double generic_pdf(const GenericMixture& m, const double (&a)[kN]) {
#pragma clang fp contract(off)
    double lik = 0.0;
    for (int d = 0; d < m.nd; ++d) {
        double q = 0.0;
#pragma unroll
        for (int j = 0; j < kN; ++j) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < kN; ++i) t = t + a[i] * m.inv[d][i][j];
            q = q + t * a[j];
        }
        lik = lik + m.cw[d] * exp(-0.5 * q);
    }
    return lik;
}

// x <- fp32(x + f(x, u, dt)) on one fp32 row
__device__ __forceinline__ void generic_step_row(float (&x)[kGNx], const GenericIO& io) {
#pragma clang fp contract(off)
    double xd[kGNx], dx[kGNx], ud[kGNu];
#pragma unroll
    for (int k = 0; k < kGNx; ++k) xd[k] = (double)x[k];
#pragma unroll
    for (int k = 0; k < kGNu; ++k) ud[k] = io.u[k];
    generic_increment(xd, ud, io.dt, dx);
#pragma unroll
    for (int k = 0; k < kGNx; ++k) x[k] = (float)(xd[k] + dx[k]);
}

__device__ __forceinline__ void generic_output_row(const float (&x)[kGNx], const GenericIO& io, float (&y)[kGNy]) {
    double xd[kGNx], ud[kGNu];
#pragma unroll
    for (int k = 0; k < kGNx; ++k) xd[k] = (double)x[k];
#pragma unroll
    for (int k = 0; k < kGNu; ++k) ud[k] = io.u[k];
    generic_output(xd, ud, y);
}

// ================================================================================================ particle filter
// predict (filter/particle.py:265-277): x_i += f(x_i, u, dt), then x += a draw of the state noise (injected: inj [N][NX])
template <bool kInjected>
__global__ __launch_bounds__(256) void generic_pf_predict_kernel(float* __restrict__ x, int64_t N, const GenericIO io,
                                                                 const GenericMixture* __restrict__ noise, const RngKey rng,
                                                                 const float* __restrict__ inj) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float row[kGNx], e[kGNx];
#pragma unroll
    for (int k = 0; k < kGNx; ++k) row[k] = x[i * kGNx + k];
    generic_step_row(row, io);
    if (kInjected) {
#pragma unroll
        for (int k = 0; k < kGNx; ++k) e[k] = inj[i * kGNx + k];
    } else {
        generic_draw(*noise, rng, (uint64_t)i, e);
    }
#pragma unroll
    for (int k = 0; k < kGNx; ++k) x[i * kGNx + k] = row[k] + e[k];
}

// update (filter/particle.py:279-294): w_i *= pdf(z - g(x_i, u))
__global__ __launch_bounds__(256) void generic_pf_update_kernel(const float* __restrict__ x, float* __restrict__ w, int64_t N,
                                                                const GenericIO io, const GenericMixture* __restrict__ meas) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    float row[kGNx], y[kGNy];
#pragma unroll
    for (int k = 0; k < kGNx; ++k) row[k] = x[i * kGNx + k];
    generic_output_row(row, io, y);
    double a[kGNy];
#pragma unroll
    for (int k = 0; k < kGNy; ++k) a[k] = io.z[k] - (double)y[k];
    w[i] = (float)((double)w[i] * generic_pdf<kGNy>(*meas, a));
}

// rows of `width` floats: out[j] = in[idx[j]];  w_out[j] = 1 / N (resample's copy, filter/particle.py:311-316)
__global__ __launch_bounds__(256) void generic_gather_kernel(const float* __restrict__ in, const int64_t* __restrict__ idx,
                                                             float* __restrict__ out, float* __restrict__ w_out, int64_t N,
                                                             int width) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= N) return;
    const int64_t a = idx[j];
    for (int k = 0; k < width; ++k) out[j * width + k] = in[a * width + k];
    if (w_out) w_out[j] = (float)(1.0 / (double)N);
}

// partial sums { sum w, sum w d (NX), upper(sum w d d^T) } with d = x - x_0, fp64, one row of 1 + NX + NX (NX + 1) / 2 doubles per
// workgroup (the host adds the rows in order): point_estimate / point_covariance (filter/particle.py:318-327)
constexpr int kGMomWidth = 1 + kGNx + kGNx * (kGNx + 1) / 2;
__global__ __launch_bounds__(256) void generic_moments_kernel(const float* __restrict__ x, const float* __restrict__ w, int64_t N,
                                                              double* __restrict__ partials) {
    __shared__ double red[4][kGMomWidth];
    double acc[kGMomWidth];
#pragma unroll
    for (int c = 0; c < kGMomWidth; ++c) acc[c] = 0.0;
    double pv[kGNx];
#pragma unroll
    for (int k = 0; k < kGNx; ++k) pv[k] = (double)x[k];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
        const double wi = (double)w[i];
        double d[kGNx];
#pragma unroll
        for (int k = 0; k < kGNx; ++k) d[k] = (double)x[i * kGNx + k] - pv[k];
        acc[0] += wi;
        int c = 1 + kGNx;
#pragma unroll
        for (int k = 0; k < kGNx; ++k) {
            const double wd = wi * d[k];
            acc[1 + k] += wd;
#pragma unroll
            for (int l = k; l < kGNx; ++l) acc[c++] += wd * d[l];
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < kGMomWidth; ++c) {
        const double t = wave_sum(acc[c]);
        if (lane == 0) red[wv][c] = t;
    }
    __syncthreads();
    if ((int)threadIdx.x < kGMomWidth) {
        const int c = threadIdx.x;
        partials[(int64_t)blockIdx.x * kGMomWidth + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    }
}

// f / g on rows (f_vectorize / g_vectorize: filter/particle.py:176-208)
__global__ __launch_bounds__(256) void generic_f_rows_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t M,
                                                             const GenericIO io) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    double xd[kGNx], dx[kGNx], ud[kGNu];
#pragma unroll
    for (int k = 0; k < kGNx; ++k) xd[k] = (double)x[i * kGNx + k];
#pragma unroll
    for (int k = 0; k < kGNu; ++k) ud[k] = io.u[k];
    generic_increment(xd, ud, io.dt, dx);
#pragma unroll
    for (int k = 0; k < kGNx; ++k) out[i * kGNx + k] = (float)dx[k];
}

__global__ __launch_bounds__(256) void generic_g_rows_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t M,
                                                             const GenericIO io) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= M) return;
    float row[kGNx], y[kGNy];
#pragma unroll
    for (int k = 0; k < kGNx; ++k) row[k] = x[i * kGNx + k];
    generic_output_row(row, io, y);
#pragma unroll
    for (int k = 0; k < kGNy; ++k) out[i * kGNy + k] = y[k];
}

// ================================================================================================ Gaussian-sum UKF
// Cholesky factor (lower) of P in fp32, column by column; a pivot that is not positive: false.  jitter is added to the diagonal
// (the reference retries with 1e-10 I: filter/gs_ukf.py:72-75).
__device__ __forceinline__ bool generic_cholesky(const float (&P)[kGNx][kGNx], float jitter, float (&L)[kGNx][kGNx]) {
#pragma clang fp contract(off)
    bool ok = true;
#pragma unroll
    for (int j = 0; j < kGNx; ++j) {
        float s = P[j][j] + jitter;
#pragma unroll
        for (int k = 0; k < j; ++k) s = s - L[j][k] * L[j][k];
        ok = ok && (s > 0.0f);
        const float dj = sqrtf(s);
        L[j][j] = dj;
#pragma unroll
        for (int i = j + 1; i < kGNx; ++i) {
            float t = P[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) t = t - L[i][k] * L[j][k];
            L[i][j] = t / dj;
        }
#pragma unroll
        for (int i = 0; i < j; ++i) L[i][j] = 0.0f;
    }
    return ok;
}

// sigma points (filter/gs_ukf.py:69-80): s_0 = mu, s_{1 + j} = mu + L[:, j], s_{1 + NX + j} = mu - L[:, j]
__device__ __forceinline__ void generic_sigma_points(const float (&mu)[kGNx], const float (&P)[kGNx][kGNx], float (&s)[kGNs][kGNx],
                                                     int32_t* status) {
    float L[kGNx][kGNx];
    if (!generic_cholesky(P, 0.0f, L)) {
        if (!generic_cholesky(P, 1e-10f, L)) atomicAdd(status, 1);
    }
#pragma unroll
    for (int k = 0; k < kGNx; ++k) s[0][k] = mu[k];
#pragma unroll
    for (int j = 0; j < kGNx; ++j)
#pragma unroll
        for (int k = 0; k < kGNx; ++k) {
            s[1 + j][k] = mu[k] + L[k][j];
            s[1 + kGNx + j][k] = mu[k] - L[k][j];
        }
}

__device__ __forceinline__ void generic_load_gaussian(const float* __restrict__ mu, const float* __restrict__ P, int64_t g,
                                                      float (&m)[kGNx], float (&C)[kGNx][kGNx]) {
#pragma unroll
    for (int k = 0; k < kGNx; ++k) m[k] = mu[g * kGNx + k];
#pragma unroll
    for (int r = 0; r < kGNx; ++r)
#pragma unroll
        for (int c = 0; c < kGNx; ++c) C[r][c] = P[(g * kGNx + r) * kGNx + c];
}

__global__ __launch_bounds__(64) void generic_gsukf_sigma_kernel(const float* __restrict__ mu, const float* __restrict__ P,
                                                                 float* __restrict__ out, int64_t N, int32_t* status) {
    const int64_t g = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (g >= N) return;
    float m[kGNx], C[kGNx][kGNx], s[kGNs][kGNx];
    generic_load_gaussian(mu, P, g, m, C);
    generic_sigma_points(m, C, s, status);
    for (int j = 0; j < kGNs; ++j)
        for (int k = 0; k < kGNx; ++k) out[(g * kGNs + j) * kGNx + k] = s[j][k];
}

// predict (filter/gs_ukf.py:82-103): every sigma point through f, + a draw of the state noise each; mean and covariance of the
// weighted points (w_0 for the centre, w_1 for the others: :62-67)
template <bool kInjected>
__global__ __launch_bounds__(64) void generic_gsukf_predict_kernel(float* __restrict__ mu, float* __restrict__ P, int64_t N,
                                                                   const GenericIO io, float w0, float w1,
                                                                   const GenericMixture* __restrict__ noise, const RngKey rng,
                                                                   const float* __restrict__ inj, int32_t* status) {
#pragma clang fp contract(off)
    const int64_t g = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (g >= N) return;
    float m[kGNx], C[kGNx][kGNx], s[kGNs][kGNx];
    generic_load_gaussian(mu, P, g, m, C);
    generic_sigma_points(m, C, s, status);
    for (int j = 0; j < kGNs; ++j) {
        generic_step_row(s[j], io);
        float e[kGNx];
        if (kInjected) {
#pragma unroll
            for (int k = 0; k < kGNx; ++k) e[k] = inj[(g * kGNs + j) * kGNx + k];
        } else {
            generic_draw(*noise, rng, (uint64_t)(g * kGNs + j), e);
        }
#pragma unroll
        for (int k = 0; k < kGNx; ++k) s[j][k] = s[j][k] + e[k];
    }
    double mean[kGNx];
#pragma unroll
    for (int k = 0; k < kGNx; ++k) {
        double a = 0.0;
        for (int j = 0; j < kGNs; ++j) a = a + (double)(j ? w1 : w0) * (double)s[j][k];
        mean[k] = a;
        mu[g * kGNx + k] = (float)a;
    }
#pragma unroll
    for (int r = 0; r < kGNx; ++r)
#pragma unroll
        for (int c = 0; c < kGNx; ++c) {
            double a = 0.0;
            for (int j = 0; j < kGNs; ++j)
                a = a + ((double)s[j][r] - mean[r]) * ((double)(j ? w1 : w0) * ((double)s[j][c] - mean[c]));
            P[(g * kGNx + r) * kGNx + c] = (float)a;
        }
}

// update (filter/gs_ukf.py:105-149): local unscented update of every Gaussian, then w *= pdf(z - g(mu', u))
__global__ __launch_bounds__(64) void generic_gsukf_update_kernel(float* __restrict__ mu, float* __restrict__ P, float* __restrict__ w,
                                                                  int64_t N, const GenericIO io, float w0, float w1,
                                                                  const GenericMixture* __restrict__ meas, int32_t* status) {
#pragma clang fp contract(off)
    const int64_t g = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (g >= N) return;
    float m[kGNx], C[kGNx][kGNx], s[kGNs][kGNx];
    generic_load_gaussian(mu, P, g, m, C);
    generic_sigma_points(m, C, s, status);
    double eta[kGNs][kGNy], eta_mean[kGNy];
    for (int j = 0; j < kGNs; ++j) {
        float y[kGNy];
        generic_output_row(s[j], io, y);
#pragma unroll
        for (int k = 0; k < kGNy; ++k) eta[j][k] = (double)y[k];
    }
#pragma unroll
    for (int k = 0; k < kGNy; ++k) {
        double a = 0.0;
        for (int j = 0; j < kGNs; ++j) a = a + (double)(j ? w1 : w0) * eta[j][k];
        eta_mean[k] = a;
    }
    double Pxy[kGNx][kGNy], Pyy[kGNy][kGNy];
#pragma unroll
    for (int r = 0; r < kGNx; ++r)
#pragma unroll
        for (int c = 0; c < kGNy; ++c) {
            double a = 0.0;
            for (int j = 0; j < kGNs; ++j)
                a = a + ((double)s[j][r] - (double)m[r]) * ((double)(j ? w1 : w0) * (eta[j][c] - eta_mean[c]));
            Pxy[r][c] = a;
        }
#pragma unroll
    for (int r = 0; r < kGNy; ++r)
#pragma unroll
        for (int c = 0; c < kGNy; ++c) {
            double a = 0.0;
            for (int j = 0; j < kGNs; ++j)
                a = a + (eta[j][r] - eta_mean[r]) * ((double)(j ? w1 : w0) * (eta[j][c] - eta_mean[c]));
            Pyy[r][c] = a;
        }
    // K = Pxy Pyy^-1 through the Cholesky factor of Pyy (symmetric positive definite for independent outputs; the reference
    // takes the pseudo-inverse, which is the inverse then): Pyy = G G^T, K G G^T = Pxy
    double G[kGNy][kGNy];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < kGNy; ++j) {
        double t = Pyy[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) t = t - G[j][k] * G[j][k];
        ok = ok && (t > 0.0);
        const double dj = sqrt(t);
        G[j][j] = dj;
#pragma unroll
        for (int i = j + 1; i < kGNy; ++i) {
            double v = Pyy[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) v = v - G[i][k] * G[j][k];
            G[i][j] = v / dj;
        }
    }
    if (!ok) atomicAdd(status, 1);
    double K[kGNx][kGNy];
#pragma unroll
    for (int r = 0; r < kGNx; ++r) {
        double t[kGNy];                                   // t G^T = Pxy[r]  (forward), K[r] G = t (backward)
#pragma unroll
        for (int c = 0; c < kGNy; ++c) {
            double v = Pxy[r][c];
#pragma unroll
            for (int k = 0; k < c; ++k) v = v - t[k] * G[c][k];
            t[c] = v / G[c][c];
        }
#pragma unroll
        for (int c = kGNy - 1; c >= 0; --c) {
            double v = t[c];
#pragma unroll
            for (int k = c + 1; k < kGNy; ++k) v = v - K[r][k] * G[k][c];
            K[r][c] = v / G[c][c];
        }
    }
    float mnew[kGNx];
#pragma unroll
    for (int r = 0; r < kGNx; ++r) {
        double a = (double)m[r];
#pragma unroll
        for (int c = 0; c < kGNy; ++c) a = a + K[r][c] * (io.z[c] - eta_mean[c]);
        mnew[r] = (float)a;
        mu[g * kGNx + r] = mnew[r];
    }
    double KP[kGNx][kGNy];                                // K Pyy
#pragma unroll
    for (int r = 0; r < kGNx; ++r)
#pragma unroll
        for (int c = 0; c < kGNy; ++c) {
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < kGNy; ++k) a = a + K[r][k] * Pyy[k][c];
            KP[r][c] = a;
        }
#pragma unroll
    for (int r = 0; r < kGNx; ++r)
#pragma unroll
        for (int c = 0; c < kGNx; ++c) {
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < kGNy; ++k) a = a + KP[r][k] * K[c][k];
            P[(g * kGNx + r) * kGNx + c] = (float)((double)C[r][c] - a);
        }
    float y[kGNy];
    generic_output_row(mnew, io, y);
    double a[kGNy];
#pragma unroll
    for (int k = 0; k < kGNy; ++k) a[k] = io.z[k] - (double)y[k];
    w[g] = (float)((double)w[g] * generic_pdf<kGNy>(*meas, a));
}

// partial sums for point_estimate / point_covariance (filter/gs_ukf.py:438-449): { sum w, sum w d, upper(sum w (P + d d^T)) },
// d = mu - mu_0; same row layout and reduction as generic_moments_kernel
__global__ __launch_bounds__(256) void generic_gsukf_moments_kernel(const float* __restrict__ mu, const float* __restrict__ P,
                                                                    const float* __restrict__ w, int64_t N,
                                                                    double* __restrict__ partials) {
    __shared__ double red[4][kGMomWidth];
    double acc[kGMomWidth];
#pragma unroll
    for (int c = 0; c < kGMomWidth; ++c) acc[c] = 0.0;
    double pv[kGNx];
#pragma unroll
    for (int k = 0; k < kGNx; ++k) pv[k] = (double)mu[k];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (int64_t)gridDim.x * 256) {
        const double wi = (double)w[i];
        double d[kGNx];
#pragma unroll
        for (int k = 0; k < kGNx; ++k) d[k] = (double)mu[i * kGNx + k] - pv[k];
        acc[0] += wi;
        int c = 1 + kGNx;
#pragma unroll
        for (int k = 0; k < kGNx; ++k) {
            acc[1 + k] += wi * d[k];
#pragma unroll
            for (int l = k; l < kGNx; ++l) acc[c++] += wi * ((double)P[(i * kGNx + k) * kGNx + l] + d[k] * d[l]);
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < kGMomWidth; ++c) {
        const double t = wave_sum(acc[c]);
        if (lane == 0) red[wv][c] = t;
    }
    __syncthreads();
    if ((int)threadIdx.x < kGMomWidth) {
        const int c = threadIdx.x;
        partials[(int64_t)blockIdx.x * kGMomWidth + c] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    }
}
#endif  // GSE_GENERIC_MODEL

}  // namespace gse
