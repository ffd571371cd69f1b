// gse_gsukf.hip -- Gaussian-sum unscented Kalman filter on gfx950.
//
// A bank of N Gaussians (mu_i, P_i, w_i); each is a UKF with 2*5+1 = 11 unscaled sigma points
// (filter/gs_ukf.py).  Mapping: 16 lanes per Gaussian (a quarter wavefront), lane j < 11 owns sigma point j, so
// the 64-Gaussian configuration runs as 16 wavefronts side by side instead of 64 serial sigma loops -- that
// configuration is launch-latency-bound and wants the shortest critical path per launch.
// Per Gaussian: the 5x5 Cholesky is done redundantly in every lane's registers (35 flops); sigma points go
// through the model in registers; the 11-term contractions (mean, P, P_xy, P_yy) are staged through an LDS tile
// and summed in the reference's order (j = 0..10) so that results track the NumPy class closely; lane l < 15
// owns entry l of the symmetric 5x5 output.  These contractions are 5x11x5 at most: far too small to feed
// the matrix cores (a 16x16x4 fp32 MFMA tile would be >90 % padding) and nowhere near a bound.
#include "gse_host.h"

namespace gse {

constexpr int kGroup = 16;                       // lanes per Gaussian
constexpr int kGaussPerBlock = kBlock / kGroup;  // 16

// Packed lower-triangular index helpers: entry (i, j), j <= i, lives at i*(i+1)/2 + j.
__device__ __forceinline__ constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }

// Cholesky factor of a symmetric 5x5 matrix given by its packed lower triangle (fp32, like LAPACK spotrf on
// the fp32 covariances).  Returns false when a pivot is not positive.
__device__ __forceinline__ bool chol5(const float (&A)[15], float (&L)[15]) {
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) {
            float s = A[tri(i, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= L[tri(i, k)] * L[tri(j, k)];
            if (i == j) {
                ok = ok && (s > 0.0f);
                L[tri(i, i)] = sqrtf(s);
            } else {
                L[tri(i, j)] = s / L[tri(j, j)];
            }
        }
    }
    return ok;
}

// Loads (mu, lower triangle of P) of one Gaussian and factorises P, retrying once with 1e-10 on the diagonal
// (filter/gs_ukf.py:72-75).  ok = false if the matrix is still not positive definite.
__device__ __forceinline__ void load_and_factor(const float* __restrict__ mu_g, const float* __restrict__ P_g,
                                                float (&mu)[5], float (&L)[15], bool& ok) {
    float A[15];
#pragma unroll
    for (int k = 0; k < 5; ++k) mu[k] = mu_g[k];
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) A[tri(i, j)] = P_g[i * 5 + j];
    ok = chol5(A, L);
    if (!ok) {
#pragma unroll
        for (int i = 0; i < 5; ++i) A[tri(i, i)] += 1e-10f;
        ok = chol5(A, L);
    }
}

// Sigma point j (0..10) of N(mu, L L^T): mu, mu + L[:, c], mu - L[:, c]   (filter/gs_ukf.py:76-80).
__device__ __forceinline__ void sigma_point(const float (&mu)[5], const float (&L)[15], int j, float (&s)[5]) {
    const int c = (j == 0) ? -1 : (j - 1) % 5;
    const float sign = (j <= 5) ? 1.0f : -1.0f;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        float col = 0.0f;
#pragma unroll
        for (int cc = 0; cc <= k; ++cc) col = (c == cc) ? L[tri(k, cc)] : col;
        s[k] = (c < 0) ? mu[k] : mu[k] + sign * col;
    }
}

// (row, col) of packed symmetric entry l (0..14), row <= col, row-major upper triangle; nibble tables.
__device__ __forceinline__ void sym_entry(int l, int& r, int& c) {
    const uint64_t R = 0x433222111100000ull, C = 0x443432432143210ull;
    r = (int)((R >> (4 * l)) & 0xf);
    c = (int)((C >> (4 * l)) & 0xf);
}

struct SigmaWeights {
    float w[GSE_NSIGMA];
    float scl32;          // sum(w) in fp32  -- the divisor numpy.average uses for fp32 data
    double scl64;         // sum(w) accumulated in fp64 -- its divisor for fp64 data
};

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void gsukf_sigma_kernel(const float* __restrict__ mu, const float* __restrict__ P,
                                                             float* __restrict__ sigmas, int64_t N,
                                                             int32_t* __restrict__ status) {
    const int b = blockIdx.y;
    const int j = threadIdx.x & (kGroup - 1);
    const int64_t g = (int64_t)blockIdx.x * kGaussPerBlock + (threadIdx.x / kGroup);
    if (g >= N) return;
    const int64_t gi = (int64_t)b * N + g;
    float m[5], L[15], s[5];
    bool ok;
    load_and_factor(mu + gi * 5, P + gi * 25, m, L, ok);
    if (j >= GSE_NSIGMA) return;
    sigma_point(m, L, j, s);
    if (!ok && j == 0 && status) atomicAdd(status, 1);
    float* out = sigmas + (gi * GSE_NSIGMA + j) * 5;
#pragma unroll
    for (int k = 0; k < 5; ++k) out[k] = s[k];
}

// ------------------------------------------------------------------------------------------------
struct GsukfPredictArgs {
    FilterIOs io;
    DrawParams draw;
    RngKey rng;
    SigmaWeights sw;
    double dt;
};

// Predict of ONE Gaussian by its 16 lanes.  mu_g / P_g: where this Gaussian is read, mu_o / P_o: where it is written
// (may be the same memory: every read precedes the workgroup barrier inside, every write follows it).  tile_g: this
// Gaussian's 56 floats of LDS.  g: its index in the filter (RNG element), nz: its injected noise row [11][5] or null.
// Must be called by every thread of the workgroup.
template <bool kInjected>
__device__ __forceinline__ void gsukf_predict_phase(const float* mu_g, const float* P_g, float* mu_o, float* P_o,
                                                    float* tile_g, bool live, int j, int b, int64_t g, const float* nz,
                                                    const DrawParams& dp, int32_t* status, const GsukfPredictArgs& a) {
    if (live && j < GSE_NSIGMA) {
        float m[5], L[15], s[5];
        bool ok;
        load_and_factor(mu_g, P_g, m, L, ok);
        if (!ok && j == 0 && status) atomicAdd(status, 1);
        sigma_point(m, L, j, s);
        bioreactor_step(s, a.io.f[b].u0, a.io.f[b].u1, a.dt);
        float e[5];
        if (kInjected) {
#pragma unroll
            for (int k = 0; k < 5; ++k) e[k] = nz[j * 5 + k];
        } else {
            uint32_t wd[8];
            rng_words8(a.rng, (uint64_t)(g * GSE_NSIGMA + j), (uint32_t)b, wd);
            mixture_draw5(dp, wd, e);
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) tile_g[j * 5 + k] = s[k] + e[k];
    }
    __syncthreads();
    if (!live || j >= 15) return;

    // mean = (sum_j sigma_j * w_j) / sum(w)  -- fp32, j in order, no FMA contraction  (numpy.average)
    float mean[5];
    {
#pragma clang fp contract(off)
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            float acc = tile_g[k] * a.sw.w[0];
#pragma unroll
            for (int q = 1; q < GSE_NSIGMA; ++q) acc = acc + tile_g[q * 5 + k] * a.sw.w[q];
            mean[k] = acc / a.sw.scl32;
        }
    }
    // lane l owns symmetric entry (r, c) of P = sum_j d_j (d_j w_j)^T
    int r, c;
    sym_entry(j, r, c);
    float mr = 0.f, mc = 0.f;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        mr = (k == r) ? mean[k] : mr;
        mc = (k == c) ? mean[k] : mc;
    }
    float acc = 0.0f;
#pragma unroll
    for (int q = 0; q < GSE_NSIGMA; ++q) {
        const float dr = tile_g[q * 5 + r] - mr;
        const float dc = tile_g[q * 5 + c] - mc;
        acc = fmaf(dr, dc * a.sw.w[q], acc);
    }
    P_o[r * 5 + c] = acc;
    P_o[c * 5 + r] = acc;
    if (j < 5) mu_o[j] = mean[j];
}

template <bool kInjected>
__global__ __launch_bounds__(kBlock) void gsukf_predict_kernel(float* __restrict__ mu, float* __restrict__ P,
                                                               const float* __restrict__ noise, int64_t N,
                                                               int32_t* __restrict__ status, const GsukfPredictArgs a) {
    __shared__ DrawParams dp;
    __shared__ float tile[kGaussPerBlock][GSE_NSIGMA * 5 + 1];
    if (!kInjected) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&a.draw);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&dp);
        for (int i = threadIdx.x; i < (int)(sizeof(DrawParams) / 4); i += kBlock) dst[i] = src[i];
        __syncthreads();
    }
    const int b = blockIdx.y;
    const int j = threadIdx.x & (kGroup - 1);
    const int gl = threadIdx.x / kGroup;
    const int64_t g = (int64_t)blockIdx.x * kGaussPerBlock + gl;
    const bool live = g < N;
    const int64_t gi = (int64_t)b * N + (live ? g : 0);
    gsukf_predict_phase<kInjected>(mu + gi * 5, P + gi * 25, mu + gi * 5, P + gi * 25, tile[gl], live, j, b, g,
                                   kInjected ? noise + gi * GSE_NSIGMA * 5 : nullptr, dp, status, a);
}

// ------------------------------------------------------------------------------------------------
struct GsukfUpdateArgs {
    FilterIOs io;
    Pdf2 pdf;
    SigmaWeights sw;
};

// Pseudo-inverse of a symmetric 2x2 matrix with numpy.linalg.pinv's default cut-off (singular values
// <= 1e-15 * largest are dropped; filter/gs_ukf.py:132).
__device__ __forceinline__ void pinv_sym2(double a, double s, double d, double (&X)[2][2]) {
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

// Update of ONE Gaussian by its 16 lanes, in place in mu_g / P_g (global memory or LDS); the weight is read from w_i
// and written to w_o.  dev_g / eta_g: this Gaussian's LDS scratch (56 floats, 22 doubles).  Must be called by every
// thread of the workgroup.
__device__ __forceinline__ void gsukf_update_phase(float* mu_g, float* P_g, const float* w_i, float* w_o, float* dev_g,
                                                   double* eta_g, bool live, int j, int b, int32_t* status,
                                                   const FilterIOs& io, const Pdf2& pdf, const SigmaWeights& sw) {
    float m[5];
    if (live) {
        float L[15], s[5];
        bool ok;
        load_and_factor(mu_g, P_g, m, L, ok);
        if (j < GSE_NSIGMA) {
            if (!ok && j == 0 && status) atomicAdd(status, 1);
            sigma_point(m, L, j, s);
            float y0, y1;
            bioreactor_output(s[0], s[2], y0, y1);
            eta_g[j * 2 + 0] = (double)y0;
            eta_g[j * 2 + 1] = (double)y1;
#pragma unroll
            for (int k = 0; k < 5; ++k) dev_g[j * 5 + k] = s[k] - m[k];
        }
    }
    __syncthreads();
    if (!live || j >= 15) return;

    // eta_mean, P_yy, P_xy in fp64, j in order (the CPU class keeps etas in fp64: filter/gs_ukf.py:116-131)
    double em[2];
#pragma unroll
    for (int q2 = 0; q2 < 2; ++q2) {
#pragma clang fp contract(off)
        double acc = eta_g[q2] * (double)sw.w[0];
#pragma unroll
        for (int q = 1; q < GSE_NSIGMA; ++q) acc = acc + eta_g[q * 2 + q2] * (double)sw.w[q];
        em[q2] = acc / sw.scl64;
    }
    double pyy[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
    double pxy[5][2];
#pragma unroll
    for (int k = 0; k < 5; ++k) pxy[k][0] = pxy[k][1] = 0.0;
#pragma unroll
    for (int q = 0; q < GSE_NSIGMA; ++q) {
        const double wq = (double)sw.w[q];
        const double e0 = eta_g[q * 2 + 0] - em[0], e1 = eta_g[q * 2 + 1] - em[1];
        const double ew0 = e0 * wq, ew1 = e1 * wq;
        pyy[0][0] = fma(e0, ew0, pyy[0][0]);
        pyy[0][1] = fma(e0, ew1, pyy[0][1]);
        pyy[1][0] = fma(e1, ew0, pyy[1][0]);
        pyy[1][1] = fma(e1, ew1, pyy[1][1]);
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const double dk = (double)dev_g[q * 5 + k];
            pxy[k][0] = fma(dk, ew0, pxy[k][0]);
            pxy[k][1] = fma(dk, ew1, pxy[k][1]);
        }
    }
    double X[2][2];
    pinv_sym2(pyy[0][0], 0.5 * (pyy[0][1] + pyy[1][0]), pyy[1][1], X);
    double K[5][2];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        K[k][0] = pxy[k][0] * X[0][0] + pxy[k][1] * X[1][0];
        K[k][1] = pxy[k][0] * X[0][1] + pxy[k][1] * X[1][1];
    }
    const double z0 = io.f[b].z0, z1 = io.f[b].z1;
    const double i0 = z0 - em[0], i1 = z1 - em[1];
    float mnew[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) mnew[k] = (float)((double)m[k] + (K[k][0] * i0 + K[k][1] * i1));

    // P -= (K P_yy) K^T : lane l owns entry (r, c)
    int r, c;
    sym_entry(j, r, c);
    double kr0 = 0, kr1 = 0, kc0 = 0, kc1 = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        kr0 = (k == r) ? K[k][0] : kr0; kr1 = (k == r) ? K[k][1] : kr1;
        kc0 = (k == c) ? K[k][0] : kc0; kc1 = (k == c) ? K[k][1] : kc1;
    }
    const double t0 = kr0 * pyy[0][0] + kr1 * pyy[1][0];
    const double t1 = kr0 * pyy[0][1] + kr1 * pyy[1][1];
    const double corr = t0 * kc0 + t1 * kc1;
    float* Pg = P_g;
    const float prc = (float)((double)Pg[r * 5 + c] - corr);
    float pcr = prc;
    if (r != c) {
        // the mirrored entry uses (K P_yy)[c] . K[r]; identical when P_yy is symmetric to rounding
        const double u0 = kc0 * pyy[0][0] + kc1 * pyy[1][0];
        const double u1 = kc0 * pyy[0][1] + kc1 * pyy[1][1];
        pcr = (float)((double)Pg[c * 5 + r] - (u0 * kr0 + u1 * kr1));
    }
    // Hazard-free without another barrier: every read of P by OTHER lanes (load_and_factor) precedes the
    // barrier above, and entries (r, c) / (c, r) are read and written by this lane only.
    Pg[r * 5 + c] = prc;
    if (r != c) Pg[c * 5 + r] = pcr;
    if (j < 5) mu_g[j] = mnew[j];
    if (j == 0) {
        // global update: w *= pdf(z - g(mu_new))
        float y0, y1;
        bioreactor_output(mnew[0], mnew[2], y0, y1);
        const double lik = mixture_pdf2<0>(pdf, z0 - (double)y0, z1 - (double)y1);
        *w_o = (float)((double)(*w_i) * lik);
    }
}

__global__ __launch_bounds__(kBlock) void gsukf_update_kernel(float* __restrict__ mu, float* __restrict__ P,
                                                              float* __restrict__ w, int64_t N,
                                                              int32_t* __restrict__ status, const GsukfUpdateArgs a) {
    __shared__ float dev[kGaussPerBlock][GSE_NSIGMA * 5 + 1];     // sigma_j - mu
    __shared__ double eta[kGaussPerBlock][GSE_NSIGMA * 2];        // g(sigma_j)
    const int b = blockIdx.y;
    const int j = threadIdx.x & (kGroup - 1);
    const int gl = threadIdx.x / kGroup;
    const int64_t g = (int64_t)blockIdx.x * kGaussPerBlock + gl;
    const bool live = g < N;
    const int64_t gi = (int64_t)b * N + (live ? g : 0);
    gsukf_update_phase(mu + gi * 5, P + gi * 25, w + gi, w + gi, dev[gl], eta[gl], live, j, b, status, a.io, a.pdf, a.sw);
}

// ------------------------------------------------------------------------------------------------
// The whole cycle predict -> update -> resample of a bank of at most 64 Gaussians in ONE workgroup of 1024 threads
// (64 groups of 16 lanes): BASELINE config 3 is launch-latency-bound, so one launch instead of five is the lever.
// mu, P live in LDS between the phases; the resample is the arithmetic of cumsum_kernel + the gather for a single
// tile (all per-thread sums sit in wavefront 0), so the result is bit-identical to the separate launches.
constexpr int kSmallBank = 64;
constexpr int kSmallBlock = kSmallBank * kGroup;      // 1024

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
    __shared__ float tile[kSmallBank][GSE_NSIGMA * 5 + 1];       // predict: sigma points; update: deviations
    __shared__ double eta[kSmallBank][GSE_NSIGMA * 2];
    __shared__ float mu_s[kSmallBank][5];
    __shared__ float P_s[kSmallBank][25];
    __shared__ float w_s[kSmallBank];
    __shared__ double c_s[kSmallBank];
    __shared__ int anc_s[kSmallBank];
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
    // 1. predict: global -> LDS
    gsukf_predict_phase<kInjected>(mu + gi * 5, P + gi * 25, mu_s[gl], P_s[gl], tile[gl], live, j, b, gl,
                                   kInjected ? noise + gi * GSE_NSIGMA * 5 : nullptr, dp, status, a.pred);
    __syncthreads();
    // 2. update in LDS; the weight goes global -> LDS
    gsukf_update_phase(mu_s[gl], P_s[gl], w + gi, &w_s[gl], tile[gl], eta[gl], live, j, b, status, a.pred.io, a.pdf,
                       a.pred.sw);
    __syncthreads();
    // 3. resample, one tile: thread t < 16 of wavefront 0 owns weights 4t..4t+3, exactly as cumsum_kernel does
    __shared__ double total_s;
    if (tid < 64) {
        const int i0 = tid * 4;
        double sc[4];
        sc[0] = (i0 + 0 < N) ? (double)w_s[min(i0 + 0, kSmallBank - 1)] : 0.0;
        sc[1] = sc[0] + ((i0 + 1 < N) ? (double)w_s[min(i0 + 1, kSmallBank - 1)] : 0.0);
        sc[2] = sc[1] + ((i0 + 2 < N) ? (double)w_s[min(i0 + 2, kSmallBank - 1)] : 0.0);
        sc[3] = sc[2] + ((i0 + 3 < N) ? (double)w_s[min(i0 + 3, kSmallBank - 1)] : 0.0);
        const double inc = wave_inclusive_scan(sc[3], tid);
        const double W = __shfl(inc, 63, 64);                       // workgroup total: wavefronts 1.. hold zeros
        double excl = __shfl_up(inc, 1, 64);
        if (tid == 0) excl = 0.0;
        const double top = W / W;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            if (i0 + p < N) c_s[i0 + p] = (i0 + p == N - 1) ? top : fmin((0.0 + (excl + sc[p])) / W, top);
        if (tid == 0) total_s = W;
    }
    __syncthreads();
    if (tid < N) {                                                   // ancestor of output tid
        const double W = total_s;
        int k = 0;
        if ((W > 0.0) && (W < INFINITY)) {
            const double u = ((double)tid + a.r[b]) / (double)N;
            int lo = 0, hi = N - 1;
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (c_s[mid] < u) lo = mid + 1; else hi = mid;
            }
            k = lo;
        }
        anc_s[tid] = k;
        if (idx_out) idx_out[(int64_t)b * N + tid] = k;
        w[(int64_t)b * N + tid] = (float)(1.0 / (double)N);
    }
    __syncthreads();
    // 4. gather LDS -> global (in place: the source is LDS)
    if (live) {
        const int src = anc_s[gl];
        for (int e = j; e < 30; e += kGroup) {
            if (e < 5) mu[gi * 5 + e] = mu_s[src][e];
            else P[gi * 25 + (e - 5)] = P_s[src][e - 5];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// moments: one thread per Gaussian, blocks stride; 36 fp64 sums
__global__ __launch_bounds__(kBlock) void gsukf_moments_kernel(const float* __restrict__ mu, const float* __restrict__ P,
                                                               const float* __restrict__ w,
                                                               double* __restrict__ partials, int64_t N) {
    __shared__ double red[4 * 36];
    const int b = blockIdx.y;
    const float* mb = mu + (int64_t)b * N * 5;
    const float* Pb = P + (int64_t)b * N * 25;
    const float* wb = w + (int64_t)b * N;
    double pv[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) pv[k] = (double)mb[k];
    double acc[36];
#pragma unroll
    for (int c = 0; c < 36; ++c) acc[c] = 0.0;
    for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < N; g += (int64_t)gridDim.x * kBlock) {
        const double wp = (double)wb[g];
        double d[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) d[k] = (double)mb[g * 5 + k] - pv[k];
        acc[0] += wp;
        int c = 6;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const double wd = wp * d[k];
            acc[1 + k] += wd;
#pragma unroll
            for (int l = k; l < 5; ++l) acc[c++] += wd * d[l];
        }
#pragma unroll
        for (int k = 0; k < 5; ++k)
#pragma unroll
            for (int l = k; l < 5; ++l) acc[c++] += wp * (double)Pb[g * 25 + k * 5 + l];
    }
    double* out = partials + ((int64_t)b * kMomentBlocks + blockIdx.x) * kMomentWidth;
    block_sums<36>(acc, red, [&](int c, double total) { out[c] = total; });
}

static int pack_sigma_weights(SigmaWeights& sw, const float* w_sigma) {
    float s32 = 0.0f;
    double s64 = 0.0;
    // numpy sums 11 contiguous values pairwise only above 8 elements per block of its unrolled loop; for n = 11
    // its pairwise routine reduces to a plain left-to-right loop (n < 8 is the unrolled threshold... n < 128
    // takes the simple path with 8 accumulators only when n >= 8): replicate that order.
    // numpy pairwise_sum, n in [8, 128): r[0..7] = a[0..7]; for i = 8; i < n - n%8: r[k] += a[i+k];
    // res = ((r0+r1)+(r2+r3)) + ((r4+r5)+(r6+r7)); then the remaining n%8 elements are added in order.
    {
        float r[8];
        for (int k = 0; k < 8; ++k) r[k] = w_sigma[k];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (int i = 8; i < GSE_NSIGMA; ++i) res += w_sigma[i];
        s32 = res;
        double q[8];
        for (int k = 0; k < 8; ++k) q[k] = (double)w_sigma[k];
        double resd = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
        for (int i = 8; i < GSE_NSIGMA; ++i) resd += (double)w_sigma[i];
        s64 = resd;
    }
    for (int i = 0; i < GSE_NSIGMA; ++i) sw.w[i] = w_sigma[i];
    sw.scl32 = s32;
    sw.scl64 = s64;
    return 0;
}

static int check_gsukf(const char* fn, int B, int64_t N) {
    GSE_REQUIRE(B >= 1 && B <= GSE_MAX_B, "%s: B = %d is outside 1..%d", fn, B, GSE_MAX_B);
    GSE_REQUIRE(N >= 1 && N < ((int64_t)1 << 31) / GSE_NSIGMA, "%s: N = %lld out of range", fn, (long long)N);
    return GSE_OK;
}

}  // namespace gse

using namespace gse;

extern "C" {

int gse_gsukf_sigma_points(const float* mu, const float* P, float* sigmas, int B, int64_t N, int32_t* status,
                           void* stream) {
    if (int rc = check_gsukf("gse_gsukf_sigma_points", B, N)) return rc;
    GSE_REQUIRE(mu && P && sigmas, "gse_gsukf_sigma_points: null argument");
    const unsigned gx = (unsigned)((N + kGaussPerBlock - 1) / kGaussPerBlock);
    hipLaunchKernelGGL(gsukf_sigma_kernel, dim3(gx, B), dim3(kBlock), 0, (hipStream_t)stream, mu, P, sigmas, N, status);
    GSE_LAUNCH_CHECK("gsukf_sigma_kernel");
    return GSE_OK;
}

int gse_gsukf_predict(float* mu, float* P, int B, int64_t N, const double* u, double dt, const float* w_sigma,
                      const gse_mixture* state_pdf, const gse_rng* rng, const float* noise, int32_t* status,
                      void* stream) {
    if (int rc = check_gsukf("gse_gsukf_predict", B, N)) return rc;
    GSE_REQUIRE(mu && P && u && w_sigma, "gse_gsukf_predict: null argument");
    GSE_REQUIRE(noise || (state_pdf && rng), "gse_gsukf_predict: need either noise or (state_pdf, rng)");
    GsukfPredictArgs a;
    pack_io(a.io, B, u, nullptr);
    pack_sigma_weights(a.sw, w_sigma);
    a.dt = dt;
    const unsigned gx = (unsigned)((N + kGaussPerBlock - 1) / kGaussPerBlock);
    if (noise) {
        a.draw = DrawParams{};
        a.rng = RngKey{};
        hipLaunchKernelGGL(gsukf_predict_kernel<true>, dim3(gx, B), dim3(kBlock), 0, (hipStream_t)stream, mu, P, noise, N,
                           status, a);
    } else {
        if (int rc = check_mixture("gse_gsukf_predict", state_pdf, GSE_NX)) return rc;
        pack_draw(a.draw, *state_pdf);
        a.rng = make_rng_key(*rng, kPurposeGsukf);
        hipLaunchKernelGGL(gsukf_predict_kernel<false>, dim3(gx, B), dim3(kBlock), 0, (hipStream_t)stream, mu, P,
                           (const float*)nullptr, N, status, a);
    }
    GSE_LAUNCH_CHECK("gsukf_predict_kernel");
    return GSE_OK;
}

int gse_gsukf_update(float* mu, float* P, float* w, int B, int64_t N, const double* u, const double* z,
                     const float* w_sigma, const gse_mixture* measurement_pdf, int32_t* status, void* workspace,
                     void* stream) {
    if (int rc = check_gsukf("gse_gsukf_update", B, N)) return rc;
    GSE_REQUIRE(mu && P && w && u && z && w_sigma && measurement_pdf, "gse_gsukf_update: null argument");
    if (int rc = check_mixture("gse_gsukf_update", measurement_pdf, GSE_NY)) return rc;
    (void)workspace;
    GsukfUpdateArgs a;
    pack_io(a.io, B, u, z);
    pack_pdf2(a.pdf, *measurement_pdf);
    pack_sigma_weights(a.sw, w_sigma);
    const unsigned gx = (unsigned)((N + kGaussPerBlock - 1) / kGaussPerBlock);
    hipLaunchKernelGGL(gsukf_update_kernel, dim3(gx, B), dim3(kBlock), 0, (hipStream_t)stream, mu, P, w, N, status, a);
    GSE_LAUNCH_CHECK("gsukf_update_kernel");
    return GSE_OK;
}

int gse_gsukf_cycle(float* mu, float* P, float* w, int32_t* idx, int B, int64_t N, const double* u, double dt, const double* z,
                    const double* r, const float* w_sigma, const gse_mixture* state_pdf, const gse_rng* rng,
                    const float* noise, const gse_mixture* measurement_pdf, int32_t* status, void* stream) {
    if (int rc = check_gsukf("gse_gsukf_cycle", B, N)) return rc;
    GSE_REQUIRE(N <= kSmallBank, "gse_gsukf_cycle: at most %d Gaussians (got %lld); use the separate calls", kSmallBank,
                (long long)N);
    GSE_REQUIRE(mu && P && w && u && z && r && w_sigma && measurement_pdf, "gse_gsukf_cycle: null argument");
    GSE_REQUIRE(noise || (state_pdf && rng), "gse_gsukf_cycle: need either noise or (state_pdf, rng)");
    if (int rc = check_mixture("gse_gsukf_cycle", measurement_pdf, GSE_NY)) return rc;
    GsukfCycleArgs a;
    pack_io(a.pred.io, B, u, z);
    pack_sigma_weights(a.pred.sw, w_sigma);
    a.pred.dt = dt;
    pack_pdf2(a.pdf, *measurement_pdf);
    for (int b = 0; b < B; ++b) {
        GSE_REQUIRE(r[b] >= 0.0 && r[b] < 1.0, "gse_gsukf_cycle: r[%d] = %g is outside [0, 1)", b, r[b]);
        a.r[b] = r[b];
    }
    if (noise) {
        a.pred.draw = DrawParams{};
        a.pred.rng = RngKey{};
        hipLaunchKernelGGL(gsukf_cycle_kernel<true>, dim3(1, B), dim3(kSmallBlock), 0, (hipStream_t)stream, mu, P, w, noise,
                           idx, (int)N, status, a);
    } else {
        if (int rc = check_mixture("gse_gsukf_cycle", state_pdf, GSE_NX)) return rc;
        pack_draw(a.pred.draw, *state_pdf);
        a.pred.rng = make_rng_key(*rng, kPurposeGsukf);
        hipLaunchKernelGGL(gsukf_cycle_kernel<false>, dim3(1, B), dim3(kSmallBlock), 0, (hipStream_t)stream, mu, P, w,
                           (const float*)nullptr, idx, (int)N, status, a);
    }
    GSE_LAUNCH_CHECK("gsukf_cycle_kernel");
    return GSE_OK;
}

int gse_gsukf_resample(const float* mu_in, const float* P_in, const float* w, float* mu_out, float* P_out, float* w_out,
                       int32_t* idx, int B, int64_t N, const double* r, uint32_t flags, void* workspace, void* stream) {
    if (int rc = check_gsukf("gse_gsukf_resample", B, N)) return rc;
    GSE_REQUIRE(mu_in && P_in && w && mu_out && P_out && w_out && r && workspace, "gse_gsukf_resample: null argument");
    GSE_REQUIRE(mu_in != mu_out && P_in != P_out && w != w_out, "gse_gsukf_resample: outputs must not alias inputs");
    // (tile sums are always recomputed for the Gaussian bank: GSE_RESAMPLE_TILES_VALID is ignored)
    ResampleArgs ra;
    for (int b = 0; b < B; ++b) {
        GSE_REQUIRE(r[b] >= 0.0 && r[b] < 1.0, "gse_gsukf_resample: r[%d] = %g is outside [0, 1)", b, r[b]);
        ra.r[b] = r[b];
    }
    Workspace ws = carve_workspace(workspace, B, N);
    hipStream_t st = (hipStream_t)stream;
    if (int rc = launch_cumsum(w, N, B, N, flags & ~GSE_RESAMPLE_TILES_VALID, nullptr, ws, ws.cumsum, st)) return rc;
    GsukfPayload pl{mu_in, P_in, mu_out, P_out, N};
    return launch_gsukf_gather(pl, w_out, idx, B, N, ra, ws, st);
}

int gse_gsukf_moments(const float* mu, const float* P, const float* w, int B, int64_t N, double* out, void* workspace,
                      void* stream) {
    if (int rc = check_gsukf("gse_gsukf_moments", B, N)) return rc;
    GSE_REQUIRE(mu && P && w && out && workspace, "gse_gsukf_moments: null argument");
    Workspace ws = carve_workspace(workspace, B, N);
    const int nblk = (int)std::min<int64_t>(kMomentBlocks, (N + kBlock - 1) / kBlock);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gsukf_moments_kernel, dim3(nblk, B), dim3(kBlock), 0, st, mu, P, w, ws.partials, N);
    GSE_LAUNCH_CHECK("gsukf_moments_kernel");
    // out: S0, S1[5], S2[15], pivot[5] at 21, C[15] at 26  (the kernel's order is S0,S1,S2,C -> remapped below)
    launch_moments_finish(ws.partials, nblk, 36, out, 46, B, mu, N * 5, 1, 36, st);
    GSE_LAUNCH_CHECK("moments_finish_kernel");
    return GSE_OK;
}

}  // extern "C"
