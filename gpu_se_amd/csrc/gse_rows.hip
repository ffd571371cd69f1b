// gse_rows.hip -- the stand-alone pieces the reference's per-subroutine benchmarks call one at a time
// (results/pf_openloop/pf_run_seq.py:134-296): row-wise f / g, mixture pdf / draw, normalised cumulative sum,
// systematic index from a cumulative sum, and the raw RNG words.  The filters themselves use the fused
// kernels in gse_pf.hip / gse_gsukf.hip; these exist for API completeness and for the parity tests.
#include "gse_rows_kernels.h"
#include "gse_generic_kernels.h"
#include "gse_rtc.h"

#include <string.h>

namespace gse {


// General mixture density, nx <= 5, nd <= 4  (MultivariateGaussianSum.py:39-63)
struct PdfN {
    int32_t nd, nx;
    double cw[GSE_MAX_ND];
    float mean[GSE_MAX_ND][GSE_NX];
    double inv[GSE_MAX_ND][GSE_NX][GSE_NX];
};

template <typename T>
__global__ __launch_bounds__(kBlock) void mixture_pdf_kernel(const T* __restrict__ e, double* __restrict__ out, int64_t M,
                                                             const PdfN m) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= M) return;
    T x[GSE_NX];
    for (int k = 0; k < m.nx; ++k) x[k] = e[i * m.nx + k];
    double total = 0.0;
    for (int d = 0; d < m.nd; ++d) {
        double a[GSE_NX];
        // x - mean in the precision of x (fp32 points stay fp32 here, as in the reference), then fp64
        for (int k = 0; k < m.nx; ++k) a[k] = (double)(x[k] - (T)m.mean[d][k]);
        double q = 0.0;
        for (int j = 0; j < m.nx; ++j) {
            double t = a[0] * m.inv[d][0][j];
            for (int k = 1; k < m.nx; ++k) t = t + a[k] * m.inv[d][k][j];
            q = (j == 0) ? t * a[0] : q + t * a[j];
        }
        const double term = m.cw[d] * exp(-0.5 * q);
        total = (d == 0) ? term : total + term;
    }
    out[i] = total;
}

struct DrawArgs {
    DrawParams draw;
    RngKey rng;
};

// out: row-major [M][nx] (ld == 0) or state-major [nx][ld]
__global__ __launch_bounds__(kBlock) void mixture_draw_kernel(float* __restrict__ out, int64_t M, int64_t ld,
                                                              const DrawArgs a) {
    __shared__ DrawParams dp;
    {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&a.draw);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&dp);
        for (int i = threadIdx.x; i < (int)(sizeof(DrawParams) / 4); i += kBlock) dst[i] = src[i];
        __syncthreads();
    }
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= M) return;
    uint32_t wd[8];
    rng_words8(a.rng, (uint64_t)i, 0u, wd);
    float e[5];
    mixture_draw5(dp, wd, e);
    const int nx = dp.nx;
    if (ld == 0) {
        for (int k = 0; k < nx; ++k) out[i * nx + k] = e[k];
    } else {
        for (int k = 0; k < nx; ++k) out[(int64_t)k * ld + i] = e[k];
    }
}

__global__ __launch_bounds__(kBlock) void rng_raw_kernel(uint32_t* __restrict__ out, int64_t M, const RngKey key) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= M) return;
    uint32_t wd[8];
    rng_words8(key, (uint64_t)i, 0u, wd);
#pragma unroll
    for (int k = 0; k < 8; ++k) out[i * 8 + k] = wd[k];
}

// idx[i] = first k with c[k] >= (i + r)/N (CPU semantics, particle.py:96-100), by binary search in global memory
__global__ __launch_bounds__(kBlock) void systematic_index_kernel(const double* __restrict__ c, int64_t* __restrict__ idx,
                                                                  int64_t N, double r) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= N) return;
    const double u = ((double)i + r) / (double)N;
    int64_t lo = 0, hi = N - 1;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (c[mid] < u) lo = mid + 1; else hi = mid;
    }
    idx[i] = lo;
}

// MultivariateGaussianSum.draw / .pdf for mixtures of more than GSE_NX dimensions (gse_generic_kernels.h)
struct MixtureNArgs {
    gse_mixture_n m;
    RngKey rng;
};

__global__ __launch_bounds__(kBlock) void mixture_draw_n_kernel(float* __restrict__ out, int64_t M, const MixtureNArgs a) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= M) return;
    float e[kGMaxNx];
    generic_draw_n(a.m, a.rng, (uint64_t)i, e);
    for (int k = 0; k < a.m.n; ++k) out[i * a.m.n + k] = e[k];
}

__global__ __launch_bounds__(kBlock) void mixture_pdf_n_kernel(const double* __restrict__ e, int64_t M, const MixtureNArgs a,
                                                               double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= M) return;
    double v[kGMaxNx];
    for (int k = 0; k < kGMaxNx; ++k) v[k] = (k < a.m.n) ? e[i * a.m.n + k] : 0.0;
    out[i] = generic_pdf_n(a.m, v);
}

static int pack_pdfn(PdfN& p, const gse_mixture& m) {
    memset(&p, 0, sizeof(p));
    p.nd = m.nd;
    p.nx = m.nx;
    for (int d = 0; d < m.nd; ++d) {
        p.cw[d] = (double)(m.consts[d] * m.weights[d]);
        for (int i = 0; i < m.nx; ++i) {
            p.mean[d][i] = m.means[d][i];
            for (int j = 0; j < m.nx; ++j) p.inv[d][i][j] = m.inv_cov[d][i][j];
        }
    }
    return 0;
}

static unsigned blocks_for(int64_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

}  // namespace gse

using namespace gse;

extern "C" {

int gse_model_f(const float* x, float* out, int64_t M, const double* u, double dt, void* stream) {
    GSE_REQUIRE(x && out && u && M >= 1, "gse_model_f: bad argument");
    hipLaunchKernelGGL(model_f_kernel, dim3(blocks_for(M)), dim3(kBlock), 0, (hipStream_t)stream, x, out, M, u[0], u[1], dt);
    GSE_LAUNCH_CHECK("model_f_kernel");
    return GSE_OK;
}

int gse_model_g_u(const float* x, float* out, int64_t M, const double* u, void* stream) {
    GSE_REQUIRE(x && out && M >= 1, "gse_model_g: bad argument");
    hipLaunchKernelGGL(model_g_kernel, dim3(blocks_for(M)), dim3(kBlock), 0, (hipStream_t)stream, x, out, M,
                       u ? u[0] : 0.0, u ? u[1] : 0.0);
    GSE_LAUNCH_CHECK("model_g_kernel");
    return GSE_OK;
}

int gse_model_g(const float* x, float* out, int64_t M, void* stream) { return gse_model_g_u(x, out, M, nullptr, stream); }

// 0: the built-in bioreactor model; otherwise the tag of the user model this library was compiled for
int gse_model_f_model(void* model, const float* x, float* out, int64_t M, const double* u, double dt, void* stream) {
    GSE_REQUIRE(model && x && out && u && M >= 0, "gse_model_f_model: bad argument");
    if (M == 0) return GSE_OK;
    double fg = u[0], fm = u[1];
    void* params[] = {(void*)&x, (void*)&out, (void*)&M, (void*)&fg, (void*)&fm, (void*)&dt};
    return launch_user_kernel(static_cast<UserModel*>(model), "gse::model_f_kernel", dim3(blocks_for(M)), dim3(kBlock), 0,
                              (hipStream_t)stream, params);
}

int gse_model_g_u_model(void* model, const float* x, float* out, int64_t M, const double* u, void* stream) {
    GSE_REQUIRE(model && x && out && u && M >= 0, "gse_model_g_u_model: bad argument");
    if (M == 0) return GSE_OK;
    double u0 = u[0], u1 = u[1];
    void* params[] = {(void*)&x, (void*)&out, (void*)&M, (void*)&u0, (void*)&u1};
    return launch_user_kernel(static_cast<UserModel*>(model), "gse::model_g_kernel", dim3(blocks_for(M)), dim3(kBlock), 0,
                              (hipStream_t)stream, params);
}

// (-DGSE_USER_MODEL_TAG, gpu_se_amd/usermodel.py)
uint64_t gse_model_tag(void) {
#ifdef GSE_USER_MODEL_TAG
    return GSE_USER_MODEL_TAG;
#else
    return 0;
#endif
}

static int mixture_pdf_common(const void* e, bool f64, int64_t M, const gse_mixture* m, double* out, void* stream) {
    GSE_REQUIRE(e && out && m && M >= 1, "gse_mixture_pdf: bad argument");
    GSE_REQUIRE(m->nd >= 1 && m->nd <= GSE_MAX_ND && m->nx >= 1 && m->nx <= GSE_NX,
                "gse_mixture_pdf: mixture has nd = %d, nx = %d; supported nd 1..%d, nx 1..%d", m->nd, m->nx, GSE_MAX_ND,
                GSE_NX);
    PdfN p;
    pack_pdfn(p, *m);
    if (f64)
        hipLaunchKernelGGL(mixture_pdf_kernel<double>, dim3(blocks_for(M)), dim3(kBlock), 0, (hipStream_t)stream,
                           (const double*)e, out, M, p);
    else
        hipLaunchKernelGGL(mixture_pdf_kernel<float>, dim3(blocks_for(M)), dim3(kBlock), 0, (hipStream_t)stream,
                           (const float*)e, out, M, p);
    GSE_LAUNCH_CHECK("mixture_pdf_kernel");
    return GSE_OK;
}

int gse_mixture_pdf_f32(const float* e, int64_t M, const gse_mixture* m, double* out, void* stream) {
    return mixture_pdf_common(e, false, M, m, out, stream);
}

int gse_mixture_pdf_f64(const double* e, int64_t M, const gse_mixture* m, double* out, void* stream) {
    return mixture_pdf_common(e, true, M, m, out, stream);
}

int gse_mixture_draw(float* out, int64_t M, int64_t ld, const gse_mixture* m, const gse_rng* rng, void* stream) {
    GSE_REQUIRE(out && m && rng && M >= 1, "gse_mixture_draw: bad argument");
    GSE_REQUIRE(ld == 0 || ld >= M, "gse_mixture_draw: ld = %lld must be 0 (row-major) or >= M", (long long)ld);
    GSE_REQUIRE(m->nd >= 1 && m->nd <= GSE_MAX_ND && m->nx >= 1 && m->nx <= GSE_NX,
                "gse_mixture_draw: mixture has nd = %d, nx = %d; supported nd 1..%d, nx 1..%d", m->nd, m->nx, GSE_MAX_ND,
                GSE_NX);
    DrawArgs a;
    pack_draw(a.draw, *m);
    a.rng = make_rng_key(*rng, kPurposeDraw);
    hipLaunchKernelGGL(mixture_draw_kernel, dim3(blocks_for(M)), dim3(kBlock), 0, (hipStream_t)stream, out, M, ld, a);
    GSE_LAUNCH_CHECK("mixture_draw_kernel");
    return GSE_OK;
}

int gse_cumsum_normalised(const float* w, double* c, int64_t N, void* workspace, void* stream) {
    GSE_REQUIRE(w && c && workspace && N >= 1, "gse_cumsum_normalised: bad argument");
    Workspace ws = carve_workspace(workspace, 1, N);
    return launch_cumsum(w, N, 1, N, 0, nullptr, ws, c, (hipStream_t)stream);
}

int gse_systematic_index(const double* c, int64_t* idx, int64_t N, double r, void* stream) {
    GSE_REQUIRE(c && idx && N >= 1, "gse_systematic_index: bad argument");
    hipLaunchKernelGGL(systematic_index_kernel, dim3(blocks_for(N)), dim3(kBlock), 0, (hipStream_t)stream, c, idx, N, r);
    GSE_LAUNCH_CHECK("systematic_index_kernel");
    return GSE_OK;
}

int gse_rng_raw(uint32_t* out, int64_t M, const gse_rng* rng, uint32_t purpose, void* stream) {
    GSE_REQUIRE(out && rng && M >= 1, "gse_rng_raw: bad argument");
    const RngKey key = make_rng_key(*rng, purpose);
    hipLaunchKernelGGL(rng_raw_kernel, dim3(blocks_for(M)), dim3(kBlock), 0, (hipStream_t)stream, out, M, key);
    GSE_LAUNCH_CHECK("rng_raw_kernel");
    return GSE_OK;
}

static int check_mixture_n(const char* fn, const gse_mixture_n* m) {
    GSE_REQUIRE(m, "%s: null mixture", fn);
    GSE_REQUIRE(m->nd >= 1 && m->nd <= GSE_MAX_ND && m->n >= 1 && m->n <= GSE_N_MAX_NX, "%s: mixture of %d components in %d "
                "dimensions (at most %d, %d)", fn, m->nd, m->n, GSE_MAX_ND, GSE_N_MAX_NX);
    return GSE_OK;
}

int gse_mixture_draw_n(float* out, int64_t M, const gse_mixture_n* m, const gse_rng* rng, void* stream) {
    GSE_REQUIRE(out && rng && M >= 1, "gse_mixture_draw_n: bad argument");
    if (int rc = check_mixture_n("gse_mixture_draw_n", m)) return rc;
    MixtureNArgs a;
    a.m = *m;
    a.rng = make_rng_key(*rng, kPurposeDraw);
    hipLaunchKernelGGL(mixture_draw_n_kernel, dim3(blocks_for(M)), dim3(kBlock), 0, (hipStream_t)stream, out, M, a);
    GSE_LAUNCH_CHECK("mixture_draw_n_kernel");
    return GSE_OK;
}

int gse_mixture_pdf_n(const double* e, int64_t M, const gse_mixture_n* m, double* out, void* stream) {
    GSE_REQUIRE(e && out && M >= 1, "gse_mixture_pdf_n: bad argument");
    if (int rc = check_mixture_n("gse_mixture_pdf_n", m)) return rc;
    MixtureNArgs a;
    a.m = *m;
    a.rng = RngKey{};
    hipLaunchKernelGGL(mixture_pdf_n_kernel, dim3(blocks_for(M)), dim3(kBlock), 0, (hipStream_t)stream, e, M, a, out);
    GSE_LAUNCH_CHECK("mixture_pdf_n_kernel");
    return GSE_OK;
}

}  // extern "C"
