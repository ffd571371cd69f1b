// gse_pf.hip -- bootstrap particle filter on gfx950: predict / update / systematic resample / moments.
//
// One workgroup = 256 threads = 4 wavefronts owns one TILE of 1024 consecutive particles of one filter;
// every thread owns 4 consecutive particles, so each state row is read and written with one dwordx4 per
// lane (1 KiB per wave-instruction, fully coalesced) from the state-major layout x[b][k][i].
// Everything here is HBM-bound streaming work (140 algorithmic bytes / particle / cycle); the
// kernels keep one pass per method and no intermediate arrays:
//
//   predict   r 20 B + w 20 B   ODE step (fp64 in registers) + Philox/Box-Muller mixture draw, fused
//   update    r 8 B + rw 8 B    only the two observed state rows are read; likelihood in fp64;
//                               workgroup sum of the new weights -> one tile sum (feeds resample)
//   resample  r 4 + 20, w 24 B  tile prefix -> in-LDS fp64 scan of the tile's weights -> each output slot
//                               binary-searches its ancestor in LDS -> gather; no cumsum / index array in HBM
#include "gse_host.h"

namespace gse {

// ------------------------------------------------------------------------------------------------
// predict
// ------------------------------------------------------------------------------------------------
struct PredictArgs {
    FilterIOs io;
    DrawParams draw;
    RngKey rng;
    double dt;
};

template <bool kInjected>
__global__ __launch_bounds__(kBlock) void pf_predict_kernel(float* __restrict__ x, const float* __restrict__ noise,
                                                            int64_t N, int64_t ld, const PredictArgs a) {
    __shared__ DrawParams dp;
    if (!kInjected) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&a.draw);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&dp);
        for (int i = threadIdx.x; i < (int)(sizeof(DrawParams) / 4); i += kBlock) dst[i] = src[i];
        __syncthreads();
    }
    const int b = blockIdx.y;
    const int64_t i0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kPerThread;
    if (i0 >= N) return;
    float* xb = x + (int64_t)b * GSE_NX * ld + i0;
    const double fg = a.io.f[b].u0, fm = a.io.f[b].u1;

    float4 row[GSE_NX];
#pragma unroll
    for (int k = 0; k < GSE_NX; ++k) row[k] = *reinterpret_cast<const float4*>(xb + k * ld);
    float4 nz[GSE_NX];
    if (kInjected) {
        const float* nb = noise + (int64_t)b * GSE_NX * ld + i0;
#pragma unroll
        for (int k = 0; k < GSE_NX; ++k) nz[k] = *reinterpret_cast<const float4*>(nb + k * ld);
    }

#pragma unroll
    for (int p = 0; p < kPerThread; ++p) {
        float s[GSE_NX];
#pragma unroll
        for (int k = 0; k < GSE_NX; ++k) s[k] = reinterpret_cast<const float*>(&row[k])[p];
        bioreactor_step(s, fg, fm, a.dt);
        float e[GSE_NX];
        if (kInjected) {
#pragma unroll
            for (int k = 0; k < GSE_NX; ++k) e[k] = reinterpret_cast<const float*>(&nz[k])[p];
        } else {
            uint32_t wd[8];
            rng_words8(a.rng, (uint64_t)(i0 + p), (uint32_t)b, wd);
            mixture_draw5(dp, wd, e);
        }
#pragma unroll
        for (int k = 0; k < GSE_NX; ++k) reinterpret_cast<float*>(&row[k])[p] = s[k] + e[k];
    }
#pragma unroll
    for (int k = 0; k < GSE_NX; ++k) *reinterpret_cast<float4*>(xb + k * ld) = row[k];
}

// ------------------------------------------------------------------------------------------------
// update
// ------------------------------------------------------------------------------------------------
struct UpdateArgs {
    FilterIOs io;
    Pdf2 pdf;
};

// Sum of a thread's 4 weights (entries at or beyond N count as zero) -- the same expression in the update
// kernel and in the standalone tile-sum kernel.
__device__ __forceinline__ double masked_sum4(const float4& w, int64_t i0, int64_t N) {
    double s = 0.0;
    if (i0 + 0 < N) s = (double)w.x;
    if (i0 + 1 < N) s += (double)w.y;
    if (i0 + 2 < N) s += (double)w.z;
    if (i0 + 3 < N) s += (double)w.w;
    return s;
}

__global__ __launch_bounds__(kBlock) void pf_update_kernel(const float* __restrict__ x, float* __restrict__ w,
                                                           double* __restrict__ tile_sums, int64_t N, int64_t ld,
                                                           int T, const UpdateArgs a) {
    __shared__ double red[4];
    const int b = blockIdx.y;
    const int64_t i0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kPerThread;
    double mine = 0.0;
    if (i0 < N) {
        const float* xb = x + (int64_t)b * GSE_NX * ld + i0;
        float* wb = w + (int64_t)b * ld + i0;
        const float4 cg = *reinterpret_cast<const float4*>(xb);                 // state row 0
        const float4 cfa = *reinterpret_cast<const float4*>(xb + 2 * ld);       // state row 2
        float4 wv = *reinterpret_cast<const float4*>(wb);
        const double z0 = a.io.f[b].z0, z1 = a.io.f[b].z1;
#pragma unroll
        for (int p = 0; p < kPerThread; ++p) {
            float y0, y1;
            bioreactor_output(reinterpret_cast<const float*>(&cg)[p], reinterpret_cast<const float*>(&cfa)[p], y0, y1);
            const double lik = mixture_pdf2(a.pdf, z0 - (double)y0, z1 - (double)y1);
            float& wp = reinterpret_cast<float*>(&wv)[p];
            wp = (float)((double)wp * lik);
        }
        *reinterpret_cast<float4*>(wb) = wv;
        mine = masked_sum4(wv, i0, N);
    }
    const double total = block_sum(mine, red);
    if (threadIdx.x == 0) tile_sums[(int64_t)b * T + blockIdx.x] = total;
}

// Four consecutive weights starting at element i0 (a multiple of 4) of a row; entries at or beyond N read as 0.
// Rows with ld % 4 == 0 are 16-byte aligned and padded (one dwordx4); otherwise (row-major GS-UKF weights of
// filter b > 0) the row start is only dword aligned and has no padding, so the loads are scalar.
__device__ __forceinline__ float4 load_weights4(const float* __restrict__ row, int64_t i0, int64_t N, int64_t ldw) {
    float4 wv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i0 >= N) return wv;
    if ((ldw & 3) == 0) {
        wv = *reinterpret_cast<const float4*>(row + i0);
    } else {
        wv.x = row[i0];
        if (i0 + 1 < N) wv.y = row[i0 + 1];
        if (i0 + 2 < N) wv.z = row[i0 + 2];
        if (i0 + 3 < N) wv.w = row[i0 + 3];
    }
    return wv;
}

// Tile sums for a resample that does not directly follow an update (weights assigned by the caller).
__global__ __launch_bounds__(kBlock) void tile_sum_kernel(const float* __restrict__ w, double* __restrict__ tile_sums,
                                                          int64_t N, int64_t ldw, int T) {
    __shared__ double red[4];
    const int b = blockIdx.y;
    const int64_t i0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kPerThread;
    const float4 wv = load_weights4(w + (int64_t)b * ldw, i0, N, ldw);
    const double mine = masked_sum4(wv, i0, N);
    const double total = block_sum(mine, red);
    if (threadIdx.x == 0) tile_sums[(int64_t)b * T + blockIdx.x] = total;
}

// Exclusive prefix over the T tile sums of each filter: prefix[q] = s_0 + ... + s_{q-1}, prefix[T] = total.
// One workgroup per filter; thread j owns the contiguous chunk [j*m, (j+1)*m).
__global__ __launch_bounds__(kBlock) void tile_scan_kernel(const double* __restrict__ tile_sums,
                                                           double* __restrict__ prefix, int T) {
    __shared__ double red[4];
    const int b = blockIdx.x;
    const double* s = tile_sums + (int64_t)b * T;
    double* p = prefix + (int64_t)b * (T + 1);
    const int m = (T + kBlock - 1) / kBlock;
    const int lo = threadIdx.x * m;
    const int hi = min(lo + m, T);
    double chunk = 0.0;
    for (int q = lo; q < hi; ++q) chunk += s[q];
    double total;
    double run = block_exclusive_scan(chunk, red, total);
    if (threadIdx.x == 0) p[0] = 0.0;
    for (int q = lo; q < hi; ++q) {
        run += s[q];
        p[q + 1] = run;
    }
}

// ------------------------------------------------------------------------------------------------
// systematic resample
// ------------------------------------------------------------------------------------------------
// #{ j in [0, N) : fl(fl(j + r) / N) <= y }  -- evaluated with exactly the reference's expression for u_j
// (filter/particle.py:97 / :247): closed-form estimate, then exact fix-up by direct comparison.
__device__ __forceinline__ int64_t count_thresholds_le(double y, double r, int64_t N) {
    const double Nd = (double)N;
    double est = floor(y * Nd - r) + 1.0;
    est = fmin(fmax(est, 0.0), Nd);                       // NaN-safe clamp
    int64_t j = (int64_t)est;
    while (j > 0 && ((double)(j - 1) + r) / Nd > y) --j;
    while (j < N && ((double)j + r) / Nd <= y) ++j;
    return j;
}

template <class Payload>
__global__ __launch_bounds__(kBlock) void resample_kernel(const Payload pl, const float* __restrict__ w,
                                                          float* __restrict__ w_out, int32_t* __restrict__ idx_out,
                                                          const double* __restrict__ tile_prefix, int64_t N,
                                                          int64_t ldw, int T, const ResampleArgs ra) {
    __shared__ double sc[kTile];          // normalised cumulative weights of this tile
    __shared__ double red[4];
    __shared__ int64_t bounds[2];

    const int b = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
    const double* pf = tile_prefix + (int64_t)b * (T + 1);
    const double P0 = pf[t], P1 = pf[t + 1], W = pf[T];
    const double r = ra.r[b];
    const int64_t k0 = (int64_t)t * kTile;
    const int cnt = (int)min((int64_t)kTile, N - k0);
    const float w_uniform = (float)(1.0 / (double)N);
    float* wo = w_out + (int64_t)b * ldw;
    int32_t* io = idx_out ? idx_out + (int64_t)b * ldw : nullptr;

    // Weights that do not sum to a positive finite number: the reference's normalised cumsum is all-NaN and
    // its sweep never advances (filter/particle.py:98), so every offspring copies particle 0.
    if (!(W > 0.0) || !(W < INFINITY)) {
        if (t == 0) {
            for (int64_t j = tid; j < N; j += kBlock) {
                pl.copy(b, 0, j);
                wo[j] = w_uniform;
                if (io) io[j] = 0;
            }
        }
        return;
    }

    // 1. fp64 inclusive scan of the tile's weights: 4 per thread in registers, then across the workgroup.
    const int64_t i0 = k0 + (int64_t)tid * kPerThread;
    const float4 wv = load_weights4(w + (int64_t)b * ldw, i0, N, ldw);
    double s[kPerThread];
    s[0] = (i0 + 0 < N) ? (double)wv.x : 0.0;
    s[1] = s[0] + ((i0 + 1 < N) ? (double)wv.y : 0.0);
    s[2] = s[1] + ((i0 + 2 < N) ? (double)wv.z : 0.0);
    s[3] = s[2] + ((i0 + 3 < N) ? (double)wv.w : 0.0);
    double tile_total;
    const double excl = block_exclusive_scan(s[3], red, tile_total);

    // 2. normalised cumulative sums into LDS; the tile's last entry is pinned to prefix[t+1]/W, the value the
    //    next tile uses as its lower boundary, so the partition of the outputs among tiles is exact.
    const double top = P1 / W;
#pragma unroll
    for (int p = 0; p < kPerThread; ++p) {
        const int q = tid * kPerThread + p;
        if (q < cnt) sc[q] = (q == cnt - 1) ? top : fmin((P0 + (excl + s[p])) / W, top);
    }
    if (tid == 0) bounds[0] = (t == 0) ? 0 : count_thresholds_le(P0 / W, r, N);
    if (tid == 64) bounds[1] = (t == T - 1) ? N : count_thresholds_le(top, r, N);
    __syncthreads();
    const int64_t o0 = bounds[0], o1 = bounds[1];

    // 3. every output slot of this tile's range finds its ancestor by binary search in LDS and gathers it.
    const double Nd = (double)N;
    for (int64_t j = o0 + tid; j < o1; j += kBlock) {
        const double u = ((double)j + r) / Nd;
        int lo = 0, hi = cnt - 1;
#pragma unroll
        for (int it = 0; it < 10; ++it) {                 // 2^10 = kTile
            const int mid = (lo + hi) >> 1;
            const bool open = lo < hi;
            const bool less = open && (sc[mid] < u);
            lo = less ? mid + 1 : lo;
            hi = (open && !less) ? mid : hi;
        }
        const int64_t anc = k0 + lo;
        pl.copy(b, anc, j);
        wo[j] = w_uniform;
        if (io) io[j] = (int32_t)anc;
    }
}

// ------------------------------------------------------------------------------------------------
// moments
// ------------------------------------------------------------------------------------------------
// Per-workgroup partial sums of {w, w d, w d d^T} with d = x - pivot, fp64; blocks stride over the tiles.
__global__ __launch_bounds__(kBlock) void pf_moments_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            double* __restrict__ partials, int64_t N, int64_t ld) {
    __shared__ double red[4];
    const int b = blockIdx.y;
    const float* xb = x + (int64_t)b * GSE_NX * ld;
    const float* wb = w + (int64_t)b * ld;
    double pv[GSE_NX];
#pragma unroll
    for (int k = 0; k < GSE_NX; ++k) pv[k] = (double)xb[k * ld];          // pivot = particle 0
    double acc[21];
#pragma unroll
    for (int c = 0; c < 21; ++c) acc[c] = 0.0;
    const int64_t groups = (N + 3) / 4;
    for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < groups; g += (int64_t)gridDim.x * kBlock) {
        const int64_t i0 = g * 4;
        const float4 wv = *reinterpret_cast<const float4*>(wb + i0);
        float4 row[GSE_NX];
#pragma unroll
        for (int k = 0; k < GSE_NX; ++k) row[k] = *reinterpret_cast<const float4*>(xb + k * ld + i0);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (i0 + p < N) {
                const double wp = (double)reinterpret_cast<const float*>(&wv)[p];
                double d[GSE_NX];
#pragma unroll
                for (int k = 0; k < GSE_NX; ++k) d[k] = (double)reinterpret_cast<const float*>(&row[k])[p] - pv[k];
                acc[0] += wp;
                int c = 6;
#pragma unroll
                for (int k = 0; k < GSE_NX; ++k) {
                    const double wd = wp * d[k];
                    acc[1 + k] += wd;
#pragma unroll
                    for (int l = k; l < GSE_NX; ++l) acc[c++] += wd * d[l];
                }
            }
        }
    }
    double* out = partials + ((int64_t)b * kMomentBlocks + blockIdx.x) * kMomentWidth;
#pragma unroll
    for (int c = 0; c < 21; ++c) {
        const double tot = block_sum(acc[c], red);
        if (threadIdx.x == 0) out[c] = tot;
    }
}

// Fixed-order reduction of the per-workgroup partials -> out[b][0..width); appends `tail` pivot values.
__global__ __launch_bounds__(kBlock) void moments_finish_kernel(const double* __restrict__ partials, int nblk, int width,
                                                                double* __restrict__ out, int out_stride,
                                                                const float* __restrict__ pivot_src,
                                                                int64_t pivot_filter_stride, int64_t pivot_elem_stride,
                                                                int pivot_at) {
    __shared__ double red[4];
    const int b = blockIdx.x;
    const double* pb = partials + (int64_t)b * kMomentBlocks * kMomentWidth;
    for (int c = 0; c < width; ++c) {
        double v = 0.0;
        for (int q = threadIdx.x; q < nblk; q += kBlock) v += pb[(int64_t)q * kMomentWidth + c];
        const double tot = block_sum(v, red);
        if (threadIdx.x == 0) out[(int64_t)b * out_stride + c] = tot;
    }
    if (threadIdx.x < GSE_NX)
        out[(int64_t)b * out_stride + pivot_at + threadIdx.x] =
            (double)pivot_src[(int64_t)b * pivot_filter_stride + threadIdx.x * pivot_elem_stride];
}

// Exposed to gse_gsukf.hip
void launch_tile_sums(const float* w, double* tile_sums, int B, int64_t N, int64_t ldw, int T, hipStream_t st) {
    hipLaunchKernelGGL(tile_sum_kernel, dim3(T, B), dim3(kBlock), 0, st, w, tile_sums, N, ldw, T);
}
void launch_tile_scan(const double* tile_sums, double* prefix, int B, int T, hipStream_t st) {
    hipLaunchKernelGGL(tile_scan_kernel, dim3(B), dim3(kBlock), 0, st, tile_sums, prefix, T);
}
void launch_gsukf_resample(const GsukfPayload& pl, const float* w, float* w_out, int32_t* idx, const double* prefix,
                           int B, int64_t N, int T, const ResampleArgs& ra, hipStream_t st) {
    hipLaunchKernelGGL(resample_kernel<GsukfPayload>, dim3(T, B), dim3(kBlock), 0, st, pl, w, w_out, idx, prefix, N, N,
                       T, ra);
}
void launch_moments_finish(const double* partials, int nblk, int width, double* out, int out_stride, int B,
                           const float* pivot_src, int64_t pfs, int64_t pes, int pivot_at, hipStream_t st) {
    hipLaunchKernelGGL(moments_finish_kernel, dim3(B), dim3(kBlock), 0, st, partials, nblk, width, out, out_stride,
                       pivot_src, pfs, pes, pivot_at);
}

}  // namespace gse

using namespace gse;

// ================================================================================================
// C ABI
// ================================================================================================
extern "C" {

int gse_pf_predict(float* x, int64_t ld, int B, int64_t N, const double* u, double dt, const gse_mixture* state_pdf,
                   const gse_rng* rng, void* stream) {
    if (int rc = check_pf_geometry("gse_pf_predict", B, N, ld)) return rc;
    GSE_REQUIRE(x && u && state_pdf && rng, "gse_pf_predict: null argument");
    GSE_REQUIRE(aligned16(x), "gse_pf_predict: x must be 16-byte aligned");
    if (int rc = check_mixture("gse_pf_predict", state_pdf, GSE_NX)) return rc;
    PredictArgs a;
    pack_io(a.io, B, u, nullptr);
    pack_draw(a.draw, *state_pdf);
    a.rng = make_rng_key(*rng, kPurposePredict);
    a.dt = dt;
    hipLaunchKernelGGL(pf_predict_kernel<false>, dim3((unsigned)tiles_of(N), B), dim3(kBlock), 0, (hipStream_t)stream, x,
                       (const float*)nullptr, N, ld, a);
    GSE_LAUNCH_CHECK("pf_predict_kernel");
    return GSE_OK;
}

int gse_pf_predict_injected(float* x, int64_t ld, int B, int64_t N, const double* u, double dt, const float* noise,
                            void* stream) {
    if (int rc = check_pf_geometry("gse_pf_predict_injected", B, N, ld)) return rc;
    GSE_REQUIRE(x && u && noise, "gse_pf_predict_injected: null argument");
    GSE_REQUIRE(aligned16(x) && aligned16(noise), "gse_pf_predict_injected: x, noise must be 16-byte aligned");
    PredictArgs a;
    pack_io(a.io, B, u, nullptr);
    a.draw = DrawParams{};
    a.rng = RngKey{};
    a.dt = dt;
    hipLaunchKernelGGL(pf_predict_kernel<true>, dim3((unsigned)tiles_of(N), B), dim3(kBlock), 0, (hipStream_t)stream, x,
                       noise, N, ld, a);
    GSE_LAUNCH_CHECK("pf_predict_kernel<injected>");
    return GSE_OK;
}

int gse_pf_update(const float* x, float* w, int64_t ld, int B, int64_t N, const double* u, const double* z,
                  const gse_mixture* measurement_pdf, void* workspace, void* stream) {
    if (int rc = check_pf_geometry("gse_pf_update", B, N, ld)) return rc;
    GSE_REQUIRE(x && w && u && z && measurement_pdf && workspace, "gse_pf_update: null argument");
    GSE_REQUIRE(aligned16(x) && aligned16(w) && aligned16(workspace), "gse_pf_update: pointers must be 16-byte aligned");
    if (int rc = check_mixture("gse_pf_update", measurement_pdf, GSE_NY)) return rc;
    UpdateArgs a;
    pack_io(a.io, B, u, z);
    pack_pdf2(a.pdf, *measurement_pdf);
    const int T = (int)tiles_of(N);
    Workspace ws = carve_workspace(workspace, B, N);
    hipLaunchKernelGGL(pf_update_kernel, dim3(T, B), dim3(kBlock), 0, (hipStream_t)stream, x, w, ws.tile_sums, N, ld, T, a);
    GSE_LAUNCH_CHECK("pf_update_kernel");
    return GSE_OK;
}

int gse_pf_resample(const float* x_in, const float* w, float* x_out, float* w_out, int32_t* idx, int64_t ld, int B,
                    int64_t N, const double* r, uint32_t flags, void* workspace, void* stream) {
    if (int rc = check_pf_geometry("gse_pf_resample", B, N, ld)) return rc;
    GSE_REQUIRE(x_in && w && x_out && w_out && r && workspace, "gse_pf_resample: null argument");
    GSE_REQUIRE(x_in != x_out && w != w_out, "gse_pf_resample: x_out / w_out must not alias x_in / w");
    GSE_REQUIRE(aligned16(x_in) && aligned16(w) && aligned16(x_out) && aligned16(w_out) && aligned16(workspace),
                "gse_pf_resample: pointers must be 16-byte aligned");
    GSE_REQUIRE(N < (int64_t)1 << 31, "gse_pf_resample: N must be below 2^31");
    ResampleArgs ra;
    for (int b = 0; b < B; ++b) {
        GSE_REQUIRE(r[b] >= 0.0 && r[b] < 1.0, "gse_pf_resample: r[%d] = %g is outside [0, 1)", b, r[b]);
        ra.r[b] = r[b];
    }
    const int T = (int)tiles_of(N);
    Workspace ws = carve_workspace(workspace, B, N);
    hipStream_t st = (hipStream_t)stream;
    if (!(flags & GSE_RESAMPLE_TILES_VALID)) {
        launch_tile_sums(w, ws.tile_sums, B, N, ld, T, st);
        GSE_LAUNCH_CHECK("tile_sum_kernel");
    }
    launch_tile_scan(ws.tile_sums, ws.tile_prefix, B, T, st);
    GSE_LAUNCH_CHECK("tile_scan_kernel");
    PfPayload pl{x_in, x_out, ld};
    hipLaunchKernelGGL(resample_kernel<PfPayload>, dim3(T, B), dim3(kBlock), 0, st, pl, w, w_out, idx, ws.tile_prefix, N,
                       ld, T, ra);
    GSE_LAUNCH_CHECK("resample_kernel");
    return GSE_OK;
}

int gse_pf_moments(const float* x, const float* w, int64_t ld, int B, int64_t N, double* out, void* workspace,
                   void* stream) {
    if (int rc = check_pf_geometry("gse_pf_moments", B, N, ld)) return rc;
    GSE_REQUIRE(x && w && out && workspace, "gse_pf_moments: null argument");
    GSE_REQUIRE(aligned16(x) && aligned16(w) && aligned16(workspace), "gse_pf_moments: pointers must be 16-byte aligned");
    Workspace ws = carve_workspace(workspace, B, N);
    const int nblk = (int)std::min<int64_t>(kMomentBlocks, tiles_of(N));
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pf_moments_kernel, dim3(nblk, B), dim3(kBlock), 0, st, x, w, ws.partials, N, ld);
    GSE_LAUNCH_CHECK("pf_moments_kernel");
    launch_moments_finish(ws.partials, nblk, 21, out, 26, B, x, GSE_NX * ld, ld, 21, st);
    GSE_LAUNCH_CHECK("moments_finish_kernel");
    return GSE_OK;
}

}  // extern "C"
