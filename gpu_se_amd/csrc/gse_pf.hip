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
// Two kernels, both perfectly balanced whatever the weights look like (a degenerate posterior -- a handful of
// particles carrying all the weight -- is the normal state of this filter, see tests/test_oracle_golden.py):
//
//   cumsum_kernel   one workgroup per tile of 1024 ANCESTORS: tile prefix (from the tile sums the update kernel
//                   left behind) + fp64 scan of the tile's weights -> normalised cumulative weights c[k], fp64,
//                   monotone, c[N-1] = 1.                                     r 4 B, w 8 B per particle
//   gather_kernel   one workgroup per tile of 1024 OUTPUTS: two 64-ary wavefront searches bracket the tile's
//                   ancestor span in c[], the span is staged in LDS, every output finds its ancestor
//                   (first k with c[k] >= (j + r)/N, the reference's comparison verbatim) by binary search in
//                   LDS and copies it; stores are contiguous in j.           r 8 + 20 B, w 24 B per particle
//
// Tile prefix computed inside the cumsum kernel (T <= 4096 tiles): every workgroup scans the T tile sums itself
// -- identical code on identical data, hence identical bits in every workgroup -- and picks out
//   Pfx(t), Pfx(t+1), Pfx(T) = total,   Pfx(q) = excl[q / m] + s[(q/m)*m] + ... + s[q-1]
// (thread j owns chunk [j*m, (j+1)*m); excl = workgroup exclusive scan of the chunk sums).  Larger T: a scan
// kernel of its own writes the prefix once.
constexpr int kInlinePrefixMaxT = 4096;

__device__ __forceinline__ void inline_tile_prefix(const double* __restrict__ s, int T, int t, double* red,
                                                   double* out3) {
    const int m = (T + kBlock - 1) / kBlock;
    const int lo = threadIdx.x * m;
    const int hi = min(lo + m, T);
    double chunk = 0.0;
    for (int q = lo; q < hi; ++q) chunk += s[q];
    double total;
    const double excl = block_exclusive_scan(chunk, red, total);
    if (t >= lo && t < hi) {
        double run = excl;
        for (int q = lo; q < t; ++q) run += s[q];
        out3[0] = run;
    }
    if (t + 1 < T) {
        if (t + 1 >= lo && t + 1 < hi) {
            double run = excl;
            for (int q = lo; q <= t; ++q) run += s[q];
            out3[1] = run;
        }
    } else if (threadIdx.x == 0) {
        out3[1] = total;
    }
    if (threadIdx.x == 0) out3[2] = total;
    __syncthreads();
}

// tile_src: kInlinePrefix ? tile sums [B][T] : exclusive tile prefix [B][T + 1] (from tile_scan_kernel).
// c: [B][N] fp64.  totals: [B] fp64 (sum of the weights).  island: optional [2][B] shard bookkeeping.
template <bool kInlinePrefix>
__global__ __launch_bounds__(kBlock) void cumsum_kernel(const float* __restrict__ w, double* __restrict__ c,
                                                        const double* __restrict__ tile_src,
                                                        double* __restrict__ totals, double* __restrict__ island,
                                                        int64_t N, int64_t ldw, int T) {
    __shared__ double red[4];
    __shared__ double pfx[3];
    const int b = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
    double P0, P1, W;
    if (kInlinePrefix) {
        inline_tile_prefix(tile_src + (int64_t)b * T, T, t, red, pfx);
        P0 = pfx[0]; P1 = pfx[1]; W = pfx[2];
    } else {
        const double* pf = tile_src + (int64_t)b * (T + 1);
        P0 = pf[t]; P1 = pf[t + 1]; W = pf[T];
    }
    if (t == 0 && tid == 0) {
        totals[b] = W;
        if (island) {                          // shard bookkeeping: accumulated log-weight and this cycle's sum
            island[b] += log(W);
            island[gridDim.y + b] = W;
        }
    }
    const int64_t i0 = (int64_t)t * kTile + (int64_t)tid * kPerThread;
    const float4 wv = load_weights4(w + (int64_t)b * ldw, i0, N, ldw);
    double s[kPerThread];
    s[0] = (i0 + 0 < N) ? (double)wv.x : 0.0;
    s[1] = s[0] + ((i0 + 1 < N) ? (double)wv.y : 0.0);
    s[2] = s[1] + ((i0 + 2 < N) ? (double)wv.z : 0.0);
    s[3] = s[2] + ((i0 + 3 < N) ? (double)wv.w : 0.0);
    double tile_total;
    const double excl = block_exclusive_scan(s[3], red, tile_total);
    // The tile's last entry is pinned to prefix[t+1]/W and everything before it clamped to that value, so the
    // array is monotone across tiles although prefix and in-tile sums associate differently; c[N-1] = W/W = 1.
    const double top = P1 / W;
    const int64_t last = min((int64_t)(t + 1) * kTile, N) - 1;
    double* cb = c + (int64_t)b * N;
    double v[kPerThread];
#pragma unroll
    for (int p = 0; p < kPerThread; ++p) v[p] = (i0 + p == last) ? top : fmin((P0 + (excl + s[p])) / W, top);
    if (i0 + 3 < N && (N & 1) == 0) {          // 16-byte aligned pairs
        *reinterpret_cast<double2*>(cb + i0) = make_double2(v[0], v[1]);
        *reinterpret_cast<double2*>(cb + i0 + 2) = make_double2(v[2], v[3]);
    } else {
#pragma unroll
        for (int p = 0; p < kPerThread; ++p)
            if (i0 + p < N) cb[i0 + p] = v[p];
    }
}

// First k in [0, N) with c[k] >= u, found by one wavefront: every round the 64 lanes probe 64 evenly spaced
// elements of the remaining range and a ballot picks the sub-range (4 rounds of one L2 access for N <= 2^24,
// instead of 24 dependent ones).  Returns N - 1 if no element qualifies (cannot happen for u <= c[N-1] = 1).
__device__ __forceinline__ int64_t wave_search_ge(const double* __restrict__ c, int64_t N, double u, int lane) {
    int64_t lo = 0, hi = N;                                   // answer in [lo, hi)
    while (true) {
        const int64_t n = hi - lo;
        const int64_t step = (n + 63) >> 6;
        const int64_t probe = min(lo + (int64_t)(lane + 1) * step, hi) - 1;
        const bool ge = c[probe] >= u;
        const unsigned long long mask = __ballot(ge);
        if (mask == 0ull) return N - 1;
        const int first = __ffsll((long long)mask) - 1;
        const int64_t nlo = lo + (int64_t)first * step;
        const int64_t nhi = min(lo + (int64_t)(first + 1) * step, hi);
        if (step == 1) return nlo;
        lo = nlo;
        hi = nhi;
    }
}

constexpr int kSpanMax = 2048;            // ancestors staged in LDS per output tile (16 KiB)

template <class Payload>
__global__ __launch_bounds__(kBlock) void gather_kernel(const Payload pl, const double* __restrict__ c,
                                                        const double* __restrict__ totals, float* __restrict__ w_out,
                                                        int32_t* __restrict__ idx_out, int64_t N, int64_t ldw,
                                                        const ResampleArgs ra) {
    __shared__ double seg[kSpanMax];
    __shared__ int64_t span[2];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t j0 = (int64_t)blockIdx.x * kTile;
    const int64_t j1 = min(j0 + kTile, N);
    const double r = ra.r[b];
    const double Nd = (double)N;
    const double W = totals[b];
    const double* cb = c + (int64_t)b * N;
    const float w_uniform = (float)(1.0 / Nd);
    float* wo = w_out + (int64_t)b * ldw;
    int32_t* io = idx_out ? idx_out + (int64_t)b * ldw : nullptr;

    // Weights that do not sum to a positive finite number: the reference's normalised cumsum is all-NaN and its
    // sweep never advances (filter/particle.py:98), so every offspring copies particle 0.
    if (!(W > 0.0) || !(W < INFINITY)) {
        for (int64_t j = j0 + tid; j < j1; j += kBlock) {
            pl.copy(b, 0, j);
            wo[j] = w_uniform;
            if (io) io[j] = 0;
        }
        return;
    }

    // 1. bracket the ancestors of this output tile: wavefront 0 searches for the first output, wavefront 1 for the last
    if (wv < 2) {
        const int64_t j = (wv == 0) ? j0 : j1 - 1;
        const int64_t k = wave_search_ge(cb, N, ((double)j + r) / Nd, lane);
        if (lane == 0) span[wv] = k;
    }
    __syncthreads();
    const int64_t klo = span[0], khi = span[1];
    const int64_t len = khi - klo + 1;

    if (len <= kSpanMax) {
        // 2a. stage c[klo..khi] in LDS, binary search there (<= 11 steps)
        for (int q = tid; q < (int)len; q += kBlock) seg[q] = cb[klo + q];
        __syncthreads();
        for (int64_t j = j0 + tid; j < j1; j += kBlock) {
            const double u = ((double)j + r) / Nd;
            int lo = 0, hi = (int)len - 1;
#pragma unroll
            for (int it = 0; it < 11; ++it) {             // 2^11 = kSpanMax
                const int mid = (lo + hi) >> 1;
                const bool open = lo < hi;
                const bool less = open && (seg[mid] < u);
                lo = less ? mid + 1 : lo;
                hi = (open && !less) ? mid : hi;
            }
            const int64_t anc = klo + lo;
            pl.copy(b, anc, j);
            wo[j] = w_uniform;
            if (io) io[j] = (int32_t)anc;
        }
    } else {
        // 2b. the tile's ancestors are spread over a long stretch of (nearly) weightless particles: search c[] itself
        for (int64_t j = j0 + tid; j < j1; j += kBlock) {
            const double u = ((double)j + r) / Nd;
            int64_t lo = klo, hi = khi;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (cb[mid] < u) lo = mid + 1; else hi = mid;
            }
            pl.copy(b, lo, j);
            wo[j] = w_uniform;
            if (io) io[j] = (int32_t)lo;
        }
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
// Normalised cumulative weights of B filters into ws.cumsum (+ totals); shared by PF, GS-UKF and gse_cumsum_normalised.
int launch_cumsum(const float* w, int64_t ldw, int B, int64_t N, uint32_t flags, double* island, const Workspace& ws,
                  double* c_out, hipStream_t st) {
    const int T = (int)tiles_of(N);
    if (!(flags & GSE_RESAMPLE_TILES_VALID)) {
        hipLaunchKernelGGL(tile_sum_kernel, dim3(T, B), dim3(kBlock), 0, st, w, ws.tile_sums, N, ldw, T);
        GSE_LAUNCH_CHECK("tile_sum_kernel");
    }
    if (T <= kInlinePrefixMaxT && !(flags & GSE_RESAMPLE_SEPARATE_SCAN)) {
        hipLaunchKernelGGL(cumsum_kernel<true>, dim3(T, B), dim3(kBlock), 0, st, w, c_out, ws.tile_sums, ws.totals, island,
                           N, ldw, T);
    } else {
        hipLaunchKernelGGL(tile_scan_kernel, dim3(B), dim3(kBlock), 0, st, ws.tile_sums, ws.tile_prefix, T);
        GSE_LAUNCH_CHECK("tile_scan_kernel");
        hipLaunchKernelGGL(cumsum_kernel<false>, dim3(T, B), dim3(kBlock), 0, st, w, c_out, ws.tile_prefix, ws.totals,
                           island, N, ldw, T);
    }
    GSE_LAUNCH_CHECK("cumsum_kernel");
    return GSE_OK;
}

int launch_gsukf_gather(const GsukfPayload& pl, float* w_out, int32_t* idx, int B, int64_t N, const ResampleArgs& ra,
                        const Workspace& ws, hipStream_t st) {
    hipLaunchKernelGGL(gather_kernel<GsukfPayload>, dim3((unsigned)tiles_of(N), B), dim3(kBlock), 0, st, pl, ws.cumsum,
                       ws.totals, w_out, idx, N, N, ra);
    GSE_LAUNCH_CHECK("gather_kernel<gsukf>");
    return GSE_OK;
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
                    int64_t N, const double* r, uint32_t flags, double* island, void* workspace, void* stream) {
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
    Workspace ws = carve_workspace(workspace, B, N);
    hipStream_t st = (hipStream_t)stream;
    if (int rc = launch_cumsum(w, ld, B, N, flags, island, ws, ws.cumsum, st)) return rc;
    PfPayload pl{x_in, x_out, ld};
    hipLaunchKernelGGL(gather_kernel<PfPayload>, dim3((unsigned)tiles_of(N), B), dim3(kBlock), 0, st, pl, ws.cumsum,
                       ws.totals, w_out, idx, N, ld, ra);
    GSE_LAUNCH_CHECK("gather_kernel");
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
