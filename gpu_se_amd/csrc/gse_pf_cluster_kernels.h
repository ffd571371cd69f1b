// gse_pf_cluster_kernels.h -- the particle filter's cycle kernel for MID-SIZE ensembles (round 4).
//
// Below one generation of workgroups the single-launch cycle (pf_step_kernel) is a chain of dependent latencies, not
// bandwidth: at N = 2^16 it runs 64 workgroups on 256 CUs, one wavefront per SIMD, and every lane walks gather -> noise draw
// -> model step -> likelihood for its FOUR particles one after the other: 11.4 us at 2^16, 12.1 at 2^18 (profiles/r03_sweep.txt).
// Here a tile of 1024 particles belongs to a CLUSTER of four workgroups of 256 threads, one particle per thread: workgroup
// (tile, p) owns the outputs 256 p .. 256 p + 255 of the tile -- exactly the four particles thread t of pf_step_kernel owns in
// its strided mapping, t + 256 p, dealt to four threads on four CUs.  Same tile, same arithmetic, same association:
//
//   * Philox: thread t of quarter p draws what thread t of the tile's workgroup draws for its particle p -- group = tile * 256
//     + t, normals 5 p .. 5 p + 4 of the group's twenty and word p of block 5 (mixture_draw5x4) -- computing only the two of
//     the five blocks those normals come from;
//   * resample: every workgroup rebuilds the tile prefix and expands the ancestor tiles of ITS 256 outputs by the inverse map
//     (expand_quad with 256 slots: the keys are re-formed bit for bit as cumsum_kernel forms them), the look-up path for wide
//     ancestor ranges included;
//   * the scans a launch leaves for the next one: the four wavefront totals block_exclusive_scan adds up inside
//     pf_step_kernel are here four WORKGROUP totals, which no workgroup of the launch can combine -- so the slot is kept in
//     QUARTER FORMAT: per tile the four quarter sums q0 .. q3 (what the wavefronts' inclusive scans end on) and per quad its
//     exclusive prefix inside the quarter; the consumer forms  tile sum = ((q0 + q1) + q2) + q3  and  excl = base_v + in-quarter
//     prefix,  base_v = 0, q0, q0 + q1, (q0 + q1) + q2 -- the expressions of block_exclusive_scan, so the same bits.
//
// Identical particles, weights and ancestors to pf_step_kernel (tests/test_gpu_pf.py::test_cluster_cycle_...): the policy
// switch changes the time, not the result.  N <= 256 tiles, Philox noise, 2-component measurement mixture, built-in model.
#pragma once

#include "gse_pf_kernels.h"

namespace gse {

constexpr int kQuarterSlots = kTile / 4;           // outputs per workgroup of a cluster
constexpr int kClusterMaxT = 256;                  // tiles (the policy's range: N <= 2^18)

// block_exclusive_scan's combination of its four wavefront totals, from a tile's four quarter sums
__device__ __forceinline__ double quarters_total(const double* q) { return ((q[0] + q[1]) + q[2]) + q[3]; }
__device__ __forceinline__ double quarters_base(const double* q, int v) {
    double base = 0.0;
    if (v > 0) base = q[0];
    if (v > 1) base += q[1];
    if (v > 2) base += q[2];
    return base;
}

// The noise of particle kP of a group: normals 5 kP .. 5 kP + 4 of the group's twenty and word kP of block 5 -- the values
// mixture_draw5x4 hands its particle kP -- from two of the five blocks of normals (unused Box-Muller halves fold away).
template <int kP>
__device__ __forceinline__ void mixture_draw5_of_group(const DrawParams& dp, const RngKey& key, uint32_t group, uint32_t fb,
                                                       float (&out)[5]) {
    constexpr int first = 5 * kP, b0 = first / 4, b1 = (first + 4) / 4;
    uint32_t cw[4], wa[4], wb[4];
    rng_block_of_group(key, group, fb, 5u, cw);
    rng_block_of_group(key, group, fb, (uint32_t)b0, wa);
    rng_block_of_group(key, group, fb, (uint32_t)b1, wb);
    float na[4], nb[4];
    box_muller(wa[0], wa[1], na[0], na[1]);
    box_muller(wa[2], wa[3], na[2], na[3]);
    box_muller(wb[0], wb[1], nb[0], nb[1]);
    box_muller(wb[2], wb[3], nb[2], nb[3]);
    float n[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int at = first + i;
        n[i] = (at / 4 == b0) ? na[at % 4] : nb[at % 4];
    }
    if (mixture_is_diagonal(dp)) mixture_transform5<1>(dp, cw[kP], n, out);
    else mixture_transform5<0>(dp, cw[kP], n, out);
}

// One output's ancestor by look-up (search_ancestors for one output, quarter format): the first key with x[k] >= tau.
__device__ __forceinline__ int32_t cluster_lookup(const OffspringCount& F, const double* pfx_s, const double* qs_s,
                                                  const float* __restrict__ wrow, const double* __restrict__ qx, int32_t N,
                                                  int32_t ld, int tl0, int tl1, int32_t j) {
    const double tau = smallest_with_quotient_ge(F.u_of(j), F.W);
    int lo = tl0, hi = tl1;                        // first tile in [tl0, tl1] whose last key (Pfx[t + 1]) reaches tau
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (pfx_s[mid + 1] < tau) lo = mid + 1; else hi = mid;
    }
    const int tq = lo;
    const double P0 = pfx_s[tq], P1 = pfx_s[tq + 1];
    double qb[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) qb[v] = quarters_base(qs_s + 4 * tq, v);
    const double* qrow = qx + (int64_t)tq * kBlock;
    auto excl_of = [&](int u) { return qb[u >> 6] + qrow[u]; };
    // the quad: count the quads whose tile-level prefix (= the previous quad's last key up to rounding) lies below tau --
    // two 16-ary rounds of independent loads (quad_of)
    int blk = 0;
#pragma unroll
    for (int m = 1; m < 16; ++m) blk += (fmin(P0 + excl_of(16 * m), P1) < tau) ? 1 : 0;
    int uq = 16 * blk;
#pragma unroll
    for (int m = 1; m < 16; ++m) uq += (fmin(P0 + excl_of(16 * blk + m), P1) < tau) ? 1 : 0;
    // the keys of that quad and its two neighbours, re-formed exactly as everywhere else (window_ancestor)
    const int u0 = max(uq - 1, 0);
    const int32_t tile_first = tq * kTile;
    const int32_t tile_last = min(tile_first + kTile, N) - 1;
    int32_t below = tile_first + u0 * kPerThread - 1;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int u = min(u0 + q, kBlock - 1);
        const int32_t k0 = tile_first + u * kPerThread;
        const float4 wv = *reinterpret_cast<const float4*>(wrow + min(k0, ld - kPerThread));
        const double ex = excl_of(u);
        double sc[kPerThread];
        sc[0] = (k0 + 0 < N) ? (double)wv.x : 0.0;
        sc[1] = sc[0] + ((k0 + 1 < N) ? (double)wv.y : 0.0);
        sc[2] = sc[1] + ((k0 + 2 < N) ? (double)wv.z : 0.0);
        sc[3] = sc[2] + ((k0 + 3 < N) ? (double)wv.w : 0.0);
#pragma unroll
        for (int i = 0; i < kPerThread; ++i) {
            const double x = (k0 + i >= tile_last) ? P1 : fmin(P0 + (ex + sc[i]), P1);
            if (k0 + i <= tile_last && !(x >= tau)) below = max(below, k0 + i);
        }
    }
    return min(max(below + 1, tl0 * kTile), N - 1);
}

// The gather stage of a cluster workgroup as a function of its own -- the ancestor of output j0 + tid, for the 256 outputs
// j0 .. j0 + 255 of one quarter of a tile -- because the GS-UKF's one-lane-per-Gaussian kernel runs the same stage for its 256
// Gaussians (gse_gsukf_kernels.h, round 4).  qv: this thread's tile's four quarter sums (thread t < T holds tile t's; requested
// by the caller at kernel entry so that something else covers their latency); wrow: the weights the sums were taken from
// (rows of ld floats, 16-byte aligned); qx: [T][256] the quads' exclusive prefixes inside their quarters.  Must be called by
// all 256 threads; W: the total weight (not positive and finite: every output's ancestor is 0, particle.py:98).
struct ClusterGatherLds {
    uint32_t mark[kQuarterSlots];
    int32_t act_list[kClusterMaxT];
    int32_t act_n;
    double red[4];
    double pfx_s[kClusterMaxT + 1];
    double qs_s[4 * kClusterMaxT];
};
__device__ __forceinline__ int32_t cluster_gather(ClusterGatherLds& L, const double (&qv)[4], const float* __restrict__ wrow,
                                                  const double* __restrict__ qx, int64_t N, int64_t ld, int T, double r, int64_t j0,
                                                  int tid, double& W) {
    const int lane = tid & 63, wv = tid >> 6;
    const uint32_t jme = (uint32_t)(j0 + tid);
    if (tid < T) {
#pragma unroll
        for (int i = 0; i < 4; ++i) L.qs_s[4 * tid + i] = qv[i];
    }
    L.mark[tid] = 0u;
    if (tid == 0) L.act_n = 0;
    // tile prefix: one tile sum per thread (the chunks of pf_step_kernel's prefix are one tile long up to 256 tiles)
    const double chunk = (tid < T) ? quarters_total(qv) : 0.0;
    double total;
    const double run = block_exclusive_scan(chunk, L.red, total);
    if (tid < T) L.pfx_s[tid] = run;
    if (tid == 0) L.pfx_s[T] = total;
    __syncthreads();
    W = L.pfx_s[T];
    int32_t anc = 0;
    const int32_t j1 = (int32_t)min(j0 + kQuarterSlots, N);
    if ((W > 0.0) && (W < INFINITY) && j0 < N) {                       // (workgroup-uniform)
        OffspringCount F;
        F.init(N, r, W);
        const int32_t jf = (int32_t)j0;
        auto slot_of = [&](double x) -> int32_t { return min(max(F(x) - jf, 0), (int32_t)kQuarterSlots); };
        const double p_lo = (F.u_of(jf) * W) * (1.0 - 0x1p-50);
        const double p_hi = (F.u_of(j1 - 1) * W) * (1.0 + 0x1p-50);
        int tl0, tl1;
        wave_search_ge_pair([&](int q) { return L.pfx_s[q + 1]; }, T, p_lo, p_hi, lane, tl0, tl1);
        // the thresholds' range with a margin far above the rounding of a key (2^-52 W) and far below the distance of two
        // thresholds (W / N): what lies outside it marks nothing
        const double g_lo = p_lo - 0x1p-40 * W, g_hi = p_hi + 0x1p-40 * W;
        // ... and the first key that can be anybody's ancestor here: the first key of the first quarter of tile tl0 whose end
        // reaches g_lo (tl0 is the first TILE whose end does).  An output whose slot nobody marked descends from it: the key
        // that would have marked the slot is the last one below the first threshold, in a quarter nobody counted.
        int32_t floor_key = tl0 * kTile;
        {
            const double P0 = L.pfx_s[tl0];
#pragma unroll
            for (int v = 0; v < 3; ++v)
                if (P0 + quarters_base(L.qs_s + 4 * tl0, v + 1) < g_lo) floor_key = tl0 * kTile + (v + 1) * (kTile / 4);
        }
        const bool direct = tl1 - tl0 < 2;
        int n_act = tl1 - tl0 + 1;
        if (!direct) {
            // census of the range: a tile whose two ends fall into one slot is spoken for by its last key (pf_step_kernel)
            for (int q0 = tl0; q0 <= tl1; q0 += kBlock) {
                const int q = q0 + tid;
                const bool have = q <= tl1;
                const int32_t e0 = have ? slot_of(L.pfx_s[q]) : 0, e1 = have ? slot_of(L.pfx_s[q + 1]) : -1;
                const uint32_t next_e1 = wave_shift_down1((uint32_t)(e1 + 2));
                if (!have) continue;
                if (e0 == e1) {
                    if (e1 < kQuarterSlots && next_e1 != (uint32_t)(e1 + 2))
                        atomicMax(&L.mark[e1], (uint32_t)(min((int64_t)(q + 1) * kTile, N) - 1) + 1u);
                } else {
                    L.act_list[atomicAdd(&L.act_n, 1)] = q;
                }
            }
            __syncthreads();
            n_act = L.act_n;
        }
        const bool wide = !direct && n_act > GSE_SEARCH_FROM;          // workgroup-uniform
        if (wide) {
            anc = cluster_lookup(F, L.pfx_s, L.qs_s, wrow, qx, (int32_t)N, (int32_t)ld, tl0, tl1, min((int32_t)jme, j1 - 1));
        } else {
            // (loads unconditional, indices clamped, the next tile's in flight while this one is expanded: pf_step_kernel)
            auto tile_at = [&](int i) { return direct ? tl0 + i : L.act_list[i]; };
            auto weights_of = [&](int tq) {
                const int64_t k0 = min((int64_t)tq * kTile + (int64_t)tid * kPerThread, ld - kPerThread);
                return *reinterpret_cast<const float4*>(wrow + k0);
            };
            const int last = max(n_act - 1, 0);
            int t0 = tile_at(0);
            float4 c0 = weights_of(t0);
            double x0 = qx[t0 * kBlock + tid];
            for (int i = 0; i < n_act; ++i) {
                const int t1 = tile_at(min(i + 1, last));
                const float4 c1 = weights_of(t1);
                const double x1 = qx[t1 * kBlock + tid];
                // A wavefront holds one QUARTER of the tile's keys.  The 256 outputs of this workgroup descend from a stretch of
                // about 256 keys -- one or two quarters of the one or two listed tiles -- so most wavefronts hold keys that lie all
                // below the first threshold or all above the last: they mark nothing and need not count (a key marks the slot of
                // the first output ABOVE it; see `floor` below for the last key below the first threshold).  The counting is
                // most of what this stage issues: skipped per wavefront, the stage costs the one-lane-per-Gaussian kernel of the
                // GS-UKF, bound by its instruction count, 2 us instead of 5 at 2^18 Gaussians.
                const double qb = quarters_base(L.qs_s + 4 * t0, wv);
                const double q_lo = L.pfx_s[t0] + qb, q_hi = q_lo + L.qs_s[4 * t0 + wv];
                if (q_hi >= g_lo && q_lo <= g_hi)                         // (wavefront-uniform)
                    expand_quad<true, kQuarterSlots>(F, L.pfx_s, L.mark, N, jf, t0, tid, c0, qb + x0);
                t0 = t1; c0 = c1; x0 = x1;
            }
            __syncthreads();
            // inclusive max-scan of the 256 marks, one slot per thread
            const uint32_t m0 = L.mark[tid];
            const uint32_t inc = wave_inclusive_max(m0);
            uint32_t* wmax = reinterpret_cast<uint32_t*>(L.red);          // (free since the tile prefix)
            if (lane == 63) wmax[wv] = inc;
            __syncthreads();
            uint32_t best = inc;
            if (wv > 0) best = max(best, wmax[0]);
            if (wv > 1) best = max(best, wmax[1]);
            if (wv > 2) best = max(best, wmax[2]);
            anc = (int32_t)min(max(best, (uint32_t)floor_key), (uint32_t)N - 1u);
        }
    }
    return anc;
}

// grid (4 T, B): workgroup (tile, quarter) = (blockIdx.x >> 2, blockIdx.x & 3).  qsum_*: [B][T][4], qx_*: [B][T][256].
template <bool kGather, bool kPredict, bool kUpdate, bool kDevIO>
__global__ __launch_bounds__(kBlock) void pf_cluster_kernel(
    const float* __restrict__ x_in, float* __restrict__ x_out, const float* __restrict__ w_in, float* __restrict__ w_out,
    int32_t* __restrict__ idx_out, const double* __restrict__ qsum_in, const double* __restrict__ qx_in,
    double* __restrict__ qsum_out, double* __restrict__ qx_out, double* __restrict__ island, int64_t N, int64_t ld, int T,
    const StepArgs a) {
    constexpr int kNd = 2;
    __shared__ DrawParams dp;
    __shared__ ClusterGatherLds gl;
    __shared__ __attribute__((aligned(16))) double wq[kQuarterSlots];      // the new weights before their rounding (fp64)
    __shared__ int kexp_s;

    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int tile_id = (int)(blockIdx.x >> 2), quarter = (int)(blockIdx.x & 3u);
    const int64_t tile0 = (int64_t)tile_id * kTile;
    const int64_t j0 = tile0 + (int64_t)quarter * kQuarterSlots;            // first output of this workgroup
    const uint32_t jme = (uint32_t)(j0 + tid), Nu = (uint32_t)N;
    const bool live = jme < Nu;
    const float* xin_b = x_in + (int64_t)b * GSE_NX * ld;
    const float w_uniform = (float)(1.0 / (double)N);

    if (kPredict) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&a.draw);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&dp);
        for (int i = tid; i < (int)(sizeof(DrawParams) / 4); i += kBlock) dst[i] = src[i];
    }
    // the quarter sums of every tile (T <= 256: at most four per thread), requested before the draw, which covers their latency
    double qv[4] = {0.0, 0.0, 0.0, 0.0};
    if (kGather) {
        const double* qs = qsum_in + (int64_t)b * 4 * T;
#pragma unroll
        for (int i = 0; i < 4; ++i) qv[i] = (tid < T) ? qs[4 * tid + i] : 0.0;
    }
    // GSE_OP_WEIGHT_SCALE: this launch's binary exponent (the same 64 samples in every workgroup of the launch): requested
    // here, with the quarter sums, turned into k after the gather stage
    ScaleSample ssm;
    if (kUpdate && a.wscale && wv == 0 && (kGather || !a.k_exact))
        scale_samples_load(ssm, xin_b, w_in ? w_in + (int64_t)b * ld : nullptr, N, ld, lane);
    if (kPredict) __syncthreads();                                       // dp staged
    float e[GSE_NX] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (kPredict) {
        const uint32_t group = (uint32_t)tile_id * kBlock + (uint32_t)tid;
        switch (quarter) {                                               // (workgroup-uniform)
            case 0: mixture_draw5_of_group<0>(dp, a.rng, group, (uint32_t)b, e); break;
            case 1: mixture_draw5_of_group<1>(dp, a.rng, group, (uint32_t)b, e); break;
            case 2: mixture_draw5_of_group<2>(dp, a.rng, group, (uint32_t)b, e); break;
            default: mixture_draw5_of_group<3>(dp, a.rng, group, (uint32_t)b, e); break;
        }
    }

    // ---- gather: the ancestor of this thread's output -----------------------------------------------------------------
    int32_t anc = (int32_t)min(jme, Nu - 1u);
    if (kGather) {
        double W;
        anc = cluster_gather(gl, qv, w_in + (int64_t)b * ld, qx_in + (int64_t)b * T * kBlock, N, ld, T, a.ra.r[b], j0, tid, W);
        if (a.wstatus && blockIdx.x == 0 && tid == 0 && !(W < INFINITY)) a.wstatus[b] = GSE_WSTATUS_SUM_NOT_FINITE;
        if (island && blockIdx.x == 0 && tid == 0) {                     // shard bookkeeping (see gse_pf_resample)
            const double ws_b = a.wscale ? a.wscale[b] : 0.0;
            island[b] += log(W) + ws_b * 0.69314718055994530942;
            island[gridDim.y + b] = ldexp(W, (int)ws_b);
        }
    }

    // GSE_OP_WEIGHT_SCALE: wavefront 0 turns its samples into k; nobody waits for it here -- the power of two is applied behind
    // the barrier of the quarter scan below (pf_step_kernel does the same behind its weight exchange)
    int kexp = 0;
    if (kUpdate && a.wscale && wv == 0) {
        const int k = scale_exponent_of<kGather, kPredict, kNd, kDevIO>(ssm, w_in != nullptr, false, a, N, b);
        if (lane == 0) kexp_s = k;
    }

    // ---- the particle: state of the ancestor, model step + noise, likelihood
    float wnew = w_uniform;
    double wd = 0.0;                                                     // the new weight before its rounding to fp32
    if (live) {
        float s[GSE_NX];
#pragma unroll
        for (int k = 0; k < GSE_NX; ++k) s[k] = xin_b[(int64_t)k * ld + anc];
        if (kUpdate && !kGather) wnew = w_in[(int64_t)b * ld + jme];
        double fg = a.io.f[b].u0, fm = a.io.f[b].u1, z0 = a.io.f[b].z0, z1 = a.io.f[b].z1;
        if constexpr (kDevIO) {
            if (a.u_dev) { fg = a.u_dev[2 * b]; fm = a.u_dev[2 * b + 1]; }
            if (a.z_dev) { z0 = a.z_dev[2 * b]; z1 = a.z_dev[2 * b + 1]; }
        }
        if (kPredict) {
            model_step(s, fg, fm, a.dt);
#pragma unroll
            for (int k = 0; k < GSE_NX; ++k) s[k] += e[k];
        }
        if (kGather || kPredict) {
            float* xb = x_out + (int64_t)b * GSE_NX * ld;
#pragma unroll
            for (int k = 0; k < GSE_NX; ++k) store_out<1>(xb + (int64_t)k * ld + jme, s[k]);
        }
        if (kUpdate) {
            float obs[kObsKept], y0, y1;
            model_keep(s, obs);
            model_output_kept(obs, fg, fm, y0, y1);
            wd = (double)wnew * mixture_pdf2<kNd>(a.pdf, z0 - (double)y0, z1 - (double)y1);
        } else if (kGather) {
            if (a.wscale && blockIdx.x == 0 && tid == 0) a.wscale[b] = 0.0;
            store_out<2>(w_out + (int64_t)b * ld + jme, wnew);
        }
        if (kGather && idx_out) idx_out[(int64_t)b * ld + jme] = anc;
    }

    // ---- the new weights (one rounding to fp32, under the launch's power of two) and what the next launch resamples from:
    //      the quarter's sum and the quads' exclusive prefixes inside the quarter -- the wavefront scan of pf_step_kernel's
    //      update stage (64 lanes x 4 consecutive weights), by wavefront 0
    if (kUpdate) {
        wq[tid] = wd;
        __syncthreads();
        if (a.wscale) kexp = kexp_s;                                     // (written before wavefront 0 came to this barrier)
        if (live) store_out<2>(w_out + (int64_t)b * ld + jme, (float)ldexp(wd, -kexp));
        if (a.wscale && blockIdx.x == 0 && tid == 0) a.wscale[b] = (kGather ? 0.0 : a.wscale[b]) + (double)kexp;
        if (wv == 0) {
            const double2 t01 = *reinterpret_cast<const double2*>(&wq[lane * kPerThread]);
            const double2 t23 = *reinterpret_cast<const double2*>(&wq[lane * kPerThread + 2]);
            const float w0 = (float)ldexp(t01.x, -kexp), w1 = (float)ldexp(t01.y, -kexp);
            const float w2 = (float)ldexp(t23.x, -kexp), w3 = (float)ldexp(t23.y, -kexp);
            const int64_t i0 = j0 + (int64_t)lane * kPerThread;
            double sc = (i0 + 0 < N) ? (double)w0 : 0.0;
            sc = sc + ((i0 + 1 < N) ? (double)w1 : 0.0);
            sc = sc + ((i0 + 2 < N) ? (double)w2 : 0.0);
            sc = sc + ((i0 + 3 < N) ? (double)w3 : 0.0);
            const double inc = wave_inclusive_scan(sc, lane);
            const double ex = wave_shift_up1(inc);
            double* qo = qx_out + ((int64_t)b * T + tile_id) * kBlock + quarter * 64 + lane;
            __hip_atomic_store(qo, ex, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane == 63) qsum_out[((int64_t)b * T + tile_id) * 4 + quarter] = inc;
        }
    }
}

// Quarter format -> tile format (what pf_step_kernel's scan modes, tile_scan_kernel and the eager resample take): per tile the
// sum ((q0 + q1) + q2) + q3 and per quad the tile-level exclusive prefix base_v + in-quarter prefix -- block_exclusive_scan's
// own expressions.  One workgroup per tile.
__global__ __launch_bounds__(kBlock) void quarters_to_tiles_kernel(const double* __restrict__ qsum, const double* __restrict__ qx,
                                                                  double* __restrict__ tile_sums, double* __restrict__ excl,
                                                                  int T) {
    const int b = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
    const double* q = qsum + ((int64_t)b * T + t) * 4;
    const double q4[4] = {q[0], q[1], q[2], q[3]};
    excl[((int64_t)b * T + t) * kBlock + tid] = quarters_base(q4, tid >> 6) + qx[((int64_t)b * T + t) * kBlock + tid];
    if (tid == 0) tile_sums[(int64_t)b * T + t] = quarters_total(q4);
}

}  // namespace gse
