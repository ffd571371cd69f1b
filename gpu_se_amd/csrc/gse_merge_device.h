// gse_merge_device.h -- device functions of the posterior merge of a sharded ensemble (gpu_se_amd/distributed.py):
// shared by gse_mpc.hip (gse_merge_row / gse_merge_rows / gse_pf_merge_begin / gse_merge_end) and by the resampled-moments
// kernel of gse_pf_kernels.h, whose last workgroup runs the merge's tail itself (round 5).
#pragma once
#include "gse_host.h"

namespace gse {

struct Pivot5 { double p[GSE_NX]; };

// gse_merge_row's arithmetic (distributed.pack_merge_row, operation by operation), by one thread: moments m[26] about the shard's
// own pivot, log-weight log_a, binary exponent ws of the sums -> row[22] about the common pivot.
__device__ __forceinline__ void merge_row_thread(const double* __restrict__ m, double log_a, int ws, const Pivot5& common,
                                                 double* __restrict__ row) {
#pragma clang fp contract(off)
    const double s0 = ldexp(m[0], ws);
    double s1[GSE_NX], delta[GSE_NX];
#pragma unroll
    for (int k = 0; k < GSE_NX; ++k) {
        s1[k] = ldexp(m[1 + k], ws);
        delta[k] = m[21 + k] - common.p[k];
    }
    row[0] = log_a;
    row[1] = s0;
#pragma unroll
    for (int k = 0; k < GSE_NX; ++k) row[2 + k] = s1[k] + delta[k] * s0;
    int c = 0;
#pragma unroll
    for (int r = 0; r < GSE_NX; ++r)
#pragma unroll
        for (int q = r; q < GSE_NX; ++q, ++c)
            row[7 + c] = ((ldexp(m[6 + c], ws) + s1[r] * delta[q]) + delta[r] * s1[q]) + s0 * (delta[r] * delta[q]);
}

// gse_merge_rows' arithmetic (distributed.merge_rows / shard_shares), by one thread.  status: GSE_MERGE_STATUS_NO_WEIGHT when no
// shard is live; GSE_MERGE_STATUS_OVERFLOW when a shard's weight sum or log-weight is +inf or NaN (round 5: such a shard used
// to be dropped like one without weight -- the stored weights overflowed their power-of-two scale, which must not pass silently).
__device__ __forceinline__ void merge_rows_thread(const double* __restrict__ rows, int P, const Pivot5& common,
                                                  double* __restrict__ out, int32_t* __restrict__ status) {
#pragma clang fp contract(off)
    double top = -INFINITY;
    int32_t bad = 0;
    for (int p = 0; p < P; ++p) {
        const double la = rows[p * 22], s0 = rows[p * 22 + 1];
        const bool live = (s0 > 0.0) && (fabs(la) < INFINITY) && (s0 < INFINITY);
        if (live) top = fmax(top, la + log(s0));
        if (!(s0 < INFINITY) || !(la < INFINITY)) bad = GSE_MERGE_STATUS_OVERFLOW;       // (+inf or NaN; -inf is a dead shard)
    }
    double acc[20], total = 0.0;
#pragma unroll
    for (int k = 0; k < 20; ++k) acc[k] = 0.0;
    if (status) status[0] = ((top > -INFINITY) ? 0 : GSE_MERGE_STATUS_NO_WEIGHT) | bad;
    for (int p = 0; p < P; ++p) {
        const double la = rows[p * 22], s0 = rows[p * 22 + 1];
        const bool live = (s0 > 0.0) && (fabs(la) < INFINITY) && (s0 < INFINITY);
        if (!live) continue;
        const double a = exp((la + log(s0)) - top);
        total += a;
#pragma unroll
        for (int k = 0; k < 20; ++k) acc[k] += a * rows[p * 22 + 2 + k] / s0;
    }
    out[0] = 1.0;
#pragma unroll
    for (int k = 0; k < 20; ++k) out[1 + k] = acc[k] / total;
#pragma unroll
    for (int k = 0; k < GSE_NX; ++k) out[21 + k] = common.p[k];
}

// ------------------------------------------------------------------------------------------------
// moments_estimates_kernel's row -> [mean (5), largest |eigenvalue| of the covariance] by ONE WAVEFRONT (round 5): lane 5 r + c
// holds A[r][c]; a rotation in the (p, q) plane is two exchanges with the partner column / row (A <- J^T A J: columns first,
// then rows, as the one-thread loop does), its angle computed by every lane from three broadcast entries.  The cyclic sweeps
// stop when the off-diagonal part has vanished against the matrix (<= 1e-30 of its squared norm: three or four sweeps where the
// one-thread kernel always runs eight).  One thread needed 26 us for a row -- longer than a whole cycle of 2^20 particles.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double lane_value(double v, int src) { return __shfl(v, src, 64); }

__device__ __forceinline__ void estimates_wave(const double* __restrict__ m, double* __restrict__ out, int lane) {
    const int r = (lane < 25) ? lane / 5 : 0, c = (lane < 25) ? lane % 5 : 0;
    const double s0 = m[0];
    const double s1r = m[1 + r], s1c = m[1 + c], pr = m[21 + r], pc = m[21 + c];
    const double mean_r = s1r + pr * s0, mean_c = s1c + pc * s0;
    const double er = mean_r - pr, ec = mean_c - pc;
    const int lo = r < c ? r : c, hi = r < c ? c : r;
    const int at = 6 + lo * 5 - (lo * (lo - 1)) / 2 + (hi - lo);         // upper triangle, row-major
    const double elo = r < c ? er : ec, ehi = r < c ? ec : er, slo = r < c ? s1r : s1c, shi = r < c ? s1c : s1r;
    double a = (lane < 25) ? m[at] - slo * ehi - elo * shi + s0 * elo * ehi : 0.0;
    for (int sweep = 0; sweep < 8; ++sweep) {
        const double sq = a * a;
        const double off = wave_sum((lane < 25 && r != c) ? sq : 0.0), all = wave_sum(sq);
        if (!(off > 1e-30 * all)) break;                                  // (uniform; also ends on a zero or NaN matrix)
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int q = p + 1; q < 5; ++q) {
                const double apq = lane_value(a, p * 5 + q);
                if (apq != 0.0) {                                          // (uniform)
                    const double app = lane_value(a, p * 5 + p), aqq = lane_value(a, q * 5 + q);
                    const double theta = (aqq - app) / (2.0 * apq);
                    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                    // columns: A[k][p] <- cs A[k][p] - sn A[k][q],  A[k][q] <- sn A[k][p] + cs A[k][q]
                    double other = lane_value(a, c == p ? r * 5 + q : (c == q ? r * 5 + p : lane));
                    if (c == p) a = cs * a - sn * other;
                    else if (c == q) a = sn * other + cs * a;
                    // rows: A[p][k] <- cs A[p][k] - sn A[q][k],  A[q][k] <- sn A[p][k] + cs A[q][k]
                    other = lane_value(a, r == p ? q * 5 + c : (r == q ? p * 5 + c : lane));
                    if (r == p) a = cs * a - sn * other;
                    else if (r == q) a = sn * other + cs * a;
                    if (lane >= 25) a = 0.0;
                }
            }
    }
    double big = (lane < 25 && r == c) ? fabs(a) : 0.0;
#pragma unroll
    for (int msk = 32; msk >= 1; msk >>= 1) big = fmax(big, __shfl_xor(big, msk, 64));
    if (lane < 5) out[lane] = m[1 + lane] + m[21 + lane] * s0;
    if (lane == 5) out[5] = big;
}

// ------------------------------------------------------------------------------------------------
// The posterior merge, lane-parallel (round 5).  One thread walking through gse_merge_row + gse_merge_rows + the estimate costs
// 29 us -- every value a dependent trip through global memory, every exp / log / ldexp a serial chain; here lane c of ONE
// wavefront owns element c of the row / of the merged moments, values are handed over in registers and LDS, and each
// element is formed by the operations of merge_row_thread / merge_rows_thread in their order: the same bits.
// ------------------------------------------------------------------------------------------------
// element c of gse_merge_row's row from the shard's moments m[26] (LDS or global)
__device__ __forceinline__ double merge_row_element(const double* m, double log_a, int ws, const Pivot5& common, int c) {
#pragma clang fp contract(off)
    if (c == 0) return log_a;
    const double s0 = ldexp(m[0], ws);
    if (c == 1) return s0;
    if (c < 7) {
        const int k = c - 2;
        return ldexp(m[1 + k], ws) + (m[21 + k] - common.p[k]) * s0;
    }
    int r = 0, q = c - 7;                                   // upper triangle, row-major: (r, q) of element c - 7
    while (q >= GSE_NX - r) { q -= GSE_NX - r; ++r; }
    q += r;
    const double dr = m[21 + r] - common.p[r], dq = m[21 + q] - common.p[q];
    const double s1r = ldexp(m[1 + r], ws), s1q = ldexp(m[1 + q], ws);
    return ((ldexp(m[c - 1], ws) + s1r * dq) + dr * s1q) + s0 * (dr * dq);
}

// gse_merge_rows by one wavefront: lane p forms shard p's log-share and exp of it (64 shards at a time), lane k < 20 adds the
// shards' column k in the order p = 0, 1, ... of merge_rows_thread.  rows: global or LDS.  Returns (all lanes) the status;
// lane k < 26 returns element k of the merged moments in `mine`.
__device__ __forceinline__ int merge_rows_wave(const double* rows, int P, const Pivot5& common, int lane, double& mine) {
#pragma clang fp contract(off)
    double top = -INFINITY;
    int bad = 0;
    for (int p0 = 0; p0 < P; p0 += 64) {
        const int p = p0 + lane;
        double lw = -INFINITY;
        int b = 0;
        if (p < P) {
            const double la = rows[p * 22], s0 = rows[p * 22 + 1];
            const bool live = (s0 > 0.0) && (fabs(la) < INFINITY) && (s0 < INFINITY);
            if (live) lw = (P == 1) ? 0.0 : la + log(s0);             // (P == 1: only "is there a live shard" is asked of top)
            if (!(s0 < INFINITY) || !(la < INFINITY)) b = 1;
        }
#pragma unroll
        for (int msk = 32; msk >= 1; msk >>= 1) lw = fmax(lw, __shfl_xor(lw, msk, 64));
        top = fmax(top, lw);
        bad |= (__ballot(b) != 0ull) ? GSE_MERGE_STATUS_OVERFLOW : 0;
    }
    double acc = 0.0, total = 0.0;
    for (int p0 = 0; p0 < P; p0 += 64) {
        const int p = p0 + lane;
        double a_p = 0.0, s0_p = 1.0;
        int live_p = 0;
        if (p < P) {
            const double la = rows[p * 22], s0 = rows[p * 22 + 1];
            live_p = ((s0 > 0.0) && (fabs(la) < INFINITY) && (s0 < INFINITY)) ? 1 : 0;
            if (live_p) {
                // (one shard: (la + log(s0)) - top is 0.0 and exp of it 1.0, exactly: no need to ask)
                a_p = (P == 1) ? 1.0 : exp((la + log(s0)) - top);
                s0_p = s0;
            }
        }
        const int cnt = min(64, P - p0);
        for (int j = 0; j < cnt; ++j) {
            const int live = __shfl(live_p, j, 64);
            if (!live) continue;                            // (uniform)
            const double a = __shfl(a_p, j, 64), s0 = __shfl(s0_p, j, 64);
            total += a;
            if (lane < 20) acc += a * rows[(p0 + j) * 22 + 2 + lane] / s0;
        }
    }
    const int status = ((top > -INFINITY) ? 0 : GSE_MERGE_STATUS_NO_WEIGHT) | bad;
    const double col = acc / total;                         // (lanes >= 20: 0 / total, unused)
    // element k of the 26: [0] = 1, [1 + k] = column k, [21 + k] = pivot
    const double below = __shfl(col, lane > 0 ? lane - 1 : 0, 64);
    mine = (lane == 0) ? 1.0 : (lane <= 20 ? below : (lane < 26 ? common.p[lane - 21] : 0.0));
    return status;
}

// What the tail of gse_pf_merge_begin needs besides the finished moments (kernel argument, by value).
struct MergeTail {
    int flags;                       // GSE_MERGE_*; bit 30 (kMergeTailOn): run the tail at all
    int P, rank;
    const double *island, *wscale;
    Pivot5 common;
    double* rows;
    double *moments26, *merged26, *estimate6;
    int32_t* status;
};
constexpr int kMergeTailOn = 1 << 30;

#ifndef __HIPCC_RTC__
inline void pack_merge_tail(const gse_merge_spec* spec, MergeTail& t) {
    t.flags = spec->flags | kMergeTailOn;
    t.P = spec->P; t.rank = spec->rank;
    t.island = spec->island; t.wscale = spec->wscale;
    for (int k = 0; k < GSE_NX; ++k) t.common.p[k] = spec->common_pivot[k];
    t.rows = spec->rows;
    t.moments26 = spec->moments26; t.merged26 = spec->merged26; t.estimate6 = spec->estimate6; t.status = spec->status;
}
#endif

// The first wavefront of the finishing workgroup: sm[26] (LDS) holds the shard's moments, log_w = log of the slot's weight sum
// (pending; formed by the caller while its loads are in flight).
// Forms the shard's row (rows[rank], other rows are the caller's to zero); a single shard (P == 1) is combined and estimated.
// sm and sx: >= 32 doubles of LDS each; sm is overwritten with the merged moments.
__device__ __forceinline__ void merge_tail_wave(const MergeTail& a, double* sm, double* sx, double log_w, int lane) {
    double la = 0.0, wsd = 0.0;
    if (lane == 0) {
        la = a.island ? a.island[0] : 0.0;
        wsd = a.wscale ? a.wscale[0] : 0.0;
        if (a.flags & GSE_MERGE_PENDING) {
            // the moments are those of the resampled ensemble (weights 1 / N); the copy's launch will add this to island[0]
            // (pf_step_kernel's gather stage: island[b] += log(W) + wscale ln 2)
            la += log_w + wsd * 0.69314718055994530942;
        }
    }
    la = __shfl(la, 0, 64);
    const int ws = (a.flags & GSE_MERGE_PENDING) ? 0 : __shfl((int)wsd, 0, 64);
    if (lane < 22) {
        const double v = merge_row_element(sm, la, ws, a.common, lane);
        a.rows[(int64_t)a.rank * 22 + lane] = v;
        sx[lane] = v;
    }
    if (a.P != 1) return;                              // (uniform) the all-reduce and gse_merge_end carry on
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double mine;
    const int status = merge_rows_wave(sx, 1, a.common, lane, mine);
    if (lane == 0) a.status[0] = status;
    if (lane < 26) a.merged26[lane] = mine;
    if (!a.estimate6) return;
    __builtin_amdgcn_wave_barrier();
    if (lane < 32) sm[lane] = (lane < 26) ? mine : 0.0;   // (the shard's moments are spent: reuse for the merged ones)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (a.flags & GSE_MERGE_COV_SIZE) estimates_wave(sm, a.estimate6, lane);
    else if (lane < GSE_NX) a.estimate6[lane] = sm[1 + lane] + sm[21 + lane] * sm[0];
}

}  // namespace gse
