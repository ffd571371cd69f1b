// gse_gsukf_cycle_lanes.h -- resample + predict + update of a LARGE bank of Gaussians in one launch (round 4).
//
// Until round 4 a cycle of a bank beyond 4096 Gaussians was two launches: ancestors_scatter_kernel (the resample's ancestors
// from the weights and the quarter sums the previous launch left: 5.3 us at 2^18 Gaussians, a chain of dependent latencies on
// 256 workgroups) and gsukf_predict_update_lanes_kernel (35 us, bound by its instruction count), which read every Gaussian from
// its ancestor's row.  Here the second kernel finds the ancestors itself: a workgroup of 256 lanes owns 256 outputs -- one quarter
// of a tile of 1024, exactly what a workgroup of the particle filter's cluster kernels owns -- and runs THEIR gather stage
// (cluster_gather, gse_pf_cluster_kernels.h: tile prefix from the quarter sums, the ancestor tiles expanded by the inverse map,
// look-up path for wide ranges) on the weights, quarter sums and quad prefixes of the previous launch: the launch of its own,
// and the gap in front of it, are gone.  Same ancestors as ancestors_scatter_kernel bit for bit (both re-form cumsum_kernel's
// keys), hence the same bank.  It pays while the bank leaves the chip room -- up to 2^17 Gaussians, two workgroups per CU in one
// generation: 13.1 / 13.6 / 14.3 / 16.6 / 22.4 us per cycle at 300 / 4096 / 2^14 / 2^16 / 2^17 Gaussians for 16.2 / 18.6 /
// 16.2 / 18.3 / 23.6 of the two launches -- and not beyond: this kernel is bound by its instruction count at two wavefronts per SIMD, and while one
// of the two waits through the gather stage's chain of latencies the SIMD issues at half its rate (2^18: 40.0 us for 37.9;
// 43 / 35 us with / without the stage under the profiler, whichever order the workgroups of a CU take noise and gather in).
// The class therefore uses it for banks of at most 512 workgroups (filter/gs_ukf.py).
// Built-in model only (a user model keeps the two launches); N a multiple of 4, at most kClusterMaxT tiles.
#pragma once

#include "gse_gsukf_kernels.h"
#include "gse_pf_cluster_kernels.h"

namespace gse {

static_assert(kLaneBank == kQuarterSlots && kLaneBank == kBlock, "a workgroup of lanes is a quarter of a tile");

// grid (4 T, B).  mu_in / P_in / w_in: the bank as the previous update left it; mu / P / w_out: other arrays.  qsum_in, qx_in:
// [B][T][4], [B][T][256] of w_in (lanes_quarter_scan); qsum_out, qx_out: those of w_out.  idx_out: the ancestors, or null.
template <bool kInjected, int kDiagonal>
__global__ __launch_bounds__(kLaneBank, GSE_GSUKF_PRED_WAVES) void gsukf_cycle_lanes_kernel(
    const float* __restrict__ mu_in, const float* __restrict__ P_in, const float* __restrict__ w_in, float* __restrict__ mu,
    float* __restrict__ P, float* __restrict__ w_out, int32_t* __restrict__ idx_out, const float* __restrict__ noise, int64_t N,
    int T, int32_t* __restrict__ status, const double* __restrict__ qsum_in, const double* __restrict__ qx_in,
    double* __restrict__ qsum_out, double* __restrict__ qx_out, const ResampleArgs ra, const GsukfPredictUpdateArgs a) {
    __shared__ DrawParams dp;
    __shared__ float stage[kLaneBank * kStageStride];
    __shared__ ClusterGatherLds gl;
    __shared__ __attribute__((aligned(16))) float wq[kLaneBank];
    const int t = threadIdx.x, b = blockIdx.y;
    const int64_t g0 = (int64_t)blockIdx.x * kLaneBank;
    const int nloc = (int)max((int64_t)0, min((int64_t)kLaneBank, N - g0));      // (the last tile's quarters beyond N: 0)
    // the quarter sums of every tile (thread t < T holds tile t's), requested first
    double qv[4] = {0.0, 0.0, 0.0, 0.0};
    {
        const double* qs = qsum_in + (int64_t)b * 4 * T;
#pragma unroll
        for (int i = 0; i < 4; ++i) qv[i] = (t < T) ? qs[4 * t + i] : 0.0;
    }
    if (!kInjected) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&a.pred.draw);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&dp);
        for (int i = t; i < (int)(sizeof(DrawParams) / 4); i += kLaneBank) dst[i] = src[i];
    }
    double W;
    const int32_t anc = cluster_gather(gl, qv, w_in + (int64_t)b * N, qx_in + (int64_t)b * T * kBlock, N, N, T, ra.r[b], g0, t, W);
    // (cluster_gather's barriers have also published dp)
    const FilterIO bio = bank_inputs(a.pred.io, a.pred.u_dev, a.pred.z_dev, b);
    float w_new = 0.0f;
    if (t < nloc) {
        float* mine = stage + t * kStageStride;
        const int64_t row = (int64_t)b * N + anc;
        if (idx_out) idx_out[(int64_t)b * N + g0 + t] = anc;
        gsukf_predict_lane<kInjected, kDiagonal>(mu_in + row * kNx, P_in + row * kNxx, mine, b, g0 + t, N, noise, dp, status, a.pred,
                                                 bio);
        w_new = (float)(1.0 / (double)N);                                   // (every Gaussian weighs 1 / N after the resample)
        gsukf_update_lane(mine, &w_new, bio.u0, bio.u1, bio.z0, bio.z1, status, a.pdf, a.pred.sw);
        w_out[(int64_t)b * N + g0 + t] = w_new;
    }
    wq[t] = w_new;
    __syncthreads();
    if (t < 64) lanes_quarter_scan(wq, g0, N, b, (int)blockIdx.x, t, qsum_out, qx_out);
    if (t < nloc) lane_row_out(stage + t * kStageStride, mu + ((int64_t)b * N + g0 + t) * kNx, P + ((int64_t)b * N + g0 + t) * kNxx);
}

}  // namespace gse
