// gse_pf.hip -- host side of the particle filter: launches and the C ABI.  The kernels are in gse_pf_kernels.h.
#include "gse_pf_kernels.h"
#include "gse_pf_cluster_kernels.h"
#include "gse_gsukf_cycle_lanes.h"
#include "gse_rtc.h"

#include <stdlib.h>

namespace gse {

// Exposed to gse_gsukf.hip
// Normalised cumulative weights of B filters into ws.cumsum (+ totals); shared by PF, GS-UKF and gse_cumsum_normalised.
int launch_cumsum(const float* w, int64_t ldw, int B, int64_t N, uint32_t flags, double* island, const Workspace& ws,
                  double* c_out, hipStream_t st) {
    const int T = (int)tiles_of(N);
    if (!(flags & GSE_RESAMPLE_TILES_VALID)) {
        hipLaunchKernelGGL(tile_sum_kernel, dim3(T, B), dim3(kBlock), 0, st, w, ws.tile_sums, N, ldw, T);
        GSE_LAUNCH_CHECK("tile_sum_kernel");
    }
    const double* wscale = (flags & GSE_RESAMPLE_WEIGHT_SCALE) ? ws.wscale : nullptr;     // (read by the island bookkeeping only)
    if (T <= kInlinePrefixMaxT && !(flags & GSE_RESAMPLE_SEPARATE_SCAN)) {
        hipLaunchKernelGGL(cumsum_kernel<true>, dim3(T, B), dim3(kBlock), 0, st, w, c_out, ws.tile_sums, ws.totals, island,
                           N, ldw, T, (int64_t)0, (int64_t)0, (int64_t)0, wscale);
    } else {
        hipLaunchKernelGGL(tile_scan_kernel, dim3(B), dim3(kBlock), 0, st, ws.tile_sums, ws.tile_prefix, T);
        GSE_LAUNCH_CHECK("tile_scan_kernel");
        hipLaunchKernelGGL(cumsum_kernel<false>, dim3(T, B), dim3(kBlock), 0, st, w, c_out, ws.tile_prefix, ws.totals,
                           island, N, ldw, T, (int64_t)(T + 1), (int64_t)0, (int64_t)T, wscale);
    }
    GSE_LAUNCH_CHECK("cumsum_kernel");
    return GSE_OK;
}

int launch_gsukf_gather(const GsukfPayload& pl, float* w_out, int32_t* idx, int B, int64_t N, const ResampleArgs& ra,
                        const Workspace& ws, hipStream_t st, UserModel* model) {
    const int64_t zero = 0;
    if (model) {                                     // the payload's row length is the model's: compiled with it
        void* params[] = {(void*)&pl, (void*)&ws.cumsum, (void*)&ws.totals, (void*)&w_out, (void*)&idx, (void*)&N, (void*)&N,
                          (void*)&ra, (void*)&zero, (void*)&N, (void*)&N};
        return launch_user_kernel(model, "gse::gather_kernel<gse::GsukfPayload>", dim3((unsigned)tiles_of(N), B), dim3(kBlock), 0,
                                  st, params);
    }
    hipLaunchKernelGGL(gather_kernel<GsukfPayload>, dim3((unsigned)tiles_of(N), B), dim3(kBlock), 0, st, pl, ws.cumsum,
                       ws.totals, w_out, idx, N, N, ra, zero, N, N);
    GSE_LAUNCH_CHECK("gather_kernel<gsukf>");
    return GSE_OK;
}

int launch_index_gather(float* w_out, int32_t* idx, int B, int64_t N, const ResampleArgs& ra, const Workspace& ws,
                        hipStream_t st) {
    const int64_t zero = 0;
    hipLaunchKernelGGL(gather_kernel<NoPayload>, dim3((unsigned)tiles_of(N), B), dim3(kBlock), 0, st, NoPayload{}, ws.cumsum,
                       ws.totals, w_out, idx, N, N, ra, zero, N, N);
    GSE_LAUNCH_CHECK("gather_kernel<index only>");
    return GSE_OK;
}

bool launch_ancestors(const float* w, float* w_out, int32_t* idx, int B, int64_t N, const ResampleArgs& ra, const Workspace& ws,
                      hipStream_t st, int sums_slot) {
    const bool quarter_sums_valid = sums_slot >= 0;
    const int T = (int)tiles_of(N);
    if (T > kAncestorsMaxT) return false;
    const size_t lds = ((size_t)(T + 1) * sizeof(double) + (size_t)T + 7) & ~(size_t)7;
    static const bool by_search = getenv("GSE_ANCESTORS_BY_SEARCH") != nullptr;     // (the round-2 kernel, kept for A/B runs)
    if (quarter_sums_valid) {           // the update launch left the quarter sums of these weights in the workspace
        if (by_search)
            hipLaunchKernelGGL(ancestors_kernel<1>, dim3(T, B), dim3(kBlock), lds, st, w, gsukf_quarter_sums(ws, sums_slot), w_out, idx, N, T, ra);
        else
            hipLaunchKernelGGL(ancestors_scatter_kernel<1>, dim3(T, B), dim3(kBlock), lds, st, w, gsukf_quarter_sums(ws, sums_slot), w_out,
                               idx, N, T, ra);
        return true;
    }
    hipLaunchKernelGGL(tile_sum_kernel, dim3(T, B), dim3(kBlock), 0, st, w, ws.tile_sums, N, N, T);
    if (by_search)
        hipLaunchKernelGGL(ancestors_kernel<0>, dim3(T, B), dim3(kBlock), lds, st, w, ws.tile_sums, w_out, idx, N, T, ra);
    else
        hipLaunchKernelGGL(ancestors_scatter_kernel<0>, dim3(T, B), dim3(kBlock), lds, st, w, ws.tile_sums, w_out, idx, N, T, ra);
    return true;
}

void launch_gsukf_cycle_lanes(int variant, const float* mu_in, const float* P_in, const float* w_in, float* mu, float* P, float* w_out,
                              int32_t* idx, const float* noise, int B, int64_t N, int32_t* status, const Workspace& ws, int slot_in,
                              const ResampleArgs& ra, const GsukfPredictUpdateArgs& a, hipStream_t st) {
    static_assert(kCycleLanesMaxT == kClusterMaxT, "the gather stage's LDS is laid out for this many tiles");
    const int T = (int)tiles_of(N), slot_out = 1 - slot_in;
    const dim3 grid((unsigned)(4 * T), (unsigned)B), block(kLaneBank);
    const float* none = nullptr;
#define GSE_CYCLE_LANES_GO(I_, D_, noise_)                                                                                          \
    hipLaunchKernelGGL((gsukf_cycle_lanes_kernel<I_, D_>), grid, block, 0, st, mu_in, P_in, w_in, mu, P, w_out, idx, noise_, N, T,     \
                       status, gsukf_quarter_sums(ws, slot_in), gsukf_quad_prefix(ws, slot_in), gsukf_quarter_sums(ws, slot_out),    \
                       gsukf_quad_prefix(ws, slot_out), ra, a)
    if (variant == 0) GSE_CYCLE_LANES_GO(true, -1, noise);
    else if (variant == 1) GSE_CYCLE_LANES_GO(false, 1, none);
    else GSE_CYCLE_LANES_GO(false, 0, none);
#undef GSE_CYCLE_LANES_GO
}

bool launch_small_resample(const float* w, float* w_out, int32_t* idx, int B, int64_t N, const ResampleArgs& ra, hipStream_t st) {
    if (tiles_of(N) > kSmallResampleTiles) return false;
    hipLaunchKernelGGL(small_resample_kernel<0>, dim3(B), dim3(kSmallResampleBlock), 0, st, w, w_out, idx, N, ra);
    return true;
}

#ifndef GSE_MOMRES_TOTAL_BLOCKS
#define GSE_MOMRES_TOTAL_BLOCKS 1024   // workgroups of the resampled-moments pass over all filters (4 per CU: one resident generation)
#endif
#ifndef GSE_MOMRES_MIN_BLOCKS
#define GSE_MOMRES_MIN_BLOCKS 32
#endif
int launch_moments_resampled_partials(const char* fn, const float* x, const float* w, int64_t ld, int B, int64_t N, const double* r,
                                      int slot, void* workspace, hipStream_t st, const gse_merge_spec* merge) {
    if (int rc = check_pf_geometry(fn, B, N, ld)) return rc;
    if (!(x && w && r && workspace)) return fail(GSE_ERR_ARG, "%s: null argument", fn);
    if (!(aligned16(x) && aligned16(w) && aligned16(workspace))) return fail(GSE_ERR_ARG, "%s: pointers must be 16-byte aligned", fn);
    const bool quarters = (slot & GSE_SLOT_QUARTERS) != 0;      // the slot is in quarter format (a GSE_OP_CLUSTER update wrote it)
    slot &= ~GSE_SLOT_QUARTERS;
    if (!(slot == 0 || slot == 1)) return fail(GSE_ERR_ARG, "%s: slot must be 0 or 1 (| GSE_SLOT_QUARTERS)", fn);
    const int T = (int)tiles_of(N);
    if (T > 2048) return fail(GSE_ERR_ARG, "%s: at most 2048 tiles (N <= 2^21)", fn);
    if (merge && B != 1) return fail(GSE_ERR_ARG, "%s: the merge is that of ONE filter (a shard)", fn);
    ResampleArgs ra;
    for (int b = 0; b < B; ++b) {
        if (!(r[b] >= 0.0 && r[b] < 1.0)) return fail(GSE_ERR_ARG, "%s: r[%d] = %g is outside [0, 1)", fn, b, r[b]);
        ra.r[b] = r[b];
    }
    Workspace ws = carve_workspace(workspace, B, N);
    // one wavefront per unit (a tile) and pass: no more workgroups than a quarter of the units (round 5: the rest only wrote
    // zeros for the reduction to read); with the merge's tail inside the launch at most 256 per half (its last workgroup adds
    // 32 partials per thread and half).  Half a tile per wavefront (GSE_MOMRES_SPLIT=2: twice the wavefronts waiting side by
    // side) is SLOWER for a filter of 2^20 particles alone on the device: 21.8 us for 19.2 (14.7 for 13.8 without the tail).
    int split = 1;
    if (const char* e = getenv("GSE_MOMRES_SPLIT")) split = atoi(e) == 2 ? 2 : 1;      // (tuning knob)
    int64_t cap = merge ? 256 * split : kMomentBlocks;
    if (const char* e = getenv("GSE_MOMRES_CAP")) cap = std::min<int64_t>(cap, std::max(1, atoi(e)));      // (tuning knob)
    const int nblk = (int)std::min<int64_t>(std::min<int64_t>(cap, ((int64_t)T * split + 3) / 4),
                                            std::max<int64_t>(GSE_MOMRES_TOTAL_BLOCKS / B, GSE_MOMRES_MIN_BLOCKS));
    MergeTail tail = {};
    if (merge) pack_merge_tail(merge, tail);
    if (getenv("GSE_MERGE_NOTAIL")) tail.flags = 0;                      // (timing experiments only: the results are not formed)
    const dim3 grid(nblk, B);
    const size_t dyn = (size_t)(T + 1) * sizeof(double);
#define GSE_MOMRES_GO(Q_, S_, sums_)                                                                                            \
    hipLaunchKernelGGL((pf_moments_resampled_kernel<Q_, S_>), grid, dim3(kBlock), dyn, st, x, w, sums_, ws.scan[slot], ws.partials, \
                       N, ld, T, ra, ws.wstatus, tail, ws.counters)
    if (quarters) {
        if (split == 2) GSE_MOMRES_GO(true, 2, ws.qsums[slot]); else GSE_MOMRES_GO(true, 1, ws.qsums[slot]);
    } else {
        if (split == 2) GSE_MOMRES_GO(false, 2, ws.scan_sums[slot]); else GSE_MOMRES_GO(false, 1, ws.scan_sums[slot]);
    }
#undef GSE_MOMRES_GO
    if (hipGetLastError() != hipSuccess) return fail(GSE_ERR_HIP, "%s: pf_moments_resampled_kernel did not launch", fn);
    return nblk;
}

#ifndef GSE_MOMENT_TOTAL_BLOCKS
#define GSE_MOMENT_TOTAL_BLOCKS 1024
#endif
// the partial sums of gse_pf_moments (live particles and weights); returns the number of partials per filter, or an error code
int launch_moments_partials(const char* fn, const float* x, const float* w, int64_t ld, int B, int64_t N, void* workspace,
                            hipStream_t st) {
    if (int rc = check_pf_geometry(fn, B, N, ld)) return rc;
    if (!(x && w && workspace)) return fail(GSE_ERR_ARG, "%s: null argument", fn);
    if (!(aligned16(x) && aligned16(w) && aligned16(workspace))) return fail(GSE_ERR_ARG, "%s: pointers must be 16-byte aligned", fn);
    Workspace ws = carve_workspace(workspace, B, N);
    // workgroups per filter: four per CU in total, so that with many filters each workgroup sweeps several tiles and the
    // reduction at its end is paid less often (32 filters x 2^18 particles: 60 / 42 / 38 us with 4096 / 2048 / 1024
    // workgroups in total -- 5.3 TB/s at 1024)
    const int nblk = (int)std::min<int64_t>(std::min<int64_t>(kMomentBlocks, tiles_of(N)), std::max<int64_t>(GSE_MOMENT_TOTAL_BLOCKS / B, 32));
    hipLaunchKernelGGL(pf_moments_kernel, dim3(nblk, B), dim3(kBlock), 0, st, x, w, ws.partials, N, ld);
    if (hipGetLastError() != hipSuccess) return fail(GSE_ERR_HIP, "%s: pf_moments_kernel did not launch", fn);
    return nblk;
}

void launch_moments_finish(const double* partials, int nblk, int width, double* out, int out_stride, int B,
                           const float* pivot_src, int64_t pfs, int64_t pes, int pivot_at, int pivot_count, hipStream_t st) {
    hipLaunchKernelGGL(moments_finish_kernel, dim3(B), dim3(64 * kFinishGroups), 0, st, partials, nblk, width, out,
                       out_stride, pivot_src, pfs, pes, pivot_at, pivot_count);
}

}  // namespace gse

using namespace gse;

// ================================================================================================
// C ABI
// ================================================================================================
namespace {

struct StepLaunch {
    dim3 grid;
    hipStream_t st;
    const float* x_in; float* x_out; const float* w_in; float* w_out; const float* noise; int32_t* idx;
    const Workspace* ws;
    double* island;
    int64_t N, ld;
    int T, slot_in, slot_out;
    const StepArgs* a;
    UserModel* model;      // null: the kernels compiled into this library (bioreactor model)
    int* rc;               // launch status of a user-model kernel (compilation happens on first use)
};

template <bool G, bool P, bool I, bool U, int ND, int S>
void step_go(const StepLaunch& L) {
    const Workspace& ws = *L.ws;
    const size_t dyn = (S & 8) ? (size_t)kPrefixWindow * sizeof(double) : ((S & 1) ? (size_t)(L.T + 1) * sizeof(double) : 0);
    const double* prefix_g = (S & 8) ? ws.tile_prefix : nullptr;
    if (L.model) {
        // the same kernel, compiled at run time for the user's model (gse_rtc.hip); same argument list
        char name[160];
        snprintf(name, sizeof name, "gse::pf_step_kernel<%s, %s, %s, %s, %d, %d>", G ? "true" : "false", P ? "true" : "false",
                 I ? "true" : "false", U ? "true" : "false", ND, S);
        void* params[] = {(void*)&L.x_in, (void*)&L.x_out, (void*)&L.w_in, (void*)&L.w_out, (void*)&L.noise, (void*)&L.idx,
                          (void*)&ws.cumsum, (void*)&ws.totals, (void*)&ws.tile_sums, (void*)&ws.scan_sums[L.slot_in],
                          (void*)&ws.scan[L.slot_in], (void*)&ws.scan_sums[L.slot_out], (void*)&ws.scan[L.slot_out],
                          (void*)&prefix_g, (void*)&L.island, (void*)&L.N, (void*)&L.ld, (void*)&L.T, (void*)L.a};
        *L.rc = launch_user_kernel(L.model, name, L.grid, dim3(kBlock), dyn, L.st, params);
        return;
    }
    hipLaunchKernelGGL((pf_step_kernel<G, P, I, U, ND, S>), L.grid, dim3(kBlock), dyn, L.st, L.x_in, L.x_out, L.w_in, L.w_out,
                       L.noise, L.idx, ws.cumsum, ws.totals, ws.tile_sums, ws.scan_sums[L.slot_in], ws.scan[L.slot_in],
                       ws.scan_sums[L.slot_out], ws.scan[L.slot_out], prefix_g, L.island, L.N, L.ld, L.T, *L.a);
}

// Run-time stage flags -> template instance.  Instantiated: every stage combination; the component count folded in
// (ND = 2) for update stages; the scan modes only for the Philox (not injected-noise) variants with ND = 2.
template <bool G, bool P, bool I, bool U, int ND>
bool step_pick_scan(int scan, const StepLaunch& L) {
    // Injected noise (the parity mode) has the scan modes only where a closed loop needs them: the whole single-launch
    // cycle (3, 7) and the predict + update launch that leaves tile prefixes behind (2, 6) -- so that a loop fed recorded
    // noise runs the SAME launches as one that draws from Philox.
    constexpr bool kScanOk = (!U || ND == 2) && (!I || (P && U));
    if (scan == 0) { step_go<G, P, I, U, ND, 0>(L); return true; }
    if constexpr (kScanOk && G && !I) if (scan == 1) { step_go<G, P, I, U, ND, 1>(L); return true; }
    if constexpr (kScanOk && U && (!I || !G)) if (scan == 2) { step_go<G, P, I, U, ND, 2>(L); return true; }
    if constexpr (kScanOk && G && U) if (scan == 3) { step_go<G, P, I, U, ND, 3>(L); return true; }
    if constexpr (!G) {                                          // + 4: inputs from device memory
        if (scan == 4) { step_go<G, P, I, U, ND, 4>(L); return true; }
        if constexpr (kScanOk && U) if (scan == 6) { step_go<G, P, I, U, ND, 6>(L); return true; }
    }
    if constexpr (G && P && U && kScanOk) {                      // ... also in the single-launch cycle (resident closed loop)
        if (scan == 7) { step_go<G, P, I, U, ND, 7>(L); return true; }
    }
    if constexpr (G && !I && kScanOk) {                          // + 8: the tile prefix from tile_scan_kernel (N > 2^21)
        if (scan == 9) { step_go<G, P, I, U, ND, 9>(L); return true; }
        if constexpr (U) if (scan == 11) { step_go<G, P, I, U, ND, 11>(L); return true; }
    }
    return false;
}
template <bool G, bool P, bool I, bool U>
bool step_pick_nd(bool nd2, int scan, const StepLaunch& L) {
    if constexpr (U) {
        if (nd2) return step_pick_scan<G, P, I, U, 2>(scan, L);
    }
    return step_pick_scan<G, P, I, U, 0>(scan, L);
}
template <bool G, bool P, bool I>
bool step_pick_u(bool u, bool nd2, int scan, const StepLaunch& L) {
    if constexpr (G || P) {
        if (!u) return step_pick_nd<G, P, I, false>(nd2, scan, L);
    }
    return u ? step_pick_nd<G, P, I, true>(nd2, scan, L) : false;
}
template <bool G>
bool step_pick_p(bool p, bool i, bool u, bool nd2, int scan, const StepLaunch& L) {
    if (p) return i ? step_pick_u<G, true, true>(u, nd2, scan, L) : step_pick_u<G, true, false>(u, nd2, scan, L);
    return step_pick_u<G, false, false>(u, nd2, scan, L);
}

// The one place that validates and dispatches a fused step.
int pf_step(uint32_t ops, const float* x_in, float* x_out, const float* w_in, float* w_out, int32_t* idx, int64_t ld,
            int B, int64_t N, const double* u, double dt, const double* z, const double* r, const gse_mixture* state_pdf,
            const gse_rng* rng, const float* noise, const gse_mixture* measurement_pdf, double* island, void* workspace,
            void* stream, const char* fn, UserModel* model = nullptr) {
    if (int rc = check_pf_geometry(fn, B, N, ld)) return rc;
    const bool G = ops & GSE_OP_GATHER, P = ops & GSE_OP_PREDICT, U = ops & GSE_OP_UPDATE;
    const uint32_t known = GSE_OP_GATHER | GSE_OP_PREDICT | GSE_OP_UPDATE | GSE_OP_SCAN_IN | GSE_OP_SCAN_OUT |
                           GSE_OP_SCAN_IN_SLOT1 | GSE_OP_SCAN_OUT_SLOT1 | GSE_OP_U_DEVICE | GSE_OP_Z_DEVICE |
                           GSE_OP_PREFIX_GLOBAL | GSE_OP_WEIGHT_SCALE | GSE_OP_CLUSTER | GSE_OP_SAMPLES | GSE_OP_UNIFORM_WEIGHTS;
    const bool prefix_global = (ops & GSE_OP_PREFIX_GLOBAL) != 0;
    GSE_REQUIRE((ops & (GSE_OP_GATHER | GSE_OP_PREDICT | GSE_OP_UPDATE)) != 0 && (ops & ~known) == 0,
                "%s: bad ops mask 0x%x", fn, ops);
    const int scan = ((ops & GSE_OP_SCAN_IN) ? 1 : 0) | ((ops & GSE_OP_SCAN_OUT) ? 2 : 0);
    GSE_REQUIRE(!(scan & 1) || G, "%s: GSE_OP_SCAN_IN needs GSE_OP_GATHER", fn);
    GSE_REQUIRE(!(scan & 1) || (w_in && aligned16(w_in)),
                "%s: GSE_OP_SCAN_IN re-forms the cumulative weights from w_in (the weights the tile sums and prefixes of the slot "
                "were taken from): it must be given", fn);
    GSE_REQUIRE(!(scan & 2) || U, "%s: GSE_OP_SCAN_OUT needs GSE_OP_UPDATE", fn);
    GSE_REQUIRE(!prefix_global || (scan & 1), "%s: GSE_OP_PREFIX_GLOBAL needs GSE_OP_SCAN_IN", fn);
    GSE_REQUIRE(!(scan & 1) || prefix_global || tiles_of(N) <= kScanMaxT,
                "%s: GSE_OP_SCAN_IN rebuilds the tile prefix in LDS for at most %d tiles (N <= %d); beyond, run "
                "gse_pf_resample_begin(GSE_RESAMPLE_PREFIX_FROM_SLOT) and pass GSE_OP_PREFIX_GLOBAL", fn, kScanMaxT, kScanMaxT * kTile);
    GSE_REQUIRE(!prefix_global || !((ops & GSE_OP_U_DEVICE) || (ops & GSE_OP_Z_DEVICE)),
                "%s: GSE_OP_PREFIX_GLOBAL is not available with device-resident u / z", fn);
    GSE_REQUIRE(x_in && aligned16(x_in), "%s: x_in must be a 16-byte aligned device pointer", fn);
    GSE_REQUIRE((workspace && aligned16(workspace)) || !(ops & (GSE_OP_GATHER | GSE_OP_UPDATE)),
                "%s: gather / update need a 16-byte aligned device workspace", fn);
    GSE_REQUIRE(N < ((int64_t)1 << 31), "%s: N must be below 2^31", fn);
    GSE_REQUIRE(!(ops & GSE_OP_GATHER) || N <= ((int64_t)1 << 30), "%s: gather uses 32-bit byte offsets: N must not exceed 2^30", fn);
    StepArgs a;
    const bool u_on_device = (ops & GSE_OP_U_DEVICE) != 0, z_on_device = (ops & GSE_OP_Z_DEVICE) != 0;
    GSE_REQUIRE(!u_on_device || (P && u), "%s: GSE_OP_U_DEVICE needs GSE_OP_PREDICT and a device pointer u", fn);
    GSE_REQUIRE(!z_on_device || (U && z), "%s: GSE_OP_Z_DEVICE needs GSE_OP_UPDATE and a device pointer z", fn);
    GSE_REQUIRE(!(u_on_device || z_on_device) || !G || (P && U && scan == 3) || (ops & GSE_OP_CLUSTER),
                "%s: with a gather, device-resident u / z need the whole single-launch cycle (gather + predict + update, scans "
                "in and out)", fn);
    pack_io(a.io, B, ((P || U) && !u_on_device ? u : nullptr), (U && !z_on_device ? z : nullptr));   // (g may read u)
    a.u_dev = u_on_device ? u : nullptr;
    a.z_dev = z_on_device ? z : nullptr;
    if (G) {
        GSE_REQUIRE(x_out && w_out && r, "%s: gather needs x_out, w_out and r", fn);
        GSE_REQUIRE(x_out != x_in && aligned16(x_out) && aligned16(w_out), "%s: x_out must be a separate aligned buffer", fn);
        GSE_REQUIRE(!U || !w_in || w_in != w_out, "%s: w_out must not alias w_in when gathering", fn);
        for (int b = 0; b < B; ++b) {
            GSE_REQUIRE(r[b] >= 0.0 && r[b] < 1.0, "%s: r[%d] = %g is outside [0, 1)", fn, b, r[b]);
            a.ra.r[b] = r[b];
        }
    }
    if (P) {
        GSE_REQUIRE(u, "%s: predict needs u", fn);
        GSE_REQUIRE(x_out && aligned16(x_out), "%s: predict needs an aligned x_out (may equal x_in without gather)", fn);
        GSE_REQUIRE(G || x_out == x_in || ((ops & GSE_OP_WEIGHT_SCALE) && U),
                    "%s: without gather the step is in place: x_out must equal x_in (except predict + update with "
                    "GSE_OP_WEIGHT_SCALE, which must NOT be in place)", fn);
        GSE_REQUIRE(!((ops & GSE_OP_WEIGHT_SCALE) && U) || x_out != x_in,
                    "%s: with GSE_OP_WEIGHT_SCALE a launch that predicts and updates reads the scale's sample particles from "
                    "x_in while other workgroups write x_out: the two must be separate buffers", fn);
        if (noise) {
            GSE_REQUIRE(aligned16(noise), "%s: noise must be 16-byte aligned", fn);
            a.draw = DrawParams{};
            a.rng = RngKey{};
        } else {
            GSE_REQUIRE(state_pdf && rng, "%s: predict needs state_pdf and rng (or injected noise)", fn);
            if (int rc = check_mixture(fn, state_pdf, model ? user_model_nx(model) : kModelNx)) return rc;
            GSE_REQUIRE((rng->element_offset & 3u) == 0, "%s: rng element_offset must be a multiple of 4", fn);
            pack_draw(a.draw, *state_pdf);
            a.rng = make_rng_key(*rng, kPurposePredict);
        }
    }
    if (U) {
        GSE_REQUIRE(z && measurement_pdf && w_out && aligned16(w_out), "%s: update needs z, measurement_pdf and w_out", fn);
        GSE_REQUIRE(G || (w_in && aligned16(w_in)), "%s: update without gather needs w_in", fn);
        if (int rc = check_mixture(fn, measurement_pdf, GSE_NY)) return rc;
        pack_pdf2(a.pdf, *measurement_pdf);
        GSE_REQUIRE(!(ops & GSE_OP_WEIGHT_SCALE) || w_in != w_out,
                    "%s: with GSE_OP_WEIGHT_SCALE an update writes its weights to another buffer (w_out != w_in): every workgroup "
                    "of the launch scores the OLD weights of the scale's sample particles", fn);
    }
    a.dt = dt;
    const bool I = P && noise != nullptr;
    const int T = (int)tiles_of(N);
    const Workspace ws = carve_workspace(workspace, B, N);
    GSE_REQUIRE(!(ops & GSE_OP_WEIGHT_SCALE) || workspace, "%s: GSE_OP_WEIGHT_SCALE keeps the exponent in the workspace", fn);
    a.wscale = (ops & GSE_OP_WEIGHT_SCALE) ? ws.wscale : nullptr;
    // the scale's sample particles travel beside the scan slots (gse_pf_kernels.h, scale_samples_carried)
    const bool samples_fit = (ops & GSE_OP_WEIGHT_SCALE) && !(ops & GSE_OP_CLUSTER) && N >= kSamplesMinN && N % kScaleSamples == 0;
    a.smp_out = (samples_fit && U && (scan & 2)) ? ws.samples[(ops & GSE_OP_SCAN_OUT_SLOT1) ? 1 : 0] : nullptr;
    a.smp_in = (samples_fit && U && (scan & 1) && (ops & GSE_OP_SAMPLES)) ? ws.samples[(ops & GSE_OP_SCAN_IN_SLOT1) ? 1 : 0] : nullptr;
    a.smp_stride = (uint32_t)(N / kScaleSamples);
    a.smp_magic = a.smp_stride ? (uint32_t)((((uint64_t)1 << 32) + a.smp_stride - 1) / a.smp_stride) : 0u;
    GSE_REQUIRE(!(a.smp_in && a.smp_out) || a.smp_in != a.smp_out, "%s: SCAN_IN and SCAN_OUT must use different slots", fn);
    a.k_exact = nullptr;
    a.wstatus = workspace ? ws.wstatus : nullptr;
    if ((ops & GSE_OP_WEIGHT_SCALE) && U && !G && !(ops & GSE_OP_UNIFORM_WEIGHTS)) {
        // an update without a resample in front of it: the scale's anchor over ALL particles, by a pass of its own
        // (scale_exact_kernel: the weights may span any number of binades by now, and 64 samples need not see the heavy ones)
        hipStream_t st = (hipStream_t)stream;
        GSE_HIP_CHECK(hipMemsetAsync(ws.kexact, 0, (size_t)B * sizeof(uint32_t), st));
        a.k_exact = ws.kexact;
        const dim3 grid((unsigned)std::min<int64_t>((N + kBlock - 1) / kBlock, 2048), (unsigned)B);
        const bool dev_io = u_on_device || z_on_device;
        if (model) {
            char name[96];
            snprintf(name, sizeof name, "gse::scale_exact_kernel<%s, %s>", P ? "true" : "false", dev_io ? "true" : "false");
            uint32_t* kx = ws.kexact;
            void* params[] = {(void*)&x_in, (void*)&w_in, (void*)&N, (void*)&ld, (void*)&kx, (void*)&a};
            if (int rc = launch_user_kernel(model, name, grid, dim3(kBlock), 0, st, params)) return rc;
        } else if (P && dev_io) {
            hipLaunchKernelGGL((scale_exact_kernel<true, true>), grid, dim3(kBlock), 0, st, x_in, w_in, N, ld, ws.kexact, a);
        } else if (P) {
            hipLaunchKernelGGL((scale_exact_kernel<true, false>), grid, dim3(kBlock), 0, st, x_in, w_in, N, ld, ws.kexact, a);
        } else if (dev_io) {
            hipLaunchKernelGGL((scale_exact_kernel<false, true>), grid, dim3(kBlock), 0, st, x_in, w_in, N, ld, ws.kexact, a);
        } else {
            hipLaunchKernelGGL((scale_exact_kernel<false, false>), grid, dim3(kBlock), 0, st, x_in, w_in, N, ld, ws.kexact, a);
        }
        GSE_LAUNCH_CHECK("scale_exact_kernel");
    }
    if (ops & GSE_OP_CLUSTER) {
        // mid-size ensembles: four workgroups per tile, one particle per thread, scan slots in quarter format
        // (gse_pf_cluster_kernels.h); the same particles, weights and ancestors as the kernels below
        GSE_REQUIRE(!model, "%s: GSE_OP_CLUSTER is for the built-in model", fn);
        GSE_REQUIRE(T <= kClusterMaxT, "%s: GSE_OP_CLUSTER takes at most %d tiles (N <= %d)", fn, kClusterMaxT, kClusterMaxT * kTile);
        GSE_REQUIRE(!I, "%s: GSE_OP_CLUSTER draws its noise (no injected-noise variant)", fn);
        GSE_REQUIRE(!prefix_global, "%s: GSE_OP_CLUSTER rebuilds the tile prefix itself", fn);
        GSE_REQUIRE(!G || (scan & 1), "%s: GSE_OP_CLUSTER gathers from a quarter-format scan slot (GSE_OP_SCAN_IN)", fn);
        GSE_REQUIRE(!U || ((scan & 2) && a.pdf.nd == 2), "%s: GSE_OP_CLUSTER updates leave a quarter-format slot (GSE_OP_SCAN_OUT) "
                    "and take a 2-component measurement mixture", fn);
        GSE_REQUIRE(G || x_out, "%s: null x_out", fn);
        const int si = (ops & GSE_OP_SCAN_IN_SLOT1) ? 1 : 0, so = (ops & GSE_OP_SCAN_OUT_SLOT1) ? 1 : 0;
        GSE_REQUIRE(!(G && U) || si != so, "%s: SCAN_IN and SCAN_OUT must use different slots", fn);
        const bool dev_io = u_on_device || z_on_device;
        const dim3 grid((unsigned)(4 * T), (unsigned)B);
        hipStream_t st = (hipStream_t)stream;
#define GSE_CLUSTER_GO(G_, P_, U_, D_)                                                                                          \
        hipLaunchKernelGGL((pf_cluster_kernel<G_, P_, U_, D_>), grid, dim3(kBlock), 0, st, x_in, x_out, w_in, w_out, idx,        \
                           ws.qsums[si], ws.scan[si], ws.qsums[so], ws.scan[so], island, N, ld, T, a)
#define GSE_CLUSTER_PICK(D_)                                                       \
        do {                                                                       \
            if (G && P && U) GSE_CLUSTER_GO(true, true, true, D_);                 \
            else if (G && P) GSE_CLUSTER_GO(true, true, false, D_);                \
            else if (G && U) GSE_CLUSTER_GO(true, false, true, D_);                \
            else if (G) GSE_CLUSTER_GO(true, false, false, D_);                    \
            else if (P && U) GSE_CLUSTER_GO(false, true, true, D_);                \
            else if (P) GSE_CLUSTER_GO(false, true, false, D_);                    \
            else GSE_CLUSTER_GO(false, false, true, D_);                           \
        } while (0)
        if (dev_io) GSE_CLUSTER_PICK(true); else GSE_CLUSTER_PICK(false);
#undef GSE_CLUSTER_PICK
#undef GSE_CLUSTER_GO
        GSE_LAUNCH_CHECK("pf_cluster_kernel");
        return GSE_OK;
    }
    StepLaunch L;
    L.grid = dim3((unsigned)T, (unsigned)B);
    L.st = (hipStream_t)stream;
    L.x_in = x_in; L.x_out = x_out; L.w_in = w_in; L.w_out = w_out; L.noise = noise; L.idx = idx;
    L.ws = &ws;
    L.island = island;
    L.N = N; L.ld = ld; L.T = T;
    L.slot_in = (ops & GSE_OP_SCAN_IN_SLOT1) ? 1 : 0;
    L.slot_out = (ops & GSE_OP_SCAN_OUT_SLOT1) ? 1 : 0;
    L.a = &a;
    int model_rc = GSE_OK;
    L.model = model;
    L.rc = &model_rc;
    GSE_REQUIRE(!(scan == 3) || L.slot_in != L.slot_out, "%s: SCAN_IN and SCAN_OUT must use different slots", fn);
    const bool nd2 = U && a.pdf.nd == 2;
    const int variant = scan | ((u_on_device || z_on_device) ? 4 : 0) | (prefix_global ? 8 : 0);
    const bool ok = G ? step_pick_p<true>(P, I, U, nd2, variant, L) : step_pick_p<false>(P, I, U, nd2, variant, L);
    if (!ok) return fail(GSE_ERR_UNSUPPORTED, "%s: stage combination 0x%x is not available (scan modes need the Philox "
                         "draw and a 2-component measurement mixture)", fn, ops);
    if (model) return model_rc;
    GSE_LAUNCH_CHECK("pf_step_kernel");
    return GSE_OK;
}

}  // namespace

extern "C" {

int gse_pf_step(uint32_t ops, const float* x_in, float* x_out, const float* w_in, float* w_out, int32_t* idx, int64_t ld,
                int B, int64_t N, const double* u, double dt, const double* z, const double* r,
                const gse_mixture* state_pdf, const gse_rng* rng, const float* noise, const gse_mixture* measurement_pdf,
                double* island, void* workspace, void* stream) {
    return pf_step(ops, x_in, x_out, w_in, w_out, idx, ld, B, N, u, dt, z, r, state_pdf, rng, noise, measurement_pdf,
                   island, workspace, stream, "gse_pf_step");
}

// The same step with the kernels compiled at run time for a user model (gse_model_create, gse_rtc.hip).
int gse_pf_step_model(void* model, uint32_t ops, const float* x_in, float* x_out, const float* w_in, float* w_out, int32_t* idx,
                      int64_t ld, int B, int64_t N, const double* u, double dt, const double* z, const double* r,
                      const gse_mixture* state_pdf, const gse_rng* rng, const float* noise,
                      const gse_mixture* measurement_pdf, double* island, void* workspace, void* stream) {
    GSE_REQUIRE(model, "gse_pf_step_model: null model");
    return pf_step(ops, x_in, x_out, w_in, w_out, idx, ld, B, N, u, dt, z, r, state_pdf, rng, noise, measurement_pdf,
                   island, workspace, stream, "gse_pf_step_model", static_cast<UserModel*>(model));
}

int gse_pf_predict(float* x, int64_t ld, int B, int64_t N, const double* u, double dt, const gse_mixture* state_pdf,
                   const gse_rng* rng, void* stream) {
    GSE_REQUIRE(state_pdf && rng, "gse_pf_predict: null argument");
    return pf_step(GSE_OP_PREDICT, x, x, nullptr, nullptr, nullptr, ld, B, N, u, dt, nullptr, nullptr, state_pdf, rng,
                   nullptr, nullptr, nullptr, nullptr, stream, "gse_pf_predict");
}

int gse_pf_predict_injected(float* x, int64_t ld, int B, int64_t N, const double* u, double dt, const float* noise,
                            void* stream) {
    GSE_REQUIRE(noise, "gse_pf_predict_injected: null argument");
    return pf_step(GSE_OP_PREDICT, x, x, nullptr, nullptr, nullptr, ld, B, N, u, dt, nullptr, nullptr, nullptr, nullptr,
                   noise, nullptr, nullptr, nullptr, stream, "gse_pf_predict_injected");
}

int gse_pf_update(const float* x, float* w, int64_t ld, int B, int64_t N, const double* u, const double* z,
                  const gse_mixture* measurement_pdf, void* workspace, void* stream) {
    GSE_REQUIRE(w, "gse_pf_update: null argument");      // (the bioreactor's g ignores u; a user model's may not)
    return pf_step(GSE_OP_UPDATE, x, nullptr, w, w, nullptr, ld, B, N, u, 0.0, z, nullptr, nullptr, nullptr, nullptr,
                   measurement_pdf, nullptr, workspace, stream, "gse_pf_update");
}

int gse_pf_slot_to_tiles(int B, int64_t N, int slot, void* workspace, void* stream) {
    GSE_REQUIRE(B >= 1 && B <= GSE_MAX_B && N >= 1 && workspace && (slot == 0 || slot == 1), "gse_pf_slot_to_tiles: bad argument");
    const int T = (int)tiles_of(N);
    const Workspace ws = carve_workspace(workspace, B, N);
    hipLaunchKernelGGL(quarters_to_tiles_kernel, dim3((unsigned)T, (unsigned)B), dim3(kBlock), 0, (hipStream_t)stream,
                       ws.qsums[slot], ws.scan[slot], ws.scan_sums[slot], ws.scan[slot], T);
    GSE_LAUNCH_CHECK("quarters_to_tiles_kernel");
    return GSE_OK;
}

int64_t gse_workspace_weight_scale_offset(int B, int64_t N) {
    if (B < 1 || B > GSE_MAX_B || N < 1) return -1;
    const Workspace ws = carve_workspace(nullptr, B, N);
    return (int64_t)(reinterpret_cast<char*>(ws.wscale) - static_cast<char*>(nullptr));
}

int gse_pf_resample_begin(const float* w, int64_t ld, int B, int64_t N, uint32_t flags, double* island, void* workspace,
                          void* stream) {
    if (int rc = check_pf_geometry("gse_pf_resample_begin", B, N, ld)) return rc;
    GSE_REQUIRE(w && workspace && aligned16(w) && aligned16(workspace), "gse_pf_resample_begin: bad pointer");
    const Workspace ws = carve_workspace(workspace, B, N);
    if (flags & GSE_RESAMPLE_PREFIX_FROM_SLOT) {
        // only the exclusive prefix over the tile sums a GSE_OP_SCAN_OUT update left in the slot: what a GSE_OP_SCAN_IN |
        // GSE_OP_PREFIX_GLOBAL gather reads (one workgroup per filter; the island bookkeeping is the gather stage's)
        const int T = (int)tiles_of(N);
        hipLaunchKernelGGL(tile_scan_kernel, dim3(B), dim3(kBlock), 0, (hipStream_t)stream,
                           ws.scan_sums[(flags & GSE_RESAMPLE_SLOT1) ? 1 : 0], ws.tile_prefix, T);
        GSE_LAUNCH_CHECK("tile_scan_kernel");
        return GSE_OK;
    }
    return launch_cumsum(w, ld, B, N, flags, island, ws, ws.cumsum, (hipStream_t)stream);
}

int gse_pf_resample(const float* x_in, const float* w, float* x_out, float* w_out, int32_t* idx, int64_t ld, int B,
                    int64_t N, const double* r, uint32_t flags, double* island, void* workspace, void* stream) {
    GSE_REQUIRE(w && w_out && w != w_out, "gse_pf_resample: w_out must be a buffer separate from w");
    GSE_REQUIRE(x_in && x_out && x_in != x_out && r, "gse_pf_resample: x_out must be a buffer separate from x_in; r is required");
    if (int rc = check_pf_geometry("gse_pf_resample", B, N, ld)) return rc;
    for (int b = 0; b < B; ++b)
        GSE_REQUIRE(r[b] >= 0.0 && r[b] < 1.0, "gse_pf_resample: r[%d] = %g is outside [0, 1)", b, r[b]);
    if (int rc = gse_pf_resample_begin(w, ld, B, N, flags, island, workspace, stream)) return rc;
    return pf_step(GSE_OP_GATHER, x_in, x_out, nullptr, w_out, idx, ld, B, N, nullptr, 0.0, nullptr, r, nullptr, nullptr,
                   nullptr, nullptr, nullptr, workspace, stream, "gse_pf_resample");
}

int gse_pf_gather_index(const float* x_in, const int32_t* idx, float* x_out, float* w_out, int64_t ld, int B, int64_t N,
                        void* stream) {
    if (int rc = check_pf_geometry("gse_pf_gather_index", B, N, ld)) return rc;
    GSE_REQUIRE(x_in && idx && x_out && w_out && x_in != x_out, "gse_pf_gather_index: x_out must be a buffer separate from x_in");
    hipLaunchKernelGGL(index_gather_kernel, dim3((unsigned)((N + kBlock - 1) / kBlock), (unsigned)B), dim3(kBlock), 0,
                       (hipStream_t)stream, x_in, idx, x_out, w_out, N, ld);
    GSE_LAUNCH_CHECK("index_gather_kernel");
    return GSE_OK;
}

// ---- exact resampling of an ensemble sharded over several GPUs (gpu_se_amd/distributed.py) ----------
int gse_pf_tile_sums(const float* w, int64_t ld, int B, int64_t N, double* sums, void* stream) {
    if (int rc = check_pf_geometry("gse_pf_tile_sums", B, N, ld)) return rc;
    GSE_REQUIRE(w && sums && aligned16(w), "gse_pf_tile_sums: bad pointer");
    const int T = (int)tiles_of(N);
    hipLaunchKernelGGL(tile_sum_kernel, dim3(T, B), dim3(kBlock), 0, (hipStream_t)stream, w, sums, N, ld, T);
    GSE_LAUNCH_CHECK("tile_sum_kernel");
    return GSE_OK;
}

int gse_tile_prefix(const double* sums, int64_t T, double* prefix, void* stream) {
    GSE_REQUIRE(sums && prefix && T >= 1 && T < ((int64_t)1 << 30), "gse_tile_prefix: bad argument");
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(kBlock), 0, (hipStream_t)stream, sums, prefix, (int)T);
    GSE_LAUNCH_CHECK("tile_scan_kernel");
    return GSE_OK;
}

int gse_pf_shard_cumsum(const float* w, int64_t N_local, const double* prefix_all, int64_t tile_offset, int64_t T_total,
                        double* c, void* stream) {
    GSE_REQUIRE(w && prefix_all && c && aligned16(w) && aligned16(c) && N_local >= 1, "gse_pf_shard_cumsum: bad argument");
    const int T = (int)tiles_of(N_local);
    GSE_REQUIRE(tile_offset >= 0 && tile_offset + T <= T_total, "gse_pf_shard_cumsum: this shard's tiles [%lld, %lld) lie "
                "outside the %lld tiles of the ensemble", (long long)tile_offset, (long long)(tile_offset + T), (long long)T_total);
    GSE_REQUIRE(N_local % kTile == 0 || tile_offset + T == T_total,
                "gse_pf_shard_cumsum: every shard but the last must hold a multiple of %d particles", kTile);
    const int64_t ldw = (N_local + 3) / 4 * 4;
    hipLaunchKernelGGL(cumsum_kernel<false>, dim3(T, 1), dim3(kBlock), 0, (hipStream_t)stream, w, c, prefix_all,
                       (double*)nullptr, (double*)nullptr, N_local, ldw, T, (int64_t)0, tile_offset, T_total,
                       (const double*)nullptr);
    GSE_LAUNCH_CHECK("cumsum_kernel<shard>");
    return GSE_OK;
}

int gse_pf_shard_gather(const float* x_in, int64_t ld, int64_t N_local, const double* c, int64_t j_base, int64_t count,
                        int64_t n_total, double r, float* rows_out, int32_t* idx_out, void* stream) {
    if (int rc = check_pf_geometry("gse_pf_shard_gather", 1, N_local, ld)) return rc;
    GSE_REQUIRE(x_in && c && rows_out && aligned16(x_in), "gse_pf_shard_gather: bad pointer");
    GSE_REQUIRE(j_base >= 0 && count >= 0 && j_base + count <= n_total && n_total >= N_local,
                "gse_pf_shard_gather: outputs [%lld, %lld) of %lld", (long long)j_base, (long long)(j_base + count),
                (long long)n_total);
    GSE_REQUIRE(r >= 0.0 && r < 1.0, "gse_pf_shard_gather: r = %g is outside [0, 1)", r);
    if (count == 0) return GSE_OK;
    RowsPayload pl{x_in, rows_out, ld};
    ResampleArgs ra;
    ra.r[0] = r;
    hipLaunchKernelGGL(gather_kernel<RowsPayload>, dim3((unsigned)tiles_of(count), 1), dim3(kBlock), 0, (hipStream_t)stream, pl,
                       c, (const double*)nullptr, (float*)nullptr, idx_out, N_local, count, ra, j_base, count, n_total);
    GSE_LAUNCH_CHECK("gather_kernel<rows>");
    return GSE_OK;
}

#ifdef GSE_STAMPS
// diagnostic build only: copy the phase stamps of the last fused launches to the host (synchronises)
int gse_debug_mstamps(unsigned long long* out32) {
    GSE_HIP_CHECK(hipDeviceSynchronize());
    GSE_HIP_CHECK(hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_mstamps), sizeof(unsigned long long) * 32));
    return GSE_OK;
}
int gse_debug_stamps(unsigned long long* out32) {
    GSE_HIP_CHECK(hipDeviceSynchronize());
    GSE_HIP_CHECK(hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 32));
    return GSE_OK;
}
int gse_debug_slow_hits(unsigned long long* out) {
    GSE_HIP_CHECK(hipDeviceSynchronize());
    GSE_HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_slow_hits), sizeof(unsigned long long)));
    return GSE_OK;
}
int gse_debug_block_times(unsigned long long* out8192, long long* info8192) {
    GSE_HIP_CHECK(hipDeviceSynchronize());
    GSE_HIP_CHECK(hipMemcpyFromSymbol(out8192, HIP_SYMBOL(g_block_times), sizeof(unsigned long long) * 8192));
    GSE_HIP_CHECK(hipMemcpyFromSymbol(info8192, HIP_SYMBOL(g_block_info), sizeof(long long) * 8192));
    return GSE_OK;
}
#endif

int gse_pf_moments(const float* x, const float* w, int64_t ld, int B, int64_t N, double* out, void* workspace,
                   void* stream) {
    GSE_REQUIRE(out, "gse_pf_moments: null argument");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = launch_moments_partials("gse_pf_moments", x, w, ld, B, N, workspace, st);
    if (nblk < 0) return nblk;
    launch_moments_finish(carve_workspace(workspace, B, N).partials, nblk, 21, out, 26, B, x, GSE_NX * ld, ld, 21, GSE_NX, st);
    GSE_LAUNCH_CHECK("moments_finish_kernel");
    return GSE_OK;
}

int gse_pf_moments_resampled(const float* x, const float* w, int64_t ld, int B, int64_t N, const double* r, int slot,
                             double* out, void* workspace, void* stream) {
    GSE_REQUIRE(out, "gse_pf_moments_resampled: null argument");
    hipStream_t st = (hipStream_t)stream;
    const int nblk = launch_moments_resampled_partials("gse_pf_moments_resampled", x, w, ld, B, N, r, slot, workspace, st);
    if (nblk < 0) return nblk;
    launch_moments_finish(carve_workspace(workspace, B, N).partials, nblk, 21, out, 26, B, x, GSE_NX * ld, ld, 21, GSE_NX, st);
    GSE_LAUNCH_CHECK("moments_finish_kernel");
    return GSE_OK;
}

}  // extern "C"
