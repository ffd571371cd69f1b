// gse_host.h -- host-side helpers of the C ABI: error reporting, argument checks, kernel-argument packing.
#pragma once

#ifndef __HIPCC_RTC__
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#endif

#include "gse.h"
#include "gse_device.h"

namespace gse {

constexpr int kMomentBlocks = 512;      // partial sums per filter of the moment kernels
constexpr int kMomentWidth = 48;

#ifndef __HIPCC_RTC__               // ---- host only (this header is also seen by the run-time compilation of the kernels)

// Thread-local message behind gse_last_error().
char* last_error_buffer();
int fail(int code, const char* fmt, ...);

#define GSE_REQUIRE(cond, ...)                                        \
    do {                                                              \
        if (!(cond)) return ::gse::fail(GSE_ERR_ARG, __VA_ARGS__);    \
    } while (0)

#define GSE_HIP_CHECK(expr)                                                                   \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return ::gse::fail(GSE_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));   \
    } while (0)

// After a kernel launch: report launch-time errors without synchronising.
#define GSE_LAUNCH_CHECK(name)                                                                       \
    do {                                                                                             \
        hipError_t e_ = hipGetLastError();                                                           \
        if (e_ != hipSuccess)                                                                        \
            return ::gse::fail(GSE_ERR_HIP, "launch of %s failed: %s", name, hipGetErrorString(e_)); \
    } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int64_t tiles_of(int64_t n) { return (n + kTile - 1) / kTile; }

// Common geometry checks for the state-major particle-filter layout.
int check_pf_geometry(const char* fn, int B, int64_t N, int64_t ld);
int check_mixture(const char* fn, const gse_mixture* m, int nx);

// Packing of host inputs into kernel-argument structs.
void pack_io(FilterIOs& io, int B, const double* u, const double* z);
void pack_draw(DrawParams& dp, const gse_mixture& m);
void pack_pdf2(Pdf2& p, const gse_mixture& m);

// Workspace carving (all offsets multiples of 256 bytes).
struct Workspace {
    double* tile_sums;     // [B][T]
    double* tile_prefix;   // [B][T + 1]   exclusive prefix of tile sums; [T] = total
    double* partials;      // [B][kMomentBlocks][kMomentWidth]
    double* totals;        // [B]          sum of the weights of each filter
    double* scan_sums[2];  // [B][T]       tile sums that go with scan[slot]
    double* scan[2];       // [B][T][256]  per-thread exclusive prefixes inside each tile (GSE_OP_SCAN_OUT / _IN), two slots
    double* cumsum;        // [B][N]       normalised cumulative weights
    double* wscale;        // [B]          GSE_OP_WEIGHT_SCALE: log2 of the factor between the stored and the true weights
    double* qsums[2];      // [B][T][4]    GSE_OP_CLUSTER: the quarter sums that go with scan[slot] in quarter format
    float* samples[2];     // [B][64][8]   GSE_OP_SAMPLES: the weight scale's sample particles of the launch that filled scan[slot]
    uint32_t* kexact;      // [B]          GSE_OP_WEIGHT_SCALE, updates without a resample in front: scale_exact_kernel's maximum
    uint32_t* wstatus;     // [B]          sticky status of each filter (GSE_WSTATUS_*), raised by the kernels, cleared by the caller
    uint32_t* counters;    // [B]          tickets of the workgroups of a launch whose last one finishes the job (gse_pf_merge_begin);
                           //              zero between launches: the workspace is zero-initialised ONCE by its owner
    int64_t bytes;
};
Workspace carve_workspace(void* base, int B, int64_t N);

#endif  // !__HIPCC_RTC__

// ---- shared between gse_pf.hip (definitions) and gse_gsukf.hip ------------------------------------------
struct ResampleArgs {
    double r[GSE_MAX_B];
};

// Payload = what the resample kernel copies from ancestor to offspring.
struct PfPayload {                    // state-major particle rows
    const float* __restrict__ in;
    float* __restrict__ out;
    int64_t ld;
    __device__ __forceinline__ void copy(int b, int64_t anc, int64_t j) const {
        const float* s = in + (int64_t)b * GSE_NX * ld + anc;
        float* d = out + (int64_t)b * GSE_NX * ld + j;
        float v[GSE_NX];
#pragma unroll
        for (int k = 0; k < GSE_NX; ++k) v[k] = s[k * ld];
#pragma unroll
        for (int k = 0; k < GSE_NX; ++k) d[k * ld] = v[k];
    }
};

struct RowsPayload {                  // state-major particles in, row-major offspring out (the send buffer of a shard)
    const float* __restrict__ in;
    float* __restrict__ out;
    int64_t ld;
    __device__ __forceinline__ void copy(int b, int64_t anc, int64_t j) const {
        (void)b;
        float v[GSE_NX];
#pragma unroll
        for (int k = 0; k < GSE_NX; ++k) v[k] = in[k * ld + anc];
#pragma unroll
        for (int k = 0; k < GSE_NX; ++k) out[j * GSE_NX + k] = v[k];
    }
};

struct GsukfPayload {                 // row-major (mu[nx], P[nx][nx]) per Gaussian, nx = states of the model
    const float* __restrict__ mu_in;
    const float* __restrict__ P_in;
    float* __restrict__ mu_out;
    float* __restrict__ P_out;
    int64_t N;
    __device__ __forceinline__ void copy(int b, int64_t anc, int64_t j) const {
        constexpr int nx = kModelNx, nxx = nx * nx;
        const float* sm = mu_in + ((int64_t)b * N + anc) * nx;
        const float* sp = P_in + ((int64_t)b * N + anc) * nxx;
        float* dm = mu_out + ((int64_t)b * N + j) * nx;
        float* dp = P_out + ((int64_t)b * N + j) * nxx;
        float v[nx + nxx];
#pragma unroll
        for (int k = 0; k < nx; ++k) v[k] = sm[k];
#pragma unroll
        for (int k = 0; k < nxx; ++k) v[nx + k] = sp[k];
#pragma unroll
        for (int k = 0; k < nx; ++k) dm[k] = v[k];
#pragma unroll
        for (int k = 0; k < nxx; ++k) dp[k] = v[nx + k];
    }
};

struct NoPayload {                    // the ancestors only (idx_out of gather_kernel): a resample whose copy is deferred
    __device__ __forceinline__ void copy(int, int64_t, int64_t) const {}
};

#ifndef __HIPCC_RTC__
int launch_cumsum(const float* w, int64_t ldw, int B, int64_t N, uint32_t flags, double* island, const Workspace& ws,
                  double* c_out, hipStream_t st);
struct UserModel;
int launch_gsukf_gather(const GsukfPayload& pl, float* w_out, int32_t* idx, int B, int64_t N, const ResampleArgs& ra,
                        const Workspace& ws, hipStream_t st, UserModel* model = nullptr);
int launch_index_gather(float* w_out, int32_t* idx, int B, int64_t N, const ResampleArgs& ra, const Workspace& ws,
                        hipStream_t st);
// cumulative weights + ancestors of a bank of at most 4 tiles in ONE launch (small_resample_kernel); false: too large
// ... of a bank of up to 2048 tiles from its weights and tile sums, no cumulative weights written (ancestors_scatter_kernel;
// GSE_ANCESTORS_BY_SEARCH=1 in the environment: round 2's ancestors_kernel)
bool launch_ancestors(const float* w, float* w_out, int32_t* idx, int B, int64_t N, const ResampleArgs& ra, const Workspace& ws,
                      hipStream_t st, int sums_slot);   // sums_slot: the slot whose quarter sums describe w, or -1
// where the one-lane-per-Gaussian update launch leaves the quarter sums of its new weights ([B][4 T] doubles): the
// workspace's first slot of per-thread tile prefixes, which only the particle filter uses
// resample + predict + update of a large GS-UKF bank in one launch (gse_gsukf_cycle_lanes.h; the kernel lives in gse_pf.hip's
// translation unit, beside the gather stage it shares with the particle filter's cluster kernels)
struct GsukfPredictUpdateArgs;
constexpr int kCycleLanesMaxT = 256;
void launch_gsukf_cycle_lanes(int variant, const float* mu_in, const float* P_in, const float* w_in, float* mu, float* P, float* w_out,
                              int32_t* idx, const float* noise, int B, int64_t N, int32_t* status, const Workspace& ws, int slot_in,
                              const ResampleArgs& ra, const GsukfPredictUpdateArgs& a, hipStream_t st);
inline double* gsukf_quarter_sums(const Workspace& ws, int slot = 0) { return ws.qsums[slot]; }    // [B][T][4]
inline double* gsukf_quad_prefix(const Workspace& ws, int slot = 0) { return ws.scan[slot]; }      // [B][T][256]
// the partial sums of gse_pf_moments_resampled without the reduction that finishes them (gse_pf.hip); returns the number of
// partials per filter, or a negative error code
// merge (optional): the launch's last workgroup finishes the sums and runs the tail of gse_pf_merge_begin (B = 1)
int launch_moments_resampled_partials(const char* fn, const float* x, const float* w, int64_t ld, int B, int64_t N, const double* r,
                                      int slot, void* workspace, hipStream_t st, const gse_merge_spec* merge = nullptr);
int launch_moments_partials(const char* fn, const float* x, const float* w, int64_t ld, int B, int64_t N, void* workspace,
                            hipStream_t st);
bool launch_small_resample(const float* w, float* w_out, int32_t* idx, int B, int64_t N, const ResampleArgs& ra, hipStream_t st);
void launch_moments_finish(const double* partials, int nblk, int width, double* out, int out_stride, int B,
                           const float* pivot_src, int64_t pfs, int64_t pes, int pivot_at, int pivot_count, hipStream_t st);

#endif  // !__HIPCC_RTC__

}  // namespace gse
