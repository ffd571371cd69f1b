// gse_gsukf.hip -- Gaussian-sum unscented Kalman filter on gfx950: the C ABI over the kernels of gse_gsukf_kernels.h.
//
// Every entry point exists twice: `gse_gsukf_x(...)` launches the kernels compiled into this library (the model it was
// built for: the bioreactor, or -- the hipcc backend of gpu_se_amd/usermodel.py -- a user model), `gse_gsukf_x_model(model,
// ...)` launches the same kernels compiled at run time for a user model (gse_rtc.hip) with the same argument list.
#include "gse_gsukf_kernels.h"
#include "gse_rtc.h"

#include <stdlib.h>

#include <tuple>

namespace gse {

// Which family of kernels: one lane per Gaussian for everything beyond the 64 Gaussians of the one-launch cycle.  (Round 2
// first drew the line where a bank fills the chip that way -- 2^16 -- but below that a cycle is bound by its launches, and
// this family needs two of them up to 4096 Gaussians -- predict + update (+ the row copy of the resample before), and the
// ancestors -- or four beyond, against five: 17-21 us per cycle from 65 to 768 Gaussians against 23-25, 21-23 up to 4096,
// 27 up to 32768 against 30-39.)  GSE_GSUKF_LANES=0 / 1 in the environment forces one or the other (tests, A/B).
static bool lane_per_gaussian(int B, int64_t N) {
    static const int forced = [] {
        const char* e = getenv("GSE_GSUKF_LANES");
        return e ? atoi(e) : -1;
    }();
    if (forced >= 0) return forced != 0;
    return (int64_t)B * N > 64;
}

static int check_gsukf(const char* fn, int B, int64_t N) {
    GSE_REQUIRE(B >= 1 && B <= GSE_MAX_B, "%s: B = %d is outside 1..%d", fn, B, GSE_MAX_B);
    GSE_REQUIRE(N >= 1 && N < ((int64_t)1 << 31) / GSE_NSIGMA, "%s: N = %lld out of range", fn, (long long)N);
    return GSE_OK;
}

// One launch, either way: the kernel compiled into this library, or -- model != null -- the instance `name` of the same
// template compiled for the user's model at run time (first use: about a second), with the same argument list.
template <class... Params, class... Args>
static int launch(UserModel* model, const char* name, void (*kernel)(Params...), dim3 grid, dim3 block, hipStream_t st,
                  Args... args) {
    if (model) {
        std::tuple<Params...> held(args...);                       // the arguments in the kernel's own parameter types
        void* params[sizeof...(Params)];
        int i = 0;
        std::apply([&](auto&... p) { ((params[i++] = (void*)&p), ...); }, held);
        return launch_user_kernel(model, name, grid, block, 0, st, params);
    }
    hipLaunchKernelGGL(kernel, grid, block, 0, st, args...);
    GSE_LAUNCH_CHECK(name);
    return GSE_OK;
}

static int states_of(UserModel* model) { return model ? user_model_nx(model) : kNx; }

// Where u and z live: host arrays (copied into the kernel arguments) or, bit by bit, DEVICE arrays [B][2] fp64 that the
// kernel reads when it runs (GSE_IO_U_DEVICE / GSE_IO_Z_DEVICE: the `_dev` entry points, a closed loop on the device).
static void pack_bank_io(FilterIOs& io, const double*& u_dev, const double*& z_dev, int B, const double* u, const double* z,
                         uint32_t where) {
    u_dev = (where & GSE_IO_U_DEVICE) ? u : nullptr;
    z_dev = (where & GSE_IO_Z_DEVICE) ? z : nullptr;
    pack_io(io, B, u_dev ? nullptr : u, z_dev ? nullptr : z);
}

#define GSE_WIDE_STR_(x) #x
#define GSE_WIDE_STR(x) GSE_WIDE_STR_(x)
#define GSE_WIDE GSE_WIDE_STR(GSE_GSUKF_WIDE_BANK)

// ------------------------------------------------------------------------------------------------ implementations
static int gsukf_sigma_points(UserModel* model, const float* mu, const float* P, float* sigmas, int B, int64_t N,
                              int32_t* status, void* stream) {
    if (int rc = check_gsukf("gse_gsukf_sigma_points", B, N)) return rc;
    GSE_REQUIRE(mu && P && sigmas, "gse_gsukf_sigma_points: null argument");
    const unsigned gx = (unsigned)((N + kGaussPerBlock - 1) / kGaussPerBlock);
    return launch(model, "gse::gsukf_sigma_kernel<0>", gsukf_sigma_kernel<0>, dim3(gx, B), dim3(kBlock), (hipStream_t)stream, mu, P,
                  sigmas, N, status);
}

static int gsukf_predict(UserModel* model, float* mu, float* P, int B, int64_t N, const double* u, double dt,
                         const float* w_sigma, const gse_mixture* state_pdf, const gse_rng* rng, const float* noise,
                         int32_t* status, void* stream, uint32_t where = 0) {
    if (int rc = check_gsukf("gse_gsukf_predict", B, N)) return rc;
    GSE_REQUIRE(mu && P && u && w_sigma, "gse_gsukf_predict: null argument");
    GSE_REQUIRE(noise || (state_pdf && rng), "gse_gsukf_predict: need either noise or (state_pdf, rng)");
    const int nx = states_of(model);
    GsukfPredictArgs a;
    pack_bank_io(a.io, a.u_dev, a.z_dev, B, u, nullptr, where & GSE_IO_U_DEVICE);
    pack_sigma_weights(a.sw, w_sigma, 2 * nx + 1);
    a.dt = dt;
    const bool lanes = lane_per_gaussian(B, N);
    const unsigned gx = lanes ? (unsigned)((N + kLaneBank - 1) / kLaneBank) : (unsigned)((N + kWideBank - 1) / kWideBank);
    const dim3 grid(gx, B), block(lanes ? kLaneBank : kWideBank * kGroup);
    hipStream_t st = (hipStream_t)stream;
    if (noise) {
        a.draw = DrawParams{};
        a.rng = RngKey{};
        if (lanes) return launch(model, "gse::gsukf_predict_lanes_kernel<true>", gsukf_predict_lanes_kernel<true>, grid, block, st, mu, P, noise, N, status, a);
        return launch(model, "gse::gsukf_predict_kernel<true, " GSE_WIDE ">", gsukf_predict_kernel<true, kWideBank>, grid, block, st, mu, P, noise, N, status, a);
    }
    if (int rc = check_mixture("gse_gsukf_predict", state_pdf, nx)) return rc;
    pack_draw(a.draw, *state_pdf);
    a.rng = make_rng_key(*rng, kPurposeGsukf);
    const float* none = nullptr;
    if (lanes) return launch(model, "gse::gsukf_predict_lanes_kernel<false>", gsukf_predict_lanes_kernel<false>, grid, block, st, mu, P, none, N, status, a);
    return launch(model, "gse::gsukf_predict_kernel<false, " GSE_WIDE ">", gsukf_predict_kernel<false, kWideBank>, grid, block, st, mu, P, none, N, status, a);
}

static int gsukf_update(UserModel* model, float* mu, float* P, float* w, int B, int64_t N, const double* u, const double* z,
                        const float* w_sigma, const gse_mixture* measurement_pdf, int32_t* status, void* workspace,
                        void* stream, uint32_t where = 0) {
    if (int rc = check_gsukf("gse_gsukf_update", B, N)) return rc;
    GSE_REQUIRE(mu && P && w && u && z && w_sigma && measurement_pdf, "gse_gsukf_update: null argument");
    if (int rc = check_mixture("gse_gsukf_update", measurement_pdf, GSE_NY)) return rc;
    GsukfUpdateArgs a;
    pack_bank_io(a.io, a.u_dev, a.z_dev, B, u, z, where);
    pack_pdf2(a.pdf, *measurement_pdf);
    pack_sigma_weights(a.sw, w_sigma, 2 * states_of(model) + 1);
    hipStream_t st = (hipStream_t)stream;
    (void)workspace;
    if (lane_per_gaussian(B, N))
        return launch(model, "gse::gsukf_update_lanes_kernel<0>", gsukf_update_lanes_kernel<0>,
                      dim3((unsigned)((N + kLaneBank - 1) / kLaneBank), B), dim3(kLaneBank), st, mu, P, w, N, status, a);
    return launch(model, "gse::gsukf_update_kernel<" GSE_WIDE ">", gsukf_update_kernel<kWideBank>,
                  dim3((unsigned)((N + kWideBank - 1) / kWideBank), B), dim3(kWideBank * kGroup), st, mu, P, w, N, status, a);
}

static int gsukf_predict_update(UserModel* model, float* mu, float* P, float* w, int B, int64_t N, const double* u, double dt,
                                const double* z, const float* w_sigma, const gse_mixture* state_pdf, const gse_rng* rng,
                                const float* noise, const gse_mixture* measurement_pdf, int32_t* status, void* stream,
                                uint32_t where = 0, const float* mu_in = nullptr, const float* P_in = nullptr,
                                const int32_t* anc = nullptr, void* workspace = nullptr) {
    const int slot = (where & GSE_IO_SUMS_SLOT1) ? 1 : 0;
    GSE_REQUIRE(!(where & GSE_IO_SUMS_SLOT1) || workspace, "gse_gsukf_predict_update: GSE_IO_SUMS_SLOT1 belongs to the calls that leave sums");
    where &= ~GSE_IO_SUMS_SLOT1;
    if (int rc = check_gsukf("gse_gsukf_predict_update", B, N)) return rc;
    GSE_REQUIRE(mu && P && w && u && z && w_sigma && measurement_pdf, "gse_gsukf_predict_update: null argument");
    GSE_REQUIRE(!anc || (mu_in && P_in && mu_in != mu && P_in != P),
                "gse_gsukf_predict_update_from: with ancestors the source arrays must be given and differ from the outputs");
    if (!anc) { mu_in = mu; P_in = P; }
    GSE_REQUIRE(noise || (state_pdf && rng), "gse_gsukf_predict_update: need either noise or (state_pdf, rng)");
    if (int rc = check_mixture("gse_gsukf_predict_update", measurement_pdf, GSE_NY)) return rc;
    const int nx = states_of(model);
    GsukfPredictUpdateArgs a;
    pack_bank_io(a.pred.io, a.pred.u_dev, a.pred.z_dev, B, u, z, where);
    pack_sigma_weights(a.pred.sw, w_sigma, 2 * nx + 1);
    a.pred.dt = dt;
    pack_pdf2(a.pdf, *measurement_pdf);
    const dim3 grid((unsigned)((N + kLaneBank - 1) / kLaneBank), B), block(kLaneBank);
    hipStream_t st = (hipStream_t)stream;
    // with a workspace the launch also leaves the quarter sums of the new weights there (gse_gsukf_resample_begin_sums)
    // (... and the quads' prefixes inside their quarters: what gse_gsukf_cycle_lanes resamples from)
    double *quarters = nullptr, *quads = nullptr;
    dim3 grid_sums = grid;
    if (workspace) {
        const Workspace ws = carve_workspace(workspace, B, N);
        quarters = gsukf_quarter_sums(ws, slot);
        quads = gsukf_quad_prefix(ws, slot);
        // every quarter of every tile gets its sum, also the quarters of the last tile that hold no Gaussian (0): the consumers
        // read four per tile
        grid_sums.x = (unsigned)(4 * tiles_of(N));
    }
    if (noise) {
        a.pred.draw = DrawParams{};
        a.pred.rng = RngKey{};
        return launch(model, "gse::gsukf_predict_update_lanes_kernel<true>", gsukf_predict_update_lanes_kernel<true>, grid_sums, block, st,
                      mu, P, w, noise, N, status, mu_in, P_in, anc, quarters, quads, a);
    }
    if (int rc = check_mixture("gse_gsukf_predict_update", state_pdf, nx)) return rc;
    pack_draw(a.pred.draw, *state_pdf);
    a.pred.rng = make_rng_key(*rng, kPurposeGsukf);
    const float* none = nullptr;
    if (a.pred.draw.diagonal)
        return launch(model, "gse::gsukf_predict_update_lanes_kernel<false, 1>", gsukf_predict_update_lanes_kernel<false, 1>, grid_sums,
                      block, st, mu, P, w, none, N, status, mu_in, P_in, anc, quarters, quads, a);
    return launch(model, "gse::gsukf_predict_update_lanes_kernel<false, 0>", gsukf_predict_update_lanes_kernel<false, 0>, grid_sums, block,
                  st, mu, P, w, none, N, status, mu_in, P_in, anc, quarters, quads, a);
}

static int gsukf_cycle(UserModel* model, float* mu, float* P, float* w, int32_t* idx, int B, int64_t N, const double* u,
                       double dt, const double* z, const double* r, const float* w_sigma, const gse_mixture* state_pdf,
                       const gse_rng* rng, const float* noise, const gse_mixture* measurement_pdf, int32_t* status,
                       void* stream, uint32_t where = 0) {
    if (int rc = check_gsukf("gse_gsukf_cycle", B, N)) return rc;
    GSE_REQUIRE(N <= kSmallBank, "gse_gsukf_cycle: at most %d Gaussians (got %lld); use the separate calls", kSmallBank,
                (long long)N);
    GSE_REQUIRE(mu && P && w && u && z && r && w_sigma && measurement_pdf, "gse_gsukf_cycle: null argument");
    GSE_REQUIRE(noise || (state_pdf && rng), "gse_gsukf_cycle: need either noise or (state_pdf, rng)");
    if (int rc = check_mixture("gse_gsukf_cycle", measurement_pdf, GSE_NY)) return rc;
    const int nx = states_of(model);
    GsukfCycleArgs a;
    pack_bank_io(a.pred.io, a.pred.u_dev, a.pred.z_dev, B, u, z, where);
    pack_sigma_weights(a.pred.sw, w_sigma, 2 * nx + 1);
    a.pred.dt = dt;
    pack_pdf2(a.pdf, *measurement_pdf);
    for (int b = 0; b < B; ++b) {
        GSE_REQUIRE(r[b] >= 0.0 && r[b] < 1.0, "gse_gsukf_cycle: r[%d] = %g is outside [0, 1)", b, r[b]);
        a.r[b] = r[b];
    }
    hipStream_t st = (hipStream_t)stream;
    const int n32 = (int)N;
    if (noise) {
        a.pred.draw = DrawParams{};
        a.pred.rng = RngKey{};
        return launch(model, "gse::gsukf_cycle_kernel<true>", gsukf_cycle_kernel<true>, dim3(1, B), dim3(kSmallBlock), st, mu, P, w,
                      noise, idx, n32, status, a);
    }
    if (int rc = check_mixture("gse_gsukf_cycle", state_pdf, nx)) return rc;
    pack_draw(a.pred.draw, *state_pdf);
    a.pred.rng = make_rng_key(*rng, kPurposeGsukf);
    const float* none = nullptr;
    return launch(model, "gse::gsukf_cycle_kernel<false>", gsukf_cycle_kernel<false>, dim3(1, B), dim3(kSmallBlock), st, mu, P, w,
                  none, idx, n32, status, a);
}

static int gsukf_resample(UserModel* model, const float* mu_in, const float* P_in, const float* w, float* mu_out, float* P_out,
                          float* w_out, int32_t* idx, int B, int64_t N, const double* r, uint32_t flags, void* workspace,
                          void* stream) {
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
    // up to 4096 Gaussians per bank: ancestors in one launch, rows in a second (the workspace's cumulative-weight array,
    // unused on this route, holds the ancestors if the caller does not want them)
    int32_t* anc = idx ? idx : reinterpret_cast<int32_t*>(ws.cumsum);
    if (launch_small_resample(w, w_out, anc, B, N, ra, st)) {
        GSE_LAUNCH_CHECK("small_resample_kernel");
        return launch(model, "gse::gsukf_gather_rows_kernel<0>", gsukf_gather_rows_kernel<0>,
                      dim3((unsigned)((N + kLaneBank - 1) / kLaneBank), B), dim3(kLaneBank), st, mu_in, P_in, (const int32_t*)anc,
                      mu_out, P_out, N);
    }
    if (int rc = launch_cumsum(w, N, B, N, flags & ~GSE_RESAMPLE_TILES_VALID, nullptr, ws, ws.cumsum, st)) return rc;
    GsukfPayload pl{mu_in, P_in, mu_out, P_out, N};
    return launch_gsukf_gather(pl, w_out, idx, B, N, ra, ws, st, model);
}

// The first half of a resample: cumulative weights and ANCESTORS (idx), weights reset to 1 / N -- no row is copied; the copy
// rides the next gse_gsukf_predict_update_from, or gse_gsukf_gather_rows does it on its own.
static int gsukf_resample_begin(const float* w, float* w_out, int32_t* idx, int B, int64_t N, const double* r, void* workspace,
                                void* stream, int sums_slot = -1) {
    if (int rc = check_gsukf("gse_gsukf_resample_begin", B, N)) return rc;
    GSE_REQUIRE(w && w_out && idx && r && workspace && w != w_out, "gse_gsukf_resample_begin: null or aliasing argument");
    ResampleArgs ra;
    for (int b = 0; b < B; ++b) {
        GSE_REQUIRE(r[b] >= 0.0 && r[b] < 1.0, "gse_gsukf_resample_begin: r[%d] = %g is outside [0, 1)", b, r[b]);
        ra.r[b] = r[b];
    }
    Workspace ws = carve_workspace(workspace, B, N);
    hipStream_t st = (hipStream_t)stream;
    if (launch_small_resample(w, w_out, idx, B, N, ra, st)) {          // up to 4096 Gaussians per bank: one launch
        GSE_LAUNCH_CHECK("small_resample_kernel");
        return GSE_OK;
    }
    if (launch_ancestors(w, w_out, idx, B, N, ra, ws, st, sums_slot)) {   // up to 2^21: tile sums, then the ancestors
        GSE_LAUNCH_CHECK("ancestors_scatter_kernel");
        return GSE_OK;
    }
    if (int rc = launch_cumsum(w, N, B, N, 0, nullptr, ws, ws.cumsum, st)) return rc;
    return launch_index_gather(w_out, idx, B, N, ra, ws, st);
}

static int gsukf_gather_rows(UserModel* model, const float* mu_in, const float* P_in, const int32_t* idx, float* mu_out,
                             float* P_out, int B, int64_t N, void* stream) {
    if (int rc = check_gsukf("gse_gsukf_gather_rows", B, N)) return rc;
    GSE_REQUIRE(mu_in && P_in && idx && mu_out && P_out && mu_in != mu_out && P_in != P_out,
                "gse_gsukf_gather_rows: null or aliasing argument");
    return launch(model, "gse::gsukf_gather_rows_kernel<0>", gsukf_gather_rows_kernel<0>,
                  dim3((unsigned)((N + kLaneBank - 1) / kLaneBank), B), dim3(kLaneBank), (hipStream_t)stream, mu_in, P_in, idx, mu_out,
                  P_out, N);
}

static int gsukf_moments(UserModel* model, const float* mu, const float* P, const float* w, int B, int64_t N, double* out,
                         void* workspace, void* stream, bool as_particles = false) {
    if (int rc = check_gsukf("gse_gsukf_moments", B, N)) return rc;
    GSE_REQUIRE(mu && P && w && out && workspace, "gse_gsukf_moments: null argument");
    Workspace ws = carve_workspace(workspace, B, N);
    const int nblk = (int)std::min<int64_t>(kMomentBlocks, (N + kBlock - 1) / kBlock);
    hipStream_t st = (hipStream_t)stream;
    const int nx = states_of(model), width = 1 + nx + nx * (nx + 1);        // S0, S1[nx], S2[tri], C[tri]
    if (as_particles) {
        // the particle filter's row (gse_pf_moments): S0, S1, S2 + C, pivot -- 1 + nx + tri + nx doubles in a row of 26
        const int w26 = 1 + nx + nx * (nx + 1) / 2;
        if (int rc = launch(model, "gse::gsukf_moments_kernel<1>", gsukf_moments_kernel<1>, dim3(nblk, B), dim3(kBlock), st, mu, P,
                            w, ws.partials, N))
            return rc;
        launch_moments_finish(ws.partials, nblk, w26, out, 26, B, mu, N * nx, 1, w26, nx, st);
        GSE_LAUNCH_CHECK("moments_finish_kernel");
        return GSE_OK;
    }
    if (int rc = launch(model, "gse::gsukf_moments_kernel<0>", gsukf_moments_kernel<0>, dim3(nblk, B), dim3(kBlock), st, mu, P, w,
                        ws.partials, N))
        return rc;
    // out[b]: the sums, then the pivot (mu of Gaussian 0) at [width, width + nx); rows are GSE_GSUKF_MOMENTS doubles apart
    launch_moments_finish(ws.partials, nblk, width, out, GSE_GSUKF_MOMENTS, B, mu, N * nx, 1, width, nx, st);
    GSE_LAUNCH_CHECK("moments_finish_kernel");
    return GSE_OK;
}

}  // namespace gse

using namespace gse;

extern "C" {

int gse_gsukf_sigma_points(const float* mu, const float* P, float* sigmas, int B, int64_t N, int32_t* status, void* stream) {
    return gsukf_sigma_points(nullptr, mu, P, sigmas, B, N, status, stream);
}
int gse_gsukf_predict(float* mu, float* P, int B, int64_t N, const double* u, double dt, const float* w_sigma,
                      const gse_mixture* state_pdf, const gse_rng* rng, const float* noise, int32_t* status, void* stream) {
    return gsukf_predict(nullptr, mu, P, B, N, u, dt, w_sigma, state_pdf, rng, noise, status, stream);
}
int gse_gsukf_update(float* mu, float* P, float* w, int B, int64_t N, const double* u, const double* z, const float* w_sigma,
                     const gse_mixture* measurement_pdf, int32_t* status, void* workspace, void* stream) {
    return gsukf_update(nullptr, mu, P, w, B, N, u, z, w_sigma, measurement_pdf, status, workspace, stream);
}
int gse_gsukf_predict_update(float* mu, float* P, float* w, int B, int64_t N, const double* u, double dt, const double* z,
                             const float* w_sigma, const gse_mixture* state_pdf, const gse_rng* rng, const float* noise,
                             const gse_mixture* measurement_pdf, int32_t* status, void* stream) {
    return gsukf_predict_update(nullptr, mu, P, w, B, N, u, dt, z, w_sigma, state_pdf, rng, noise, measurement_pdf, status, stream);
}
int gse_gsukf_cycle(float* mu, float* P, float* w, int32_t* idx, int B, int64_t N, const double* u, double dt, const double* z,
                    const double* r, const float* w_sigma, const gse_mixture* state_pdf, const gse_rng* rng,
                    const float* noise, const gse_mixture* measurement_pdf, int32_t* status, void* stream) {
    return gsukf_cycle(nullptr, mu, P, w, idx, B, N, u, dt, z, r, w_sigma, state_pdf, rng, noise, measurement_pdf, status, stream);
}
int gse_gsukf_resample(const float* mu_in, const float* P_in, const float* w, float* mu_out, float* P_out, float* w_out,
                       int32_t* idx, int B, int64_t N, const double* r, uint32_t flags, void* workspace, void* stream) {
    return gsukf_resample(nullptr, mu_in, P_in, w, mu_out, P_out, w_out, idx, B, N, r, flags, workspace, stream);
}
int gse_gsukf_moments(const float* mu, const float* P, const float* w, int B, int64_t N, double* out, void* workspace,
                      void* stream) {
    return gsukf_moments(nullptr, mu, P, w, B, N, out, workspace, stream);
}

// ---- inputs and measurements in DEVICE memory (`where`: GSE_IO_U_DEVICE | GSE_IO_Z_DEVICE says which of u, z are device
//      pointers [B][2] fp64, read when the kernel runs): a closed loop whose controller and plant are kernels too
int gse_gsukf_predict_dev(float* mu, float* P, int B, int64_t N, const double* u, double dt, const float* w_sigma,
                          const gse_mixture* state_pdf, const gse_rng* rng, const float* noise, int32_t* status, uint32_t where,
                          void* stream) {
    return gsukf_predict(nullptr, mu, P, B, N, u, dt, w_sigma, state_pdf, rng, noise, status, stream, where);
}
int gse_gsukf_update_dev(float* mu, float* P, float* w, int B, int64_t N, const double* u, const double* z, const float* w_sigma,
                         const gse_mixture* measurement_pdf, int32_t* status, void* workspace, uint32_t where, void* stream) {
    return gsukf_update(nullptr, mu, P, w, B, N, u, z, w_sigma, measurement_pdf, status, workspace, stream, where);
}
int gse_gsukf_predict_update_dev(float* mu, float* P, float* w, int B, int64_t N, const double* u, double dt, const double* z,
                                 const float* w_sigma, const gse_mixture* state_pdf, const gse_rng* rng, const float* noise,
                                 const gse_mixture* measurement_pdf, int32_t* status, uint32_t where, void* stream) {
    return gsukf_predict_update(nullptr, mu, P, w, B, N, u, dt, z, w_sigma, state_pdf, rng, noise, measurement_pdf, status, stream,
                                where);
}
int gse_gsukf_cycle_dev(float* mu, float* P, float* w, int32_t* idx, int B, int64_t N, const double* u, double dt, const double* z,
                        const double* r, const float* w_sigma, const gse_mixture* state_pdf, const gse_rng* rng,
                        const float* noise, const gse_mixture* measurement_pdf, int32_t* status, uint32_t where, void* stream) {
    return gsukf_cycle(nullptr, mu, P, w, idx, B, N, u, dt, z, r, w_sigma, state_pdf, rng, noise, measurement_pdf, status, stream,
                       where);
}
// ---- a resample whose copy is deferred (large banks): ancestors now, rows with the next predict + update
int gse_gsukf_resample_begin(const float* w, float* w_out, int32_t* idx, int B, int64_t N, const double* r, void* workspace,
                             void* stream) {
    return gsukf_resample_begin(w, w_out, idx, B, N, r, workspace, stream);
}
int gse_gsukf_gather_rows(const float* mu_in, const float* P_in, const int32_t* idx, float* mu_out, float* P_out, int B, int64_t N,
                          void* stream) {
    return gsukf_gather_rows(nullptr, mu_in, P_in, idx, mu_out, P_out, B, N, stream);
}
int gse_gsukf_predict_update_from(const float* mu_in, const float* P_in, const int32_t* idx, float* mu, float* P, float* w, int B,
                                  int64_t N, const double* u, double dt, const double* z, const float* w_sigma,
                                  const gse_mixture* state_pdf, const gse_rng* rng, const float* noise,
                                  const gse_mixture* measurement_pdf, int32_t* status, uint32_t where, void* stream) {
    return gsukf_predict_update(nullptr, mu, P, w, B, N, u, dt, z, w_sigma, state_pdf, rng, noise, measurement_pdf, status, stream,
                                where, mu_in, P_in, idx);
}
// ... and the pair that saves the tile-sum launch of a large bank's resample: the predict + update launch leaves the quarter
// sums of its new weights in the workspace, the resample that DIRECTLY follows (same weights, same workspace) takes them
int gse_gsukf_predict_update_sums(const float* mu_in, const float* P_in, const int32_t* idx, float* mu, float* P, float* w, int B,
                                  int64_t N, const double* u, double dt, const double* z, const float* w_sigma,
                                  const gse_mixture* state_pdf, const gse_rng* rng, const float* noise,
                                  const gse_mixture* measurement_pdf, int32_t* status, void* workspace, uint32_t where,
                                  void* stream) {
    GSE_REQUIRE(workspace && aligned16(workspace), "gse_gsukf_predict_update_sums: needs the 16-byte aligned device workspace");
    return gsukf_predict_update(nullptr, mu, P, w, B, N, u, dt, z, w_sigma, state_pdf, rng, noise, measurement_pdf, status, stream,
                                where, mu_in, P_in, idx, workspace);
}
int gse_gsukf_resample_begin_sums(const float* w, float* w_out, int32_t* idx, int B, int64_t N, const double* r, void* workspace,
                                  void* stream) {
    return gsukf_resample_begin(w, w_out, idx, B, N, r, workspace, stream, 0);
}
int gse_gsukf_resample_begin_slot(const float* w, float* w_out, int32_t* idx, int B, int64_t N, const double* r, void* workspace,
                                  int slot, void* stream) {
    GSE_REQUIRE(slot == 0 || slot == 1, "gse_gsukf_resample_begin_slot: slot must be 0 or 1");
    return gsukf_resample_begin(w, w_out, idx, B, N, r, workspace, stream, slot);
}
// Resample + predict + update of a large bank in ONE launch (gse_gsukf_cycle_lanes.h): the resample is the one the sums of
// slot s describe (s = 1 with GSE_IO_SUMS_SLOT1, else 0: left by gse_gsukf_predict_update_sums or by this function, on w_in),
// the new sums go to the other slot.
int gse_gsukf_cycle_lanes(const float* mu_in, const float* P_in, const float* w_in, float* mu, float* P, float* w_out,
                          int32_t* idx, int B, int64_t N, const double* u, double dt, const double* z, const double* r,
                          const float* w_sigma, const gse_mixture* state_pdf, const gse_rng* rng, const float* noise,
                          const gse_mixture* measurement_pdf, int32_t* status, void* workspace, uint32_t where, void* stream) {
    const char* fn = "gse_gsukf_cycle_lanes";
    if (int rc = check_gsukf(fn, B, N)) return rc;
    GSE_REQUIRE(mu_in && P_in && w_in && mu && P && w_out && u && z && r && w_sigma && measurement_pdf && workspace,
                "%s: null argument", fn);
    GSE_REQUIRE(mu != mu_in && P != P_in && w_out != w_in, "%s: the outputs must be other arrays than the inputs", fn);
    GSE_REQUIRE(aligned16(workspace) && aligned16(w_in), "%s: workspace and w_in must be 16-byte aligned", fn);
    const int T = (int)tiles_of(N);
    GSE_REQUIRE(N % 4 == 0 && T <= kCycleLanesMaxT && lane_per_gaussian(B, N),
                "%s: for banks of 65 .. %d Gaussians, N a multiple of 4 (otherwise gse_gsukf_resample_begin_slot + "
                "gse_gsukf_predict_update_sums)", fn, kCycleLanesMaxT * kTile);
    GSE_REQUIRE(noise || (state_pdf && rng), "%s: need either noise or (state_pdf, rng)", fn);
    if (int rc = check_mixture(fn, measurement_pdf, GSE_NY)) return rc;
    ResampleArgs ra;
    for (int b = 0; b < B; ++b) {
        GSE_REQUIRE(r[b] >= 0.0 && r[b] < 1.0, "%s: r[%d] = %g is outside [0, 1)", fn, b, r[b]);
        ra.r[b] = r[b];
    }
    const int slot_in = (where & GSE_IO_SUMS_SLOT1) ? 1 : 0;
    where &= ~GSE_IO_SUMS_SLOT1;
    GsukfPredictUpdateArgs a;
    pack_bank_io(a.pred.io, a.pred.u_dev, a.pred.z_dev, B, u, z, where);
    pack_sigma_weights(a.pred.sw, w_sigma, 2 * kNx + 1);
    a.pred.dt = dt;
    pack_pdf2(a.pdf, *measurement_pdf);
    const Workspace ws = carve_workspace(workspace, B, N);
    hipStream_t st = (hipStream_t)stream;
    int variant;                                            // 0 injected noise, 1 Philox with diagonal factors, 2 Philox
    if (noise) {
        a.pred.draw = DrawParams{};
        a.pred.rng = RngKey{};
        variant = 0;
    } else {
        if (int rc = check_mixture(fn, state_pdf, kNx)) return rc;
        pack_draw(a.pred.draw, *state_pdf);
        a.pred.rng = make_rng_key(*rng, kPurposeGsukf);
        variant = a.pred.draw.diagonal ? 1 : 2;
    }
    launch_gsukf_cycle_lanes(variant, mu_in, P_in, w_in, mu, P, w_out, idx, noise, B, N, status, ws, slot_in, ra, a, st);
    GSE_LAUNCH_CHECK("gsukf_cycle_lanes_kernel");
    return GSE_OK;
}
// gse_gsukf_moments in the particle filter's layout (rows of 26 doubles: S0, S1[5], S2[15] + C[15], pivot[5]): what
// gse_loop_control and the host post-processing of a closed loop take from either estimator
int gse_gsukf_moments26(const float* mu, const float* P, const float* w, int B, int64_t N, double* out, void* workspace,
                        void* stream) {
    return gsukf_moments(nullptr, mu, P, w, B, N, out, workspace, stream, true);
}

// ---- the same calls for a user model compiled at run time (handle from gse_model_create); shapes follow the model's
//      number of states nx: mu [B][N][nx], P [B][N][nx][nx], sigma points [B][N][2 nx + 1][nx], w_sigma [2 nx + 1]
#define GSE_MODEL_ARG(model) \
    GSE_REQUIRE(model, "%s: null model handle", __func__); \
    UserModel* m_ = static_cast<UserModel*>(model)

int gse_gsukf_sigma_points_model(void* model, const float* mu, const float* P, float* sigmas, int B, int64_t N, int32_t* status,
                                 void* stream) {
    GSE_MODEL_ARG(model);
    return gsukf_sigma_points(m_, mu, P, sigmas, B, N, status, stream);
}
int gse_gsukf_predict_model(void* model, float* mu, float* P, int B, int64_t N, const double* u, double dt, const float* w_sigma,
                            const gse_mixture* state_pdf, const gse_rng* rng, const float* noise, int32_t* status,
                            void* stream) {
    GSE_MODEL_ARG(model);
    return gsukf_predict(m_, mu, P, B, N, u, dt, w_sigma, state_pdf, rng, noise, status, stream);
}
int gse_gsukf_update_model(void* model, float* mu, float* P, float* w, int B, int64_t N, const double* u, const double* z,
                           const float* w_sigma, const gse_mixture* measurement_pdf, int32_t* status, void* workspace,
                           void* stream) {
    GSE_MODEL_ARG(model);
    return gsukf_update(m_, mu, P, w, B, N, u, z, w_sigma, measurement_pdf, status, workspace, stream);
}
int gse_gsukf_predict_update_model(void* model, float* mu, float* P, float* w, int B, int64_t N, const double* u, double dt,
                                   const double* z, const float* w_sigma, const gse_mixture* state_pdf, const gse_rng* rng,
                                   const float* noise, const gse_mixture* measurement_pdf, int32_t* status, void* stream) {
    GSE_MODEL_ARG(model);
    return gsukf_predict_update(m_, mu, P, w, B, N, u, dt, z, w_sigma, state_pdf, rng, noise, measurement_pdf, status, stream);
}
int gse_gsukf_cycle_model(void* model, float* mu, float* P, float* w, int32_t* idx, int B, int64_t N, const double* u, double dt,
                          const double* z, const double* r, const float* w_sigma, const gse_mixture* state_pdf,
                          const gse_rng* rng, const float* noise, const gse_mixture* measurement_pdf, int32_t* status,
                          void* stream) {
    GSE_MODEL_ARG(model);
    return gsukf_cycle(m_, mu, P, w, idx, B, N, u, dt, z, r, w_sigma, state_pdf, rng, noise, measurement_pdf, status, stream);
}
int gse_gsukf_resample_model(void* model, const float* mu_in, const float* P_in, const float* w, float* mu_out, float* P_out,
                             float* w_out, int32_t* idx, int B, int64_t N, const double* r, uint32_t flags, void* workspace,
                             void* stream) {
    GSE_MODEL_ARG(model);
    return gsukf_resample(m_, mu_in, P_in, w, mu_out, P_out, w_out, idx, B, N, r, flags, workspace, stream);
}
int gse_gsukf_gather_rows_model(void* model, const float* mu_in, const float* P_in, const int32_t* idx, float* mu_out, float* P_out,
                                int B, int64_t N, void* stream) {
    GSE_MODEL_ARG(model);
    return gsukf_gather_rows(m_, mu_in, P_in, idx, mu_out, P_out, B, N, stream);
}
int gse_gsukf_predict_update_from_model(void* model, const float* mu_in, const float* P_in, const int32_t* idx, float* mu, float* P,
                                        float* w, int B, int64_t N, const double* u, double dt, const double* z,
                                        const float* w_sigma, const gse_mixture* state_pdf, const gse_rng* rng,
                                        const float* noise, const gse_mixture* measurement_pdf, int32_t* status, uint32_t where,
                                        void* stream) {
    GSE_MODEL_ARG(model);
    GSE_REQUIRE(where == 0, "gse_gsukf_predict_update_from_model: inputs in device memory are read by the built-in model only");
    return gsukf_predict_update(m_, mu, P, w, B, N, u, dt, z, w_sigma, state_pdf, rng, noise, measurement_pdf, status, stream, 0,
                                mu_in, P_in, idx);
}
int gse_gsukf_moments_model(void* model, const float* mu, const float* P, const float* w, int B, int64_t N, double* out,
                            void* workspace, void* stream) {
    GSE_MODEL_ARG(model);
    return gsukf_moments(m_, mu, P, w, B, N, out, workspace, stream);
}

// K open-loop cycles of B banks enqueued from C++: filter/gs_ukf.py's deferred stages restated for its three regimes on the entry
// points above (include/gse.h).  What the class's predict / update / resample launch, in their order.
int gse_gsukf_cycles(gse_gsukf_cycles_spec* s, int K, const double* u, const double* z, void* stream) {
    GSE_REQUIRE(s && u && z && K >= 1, "gse_gsukf_cycles: bad argument");
    GSE_REQUIRE(s->B >= 1 && s->B <= GSE_MAX_B && s->N >= 1, "gse_gsukf_cycles: bad B / N");
    GSE_REQUIRE(s->mu[0] && s->P[0] && s->w[0] && s->status && s->w_sigma && s->state_pdf && s->measurement_pdf && s->r,
                "gse_gsukf_cycles: null argument");
    GSE_REQUIRE(s->regime == GSE_GSUKF_REGIME_CYCLE || (s->mu[1] && s->P[1] && s->w[1] && s->workspace),
                "gse_gsukf_cycles: this regime needs the second buffers and the workspace");
    GSE_REQUIRE(s->regime != GSE_GSUKF_REGIME_TWO || s->idx, "gse_gsukf_cycles: the two-launch regime needs idx");
    GSE_REQUIRE(s->regime != GSE_GSUKF_REGIME_CYCLE || s->N <= 64, "gse_gsukf_cycles: GSE_GSUKF_REGIME_CYCLE takes at most 64 Gaussians");
    const int B = s->B;
    const int64_t N = s->N;
    float *mu = s->mu[0], *mu_alt = s->mu[1], *P = s->P[0], *P_alt = s->P[1], *w = s->w[0], *w_alt = s->w[1];
    bool pend_resample = s->pend_resample != 0, pend_gather = s->pend_gather != 0, sums_valid = s->sums_valid != 0;
    int slot = s->sums_slot;
    const double* r_pend = s->r_pending;
    gse_rng rng = s->rng;
    for (int k = 0; k < K; ++k) {
        const double *uk = u + (int64_t)k * B * GSE_NU, *zk = z + (int64_t)k * B * GSE_NY, *rk = s->r + (int64_t)k * B;
        if (s->regime == GSE_GSUKF_REGIME_CYCLE) {
            if (int rc = gse_gsukf_cycle(mu, P, w, nullptr, B, N, uk, s->dt, zk, rk, s->w_sigma, s->state_pdf, &rng, nullptr,
                                         s->measurement_pdf, s->status, stream))
                return rc;
            rng.step += 1;
            continue;
        }
        // predict(u, dt) is deferred; update(u, z) launches predict + update, with whatever the last resample left to it
        if (s->regime == GSE_GSUKF_REGIME_LANES && pend_resample) {
            if (int rc = gse_gsukf_cycle_lanes(mu, P, w, mu_alt, P_alt, w_alt, nullptr, B, N, uk, s->dt, zk, r_pend, s->w_sigma,
                                               s->state_pdf, &rng, nullptr, s->measurement_pdf, s->status, s->workspace,
                                               slot ? GSE_IO_SUMS_SLOT1 : 0u, stream))
                return rc;
            std::swap(mu, mu_alt); std::swap(P, P_alt); std::swap(w, w_alt);
            pend_resample = false;
            sums_valid = true;
            slot = 1 - slot;
        } else {
            const int32_t* idx = pend_gather ? s->idx : nullptr;
            float *out_m = idx ? mu_alt : mu, *out_c = idx ? P_alt : P;
            if (int rc = gse_gsukf_predict_update_sums(mu, P, idx, out_m, out_c, w, B, N, uk, s->dt, zk, s->w_sigma, s->state_pdf,
                                                       &rng, nullptr, s->measurement_pdf, s->status, s->workspace, 0u, stream))
                return rc;
            if (idx) { std::swap(mu, mu_alt); std::swap(P, P_alt); }
            pend_gather = false;
            sums_valid = true;
            slot = 0;
        }
        rng.step += 1;
        // resample(r)
        if (s->regime == GSE_GSUKF_REGIME_LANES) {
            pend_resample = true;                              // nothing to launch: the next cycle's launch finds the ancestors
            r_pend = rk;
        } else {
            if (int rc = gse_gsukf_resample_begin_slot(w, w_alt, s->idx, B, N, rk, s->workspace, slot, stream)) return rc;
            std::swap(w, w_alt);
            sums_valid = false;
            pend_gather = true;
        }
    }
    s->bank_current = (mu == s->mu[0]) ? 0 : 1;
    s->w_current = (w == s->w[0]) ? 0 : 1;
    s->pend_resample = pend_resample ? 1 : 0;
    s->pend_gather = pend_gather ? 1 : 0;
    s->sums_valid = sums_valid ? 1 : 0;
    s->sums_slot = slot;
    s->rng.step = rng.step;
    if (pend_resample)
        for (int b = 0; b < B; ++b) s->r_pending[b] = r_pend[b];
    return GSE_OK;
}

}  // extern "C"
