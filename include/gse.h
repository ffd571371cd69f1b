/*
 * gse.h -- C ABI of libgse_hip.so: the MI355X (gfx950) particle-filter / Gaussian-sum-UKF engine.
 *
 * The reference (AlgorithmicAmoeba/gpu_se) has no FFI layer: its device boundary is the Python class
 * API that sim_base.get_parts selects with gpu=True (sim_base.py:89-112).  Each entry point below
 * replaces the device work behind one method of those classes; the reference site is cited per
 * function.  The Python mirror of the class API (gpu_se_amd/filter, gpu_se_amd/gaussian_sum_dist,
 * gpu_se_amd/sim_base) binds these through ctypes; INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - every function returns 0 on success, a negative gse_status otherwise; gse_last_error() returns a
 *     thread-local message for the last failure on the calling thread;
 *   - array arguments are caller-owned DEVICE pointers (16-byte aligned) unless marked "host"; nothing is
 *     retained after the call returns; scalars and the small per-filter inputs (u, z, r) are HOST values
 *     and travel in the kernel-argument segment -- there is no host->device copy and no synchronisation;
 *   - the last argument is a hipStream_t (as void*); all work is enqueued asynchronously on it;
 *   - B independent filters are processed per call (leading dimension everywhere, 1 <= B <= GSE_MAX_B);
 *   - no function allocates, frees or synchronises: every call is hipGraph-capturable.
 *
 * Layouts
 *   particle filter   x  : [B][5][ld]  fp32, state-major ("SoA"): state k of particle i of filter b at
 *                          x[(b*5 + k)*ld + i];  ld >= N, ld % 4 == 0 (the padding is never interpreted)
 *                     w  : [B][ld]     fp32 weights
 *   GS-UKF            mu : [B][N][5]   fp32,  P : [B][N][5][5] fp32,  w : [B][N] fp32   (row-major "AoS")
 *   row helpers (gse_model_f/g, gse_mixture_pdf/draw): plain row-major [M][Nx] arrays.
 */
#ifndef GSE_H
#define GSE_H

#ifdef __HIPCC_RTC__            /* compiled at run time (hiprtc has no C library headers): the types of this ABI */
typedef int int32_t;
typedef unsigned int uint32_t;
typedef long int64_t;
typedef unsigned long uint64_t;
#else
#include <stdint.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define GSE_VERSION 100          /* 0.1.0 */
#define GSE_NX 5                 /* bioreactor states  Cg, Cx, Cfa, Ce, Ch  (model/BioreactorModel.py:170) */
#define GSE_NU 2                 /* inputs             Fg_in, Fm_in                                            */
#define GSE_NY 2                 /* outputs            Cg*180, Cfa*116     (model/BioreactorModel.py:233) */
#define GSE_NSIGMA 11            /* 2*Nx + 1 sigma points                  (filter/gs_ukf.py:60)          */
#define GSE_MAX_ND 4             /* mixture components per Gaussian sum                                     */
#define GSE_MAX_B 32             /* independent filters per call                                            */
#define GSE_TILE 1024            /* particles per weight tile (one partial sum per tile)                    */

typedef enum gse_status {
    GSE_OK = 0,
    GSE_ERR_ARG = -1,            /* bad argument (null pointer, size, alignment, B, Nd, ...)   */
    GSE_ERR_HIP = -2,            /* a HIP call failed; message carries hipGetErrorString       */
    GSE_ERR_UNSUPPORTED = -3
} gse_status;

/*
 * A Gaussian sum, as MultivariateGaussianSum holds it (gaussian_sum_dist/MultivariateGaussianSum.py:27-37).
 * HOST memory, passed by pointer, read during the call only.
 *   weights/means/consts are the reference's fp32 attributes; inv_cov its fp64 _inverse_covariances;
 *   chol is the lower Cholesky factor of each (fp32) covariance, used by the draw kernels.
 * Arrays are dimensioned for GSE_MAX_ND x GSE_NX; only the leading nd x nx part is read.
 */
typedef struct gse_mixture {
    int32_t nd;                                      /* components, 1..GSE_MAX_ND */
    int32_t nx;                                      /* dimension,  1..GSE_NX     */
    float   weights[GSE_MAX_ND];
    float   consts[GSE_MAX_ND];                      /* (2 pi)^(-nx/2) / sqrt(det cov)                      */
    float   means[GSE_MAX_ND][GSE_NX];
    float   chol[GSE_MAX_ND][GSE_NX][GSE_NX];        /* lower triangular                                     */
    double  inv_cov[GSE_MAX_ND][GSE_NX][GSE_NX];
} gse_mixture;

/* Counter-based RNG stream: Philox4x32-10, key = seed, counter = (element, step, purpose, filter). */
typedef struct gse_rng {
    uint64_t seed;
    uint64_t step;               /* caller increments per call that consumes randomness               */
    uint64_t element_offset;     /* global index of this call's element 0 (sharded ensembles: rank*N) */
    uint32_t filter_offset;      /* global id of this call's filter 0                                 */
    uint32_t reserved;
} gse_rng;

int         gse_version(void);
const char *gse_last_error(void);
/* Wait until everything enqueued on `stream` has finished; spin != 0: by polling the stream (the reference's classes are
 * synchronous at .get() / float(cupy scalar), filter/particle.py:307, :320 -- a caller that times single method calls,
 * results/pf_openloop/pf_run_seq.py:15-130, sees the wait's wake-up latency in every sample). */
int         gse_stream_synchronize(void *stream, int spin);

/* Bytes of device workspace the particle-filter / GS-UKF calls below need for (B, N): per filter the tile
 * sums, their prefix and the moment partials.  The workspace carries state only BETWEEN gse_*_update and a
 * gse_*_resample(..., GSE_RESAMPLE_TILES_VALID) that directly follows it -- plus three small per-filter words the library
 * maintains across calls (weight-scale exponent, sticky status, the ticket counter of gse_pf_merge_begin): the owner
 * ZERO-INITIALISES the workspace once, when it allocates it. */
int64_t gse_workspace_bytes(int B, int64_t N);

/* ---------------------------------------------------------------- particle filter ---------------- */

/* ParallelParticleFilter.predict (filter/particle.py:265-277): x_i += f(x_i, u, dt) (Euler increment of
 * Bioreactor.homeostatic_DEs, fp64 arithmetic, one rounding to fp32), then x_i += a draw from state_pdf
 * (MultivariateGaussianSum.draw, MultivariateGaussianSum.py:65-97; component i.i.d. per particle).
 * u: host [B][2]. */
int gse_pf_predict(float *x, int64_t ld, int B, int64_t N, const double *u, double dt,
                   const gse_mixture *state_pdf, const gse_rng *rng, void *stream);

/* Same, with the noise supplied: noise is [B][5][ld] fp32, laid out like x.  This is the parity mode -- the
 * counterpart of running the reference with DeterministicGaussianSum (DeterministicGaussianSum.py:32-65). */
int gse_pf_predict_injected(float *x, int64_t ld, int B, int64_t N, const double *u, double dt,
                            const float *noise, void *stream);

/* ParallelParticleFilter.update (filter/particle.py:279-294): w_i *= measurement_pdf.pdf(z - g(x_i, u))
 * with the mixture density evaluated in fp64 (MultivariateGaussianSum.py:39-63).  Also leaves the per-tile
 * sums of the new weights in the workspace.  u, z: host [B][2]. */
int gse_pf_update(const float *x, float *w, int64_t ld, int B, int64_t N, const double *u, const double *z,
                  const gse_mixture *measurement_pdf, void *workspace, void *stream);

#define GSE_RESAMPLE_TILES_VALID 1u   /* workspace tile sums were left by the preceding gse_*_update on w */
#define GSE_RESAMPLE_SEPARATE_SCAN 2u /* scan the tile sums in a kernel of its own even when N is small enough
                                         (<= 4096 tiles) for the resample kernel to do it itself (test hook) */
#define GSE_RESAMPLE_PREFIX_FROM_SLOT 4u /* gse_pf_resample_begin only: no cumulative weights -- just the exclusive prefix over
                                         the tile sums a GSE_OP_SCAN_OUT update left in scan slot 0 (1 with _SLOT1), for a
                                         GSE_OP_SCAN_IN | GSE_OP_PREFIX_GLOBAL gather */
#define GSE_RESAMPLE_SLOT1 8u
#define GSE_RESAMPLE_WEIGHT_SCALE 16u /* the weights carry the workspace's binary exponent (GSE_OP_WEIGHT_SCALE below): the island
                                         bookkeeping adds it to the logarithm of their sum */

/* ParallelParticleFilter.resample (filter/particle.py:296-316 with _parallel_resample :223-263):
 * c = cumsum(w) / sum(w) in fp64; ancestor of output j = first k with c_k >= (j + r)/N;
 * x_out[:, j] = x_in[:, ancestor]; w_out[j] = 1/N.  r: host [B], each in [0,1).
 * idx (optional, may be NULL): [B][ld] int32 ancestor indices.  x_out / w_out must not alias x_in / w
 * (a workgroup writes the offspring range of its ancestor tile, which overlaps other tiles' inputs).
 * island (optional, may be NULL): device fp64 [2][B], the bookkeeping of a SHARDED ensemble whose shards resample
 * locally: island[0][b] += log(sum of w of filter b) (the shard's accumulated log-weight, never reset by the
 * library), island[1][b] = that sum itself (what the ranks all-reduce for the global weight normalisation). */
int gse_pf_resample(const float *x_in, const float *w, float *x_out, float *w_out, int32_t *idx,
                    int64_t ld, int B, int64_t N, const double *r, uint32_t flags, double *island,
                    void *workspace, void *stream);

/* The copy of a resample whose ancestors are GIVEN: x_out[:, j] = x_in[:, idx[j]], w_out[j] = 1 / N; idx [B][ld] int32
 * (entries clamped to [0, N)).  For replaying a recorded run of the reference -- the ancestors its own
 * filter/particle.py:296-316 chose -- through the drivers; x_out / w_out separate buffers. */
int gse_pf_gather_index(const float *x_in, const int32_t *idx, float *x_out, float *w_out, int64_t ld, int B, int64_t N,
                        void *stream);

/* First half of a resample only: the normalised cumulative weights (and the weight sum) go to the workspace, where a
 * following gse_pf_step(GSE_OP_GATHER | ...) finds them.  flags / island as for gse_pf_resample. */
int gse_pf_resample_begin(const float *w, int64_t ld, int B, int64_t N, uint32_t flags, double *island,
                          void *workspace, void *stream);

/* One fused pass over the particles that performs any subset, in this order, of
 *   GSE_OP_GATHER   second half of resample: offspring j copies its ancestor (needs gse_pf_resample_begin on the same
 *                   workspace first; r as for gse_pf_resample); the offspring weight is 1/N;
 *   GSE_OP_PREDICT  gse_pf_predict (noise == NULL: Philox draw from state_pdf) or gse_pf_predict_injected (noise given);
 *   GSE_OP_UPDATE   gse_pf_update (leaves the tile sums of the new weights in the workspace).
 * It computes exactly what the separate calls compute, with one launch and one trip of the state through HBM; the
 * steady-state cycle resample -> predict -> update is gse_pf_resample_begin + ONE gse_pf_step of all three stages.
 * With GATHER: x_out, w_out are separate buffers, w_in is not read.  Without: in place (x_out == x_in; w_out may
 * equal w_in).  Arguments of stages that are not requested may be NULL.
 *
 * Single-launch cycle (N <= 4096 * GSE_TILE, Philox draw, 2-component measurement mixture):
 *   GSE_OP_SCAN_OUT  (with UPDATE)  also leave, in workspace slot 0 (or 1 with GSE_OP_SCAN_OUT_SLOT1), the tile sums of the
 *                    new weights and -- per thread of the update stage, i.e. per 4 consecutive particles -- the exclusive
 *                    prefix of its weights inside the tile of 1024: 8 B per FOUR particles (round 1 stored the tile-local
 *                    inclusive scans, 8 B per particle);
 *   GSE_OP_SCAN_IN   (with GATHER)  resample from slot 0 (1 with GSE_OP_SCAN_IN_SLOT1) and w_in instead of from the
 *                    cumulative weights of gse_pf_resample_begin: the key of particle k is re-formed as
 *                    min(tile prefix + (thread prefix + running sum of the thread's weights), next tile prefix) / sum --
 *                    cumsum_kernel's arithmetic and bits -- and the ancestors follow by the inverse map (offspring counts
 *                    per key; per-output look-up for wide ancestor ranges), never by materialising c[].
 * Alternating the slots, update -> resample -> predict -> update -> ... is ONE launch per cycle.  Injected noise (the parity
 * mode) has these modes where a closed loop needs them: the whole cycle and predict + update.  island: as for
 * gse_pf_resample, maintained by the GATHER stage when GSE_OP_SCAN_IN is set. */
#define GSE_OP_GATHER  1u
#define GSE_OP_PREDICT 2u
#define GSE_OP_UPDATE  4u
#define GSE_OP_SCAN_IN 8u
#define GSE_OP_SCAN_OUT 16u
#define GSE_OP_SCAN_IN_SLOT1 32u
#define GSE_OP_SCAN_OUT_SLOT1 64u
#define GSE_OP_PREFIX_GLOBAL 512u /* (with SCAN_IN) the tile prefix comes from gse_pf_resample_begin(..., GSE_RESAMPLE_PREFIX_FROM_SLOT
                                  *  [| GSE_RESAMPLE_SLOT1]) instead of being rebuilt by every workgroup: any N; what the class
                                  *  uses beyond 2^21 particles (2048 tiles) */
#define GSE_OP_U_DEVICE 128u     /* (with PREDICT) u is a DEVICE pointer [B][2] fp64, read when the kernel runs */
#define GSE_OP_Z_DEVICE 256u     /* (with UPDATE)  z is a DEVICE pointer [B][2] fp64                            */
                                 /* (both: with GSE_OP_GATHER only in the whole single-launch cycle              */
                                 /*  GATHER | PREDICT | UPDATE | SCAN_IN | SCAN_OUT)                             */
/* The reference's weight RANGE (filter/particle.py:103, :316 re-create the weights as fp64 after every resample, so
 * ``weights *= pdf`` at :294 survives likelihoods down to 1e-308; fp32 storage flushes below 1e-38 N):
 *   GSE_OP_WEIGHT_SCALE  (with UPDATE: stores fp32(w * pdf * 2^-k) instead of fp32(w * pdf), k = the binary exponent that
 *                    brings the best of 64 sample particles of the filter -- old weight x likelihood after this launch's model
 *                    step -- to about 1 / N: one value per filter and launch, a power of two, so no mantissa bit of any weight
 *                    changes; k = 0 unless that likelihood is below 2^-40, i.e. launches whose weights fp32 holds anyway are
 *                    bit for bit what they are without the flag; an UPDATE without GATHER looks at ALL particles first, see
 *                    GSE_OP_UNIFORM_WEIGHTS) the accumulated exponent
 *                    lives in the workspace at gse_workspace_weight_scale_offset(B, N): [B] doubles, true weight = stored
 *                    weight * 2^scale[b].  The CALLER zeroes it when it sets the weights itself; UPDATE adds its k, GATHER
 *                    resets it (offspring weigh 1 / N); the island bookkeeping includes it.  Everything normalised (resampling,
 *                    offspring counts, S1 / S0) is invariant; the un-normalised sums of gse_pf_moments are those of the STORED
 *                    weights: multiply by 2^scale[b].  The scale's anchor is read from x_in and w_in while other workgroups
 *                    write: an UPDATE then needs w_out != w_in, and PREDICT | UPDATE without GATHER x_out != x_in. */
#define GSE_OP_WEIGHT_SCALE 1024u
/* Mid-size ensembles (N <= 2^18: at most 256 tiles; Philox noise, 2-component measurement mixture):
 *   GSE_OP_CLUSTER   the same stages by FOUR workgroups per tile of 1024 particles, one particle per thread -- below one
 *                    generation of workgroups the cycle is a chain of dependent latencies (11-12 us from 2^14 to 2^18
 *                    particles), and a lane that walks it for one particle instead of four on four times the CUs is through
 *                    sooner.  Same particles, weights and ancestors, bit for bit.  The scan slots such a launch writes
 *                    (SCAN_OUT) and reads (SCAN_IN; a GATHER needs it) are in QUARTER FORMAT -- per tile the four sums of its
 *                    quarters and per 4 particles the exclusive prefix inside the quarter, which the consumer combines as the
 *                    one-workgroup-per-tile kernel combines its four wavefronts -- so a slot is read by the kind of launch that
 *                    wrote it; gse_pf_moments_resampled takes such a slot as  slot | GSE_SLOT_QUARTERS,  and
 *                    gse_pf_slot_to_tiles converts one in place (tile sums + tile-level prefixes) for everything else.
 *                    Device-resident u / z with any stage combination. */
#define GSE_OP_CLUSTER 2048u
/*   GSE_OP_SAMPLES   (with WEIGHT_SCALE, SCAN_IN) the caller's word that x_in is exactly what the launch that filled the slot
 *                    left in its x_out: that launch (WEIGHT_SCALE | UPDATE | SCAN_OUT, N >= 65536 a multiple of 64, not
 *                    CLUSTER) put the scale's 64 sample particles beside the slot, and this one takes them from there instead
 *                    of fetching them from x_in / w_in (64 cache lines per row and a model step on the critical path of every
 *                    workgroup: 3-4 % of the launch).  Speed only; ignored where no samples can have been left. */
#define GSE_OP_SAMPLES 4096u
/*   GSE_OP_UNIFORM_WEIGHTS  (with WEIGHT_SCALE | UPDATE, no GATHER) the caller's word that w_in holds 1 / N everywhere (a
 *                    resample wrote it and nothing has touched it since): the scale's 64 samples then say enough, and the
 *                    pass over all particles that such a launch otherwise runs first for its anchor (weights of any spread:
 *                    scale_exact_kernel, one more kernel inside this call) is left out.  Speed only. */
#define GSE_OP_UNIFORM_WEIGHTS 8192u
#define GSE_SLOT_QUARTERS 2
int gse_pf_slot_to_tiles(int B, int64_t N, int slot, void *workspace, void *stream);
int64_t gse_workspace_weight_scale_offset(int B, int64_t N);   /* byte offset of scale[B] inside the workspace; < 0: bad B / N */
/* Sticky status of every filter, uint32 [B] inside the workspace at this byte offset (round 5).  The kernels only ever RAISE bits,
 * the caller reads them where it synchronises anyway and clears them:
 *   GSE_WSTATUS_SUM_NOT_FINITE  a resample (its gather stage, gse_pf_resample_begin, gse_pf_moments_resampled) met weights whose
 *                    sum is +inf or NaN.  With GSE_OP_WEIGHT_SCALE that is an OVERFLOW of the stored fp32 weights: a particle
 *                    weighed 2^127 more than the launch's anchor (64 sample particles).  The resample itself then does what the
 *                    reference's does with a NaN cumulative sum (every offspring copies particle 0, filter/particle.py:98) -- the
 *                    ensemble collapses -- so the drivers turn this bit into FloatingPointError instead of carrying on. */
#define GSE_WSTATUS_SUM_NOT_FINITE 1u
int64_t gse_workspace_status_offset(int B, int64_t N);
int gse_pf_step(uint32_t ops, const float *x_in, float *x_out, const float *w_in, float *w_out, int32_t *idx,
                int64_t ld, int B, int64_t N, const double *u, double dt, const double *z, const double *r,
                const gse_mixture *state_pdf, const gse_rng *rng, const float *noise,
                const gse_mixture *measurement_pdf, double *island, void *workspace, void *stream);

/* Moments behind point_estimate / point_covariance (filter/particle.py:318-327), fp64 accumulation:
 * out [B][26] device = { S0 = sum w, S1[5] = sum w (x - p), S2[15] = upper triangle (row-major) of
 * sum w (x - p)(x - p)^T, p[5] } with the pivot p = particle 0 (keeps the second moments free of
 * cancellation).  The host forms  w@x = S1 + p S0  and the covariance from these. */
int gse_pf_moments(const float *x, const float *w, int64_t ld, int B, int64_t N, double *out,
                   void *workspace, void *stream);

/* The same moments for the ensemble a systematic resample of (x, w) WOULD produce (every offspring weighing
 * fp32(1 / N)), without moving a particle: particle k enters with its offspring count n_k = F(k) - F(k-1), the counts of
 * the very resample the next GSE_OP_GATHER | GSE_OP_SCAN_IN launch will perform with the same r (same tile sums and tile
 * prefixes: `slot` = the scan slot the preceding GSE_OP_SCAN_OUT update wrote).  This is what lets a closed loop on the
 * device run  update -> resample -> point_estimate -> control -> predict  (sim_base.py:253-262) with the resample's copy
 * deferred into the next cycle's single launch.  Sums are formed in a different order than gse_pf_moments forms them after
 * the copy: equal to fp64 rounding, not to the bit.  N <= 2^21.  r: host, [B]. */
int gse_pf_moments_resampled(const float *x, const float *w, int64_t ld, int B, int64_t N, const double *r, int slot,
                             double *out, void *workspace, void *stream);

/* ---------------------------------------------------------------- Gaussian-sum UKF --------------- */

/* Sigma points (filter/gs_ukf.py:332-346): L = chol(P) (retry with P + 1e-10 I for a Gaussian whose
 * factorisation fails); sigma_0 = mu, sigma_j = mu + L[:, j-1], sigma_{5+j} = mu - L[:, j-1].
 * sigmas: [B][N][11][5] fp32.  status (optional, may be NULL): device int32[1], incremented once per
 * Gaussian whose covariance is still not positive definite after the retry (its sigma points are NaN). */
int gse_gsukf_sigma_points(const float *mu, const float *P, float *sigmas, int B, int64_t N,
                           int32_t *status, void *stream);

/* ParallelGaussianSumUnscentedKalmanFilter.predict (filter/gs_ukf.py:348-367): sigma points through f,
 * plus a noise DRAW per sigma point, mean = average(sigma, weights=w_sigma), P = sum_j w_j d_j d_j^T.
 * w_sigma: host [11] fp32 (gs_ukf.py:62-67).  noise: NULL (Philox draw from state_pdf) or device
 * [B][N][11][5] fp32 (parity mode).  u: host [B][2]. */
int gse_gsukf_predict(float *mu, float *P, int B, int64_t N, const double *u, double dt,
                      const float *w_sigma, const gse_mixture *state_pdf, const gse_rng *rng,
                      const float *noise, int32_t *status, void *stream);

/* ...update (filter/gs_ukf.py:369-407): per Gaussian eta = g(sigma), P_xy, P_yy (no R), K = P_xy pinv(P_yy)
 * (CPU class, gs_ukf.py:132), mu += K (z - eta_mean), P -= K P_yy K^T in fp64; then the global update
 * w *= measurement_pdf.pdf(z - g(mu_new)).  Leaves tile sums of w in the workspace.  u, z: host [B][2]. */
int gse_gsukf_update(float *mu, float *P, float *w, int B, int64_t N, const double *u, const double *z,
                     const float *w_sigma, const gse_mixture *measurement_pdf, int32_t *status,
                     void *workspace, void *stream);
/* gse_gsukf_predict followed by gse_gsukf_update (same u) in ONE launch, one lane per Gaussian: the same bits, without
 * writing (mu, P) out and reading them back in between.  For large banks (the class uses it from 2^16 Gaussians). */
int gse_gsukf_predict_update(float *mu, float *P, float *w, int B, int64_t N, const double *u, double dt,
                             const double *z, const float *w_sigma, const gse_mixture *state_pdf,
                             const gse_rng *rng, const float *noise, const gse_mixture *measurement_pdf,
                             int32_t *status, void *stream);

/* The whole cycle predict -> update -> resample (filter/gs_ukf.py:348-436) of a bank of at most 64 Gaussians in ONE
 * launch of one workgroup per filter (mu, P stay in LDS between the phases): the 64-Gaussian configuration is
 * launch-latency-bound, five launches become one.  Same arithmetic and bits as the three separate calls, in place.
 * r: host [B]; idx (optional): device int32 [B][N] ancestors; other arguments as in the separate calls. */
int gse_gsukf_cycle(float *mu, float *P, float *w, int32_t *idx, int B, int64_t N, const double *u, double dt,
                    const double *z, const double *r, const float *w_sigma, const gse_mixture *state_pdf,
                    const gse_rng *rng, const float *noise, const gse_mixture *measurement_pdf, int32_t *status,
                    void *stream);

/* ...resample (filter/gs_ukf.py:409-436): as gse_pf_resample, gathering (mu, P) rows. */
int gse_gsukf_resample(const float *mu_in, const float *P_in, const float *w, float *mu_out, float *P_out,
                       float *w_out, int32_t *idx, int B, int64_t N, const double *r, uint32_t flags,
                       void *workspace, void *stream);

/* Moments behind point_estimate / point_covariance (filter/gs_ukf.py:438-449): out [B][GSE_GSUKF_MOMENTS] device =
 * { S0, S1[5] = sum w (mu - p), S2[15] = upper triangle of sum w (mu - p)(mu - p)^T,
 *   C[15] = upper triangle of sum w P, p[5] (= mu_0, the pivot), 5 reserved }. */
#define GSE_GSUKF_MOMENTS 46
int gse_gsukf_moments(const float *mu, const float *P, const float *w, int B, int64_t N, double *out,
                      void *workspace, void *stream);

/* Inputs and measurements in DEVICE memory: `where` says which of u, z are device pointers [B][2] fp64 that the kernel
 * reads when it runs (a closed loop whose controller and plant are kernels never brings them to the host: the counterpart
 * of GSE_OP_U_DEVICE / GSE_OP_Z_DEVICE of gse_pf_step).  Otherwise as the calls without `_dev`. */
#define GSE_IO_U_DEVICE 1u
#define GSE_IO_Z_DEVICE 2u
#define GSE_IO_SUMS_SLOT1 4u   /* (calls that leave or take quarter sums) slot 1 of the workspace instead of slot 0 */
int gse_gsukf_predict_dev(float *mu, float *P, int B, int64_t N, const double *u, double dt, const float *w_sigma,
                          const gse_mixture *state_pdf, const gse_rng *rng, const float *noise, int32_t *status,
                          uint32_t where, void *stream);
int gse_gsukf_update_dev(float *mu, float *P, float *w, int B, int64_t N, const double *u, const double *z,
                         const float *w_sigma, const gse_mixture *measurement_pdf, int32_t *status, void *workspace,
                         uint32_t where, void *stream);
int gse_gsukf_predict_update_dev(float *mu, float *P, float *w, int B, int64_t N, const double *u, double dt,
                                 const double *z, const float *w_sigma, const gse_mixture *state_pdf,
                                 const gse_rng *rng, const float *noise, const gse_mixture *measurement_pdf,
                                 int32_t *status, uint32_t where, void *stream);
int gse_gsukf_cycle_dev(float *mu, float *P, float *w, int32_t *idx, int B, int64_t N, const double *u, double dt,
                        const double *z, const double *r, const float *w_sigma, const gse_mixture *state_pdf,
                        const gse_rng *rng, const float *noise, const gse_mixture *measurement_pdf, int32_t *status,
                        uint32_t where, void *stream);
/* A resample of a large bank whose row copy is deferred (what gse_pf_resample_begin is to the particle filter):
 * gse_gsukf_resample_begin computes the cumulative weights and the ANCESTORS idx [B][N] int32 of filter/gs_ukf.py:409-436 and
 * sets w_out to 1 / N; the rows are copied by the next gse_gsukf_predict_update_from (predict + update read every Gaussian
 * from its ancestor's row in mu_in / P_in and write to mu / P -- the 120 bytes per Gaussian of the copy are neither written
 * nor read back), or by gse_gsukf_gather_rows on its own.  idx == NULL in gse_gsukf_predict_update_from: no gather (mu_in, P_in
 * ignored), i.e. gse_gsukf_predict_update_dev.  Same bits as gse_gsukf_resample followed by gse_gsukf_predict_update. */
int gse_gsukf_resample_begin(const float *w, float *w_out, int32_t *idx, int B, int64_t N, const double *r,
                             void *workspace, void *stream);
int gse_gsukf_gather_rows(const float *mu_in, const float *P_in, const int32_t *idx, float *mu_out, float *P_out,
                          int B, int64_t N, void *stream);
int gse_gsukf_predict_update_from(const float *mu_in, const float *P_in, const int32_t *idx, float *mu, float *P,
                                  float *w, int B, int64_t N, const double *u, double dt, const double *z,
                                  const float *w_sigma, const gse_mixture *state_pdf, const gse_rng *rng,
                                  const float *noise, const gse_mixture *measurement_pdf, int32_t *status,
                                  uint32_t where, void *stream);
/* The same launch, which also leaves the quarter sums of the new weights ([B][4 T] doubles: what the four wavefronts of the
 * tile-sum kernel would find per tile of 1024) in the workspace; gse_gsukf_resample_begin_sums -- called DIRECTLY after it on
 * the same weights and workspace -- forms the tile sums from them (same association, same bits) and saves that kernel's
 * launch: 4.7 us of a cycle at 2^18 Gaussians (filter/gs_ukf.py:369-436: update, then resample). */
int gse_gsukf_predict_update_sums(const float *mu_in, const float *P_in, const int32_t *idx, float *mu, float *P, float *w, int B,
                                  int64_t N, const double *u, double dt, const double *z, const float *w_sigma,
                                  const gse_mixture *state_pdf, const gse_rng *rng, const float *noise,
                                  const gse_mixture *measurement_pdf, int32_t *status, void *workspace, uint32_t where,
                                  void *stream);
int gse_gsukf_resample_begin_sums(const float *w, float *w_out, int32_t *idx, int B, int64_t N, const double *r, void *workspace,
                                  void *stream);
/* Two slots for those sums (with every quad's exclusive prefix inside its quarter beside them), so that ONE launch can resample
 * from the sums of the previous update while it leaves those of its own: gse_gsukf_predict_update_sums takes
 * GSE_IO_SUMS_SLOT1 in `where`; gse_gsukf_resample_begin_slot reads the slot it is told; and
 * gse_gsukf_cycle_lanes is resample(r) + predict(u, dt) + update(u, z) of a bank of 65 .. 2^18 Gaussians (N a multiple of 4,
 * the built-in model) in one launch: the ancestors are found by the workgroups of the predict + update kernel themselves
 * (the gather stage of the particle filter's cluster kernels), from w_in and the sums of slot 0 (1 with GSE_IO_SUMS_SLOT1);
 * the bank is read from (mu_in, P_in) at its ancestors' rows, results and new weights go to (mu, P, w_out), the new sums to
 * the OTHER slot, the ancestors to idx if it is not NULL.  Same bits as gse_gsukf_resample_begin_slot +
 * gse_gsukf_predict_update_sums, one launch and 4-5 us less per cycle at 2^18 Gaussians (filter/gs_ukf.py:348-436). */
int gse_gsukf_resample_begin_slot(const float *w, float *w_out, int32_t *idx, int B, int64_t N, const double *r, void *workspace,
                                  int slot, void *stream);
int gse_gsukf_cycle_lanes(const float *mu_in, const float *P_in, const float *w_in, float *mu, float *P, float *w_out,
                          int32_t *idx, int B, int64_t N, const double *u, double dt, const double *z, const double *r /* host [B] */,
                          const float *w_sigma, const gse_mixture *state_pdf, const gse_rng *rng, const float *noise,
                          const gse_mixture *measurement_pdf, int32_t *status, void *workspace, uint32_t where, void *stream);

/* K open-loop cycles predict(u_k, dt) -> update(u_k, z_k) -> resample(r_k) of B banks enqueued from C++ (round 5): the loop
 * results/gsf_openloop/gsf_run_seq.py:16-134 runs over the three methods from Python -- where the host spends 20-50 us per
 * cycle between the calls, more than the device needs up to 2^17 Gaussians -- restated on the entry points above, in the
 * class's three regimes (filter/gs_ukf.py of this build): the same launches with the same arguments in the same order, hence the
 * same banks, weights and ancestors bit for bit.
 *   GSE_GSUKF_REGIME_CYCLE  (N <= 64)      one gse_gsukf_cycle per cycle, in place;
 *   GSE_GSUKF_REGIME_LANES  (gse_gsukf_cycle_lanes' range) one gse_gsukf_cycle_lanes per cycle (the resample of cycle k rides
 *                           the launch of cycle k + 1; the first cycle, with no resample pending: gse_gsukf_predict_update_sums);
 *   GSE_GSUKF_REGIME_TWO    (beyond)       gse_gsukf_predict_update_sums (+ the pending row copy) and
 *                           gse_gsukf_resample_begin_slot per cycle.
 * Philox noise, the built-in model.  u, z: HOST [K][B][2]; spec->r: HOST [K][B].  The trailing fields of the spec are in / out:
 * what is pending on entry (a resample whose ancestors are still to be found, or whose row copy is) and on return. */
#define GSE_GSUKF_REGIME_CYCLE 0
#define GSE_GSUKF_REGIME_LANES 1
#define GSE_GSUKF_REGIME_TWO 2
typedef struct gse_gsukf_cycles_spec {
    int32_t B, regime;
    int64_t N;
    double dt;
    float *mu[2], *P[2], *w[2];                  /* the bank's buffer pairs [B][N][5], [B][N][5][5], [B][N]; [0] holds the bank */
    int32_t *idx;                                /* [B][N] ancestors (regime TWO) */
    void *workspace;                             /* gse_workspace_bytes(B, N) */
    int32_t *status;                             /* [1] counts covariances that could not be factorised */
    const float *w_sigma;                        /* host [11] */
    const gse_mixture *state_pdf, *measurement_pdf;          /* host */
    gse_rng rng;                                 /* step = that of the next predict (in / out) */
    const double *r;                             /* HOST [K][B] */
    /* in / out */
    int32_t bank_current, w_current;             /* out: which of mu[] / P[] and of w[] holds the bank now */
    int32_t pend_resample;                       /* LANES: a resample whose ancestors the next launch finds (r_pending) */
    int32_t pend_gather;                         /* TWO: ancestors in idx, the row copy rides the next launch */
    int32_t sums_valid, sums_slot;               /* the workspace holds the quarter sums of the current weights, in this slot */
    double r_pending[GSE_MAX_B];
} gse_gsukf_cycles_spec;
int gse_gsukf_cycles(gse_gsukf_cycles_spec *spec /* host, in / out */, int K, const double *u, const double *z, void *stream);

/* gse_gsukf_moments in the particle filter's layout -- rows of 26 doubles { S0, S1[5], S2[15] + C[15], p[5] }, so that
 * S2 - S1 e^T - e S1^T + S0 e e^T is the whole covariance (filter/gs_ukf.py:442-449): what gse_loop_control and the host
 * post-processing of a closed loop (sim_base.py:243-309) take from either estimator. */
int gse_gsukf_moments26(const float *mu, const float *P, const float *w, int B, int64_t N, double *out,
                        void *workspace, void *stream);

/* The GS-UKF for a user model compiled at run time (filter/gs_ukf.py:243-275 lets Numba JIT any f / g): the calls above
 * with the handle of gse_model_create in front.  Shapes follow the model's number of states nx (1..5): mu [B][N][nx],
 * P [B][N][nx][nx], sigma points and injected noise [B][N][2 nx + 1][nx], w_sigma host [2 nx + 1]; the moments row is
 * { S0, S1[nx], S2[nx (nx + 1) / 2], C[nx (nx + 1) / 2], p[nx] } in a row of GSE_GSUKF_MOMENTS doubles.  The kernels are
 * those of the built-in model (gse_gsukf_kernels.h) instantiated for the user's f and g on first use. */
int gse_gsukf_sigma_points_model(void *model, const float *mu, const float *P, float *sigmas, int B, int64_t N,
                                 int32_t *status, void *stream);
int gse_gsukf_predict_model(void *model, float *mu, float *P, int B, int64_t N, const double *u, double dt,
                            const float *w_sigma, const gse_mixture *state_pdf, const gse_rng *rng,
                            const float *noise, int32_t *status, void *stream);
int gse_gsukf_update_model(void *model, float *mu, float *P, float *w, int B, int64_t N, const double *u,
                           const double *z, const float *w_sigma, const gse_mixture *measurement_pdf,
                           int32_t *status, void *workspace, void *stream);
int gse_gsukf_predict_update_model(void *model, float *mu, float *P, float *w, int B, int64_t N, const double *u,
                                   double dt, const double *z, const float *w_sigma, const gse_mixture *state_pdf,
                                   const gse_rng *rng, const float *noise, const gse_mixture *measurement_pdf,
                                   int32_t *status, void *stream);
int gse_gsukf_cycle_model(void *model, float *mu, float *P, float *w, int32_t *idx, int B, int64_t N, const double *u,
                          double dt, const double *z, const double *r, const float *w_sigma,
                          const gse_mixture *state_pdf, const gse_rng *rng, const float *noise,
                          const gse_mixture *measurement_pdf, int32_t *status, void *stream);
int gse_gsukf_resample_model(void *model, const float *mu_in, const float *P_in, const float *w, float *mu_out,
                             float *P_out, float *w_out, int32_t *idx, int B, int64_t N, const double *r,
                             uint32_t flags, void *workspace, void *stream);
int gse_gsukf_gather_rows_model(void *model, const float *mu_in, const float *P_in, const int32_t *idx, float *mu_out,
                                float *P_out, int B, int64_t N, void *stream);
int gse_gsukf_predict_update_from_model(void *model, const float *mu_in, const float *P_in, const int32_t *idx,
                                        float *mu, float *P, float *w, int B, int64_t N, const double *u, double dt,
                                        const double *z, const float *w_sigma, const gse_mixture *state_pdf,
                                        const gse_rng *rng, const float *noise, const gse_mixture *measurement_pdf,
                                        int32_t *status, uint32_t where, void *stream);
int gse_gsukf_moments_model(void *model, const float *mu, const float *P, const float *w, int B, int64_t N,
                            double *out, void *workspace, void *stream);

/* ---------------------------------------------------------------- row helpers -------------------- */
/* The pieces the reference's benchmark scripts call one by one (results/pf_openloop/pf_run_seq.py:134-296). */

/* f_vectorize (filter/particle.py:176-191): out[i] = f(x[i], u, dt); x, out: [M][5] fp32; u host [2]. */
int gse_model_f(const float *x, float *out, int64_t M, const double *u, double dt, void *stream);
/* g_vectorize (filter/particle.py:193-208): out[i] = g(x[i], u); x: [M][5], out: [M][2] fp32. */
int gse_model_g(const float *x, float *out, int64_t M, void *stream);
/* The same with the inputs g may depend on (u host [2] or NULL = zeros); the bioreactor's g ignores them. */
int gse_model_g_u(const float *x, float *out, int64_t M, const double *u, void *stream);
/* Which model this library's particle-filter kernels were compiled for: 0 = the bioreactor (the shipped library);
 * otherwise the tag of a user model translated and compiled at run time (gpu_se_amd/usermodel.py -- the counterpart of
 * the reference's Numba JIT of arbitrary f / g, filter/particle.py:176-208). */
uint64_t gse_model_tag(void);

/* ---------------------------------------------------------------- user models at run time --------- */

/* The reference hands arbitrary Python f / g to Numba's JIT (filter/particle.py:176-208).  Here a model is two device
 * functions, `user_increment` and `user_output`, in a generated header `model_header` (gpu_se_amd/usermodel.py translates a
 * restricted form of Python; GSE_USER_NX / GSE_USER_NU inside say how many states and inputs it uses), and the
 * particle-filter kernels are compiled FOR it in the process by hiprtc (opened from `hiprtc_path`, or found by name) the
 * first time a launch needs them -- one template instance at a time, about a second each, no compiler binary and no files.
 * include_csrc / include_abi: the directories of the kernel sources (gse_pf_kernels.h, ...) and of this header.
 * A model handle is not thread-safe against itself beyond its internal lock; destroy it when no launch is in flight. */
int gse_model_create(const char *model_header, int nx, int nu, uint64_t tag, const char *include_csrc,
                     const char *include_abi, const char *hiprtc_path, void **model_out);
int gse_model_destroy(void *model);
/* Models BEYOND the fast path's bounds (round 5): up to 8 states, 4 inputs, 4 outputs.  Their header defines GSE_G_NX / GSE_G_NU /
 * GSE_G_NY, generic_increment() and generic_output(); their kernels are the dimension-generic ones of
 * gpu_se_amd/csrc/gse_generic_kernels.h (both filters: one thread per particle / Gaussian, rows of NX floats), compiled by
 * hiprtc with those constants.  gse_model_launch starts a kernel of a model's translation unit by its C++ name, e.g.
 * "gse::generic_pf_predict_kernel<false>" (compiled on first use): `params` holds one pointer per kernel argument, as
 * hipModuleLaunchKernel takes them.  The Python classes of gpu_se_amd/filter/generic.py drive these kernels the way the
 * reference drives its Numba kernels -- from Python -- with the resample's cumulative weights and ancestors from the entry
 * points above (gse_cumsum_normalised, gse_systematic_index), which do not depend on the model. */
int gse_generic_model_create(const char *model_header, int nx, int nu, int ny, uint64_t tag, const char *include_csrc,
                             const char *include_abi, const char *hiprtc_path, void **model_out);
int gse_model_launch(void *model, const char *name_expression, uint32_t gx, uint32_t gy, uint32_t gz, uint32_t bx,
                     uint32_t by, uint32_t bz, uint32_t dynamic_lds, void **params, void *stream);
/* Compile and load a kernel instance now, e.g. "gse::pf_step_kernel<true, true, false, true, 2, 3>" (otherwise: first use). */
int gse_model_prepare(void *model, const char *name_expression);
/* Compile only -- works without a device; returns the size of the code object in bytes, or a negative status. */
int64_t gse_model_compile_check(void *model, const char *name_expression);
int64_t gse_model_kernels(void *model);            /* instances compiled and loaded so far */
/* gse_pf_step / gse_model_f / gse_model_g_u with the model's kernels: same arguments, same checks, same results as a
 * library built for the model would give. */
int gse_pf_step_model(void *model, uint32_t ops, const float *x_in, float *x_out, const float *w_in, float *w_out,
                      int32_t *idx, int64_t ld, int B, int64_t N, const double *u, double dt, const double *z,
                      const double *r, const gse_mixture *state_pdf, const gse_rng *rng, const float *noise,
                      const gse_mixture *measurement_pdf, double *island, void *workspace, void *stream);
int gse_model_f_model(void *model, const float *x, float *out, int64_t M, const double *u, double dt, void *stream);
int gse_model_g_u_model(void *model, const float *x, float *out, int64_t M, const double *u, void *stream);
/* MultivariateGaussianSum.pdf (MultivariateGaussianSum.py:39-63): e [M][nx] fp32 or fp64 -> out [M] fp64. */
int gse_mixture_pdf_f32(const float *e, int64_t M, const gse_mixture *m, double *out, void *stream);
int gse_mixture_pdf_f64(const double *e, int64_t M, const gse_mixture *m, double *out, void *stream);
/* MultivariateGaussianSum.draw (MultivariateGaussianSum.py:65-97): out [M][nx] fp32 (row-major), or with
 * ld > 0 state-major [nx][ld] (the layout gse_pf_* uses; this is how the filters draw their prior). */
int gse_mixture_draw(float *out, int64_t M, int64_t ld, const gse_mixture *m, const gse_rng *rng,
                     void *stream);
/* The same two for mixtures of MORE than GSE_NX dimensions (round 5: user models of up to 8 states run the dimension-generic
 * kernels, gpu_se_amd/filter/generic.py): the mixture as the generic kernels take it -- normalised cumulative weights for the
 * draw, constant x weight and inverse covariance for the density.  Element i of a draw takes Philox blocks sub = 0, 1, ... of
 * element rng->element_offset + i: pair p = words (2p, 2p + 1) is one Box-Muller pair = normals 2p, 2p + 1, the word behind
 * the pairs selects the component; sample = mean + L n.  e / out rows are [M][n]. */
#define GSE_N_MAX_NX 8
typedef struct gse_mixture_n {
    int32_t nd, n;                              /* components (<= GSE_MAX_ND), dimension (<= GSE_N_MAX_NX) */
    float   cumw[GSE_MAX_ND];                   /* normalised cumulative weights */
    double  cw[GSE_MAX_ND];                     /* (2 pi)^(-n/2) / sqrt(det Sigma_d) (fp32) x weight_d */
    float   mean[GSE_MAX_ND][GSE_N_MAX_NX];
    float   chol[GSE_MAX_ND][GSE_N_MAX_NX][GSE_N_MAX_NX];   /* lower Cholesky factor of Sigma_d */
    double  inv[GSE_MAX_ND][GSE_N_MAX_NX][GSE_N_MAX_NX];    /* Sigma_d^-1 */
} gse_mixture_n;
int gse_mixture_draw_n(float *out, int64_t M, const gse_mixture_n *m /* host */, const gse_rng *rng, void *stream);
int gse_mixture_pdf_n(const double *e, int64_t M, const gse_mixture_n *m /* host */, double *out, void *stream);
/* torch.cumsum + "/= cumsum[-1]" of resample (filter/particle.py:301-304): w [N] fp32 -> c [N] fp64. */
int gse_cumsum_normalised(const float *w, double *c, int64_t N, void *workspace, void *stream);
/* _parallel_resample (filter/particle.py:223-263): idx[i] = first k with c[k] >= (i + r)/N.  idx int64. */
int gse_systematic_index(const double *c, int64_t *idx, int64_t N, double r, void *stream);
/* The raw Philox words the draw kernels consume for element i: out [M][8] uint32 (tests pin these against
 * the published Philox4x32-10 algorithm). */
int gse_rng_raw(uint32_t *out, int64_t M, const gse_rng *rng, uint32_t purpose, void *stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Model-predictive controller (controller.py:9-279): the step that follows point_estimate() in the closed loop
 * (sim_base.py:262-270).  The reference hands a sparse QP in (states, outputs, inputs, input moves) to OSQP on
 * every control tick; only three small vectors of that QP change from tick to tick -- theta = (x0, u_{-1}, bias),
 * controller.py:236-262 -- so the QP is condensed ONCE on the host (gpu_se_amd/controller.py) into an explicit law
 *
 *     c0    = Gc theta + gc0                      the bounded rows at the unconstrained optimum   [n_rows]
 *     nu    = argmax of the dual:  lo <= c0 - S nu <= hi,  nu_i (c_i - bound_i) = 0             (one or two rows: the nine
 *             candidate active sets in closed form, one per lane; more rows: Hildreth sweeps)
 *     du0   = Gu theta + gu0 - Tu nu              first input move                               [n_u]
 *     y1    = Gy theta + gy0 - Ty nu              first predicted output                         [n_y]
 *
 * and this entry point evaluates the law for B rollouts, one wavefront each.  All matrices are row-major fp64 in
 * DEVICE memory, read-only; theta, du0, y1, status are device arrays, so the controller sits between the moments
 * kernel and the next predict without a host round trip.  status[b] = 0 converged, 1 not converged within
 * max_sweeps (infeasible or ill-conditioned bounds: the caller falls back as sim_base.py:271-273 does).
 */
#define GSE_MPC_MAX_ROWS 64      /* bounded rows (finite y / u / du limits) that may be ACTIVE together: one per lane; the most
                                    the fused entry points (gse_pf_moments_resampled_tail, gse_gsukf_closed_loop) take at all */
#define GSE_MPC_MAX_ROWS_TOTAL 2048 /* bounded rows gse_mpc_step / gse_loop_control take (controller.py:180-207 bounds every output
                                    over the whole horizon): beyond 64 the dual is solved over a working set of at most 64 rows,
                                    grown by the most violated row until no row outside it is violated (exact: KKT) */
#define GSE_MPC_MAX_THETA 16     /* Nx + Ni + No of the controller's (sub-selected) linear model     */
#define GSE_MPC_MAX_OUT 8        /* Ni, No                                                            */
typedef struct gse_mpc_law {
    int32_t n_theta, n_u, n_y, n_rows;
    int32_t max_sweeps;
    int32_t report_unsolved;     /* fault injection: != 0 -> every step reports status 1 (what OSQP does when it returns a
                                    non-solved status, controller.py:268-269), so that the caller's fallback branch
                                    (sim_base.py:271-273) can be driven on the device; 0 in normal use */
    double  eps_abs, eps_rel;    /* row residual <= eps_abs + eps_rel * (|c0_i| + |bound_i|) ends the sweeps */
    const double *Gc, *gc0;      /* [n_rows][n_theta], [n_rows]  */
    const double *S;             /* [n_rows][n_rows] symmetric positive semi-definite */
    const double *lo, *hi;       /* [n_rows], -inf / +inf allowed */
    const double *Gu, *gu0, *Tu; /* [n_u][n_theta], [n_u], [n_u][n_rows] */
    const double *Gy, *gy0, *Ty; /* [n_y][n_theta], [n_y], [n_y][n_rows] */
} gse_mpc_law;
int gse_mpc_step(const gse_mpc_law *law, const double *theta /* [B][n_theta] */, int64_t B, double *du0 /* [B][n_u] */,
                 double *y1 /* [B][n_y] */, int32_t *status /* [B] */, void *stream);

/*
 * The closed loop without host round trips (sim_base.py:243-309 for B rollouts): with
 * GSE_OP_U_DEVICE / GSE_OP_Z_DEVICE the estimator reads its inputs from device memory, and the two entry points below
 * are the controller call and the plant of one tick.  Everything is enqueued on the caller's stream; histories are
 * device arrays the caller slices per tick.
 *
 * gse_loop_control = sim_base.py:253-274: point estimate from the estimator's moments (gse_pf_moments layout, [B][26]),
 *   deviation variables of the sub-selected linear model (glue: lin_model.states / inputs / outputs and x_bar, u_bar,
 *   y_bar), bias = y_meas - y_predicted once a step has been solved (known[b]), the MPC law, then
 *   u[b][inputs] <- du0 + u_-1 + u_bar   (or u_fallback + u_bar when the QP is not solved, as the reference's except
 *   branch does), y_pred <- y1 - bias, known <- 1.  u is updated IN PLACE (the other inputs keep their values);
 *   theta_out (optional) receives the parameter vectors (its last n_y entries are the reference's ``biass`` log).
 * gse_plant_step = sim_base.py:278-287 + BioreactorModel.py:95-122: X += DEs(X, u) dt in fp64, X[:4] >= 0, X += a draw
 *   of the state noise, y = outputs(X) = X * molar masses, y_meas = y with a draw of the measurement noise added to
 *   outputs 0 and 2 (Cg, Cfa), z = those two (the estimator's measurement).  rng->step numbers the tick.
 */
typedef struct gse_loop_glue {
    int32_t n_state_full, n_input_full, n_output_full, reserved;    /* 5, 2, 5 for the bioreactor */
    int32_t states[GSE_MPC_MAX_OUT], inputs[GSE_MPC_MAX_OUT], outputs[GSE_MPC_MAX_OUT];
    double  x_bar[GSE_MPC_MAX_OUT], u_bar[GSE_MPC_MAX_OUT], y_bar[GSE_MPC_MAX_OUT], u_fallback[GSE_MPC_MAX_OUT];
} gse_loop_glue;
int gse_loop_control(const gse_mpc_law *law, const gse_loop_glue *glue /* host */, const double *moments /* [B][26] */,
                     const double *y_meas /* [B][n_output_full] */, double *u /* [B][n_input_full], in/out */,
                     double *y_pred /* [B][n_y], in/out */, int32_t *known /* [B], in/out */, int32_t *status /* [B] */,
                     double *theta_out /* [B][n_theta] or NULL */, int64_t B, void *stream);
int gse_plant_step(double *X /* [B][5], in/out */, const double *u /* [B][2] */, double dt, const gse_mixture *state_noise,
                   const gse_mixture *meas_noise, const gse_rng *rng, double *y /* [B][5] */, double *y_meas /* [B][5] */,
                   double *z /* [B][2] */, double *log_x /* [B][5] or NULL */, double *log_u /* [B][2] or NULL */,
                   int64_t B, void *stream);
/* The rest of a control tick of the resident particle-filter loop behind the moments of the pending resample, in two launches
 * instead of five: gse_pf_moments_resampled's partial sums, then ONE kernel that finishes them into out[B][26], makes the
 * controller call (gse_loop_control: reads y_meas, updates u / y_pred / known, writes status) and advances the plant
 * (gse_plant_step, or gse_plant_step_injected when the two draws are given) -- sim_base.py:253-287 for B rollouts.  Same
 * device functions in the same order: the records of the separate calls bit for bit.  Pointers device memory unless host. */
typedef struct gse_loop_tail {
    const gse_mpc_law *law;                      /* host struct of device pointers */
    const gse_loop_glue *glue;                   /* host */
    const double *y_meas;                        /* [B][n_output_full]: the measured outputs the controller sees */
    double *u, *y_pred;                          /* [B][n_input_full], [B][n_y]: in / out */
    int32_t *known, *status;                     /* [B] in / out, [B] out */
    double *X;                                   /* [B][5] plant state, in / out */
    double dt;
    const gse_mixture *state_noise, *meas_noise; /* host; unused with injected draws */
    const gse_rng *rng;                          /* host; rng->step numbers the tick */
    const float *state_draw, *meas_draw;         /* [B][5], [B][2] injected draws, or both NULL */
    double *y, *y_meas_out, *z;                  /* [B][5], [B][5], [B][2]: the plant's records of this tick */
    double *log_x, *log_u;                       /* [B][5], [B][2] or NULL */
} gse_loop_tail;
int gse_pf_moments_resampled_tail(const float *x, const float *w, int64_t ld, int B, int64_t N, const double *r /* host [B] */,
                                  int slot, double *out /* [B][26] */, void *workspace, const gse_loop_tail *tail /* host */,
                                  void *stream);
/* A whole closed loop in ONE launch (sim_base.py:243-309 for B rollouts whose estimator is a Gaussian-sum filter of at most 64
 * Gaussians, results/gsf_closedloop/bioreactor_performance_gsf.py:104-118): one workgroup per rollout runs every tick --
 * predict if due, on a control tick update / resample / estimate / MPC, then the plant -- with the bank in LDS throughout.
 * Rollouts do not interact, so nothing needs a launch boundary; what the launch-per-piece loop spends waiting for the host
 * to enqueue (five launches a tick) is gone.  Same device functions, same Philox elements and steps as gse_gsukf_cycle_dev +
 * gse_gsukf_moments26 + gse_loop_control + gse_plant_step: the records agree bit for bit.
 * All pointers device memory unless marked host; records are [ticks][B][...] and row 0 is the caller's (initial values);
 * schedule[t] bit 0: a predict is due at tick t, bit 1: the controller acts (sim_base.py:250, :255). */
typedef struct gse_loop_spec {
    int32_t B, N, ticks, reserved;
    double dt;
    const float *w_sigma;                        /* host [11] */
    const gse_mixture *state_pdf;                /* host: the filter's state noise */
    const gse_rng *rng;                          /* host: the filter's Philox stream; step = that of its first predict */
    const gse_mixture *measurement_pdf;          /* host */
    const gse_mixture *plant_state_noise, *plant_meas_noise;   /* host */
    const gse_rng *plant_rng;                    /* host: tick t draws with step t */
    const gse_mpc_law *law;                      /* host struct of device pointers (gse_mpc_step) */
    const gse_loop_glue *glue;                   /* host */
    const unsigned char *schedule;               /* [ticks] */
    const double *r;                             /* [updates][B] uniform numbers of the resamples, in [0, 1) */
    float *mu, *P, *w;                           /* the banks [B][N][5], [B][N][5][5], [B][N]: in / out */
    double *X, *u, *z;                           /* [B][5], [B][2], [B][2]: in / out */
    double *y_pred;                              /* [B][n_y] in / out */
    int32_t *known;                              /* [B] in / out */
    double *us, *xs, *ys, *ys_meas, *moments;    /* [ticks][B][2 | 5 | 5 | 5 | 26] */
    int32_t *status;                             /* [ticks][B]: 1 where the MPC's QP was not solved */
    int32_t *chol_status;                        /* [1]: counts covariances that could not be factorised */
    void *args_buffer;                           /* GSE_LOOP_ARGS_BYTES of device memory for the kernel's argument block */
} gse_loop_spec;
#define GSE_LOOP_ARGS_BYTES 4096
int gse_gsukf_closed_loop(const gse_loop_spec *spec /* host */, void *stream);

/* The resident closed loop of the PARTICLE filter, its ticks enqueued from C++ (sim_base.py:243-309 for B rollouts; the
 * Monte-Carlo loops of results/pf_closedloop/bioreactor_performance_pf.py:104-118): per control tick gse_pf_step (the
 * pending resample's copy + predict + update in one launch), gse_pf_moments_resampled_tail (the estimate of the resampled
 * ensemble, the MPC law, the plant), per other tick gse_plant_step + gse_pf_moments -- exactly the calls, arguments and order
 * of a host loop over those entry points that defers resample and predict into the next update's launch, so the records
 * are that loop's bit for bit; what is saved is the host's time between launches.  Philox noise, the built-in model,
 * N <= 2^21, a 2-component measurement mixture.  Pointers are device memory unless marked host.  On return the spec's
 * trailing fields say where the ensemble is and what is still pending (a resample's copy rides the NEXT launch). */
#define GSE_PF_LOOP_WEIGHT_SCALE 1u              /* launches carry GSE_OP_WEIGHT_SCALE */
#define GSE_PF_LOOP_CLUSTER 2u                   /* launches carry GSE_OP_CLUSTER (N <= 2^18; the slots are in quarter format) */
typedef struct gse_pf_loop_spec {
    int32_t B, ticks;
    int64_t N, ld;
    double dt;
    uint32_t flags;
    uint32_t samples_ok;                         /* in / out: the pending copy may carry GSE_OP_SAMPLES (the particles are what the launch
                                                    that filled gather_slot wrote); 0 if in doubt */
    float *x[2], *w[2];                          /* the filter's buffer pairs [B][5][ld], [B][ld]; [0] holds the ensemble */
    void *workspace;                             /* gse_workspace_bytes(B, N) */
    const gse_mixture *state_pdf, *measurement_pdf;          /* host */
    gse_rng rng;                                 /* the filter's Philox stream, step = that of its next predict (in / out) */
    const gse_mpc_law *law;                      /* host struct of device pointers */
    const gse_loop_glue *glue;                   /* host */
    double *X, *u, *z, *y_pred;                  /* [B][5], [B][2], [B][2], [B][n_y]: in / out */
    int32_t *known;                              /* [B] in / out */
    const gse_mixture *plant_state_noise, *plant_meas_noise; /* host */
    gse_rng plant_rng;                           /* tick t draws with step t */
    const unsigned char *schedule;               /* HOST [ticks]: bit 0 a predict is due, bit 1 the controller acts */
    const double *r;                             /* HOST [updates][B]: the uniform numbers of the resamples, in [0, 1) */
    double *us, *xs, *ys, *ys_meas, *moments;    /* records [ticks][B][2 | 5 | 5 | 5 | 26]; row 0 is the caller's */
    int32_t *status;                             /* [ticks][B]: 1 where the MPC's QP was not solved */
    /* out */
    int32_t x_current, w_current;                /* which of x[] / w[] holds the ensemble now */
    int32_t pend_gather, pend_predict;           /* a resample's copy / a predict still to be launched */
    int32_t gather_slot, scan_slot;              /* scan slot the pending copy resamples from; slot of the current weights' scans */
    double r_pending[GSE_MAX_B];                 /* the uniform numbers of the pending resample */
    /* in / out (round 5) */
    double *island;                              /* [2][B] shard bookkeeping handed to every gse_pf_step (a SHARDED ensemble's
                                                    cycles: gpu_se_amd/distributed.py), or NULL */
    int32_t w_uniform, reserved2;                /* every weight is 1 / N (a resample's copy ran, no update since): the first
                                                    update without a copy in its launch then passes GSE_OP_UNIFORM_WEIGHTS, as
                                                    the class does; 0 if in doubt */
} gse_pf_loop_spec;
int gse_pf_closed_loop(gse_pf_loop_spec *spec /* host, in / out */, void *stream);
/* K open-loop cycles predict(u_k, dt) -> update(u_k, z_k) -> resample(r_k) of the B filters in `spec` (its filter fields:
 * B, N, ld, dt, flags, x, w, workspace, state_pdf, measurement_pdf, rng, r = HOST [K][B]; on entry pend_gather / gather_slot /
 * r_pending may describe a resample whose copy is still pending -- in tile or quarter format as flags say), enqueued from
 * C++: one gse_pf_step per cycle (the previous cycle's resample copy + predict + update), the loops of
 * results/pf_openloop/pf_run_seq.py:15-130 without the host's time between the calls.  u, z: HOST [K][B][2]. */
int gse_pf_cycles(gse_pf_loop_spec *spec /* host, in / out */, int K, const double *u, const double *z, void *stream);

/* Rows of estimator moments (26 doubles each: gse_pf_moments / gse_gsukf_moments26) -> per row [mean (5), largest singular
 * value of the covariance]: point_estimate() and point_covariance() (filter/particle.py:318-327) for every tick and rollout of a
 * closed loop at once, on the device (sim_base.py:287-296 keeps both per tick). */
int gse_moments_estimates(const double *moments /* [R][26] */, int64_t R, double *out /* [R][6] */, void *stream);
/* Parity mode of gse_plant_step: the two draws of sim_base.py:282 / :285 are INPUTS (fp32, as state_pdf.draw() returns
 * them) -- what lets a closed loop on the device be compared tick by tick with oracle/closed_loop.py. */
int gse_plant_step_injected(double *X /* [B][5], in/out */, const double *u /* [B][2] */, double dt,
                            const float *state_draw /* [B][5] */, const float *meas_draw /* [B][2] */, double *y, double *y_meas,
                            double *z, double *log_x, double *log_u, int64_t B, void *stream);

/* The posterior merge of an ensemble SHARDED over several GPUs, on the device (no reference counterpart -- the
 * reference is single-GPU): between them sits ONE all-reduce of [P][22] doubles (RCCL over xGMI; 1.4 KB at P = 8).
 * gse_merge_row: this shard's moments (gse_pf_moments layout, about its own pivot), its accumulated log-weight log_island[0]
 * and the binary exponent of its stored weights (wscale[0], or NULL) -> row[22] = { log A, S0, sum w (x - p0) (5), upper
 * triangle of sum w (x - p0)(x - p0)^T (15) } about the pivot p0 = common_pivot (host [5]) every rank shares.
 * gse_merge_rows: the rows of all P shards -> the moments of the whole ensemble, 26 doubles in the gse_pf_moments layout
 * (S0 = 1, moments about p0, p0) that gse_loop_control / gse_moments_estimates take; shard p weighs A_p S0_p (log domain);
 * status[0] (optional) = 1 when no shard has weight. */
int gse_merge_row(const double *moments26, const double *log_island, const double *wscale, const double *common_pivot /* host */,
                  double *row /* [22] */, void *stream);
int gse_merge_rows(const double *rows /* [P][22] */, int P, const double *common_pivot /* host */, double *moments26_out,
                   int32_t *status, void *stream);
/* The same merge in TWO launches before the all-reduce and ONE behind it (round 5) -- what an estimate at the end of every
 * cycle costs a shard (the reference's loop asks for one every tick, sim_base.py:262, :287-295):
 * gse_pf_merge_begin: the shard's moments -- with GSE_MERGE_PENDING those of the ensemble the PENDING resample will produce
 *   (spec->flags & GSE_MERGE_PENDING; gse_pf_moments_resampled: offspring-weighted, no particle is copied, the copy keeps riding the next cycle's single
 *   launch; the row's log-weight is island[0] + log(sum of the slot's weights) + wscale ln 2, what the copy's launch will make
 *   of island[0]), otherwise gse_pf_moments of the live weights -- then ONE kernel that finishes the partial sums, writes
 *   this shard's row of rows[P][22] (gse_merge_row's arithmetic) and zeroes the other rows.  With P == 1 that kernel also
 *   combines (gse_merge_rows) and writes merged26 / estimate6, and there is nothing left to do.
 * gse_merge_end (P > 1, behind the all-reduce of rows): gse_merge_rows + the estimate in one launch.
 * estimate6 = [mean (5), largest singular value of the covariance] as gse_moments_estimates gives them; the sixth only with
 *   GSE_MERGE_COV_SIZE (the same cyclic Jacobi sweeps, 25 lanes to a 5 x 5 matrix, stopped when the off-diagonal has
 *   vanished: a serial chain of fp64 divisions and roots that costs more than the rest of the merge -- a loop that records
 *   merged26 per tick gets all its covariance sizes from ONE gse_moments_estimates call at the end, as the resident loops do).
 * status[0]: bit 0 -- no shard has weight (every likelihood underflowed); bit 1 -- a shard's weight sum is inf or NaN (the
 *   stored weights overflowed the launch's power-of-two scale, GSE_OP_WEIGHT_SCALE): such a shard is never silently dropped.
 * One filter per call (B = 1: a shard).  Pointers device memory unless marked host. */
#define GSE_MERGE_STATUS_NO_WEIGHT 1
#define GSE_MERGE_STATUS_OVERFLOW 2
#define GSE_MERGE_PENDING 1
#define GSE_MERGE_COV_SIZE 2
typedef struct gse_merge_spec {
    int32_t P, rank;                 /* shards; this shard's row */
    int32_t flags;                   /* GSE_MERGE_PENDING: moments of the pending resample's outcome (slot, r), else of the live
                                        particles / weights; GSE_MERGE_COV_SIZE: estimate6[5] is wanted (else only the mean) */
    int32_t slot;                    /* scan slot the pending resample reads (| GSE_SLOT_QUARTERS) */
    double r;                        /* the pending resample's uniform number, in [0, 1) */
    const double *island;            /* [2][1] shard bookkeeping of gse_pf_step, or NULL (log-weight 0) */
    const double *wscale;            /* [1] binary exponent of the stored weights (GSE_OP_WEIGHT_SCALE), or NULL */
    double common_pivot[5];          /* the pivot every rank shares */
    double *rows;                    /* [P][22] merge buffer (what the ranks all-reduce) */
    double *moments26;               /* [26] out: this shard's moments (gse_pf_moments layout) */
    double *merged26;                /* [26] out: the whole ensemble's (S0 = 1, about common_pivot) */
    double *estimate6;               /* [6] out, or NULL */
    int32_t *status;                 /* [1] out */
} gse_merge_spec;
int gse_pf_merge_begin(const float *x /* [5][ld] */, const float *w /* [ld] */, int64_t ld, int64_t N,
                       const gse_merge_spec *spec /* host */, void *workspace, void *stream);
int gse_merge_end(const gse_merge_spec *spec /* host */, void *stream);
/* log_island[0] += log(sum of w[0 .. N)) (fp64 sum in a fixed order; sum_out[0], optional, receives the sum): the island
 * bookkeeping of a shard whose resample does not carry it -- the sharded Gaussian-sum filter calls this right before a local
 * resample, the particle filter's gather stage does the same inside its launch (gse_pf_step's island argument). */
int gse_island_absorb(const float *w, int64_t N, double *log_island /* [1] in / out */, double *sum_out /* [1] or NULL */,
                      void *stream);

/* ---------------------------------------------------------------------------------------------------------------
 * Exact systematic resampling of an ensemble SHARDED over several GPUs: the result is, bit for
 * bit, what gse_pf_resample gives on one GPU holding the concatenation of the shards (particle.py:296-316 semantics).
 * Every rank (gpu_se_amd/distributed.py):
 *   1. gse_pf_tile_sums on its weights -> all-gather of the tile sums (8 B per 1024 particles);
 *   2. gse_tile_prefix over the tile sums of ALL shards (same summation order as the single-GPU prefix);
 *   3. gse_pf_shard_cumsum: the normalised cumulative weights of its own particles, c [N_local] fp64;
 *   4. the outputs [J_q, J_q+1) that descend from shard q follow from the prefix at the shard boundaries alone, so
 *      every rank knows all send / receive counts; gse_pf_shard_gather writes those offspring as rows [count][5]
 *      (idx_out: their LOCAL ancestor indices), and one all-to-all over xGMI delivers each rank its n outputs.
 * Every shard but the last must hold a multiple of GSE_TILE particles (tile boundaries of the concatenation).
 */
int gse_pf_tile_sums(const float *w, int64_t ld, int B, int64_t N, double *sums /* [B][T] */, void *stream);
int gse_tile_prefix(const double *sums /* [T] */, int64_t T, double *prefix /* [T + 1], [T] = total */, void *stream);
int gse_pf_shard_cumsum(const float *w, int64_t N_local, const double *prefix_all /* [T_total + 1] */, int64_t tile_offset,
                        int64_t T_total, double *c /* [N_local] */, void *stream);
int gse_pf_shard_gather(const float *x_in /* [5][ld] */, int64_t ld, int64_t N_local, const double *c, int64_t j_base,
                        int64_t count, int64_t n_total, double r, float *rows_out /* [count][5] */,
                        int32_t *idx_out /* [count] or NULL */, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GSE_H */
