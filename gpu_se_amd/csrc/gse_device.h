// gse_device.h -- device-side building blocks shared by the gfx950 kernels of libgse_hip.so.
//
//   * Philox4x32-10 counter-based RNG held entirely in registers (Salmon et al., SC'11) + Box-Muller
//   * the bioreactor model: Euler increment f (fp64 arithmetic on fp32 states) and observation g
//       -- restates Bioreactor.homeostatic_DEs / static_outputs (model/BioreactorModel.py:170-253)
//   * the 2-d measurement mixture density in fp64 (MultivariateGaussianSum.pdf, MultivariateGaussianSum.py:39-63)
//   * 64-wide wavefront reductions / scans (fp64) and their 256-thread block forms
//
// Written for CDNA4 only: wavefront = 64, no dual paths.
#pragma once

#ifndef __HIPCC_RTC__               // (hiprtc brings the device runtime itself and has no C library headers)
#include <hip/hip_runtime.h>
#include <stdint.h>
#else
#ifndef INFINITY
#define INFINITY __builtin_huge_valf()
#endif
#endif

#include "gse.h"

namespace gse {

constexpr int kBlock = 256;               // threads per workgroup (4 wavefronts)
constexpr int kPerThread = 4;             // particles per thread (one dwordx4 per state row)
constexpr int kTile = GSE_TILE;           // particles per workgroup = weight-tile size
static_assert(kBlock * kPerThread == kTile, "tile = block x 4");

// ------------------------------------------------------------------------------------------------
// Per-filter host inputs, carried in the kernel-argument segment (no H2D copy).
// ------------------------------------------------------------------------------------------------
struct FilterIO {
    double u0, u1;      // inputs Fg_in, Fm_in
    double z0, z1;      // measurement (update) -- or {r, unused} for resample
};
struct FilterIOs {
    FilterIO f[GSE_MAX_B];
};

// ------------------------------------------------------------------------------------------------
// Philox4x32-10
// ------------------------------------------------------------------------------------------------
struct RngKey {
    uint32_t k0, k1;            // seed
    uint32_t step_lo;           // counter word 2
    uint32_t hi;                // counter word 3 template: (step_hi & 0xff) << 24 | purpose << 20 | filter0
    uint64_t elem0;             // global index of element 0
};

enum : uint32_t { kPurposePrior = 0, kPurposePredict = 1, kPurposeDraw = 2, kPurposeGsukf = 3, kPurposePlantState = 4,
                  kPurposePlantMeas = 5 };

__host__ __device__ inline RngKey make_rng_key(const gse_rng& r, uint32_t purpose) {
    RngKey k;
    k.k0 = (uint32_t)r.seed;
    k.k1 = (uint32_t)(r.seed >> 32);
    k.step_lo = (uint32_t)r.step;
    k.hi = (((uint32_t)(r.step >> 32) & 0xffu) << 24) | ((purpose & 0xfu) << 20) | (r.filter_offset & 0xffffu);
    k.elem0 = r.element_offset;
    return k;
}

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    // one 32x32 -> 64 multiply per product (v_mad_u64_u32) instead of a mul_hi / mul_lo pair
    const uint64_t p0 = (uint64_t)M0 * (uint64_t)c[0];
    const uint64_t p1 = (uint64_t)M1 * (uint64_t)c[2];
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    // three-way XOR in ONE instruction: v_bitop3_b32 (new on gfx950) with truth table 0x96 = a ^ b ^ c; hipcc does not form
    // it from `a ^ b ^ c` by itself (two v_xor_b32).  The round key is uniform and rides the instruction's scalar operand.
    const uint32_t n0 = __builtin_amdgcn_bitop3_b32(hi1, c[1], k0, 0x96);
    const uint32_t n2 = __builtin_amdgcn_bitop3_b32(hi0, c[3], k1, 0x96);
    c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
}

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += W0;
        k1 += W1;
    }
}

// The 8 random words of element `elem` (filter `fb` of this call): two Philox blocks, sub = 0, 1.
__device__ __forceinline__ void rng_words8(const RngKey& key, uint64_t elem, uint32_t fb, uint32_t (&out)[8]) {
    const uint64_t e = key.elem0 + elem;
    const uint32_t hi = key.hi + fb;                // filter id lives in the low 16 bits
#pragma unroll
    for (uint32_t sub = 0; sub < 2; ++sub) {
        uint32_t c[4] = {(uint32_t)e, (uint32_t)(e >> 32), key.step_lo, hi | (sub << 16)};
        philox4x32_10(c, key.k0, key.k1);
        out[4 * sub + 0] = c[0]; out[4 * sub + 1] = c[1]; out[4 * sub + 2] = c[2]; out[4 * sub + 3] = c[3];
    }
}

// One Philox block of element `elem` (filter `fb` of this call), sub = 0 .. 15.
__device__ __forceinline__ void rng_block_of_elem(const RngKey& key, uint64_t elem, uint32_t fb, uint32_t sub, uint32_t (&c)[4]) {
    const uint64_t e = key.elem0 + elem;
    c[0] = (uint32_t)e;
    c[1] = (uint32_t)(e >> 32);
    c[2] = key.step_lo;
    c[3] = (key.hi + fb) | (sub << 16);
    philox4x32_10(c, key.k0, key.k1);
}

// The particle filter's predict stage draws the noise of FOUR particles from SIX Philox blocks (24 words: 20 for ten
// Box-Muller pairs = 20 normals, 4 for the mixture components) instead of two blocks per particle (8 words of which 7
// are used).  The four particles of a group are the ones a thread owns in the strided mapping of pf_step_kernel:
// local indices  t + 256 p  (p = 0..3) of one tile of 1024, group id = tile * 256 + t  -- a function of the particle's
// LOCAL index alone, so every launch geometry that keeps this mapping draws the same numbers.  Counter:
//   word 0 = group id (28 bits) | bits 32..35 of the shard's element offset << 28,  word 1 = element offset (low 32 bits),
//   word 2 = step,  word 3 = (step_hi & 0xff) << 24 | purpose << 20 | sub << 16 | filter,   sub = 0..5.
// (oracle: gse_oracle.predict_noise.)
__device__ __forceinline__ void rng_block_of_group(const RngKey& key, uint32_t group, uint32_t fb, uint32_t sub,
                                                   uint32_t (&c)[4]) {
    c[0] = group | ((uint32_t)(key.elem0 >> 32) << 28);
    c[1] = (uint32_t)key.elem0;
    c[2] = key.step_lo;
    c[3] = (key.hi + fb) | (sub << 16);
    philox4x32_10(c, key.k0, key.k1);
}

// Box-Muller on two 32-bit words: u1 = (a + 0.5) / 2^32 in (0, 1], angle = b / 2^32 revolutions.
// v_sin_f32 / v_cos_f32 take their argument in revolutions, so no 2*pi multiply is needed.
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
    const float u1 = fmaf((float)a, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
    const float rev = (float)b * 2.3283064365386963e-10f;
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // -2 ln2 log2(u1)
    n0 = r * __builtin_amdgcn_cosf(rev);
    n1 = r * __builtin_amdgcn_sinf(rev);
}

// Parameters of a Gaussian-sum draw, staged in LDS by each workgroup (component index is per lane).
struct DrawParams {
    int32_t nd, nx;
    float cumw[GSE_MAX_ND];                 // normalised cumulative component weights
    float mean[GSE_MAX_ND][GSE_NX];
    float chol[GSE_MAX_ND][15];             // packed lower triangle, row-major: (0,0) (1,0) (1,1) (2,0) ...
    int32_t diagonal;                       // every component's factor is diagonal (pack_draw): see mixture_transform5
    float pad;                              // sizeof = 352, a multiple of 16 (LDS carve alignment)
};
static_assert(sizeof(DrawParams) % 16 == 0, "DrawParams must keep LDS offsets 16-byte aligned");

// mean + L n of the component a selector word picks.  L n row by row, j in order, each step one FMA.  A mixture whose factors
// are all diagonal (independent noise per state: the bioreactor's, sim_base.py:117-140) takes the short way -- the entries
// below the diagonal are zeros, fma(0, n_j, acc) = acc for the finite n_j Box-Muller gives and an acc that is not -0
// (pack_draw turns a mean of -0 into +0), so the five FMAs on the diagonal give the same bits as the fifteen; the flag is the
// same for every lane, the branch is a scalar one.  n: at least 5 normals.
// kDiagonal: 1 / 0 when the caller has looked at the flag already (one branch around several draws instead of one in each: a
// branch ends a basic block, and the values alive across it are copied), -1: look here.
__device__ __forceinline__ bool mixture_is_diagonal(const DrawParams& dp) { return __builtin_amdgcn_readfirstlane(dp.diagonal) != 0; }
template <int kDiagonal = -1>
__device__ __forceinline__ void mixture_transform5(const DrawParams& dp, uint32_t selector, const float* n, float (&out)[5]) {
    const float uc = (float)(selector >> 8) * 5.9604644775390625e-08f;   // [0, 1) with 24 bits
    // (pack_draw leaves cumw[nd - 1] = 1 and 2 beyond: uc < 1 never reaches those, no need to look at nd)
    int comp = 0;
#pragma unroll
    for (int d = 0; d < GSE_MAX_ND - 1; ++d) comp += (uc >= dp.cumw[d]) ? 1 : 0;
    const float* L = dp.chol[comp];
    const float* m = dp.mean[comp];
    if (kDiagonal == 1 || (kDiagonal < 0 && mixture_is_diagonal(dp))) {
#pragma unroll
        for (int k = 0; k < 5; ++k) out[k] = fmaf(L[k * (k + 1) / 2 + k], n[k], m[k]);
        return;
    }
    int at = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        float acc = m[k];
#pragma unroll
        for (int j = 0; j <= k; ++j) acc = fmaf(L[at + j], n[j], acc);
        at += k + 1;
        out[k] = acc;
    }
}

// One draw from the mixture for a 5-d state: words -> component + 5 normals -> mean + L n.
__device__ __forceinline__ void mixture_draw5(const DrawParams& dp, const uint32_t (&wd)[8], float (&out)[5]) {
    float n[6];
    box_muller(wd[0], wd[1], n[0], n[1]);
    box_muller(wd[2], wd[3], n[2], n[3]);
    box_muller(wd[4], wd[5], n[4], n[5]);
    mixture_transform5(dp, wd[7], n, out);
}

// The state noise of the four particles of a group (see rng_block_of_group): blocks 0..4 give normals 4b .. 4b+3 (two
// Box-Muller pairs each), block 5 the four component words; particle p takes normals 5p .. 5p+4 and word p of block 5.
__device__ __forceinline__ void mixture_draw5x4(const DrawParams& dp, const RngKey& key, uint32_t group, uint32_t fb,
                                                float (&out)[4][5]) {
    uint32_t cw[4];
    rng_block_of_group(key, group, fb, 5u, cw);
    float n[20];
#pragma unroll
    for (uint32_t blk = 0; blk < 5; ++blk) {
        uint32_t w[4];
        rng_block_of_group(key, group, fb, blk, w);
        box_muller(w[0], w[1], n[4 * blk + 0], n[4 * blk + 1]);
        box_muller(w[2], w[3], n[4 * blk + 2], n[4 * blk + 3]);
    }
    if (mixture_is_diagonal(dp)) {
#pragma unroll
        for (int p = 0; p < 4; ++p) mixture_transform5<1>(dp, cw[p], n + 5 * p, out[p]);
    } else {
#pragma unroll
        for (int p = 0; p < 4; ++p) mixture_transform5<0>(dp, cw[p], n + 5 * p, out[p]);
    }
}

// ------------------------------------------------------------------------------------------------
// Bioreactor model.  fp64 arithmetic in the reference's operation order; contraction into FMA is
// switched off so that the increment is bit-identical to an IEEE evaluation on the host.
// ------------------------------------------------------------------------------------------------
// RN(x / c) for a constant c with y = RN(1 / c) precomputed (see bioreactor_increment)
__device__ __forceinline__ double div_by_constant(double x, double c, double y) {
    const double q = x * y;
    const double r = fma(-c, q, x);
    return fma(r, y, q);
}

template <class T>       // T = float (particles, sigma points: stored fp32) or double (the plant of the closed loop)
__device__ __forceinline__ void bioreactor_increment(const T (&x)[5], double fg, double fm, double dt,
                                                     double (&dx)[5]) {
#pragma clang fp contract(off)
    const double cg_in = 5000.0 / 180.0;
    const double cg = fmax((double)x[0], 0.0);
    const double cx = fmax((double)x[1], 0.0);
    const double cfa = fmax((double)x[2], 0.0);
    const double ce = fmax((double)x[3], 0.0);
    const double ch = (double)x[4];
    const double f_out = fg + fm;
    const double r_x = 0.0 * cx;
    const double r_h = 280.0 / 180.0 - cg;
    // cg / (0.01 + cg): the correctly rounded quotient without the range scaling of the general division (v_div_scale x 2,
    // v_div_fmas, v_div_fixup) -- for fp32-born states the operands sit far from the exponent limits, where the general
    // sequence reduces to exactly these operations (reciprocal estimate, two Newton steps, quotient, one residual
    // correction), hence the same bits; 8 instructions instead of 12.  (fp64 states -- the plant -- keep the general one.)
    double monod;
    if constexpr (sizeof(T) == 4) {
        const double den = 1e-2 + cg;
        const double y0 = __builtin_amdgcn_rcp(den);
        const double e0 = fma(-den, y0, 1.0);
        const double y1 = fma(y0, e0, y0);
        const double e1 = fma(-den, y1, 1.0);
        const double y2 = fma(y1, e1, y1);
        const double q = cg * y2;
        const double rr = fma(-den, q, cg);
        monod = fma(rr, y2, q);
    } else {
        monod = cg / (1e-2 + cg);
    }
    const double r_fa = (0.25 / 116.0 * cx * 24.6) * monod;
    const double th1_max = (0.4 - 0.25) / 180.0 * cx * 24.6;
    // th1_max / 2000 / (0.28/180): division by a CONSTANT c, correctly rounded without a divide (Markstein): with
    // y = RN(1/c), q = RN(x y), r = x - c q (exact in an FMA), RN(q + r y) = RN(x / c).  Same bits as the IEEE quotient
    // (checked against exact rational arithmetic on 1.4e6 inputs incl. quotients adjacent to rounding midpoints, and
    // by the bit-exact parity tests); 6 FMAs instead of two ~14-instruction fp64 divisions per particle.
    const double th1_scaled = div_by_constant(div_by_constant(th1_max, 2000.0, 1.0 / 2000.0), 0.28 / 180.0,
                                              1.0 / (0.28 / 180.0));
    const double th1_req = th1_max - (th1_scaled * r_h + 0.01 * ch);
    const double th1 = fmin(th1_max, fmax(0.0, th1_req)) * monod;
    const double e_max = 0.025 / 46.0 * cx * 24.6;
    const double e_req = th1_req - th1_max;
    const double r_e = fmin(e_max, fmax(0.0, e_req));
    const double th2_max = (0.1 - 0.025) / 180.0 * cx * 24.6;
    const double th2_req = th1_req - th1_max - r_e;
    const double th2 = fmin(th2_max, fmax(0.0, th2_req));
    const double r_g = -r_fa * (116.0 / 180.0) - th1 - r_e * (46.0 / 180.0) - th2;
    dx[0] = (fg * cg_in - f_out * cg + r_g) * dt;
    dx[1] = r_x * dt;
    dx[2] = (-f_out * cfa + r_fa) * dt;
    dx[3] = (-f_out * ce + r_e) * dt;
    dx[4] = r_h * dt;
}

// x += f(x): one rounding of the fp64 sum to fp32 (what "row_f32 += tuple_f64" does in NumPy).
__device__ __forceinline__ void bioreactor_step(float (&x)[5], double fg, double fm, double dt) {
    double dx[5];
    bioreactor_increment(x, fg, fm, dt, dx);
#pragma unroll
    for (int k = 0; k < 5; ++k) x[k] = (float)((double)x[k] + dx[k]);
}

// g(x) = (Cg*180, Cfa*116) as correctly rounded fp32 products.
__device__ __forceinline__ void bioreactor_output(float cg, float cfa, float& y0, float& y1) {
    y0 = cg * 180.0f;
    y1 = cfa * 116.0f;
}

// ------------------------------------------------------------------------------------------------
// The model the particle-filter kernels are compiled for.  Default: the bioreactor above.  With
// -DGSE_USER_MODEL='"<header>"' (gpu_se_amd/usermodel.py: a restricted Python f / g translated to HIP and the library
// rebuilt at run time -- what Numba's JIT does for the reference, particle.py:176-208) the header provides
//     template <class T> void user_increment(const T (&x)[5], double u0, double u1, double dt, double (&dx)[5]);
//     void user_output(const float (&x)[5], double u0, double u1, float& y0, float& y1);
// with unused states / inputs left at zero.  What the update stage must keep of a particle's new state differs: the
// bioreactor's g reads states 0 and 2 only (two registers per particle), a user's g may read all five.
// ------------------------------------------------------------------------------------------------
#ifdef GSE_USER_MODEL
}  // namespace gse
#include GSE_USER_MODEL
namespace gse {
constexpr int kModelNx = GSE_USER_NX;      // states the model uses (the storage keeps GSE_NX rows, the rest stay zero)
constexpr int kObsKept = GSE_NX;
template <class T>
__device__ __forceinline__ void model_increment(const T (&x)[5], double u0, double u1, double dt, double (&dx)[5]) {
    user_increment(x, u0, u1, dt, dx);
}
__device__ __forceinline__ void model_keep(const float (&s)[5], float (&obs)[kObsKept]) {
#pragma unroll
    for (int k = 0; k < GSE_NX; ++k) obs[k] = s[k];
}
__device__ __forceinline__ void model_output_kept(const float (&obs)[kObsKept], double u0, double u1, float& y0, float& y1) {
    user_output(obs, u0, u1, y0, y1);
}
__device__ __forceinline__ void model_output(const float (&x)[5], double u0, double u1, float& y0, float& y1) {
    user_output(x, u0, u1, y0, y1);
}
#else
constexpr int kModelNx = GSE_NX;
constexpr int kObsKept = 2;
template <class T>
__device__ __forceinline__ void model_increment(const T (&x)[5], double u0, double u1, double dt, double (&dx)[5]) {
    bioreactor_increment(x, u0, u1, dt, dx);
}
__device__ __forceinline__ void model_keep(const float (&s)[5], float (&obs)[kObsKept]) {
    obs[0] = s[0];
    obs[1] = s[2];
}
__device__ __forceinline__ void model_output_kept(const float (&obs)[kObsKept], double, double, float& y0, float& y1) {
    bioreactor_output(obs[0], obs[1], y0, y1);
}
__device__ __forceinline__ void model_output(const float (&x)[5], double, double, float& y0, float& y1) {
    bioreactor_output(x[0], x[2], y0, y1);
}
#endif

// The state after this launch's model step, APPROXIMATELY, for the weight scale's anchor (gse_pf_kernels.h: only the binary
// exponent of a likelihood is taken from it): the bioreactor's increment restated in fp32 with hardware reciprocals -- a
// third of the dependent latency of the fp64 evaluation, and the wavefront that computes the anchor is on its workgroup's
// critical path.  A user model takes its exact step.
#ifndef GSE_USER_MODEL
// the two rates the observed states' increments are made of -- functions of the state alone (no input, no step length)
__device__ __forceinline__ void anchor_rates(const float (&x)[5], float& r_g, float& r_fa) {
    const float cg = fmaxf(x[0], 0.f), cx = fmaxf(x[1], 0.f), ch = x[4];
    const float r_h = (float)(280.0 / 180.0) - cg;
    const float monod = cg * __builtin_amdgcn_rcpf(1e-2f + cg);
    r_fa = (float)(0.25 / 116.0 * 24.6) * cx * monod;
    const float th1_max = (float)((0.4 - 0.25) / 180.0 * 24.6) * cx;
    const float th1_req = th1_max - (th1_max * (float)(1.0 / 2000.0 / (0.28 / 180.0)) * r_h + 0.01f * ch);
    const float th1 = fminf(th1_max, fmaxf(0.f, th1_req)) * monod;
    const float r_e = fminf((float)(0.025 / 46.0 * 24.6) * cx, fmaxf(0.f, th1_req - th1_max));
    const float th2 = fminf((float)((0.1 - 0.025) / 180.0 * 24.6) * cx, fmaxf(0.f, th1_req - th1_max - r_e));
    r_g = -r_fa * (float)(116.0 / 180.0) - th1 - r_e * (float)(46.0 / 180.0) - th2;
}
// ... and the step of the two observed states from them
__device__ __forceinline__ void anchor_advance(float& cg_s, float& cfa_s, float r_g, float r_fa, double u0, double u1, double dt) {
    const float fg = (float)u0, f_out = (float)(u0 + u1), h = (float)dt;
    cg_s += (fg * (float)(5000.0 / 180.0) - f_out * fmaxf(cg_s, 0.f) + r_g) * h;
    cfa_s += (-f_out * fmaxf(cfa_s, 0.f) + r_fa) * h;
}
#endif
__device__ __forceinline__ void model_step_anchor(float (&x)[5], double u0, double u1, double dt) {
#ifdef GSE_USER_MODEL
    double dx[5];
    model_increment(x, u0, u1, dt, dx);
#pragma unroll
    for (int k = 0; k < 5; ++k) x[k] = (float)((double)x[k] + dx[k]);
#else
    float r_g, r_fa;
    anchor_rates(x, r_g, r_fa);
    anchor_advance(x[0], x[2], r_g, r_fa, u0, u1, dt);
#endif
}

// x += f(x): one rounding of the fp64 sum to fp32 (what "row_f32 += tuple_f64" does in NumPy).
__device__ __forceinline__ void model_step(float (&x)[5], double u0, double u1, double dt) {
    double dx[5];
    model_increment(x, u0, u1, dt, dx);
#pragma unroll
    for (int k = 0; k < 5; ++k) x[k] = (float)((double)x[k] + dx[k]);
}

// ------------------------------------------------------------------------------------------------
// Measurement mixture density (Ny = 2), fp64.
// ------------------------------------------------------------------------------------------------
struct Pdf2 {
    int32_t nd, pad;
    double cw[GSE_MAX_ND];                 // (double) fp32(const_d * weight_d)
    double mean[GSE_MAX_ND][2];
    double inv[GSE_MAX_ND][2][2];
    double l2cw[GSE_MAX_ND];               // log2 of cw (host; -inf for a component without weight): likelihood_exponent
};

// kNd > 0 fixes the number of components at compile time (the bioreactor's mixtures have 2): with a run-time count
// the compiler keeps the parameters of all GSE_MAX_ND components live (28 fp64 values), which overflows the scalar
// registers in the fused kernels.  kNd = 0 reads m.nd.
template <int kNd>
__device__ __forceinline__ double mixture_pdf2(const Pdf2& m, double e0, double e1) {
#pragma clang fp contract(off)
    double total = 0.0;
    const int nd = kNd > 0 ? kNd : m.nd;
    for (int d = 0; d < nd; ++d) {
        const double a0 = e0 - m.mean[d][0], a1 = e1 - m.mean[d][1];
        const double t0 = a0 * m.inv[d][0][0] + a1 * m.inv[d][1][0];
        const double t1 = a0 * m.inv[d][0][1] + a1 * m.inv[d][1][1];
        const double q = t0 * a0 + t1 * a1;
        const double term = m.cw[d] * exp(-0.5 * q);
        total = (d == 0) ? term : total + term;
    }
    return total;
}


// The reference keeps its weights in fp64 after the first resample (filter/particle.py:103, :316: numpy.full / cupy.full
// without a dtype), so ``w *= pdf`` survives likelihoods down to 1e-308; fp32 storage flushes everything below 1e-38 N to
// zero -- after a jump of the measurement EVERY particle, the weights sum to zero and the run forks from the reference's.
// GSE_OP_WEIGHT_SCALE restores the range with a launch-uniform power of two: the update stores fp32(w pdf 2^-k).  k comes
// from an ANCHOR every workgroup of the launch evaluates identically: 64 particles spread evenly over the ensemble as the
// launch found it, each scored by the binary exponent of its OLD weight plus that of its likelihood under the new measurement
// after this launch's model step -- estimated without an exponential as max_d (log2 cw_d - q_d / (2 ln 2)) -- and the
// best-scoring one (the sample that matters most to the posterior) fixes k so that ITS new weight is about 1 / N.  (The
// samples travel from launch to launch beside the scan slots, GSE_OP_SAMPLES; a launch that updates without a resample in
// front of it, where 64 samples need not see the heavy particles, takes the maximum over ALL particles from a pass of its
// own: gse_pf_kernels.h.)  Resampling, normalised estimates and the offspring counts are invariant under the scale; what is
// not (the un-normalised sums of point_estimate between update and resample, the island weight of a shard) reads the
// accumulated exponent from the workspace.  While the anchor's likelihood is 2^-kWeightScaleFrom or more nothing is scaled
// (k = 0): every launch whose weights fp32 holds anyway produces the bits it produced before.  fp32 spans 2^-149 .. 2^127
// around the anchor: particles lighter than that flush (harmless -- they are that far below the posterior's mass).
constexpr int kWeightScaleFrom = 40;
constexpr int kWeightScaleMin = -1000;      // (2^1000 cw still finite; fp64 exp() has underflowed long before)
constexpr int kScaleSamples = 64;
template <int kNd>
__device__ __forceinline__ int likelihood_exponent(const Pdf2& m, double e0, double e1) {
#pragma clang fp contract(off)
    double best = -INFINITY;
    const int nd = kNd > 0 ? kNd : m.nd;
    for (int d = 0; d < nd; ++d) {
        const double a0 = e0 - m.mean[d][0], a1 = e1 - m.mean[d][1];
        const double t0 = a0 * m.inv[d][0][0] + a1 * m.inv[d][1][0];
        const double t1 = a0 * m.inv[d][0][1] + a1 * m.inv[d][1][1];
        const double q = t0 * a0 + t1 * a1;
        best = fmax(best, m.l2cw[d] - 0.72134752044448170368 * q);     // log2(cw_d exp(-q / 2))
    }
    if (!(best > -1.0e6)) return -1000000;                               // (also: NaN)
    return (int)floor(fmin(best, 2000.0));
}

// ------------------------------------------------------------------------------------------------
// Wavefront / workgroup collectives (fp64).  All lanes of the workgroup must call these.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// Data-parallel-primitive moves (no LDS traffic): lanes without a source read 0, which is the identity of every scan here
// (sums of non-negative weights, maxima of unsigned marks).
template <int kCtrl, int kRowMask>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, kCtrl, kRowMask, 0xf, true);
}
template <int kCtrl, int kRowMask>
__device__ __forceinline__ double dpp_or_zero(double v) {
    const uint64_t bits = (uint64_t)__double_as_longlong(v);
    const uint32_t lo = dpp_or_zero<kCtrl, kRowMask>((uint32_t)bits), hi = dpp_or_zero<kCtrl, kRowMask>((uint32_t)(bits >> 32));
    return __longlong_as_double((long long)(((uint64_t)hi << 32) | lo));
}
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118;
constexpr int kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143, kDppWaveShr1 = 0x138, kDppWaveShl1 = 0x130;

// Inclusive scan across the 64 lanes of a wavefront: four shift-and-add steps inside each row of 16 lanes, then the row
// totals are handed on (lane 15 of rows 0 and 2 to rows 1 and 3, lane 31 to rows 2 and 3).  Fixed association -- every
// kernel that forms cumulative weights goes through this function, so they all produce the same bits.
__device__ __forceinline__ double wave_inclusive_scan(double v, int /*lane*/) {
    v += dpp_or_zero<kDppRowShr1, 0xf>(v);
    v += dpp_or_zero<kDppRowShr2, 0xf>(v);
    v += dpp_or_zero<kDppRowShr4, 0xf>(v);
    v += dpp_or_zero<kDppRowShr8, 0xf>(v);
    v += dpp_or_zero<kDppRowBcast15, 0xa>(v);
    v += dpp_or_zero<kDppRowBcast31, 0xc>(v);
    return v;
}
// the value of the lane below (lane 0: 0)
__device__ __forceinline__ double wave_shift_up1(double v) { return dpp_or_zero<kDppWaveShr1, 0xf>(v); }
__device__ __forceinline__ uint32_t wave_shift_up1(uint32_t v) { return dpp_or_zero<kDppWaveShr1, 0xf>(v); }
// the value of the lane above (lane 63: 0)
__device__ __forceinline__ uint32_t wave_shift_down1(uint32_t v) { return dpp_or_zero<kDppWaveShl1, 0xf>(v); }

// Sum over the 256-thread workgroup; result valid in every thread.  lds: >= 4 doubles.
__device__ __forceinline__ double block_sum(double v, double* lds) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();                      // protect lds against a previous use
    if (lane == 0) lds[wv] = v;
    __syncthreads();
    return (lds[0] + lds[1]) + (lds[2] + lds[3]);
}

// kCount sums over the 256-thread workgroup at once: wavefront reductions in registers, ONE barrier, then thread c
// adds the four wavefront totals of value c in block_sum's order ((w0 + w1) + (w2 + w3)) and hands it to emit(c, total).
// (block_sum per value costs two barriers each: 42 for the 21 moments of a particle filter.)  lds: >= 4 * kCount doubles.
template <int kCount, class Emit>
__device__ __forceinline__ void block_sums(double (&v)[kCount], double* lds, Emit emit) {
    static_assert(kCount <= 64, "one thread per value adds the wavefront totals");
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < kCount; ++c) v[c] = wave_sum(v[c]);
    __syncthreads();                      // protect lds against a previous use
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < kCount; ++c) lds[wv * kCount + c] = v[c];
    }
    __syncthreads();
    if ((int)threadIdx.x < kCount) {
        const int c = threadIdx.x;
        emit(c, (lds[c] + lds[kCount + c]) + (lds[2 * kCount + c] + lds[3 * kCount + c]));
    }
}

// block_sums with the wavefront totals formed by data-parallel-primitive moves (wave_inclusive_scan's six steps: lane 63 ends up
// with the total) instead of six butterfly exchanges through the LDS crossbar per value: 21 values cost 3.2 us that way in a
// kernel whose whole job is 13 (the resampled-moments pass of a shard's merge, round 5).  Another association of the same
// 64 numbers than block_sums: callers that must agree bit for bit with each other use the same one.
template <int kCount, class Emit>
__device__ __forceinline__ void block_sums_dpp(double (&v)[kCount], double* lds, Emit emit) {
    static_assert(kCount <= 64, "one thread per value adds the wavefront totals");
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < kCount; ++c) v[c] = wave_inclusive_scan(v[c], lane);
    __syncthreads();                      // protect lds against a previous use
    if (lane == 63) {
#pragma unroll
        for (int c = 0; c < kCount; ++c) lds[wv * kCount + c] = v[c];
    }
    __syncthreads();
    if ((int)threadIdx.x < kCount) {
        const int c = threadIdx.x;
        emit(c, (lds[c] + lds[kCount + c]) + (lds[2 * kCount + c] + lds[3 * kCount + c]));
    }
}

// Fixed-order reduction of a filter's per-workgroup moment partials -> out[b][0 .. width); appends pivot_count pivot values.
// 1024 threads: thread (g, c) = (tid / 64, tid % 64) adds the partials q = g, g + 16, ... of column c into 8 interleaved
// accumulators (8 loads in flight), folds them pairwise, and the 16 group totals are folded pairwise after one barrier: a
// fixed order, so the result does not depend on timing; <= 4 dependent load rounds for 512 partials.
constexpr int kFinishGroups = 16;
// pb: THIS filter's partials, rows of `pitch` doubles.
__device__ __forceinline__ void moments_finish_block(const double* __restrict__ pb, int pitch, int nblk, int width,
                                                     double* __restrict__ out, int out_stride,
                                                     const float* __restrict__ pivot_src, int64_t pivot_filter_stride,
                                                     int64_t pivot_elem_stride, int pivot_at, int pivot_count, int b,
                                                     double (*part)[64]) {
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    double a[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    if (c < width) {
        for (int q0 = g; q0 < nblk; q0 += 8 * kFinishGroups) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int q = q0 + j * kFinishGroups;
                if (q < nblk) a[j] += pb[(int64_t)q * pitch + c];
            }
        }
    }
    part[g][c] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (g == 0 && c < width) {
        double t[kFinishGroups];
#pragma unroll
        for (int k = 0; k < kFinishGroups; ++k) t[k] = part[k][c];
#pragma unroll
        for (int span = 1; span < kFinishGroups; span *= 2)
#pragma unroll
            for (int k = 0; k < kFinishGroups; k += 2 * span) t[k] += t[k + span];
        out[(int64_t)b * out_stride + c] = t[0];
    }
    if ((int)threadIdx.x < pivot_count)
        out[(int64_t)b * out_stride + pivot_at + threadIdx.x] =
            (double)pivot_src[(int64_t)b * pivot_filter_stride + threadIdx.x * pivot_elem_stride];
}

// Exclusive scan over the workgroup of one value per thread; also returns the workgroup total.
__device__ __forceinline__ double block_exclusive_scan(double v, double* lds, double& total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const double inc = wave_inclusive_scan(v, lane);
    __syncthreads();
    if (lane == 63) lds[wv] = inc;
    __syncthreads();
    double base = 0.0;
    const double w0 = lds[0], w1 = lds[1], w2 = lds[2], w3 = lds[3];
    if (wv > 0) base = w0;
    if (wv > 1) base += w1;
    if (wv > 2) base += w2;
    total = ((w0 + w1) + w2) + w3;
    return base + wave_shift_up1(inc);
}

// Two independent exclusive scans over the workgroup at once (one value of each per thread) with ONE exchange through
// LDS: per value the arithmetic -- and therefore every bit -- of block_exclusive_scan.  lds: >= 8 doubles.
__device__ __forceinline__ void block_exclusive_scan2(double v0, double v1, double* lds, double& e0, double& e1) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const double inc0 = wave_inclusive_scan(v0, lane), inc1 = wave_inclusive_scan(v1, lane);
    __syncthreads();
    if (lane == 63) {
        lds[wv] = inc0;
        lds[4 + wv] = inc1;
    }
    __syncthreads();
    double b0 = 0.0, b1 = 0.0;
    if (wv > 0) { b0 = lds[0]; b1 = lds[4]; }
    if (wv > 1) { b0 += lds[1]; b1 += lds[5]; }
    if (wv > 2) { b0 += lds[2]; b1 += lds[6]; }
    e0 = b0 + wave_shift_up1(inc0);
    e1 = b1 + wave_shift_up1(inc1);
}

// ------------------------------------------------------------------------------------------------
// Weight rows and searches in the normalised cumulative weights
// ------------------------------------------------------------------------------------------------
using v4f = float __attribute__((ext_vector_type(4)));     // a native 4 x fp32 register tuple (asm operands)

// Sum of a thread's 4 weights (entries at or beyond N count as zero) -- the same expression wherever tile sums
// are formed.
__device__ __forceinline__ double masked_sum4(float w0, float w1, float w2, float w3, int64_t i0, int64_t N) {
    double s = 0.0;
    if (i0 + 0 < N) s = (double)w0;
    if (i0 + 1 < N) s += (double)w1;
    if (i0 + 2 < N) s += (double)w2;
    if (i0 + 3 < N) s += (double)w3;
    return s;
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

// First k in [0, N) with c[k] >= u, found by one wavefront: every round the 64 lanes probe 64 evenly spaced
// elements of the remaining range and a ballot picks the sub-range (4 rounds of one L2 access for N <= 2^24,
// instead of 24 dependent ones).  Returns N - 1 if no element qualifies (cannot happen for u <= c[N-1] = 1).
template <class Value>     // Value: k -> c[k]
__device__ __forceinline__ int64_t wave_search_ge(Value value, int64_t N, double u, int lane) {
    int64_t lo = 0, hi = N;                                   // answer in [lo, hi)
    while (true) {
        const int64_t n = hi - lo;
        const int64_t step = (n + 63) >> 6;
        const int64_t probe = min(lo + (int64_t)(lane + 1) * step, hi) - 1;
        const bool ge = value(probe) >= u;
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

// Two searches of that kind at once in ONE monotone array of N <= 32768 values: lanes 0..31 look for the first k with
// value(k) >= u_lo, lanes 32..63 for the first k with value(k) >= u_hi, 32 probes per round each (two rounds up to 1024
// values, three beyond) -- half the dependent LDS round trips of two 64-ary searches in a row.  Results are uniform over
// the wavefront; N - 1 where no element qualifies.
template <class Value>
__device__ __forceinline__ void wave_search_ge_pair(Value value, int N, double u_lo, double u_hi, int lane, int& k_lo, int& k_hi) {
    const int half = lane >> 5, l = lane & 31;
    const double u = half ? u_hi : u_lo;
    int lo = 0, hi = N;                                           // per half: answer in [lo, hi)
    const int rounds = (N <= 32) ? 1 : ((N <= 1024) ? 2 : ((N <= 32768) ? 3 : 5));
    for (int round = 0; round < rounds; ++round) {
        const int n = hi - lo;
        const int step = (n + 31) >> 5;
        const int probe = min(lo + (l + 1) * step, hi) - 1;
        const bool ge = value(probe) >= u;
        const unsigned long long mask = __ballot(ge);
        const uint32_t mine = half ? (uint32_t)(mask >> 32) : (uint32_t)mask;
        if (mine == 0u) {                                         // nothing qualifies: settle on the last element
            lo = N - 1;
            hi = N;
        } else {
            const int first = __ffs((int)mine) - 1;
            const int nlo = lo + first * step;
            hi = min(lo + (first + 1) * step, hi);
            lo = nlo;
        }
    }
    k_lo = __builtin_amdgcn_readlane(lo, 0);
    k_hi = __builtin_amdgcn_readlane(lo, 32);
}

// Smallest fp64 x >= 0 with fl(x / W) >= u  (W > 0 finite).  x -> fl(x / W) is monotone non-decreasing, so for any
// key:  fl(key / W) >= u  <=>  key >= this value.  Start from u * W and walk by single ulps (a few steps at most).
__device__ __forceinline__ double smallest_with_quotient_ge(double u, double W) {
    if (!(u > 0.0)) return 0.0;                       // fl(0 / W) = 0 >= u
    double x = u * W;
    if (!(x < INFINITY)) return x;
    // The answer lies within an ulp or two of u * W (one rounding there, one in the quotient): u * W itself says which way to
    // look, and its two neighbours on that side are tried side by side -- three divisions, two deep, where walking by single
    // ulps chains four or five -- and the walk remains for whatever they do not settle.
    const long long bits = __double_as_longlong(x);
    if (bits > 2) {
        const bool here = x / W >= u;
        const double a = __longlong_as_double(here ? bits - 1 : bits + 1), b = __longlong_as_double(here ? bits - 2 : bits + 2);
        const bool qa = a / W >= u, qb = b / W >= u;
        if (here) {
            if (!qa) return x;
            if (!qb) return a;
        } else {
            if (qa) return a;
            if (qb) return b;
        }
    }
    // down while the predecessor still satisfies the predicate, then up until the predicate holds
    for (int it = 0; it < 8; ++it) {
        const double y = __longlong_as_double(__double_as_longlong(x) - 1);     // next below (x > 0)
        if (x > 0.0 && y / W >= u) x = y; else break;
    }
    for (int it = 0; it < 8; ++it) {
        if (x / W >= u) break;
        x = __longlong_as_double(__double_as_longlong(x) + 1);                   // next above
    }
    return x;
}

// ------------------------------------------------------------------------------------------------
// Inverse map of the systematic resample: instead of searching, for every output j, the first key c[k] >= u_j
// (particle.py:96-100), every KEY says how many outputs lie at or below it,
//     F(k) = #{ j in [0, N) : u_j <= c[k] },     u_j = fl(fl(j + r) / N),   c[k] = fl(x[k] / W)
// (x = un-normalised cumulative weight) -- the outputs F(k-1) .. F(k) - 1 are exactly the offspring of particle k.  The
// thresholds are an arithmetic progression, so F is arithmetic, not a search: F = floor(x N / W - r) + 1 whenever that real
// number is not within `guard` of an integer; inside the guard band (a few keys per million launches) the count is settled
// with the reference's own comparison on the correctly rounded quotient, so the result is EXACTLY the reference's.
// Error budget: the exact comparison differs from the real-number one by at most N 2^-52 (roundings of x / W and of j + r,
// both <= 1) and the fused evaluation of x (N / W) - r by at most N 2^-51; guard = N 2^-48.
// ------------------------------------------------------------------------------------------------
#ifdef GSE_STAMPS
static __device__ unsigned long long g_slow_hits;      // diagnostic build: guard-band evaluations
#endif
struct OffspringCount {
    double r, Nd, invN, W, NoverW, guard;
    int32_t N;
    bool pow2;
    __device__ __forceinline__ void init(int64_t N_, double r_, double W_) {
        N = (int32_t)N_; r = r_; W = W_; Nd = (double)N_;
        invN = 1.0 / Nd; NoverW = Nd / W_; guard = Nd * 0x1p-48;
        pow2 = (N_ & (N_ - 1)) == 0;
    }
    // the threshold of output j, bit for bit as the kernels and the reference form it
    __device__ __forceinline__ double u_of(int32_t j) const {
        const double jr = (double)j + r;
        return pow2 ? jr * invN : jr / Nd;
    }
    // The arithmetic part: candidate J = floor(x N / W - r) (the count is J + 1) and whether the guard band was hit.  Straight-
    // line code: callers evaluate several keys side by side and settle the rare hits afterwards, in one branch.
    __device__ __forceinline__ int32_t candidate(double x, bool& near) const {
        const double s = fma(x, NoverW, -r);
        const double fl = floor(s);
        const double frac = s - fl;
        near = !(frac >= guard && frac <= 1.0 - guard);                  // (NaN lands here too)
        return (int32_t)fl;                                              // s lies in [-1, N]: fits
    }
    // ... and the exact one, for a key inside the guard band: the reference's comparison on the correctly rounded quotient
    __device__ __forceinline__ int32_t settle(double x, int32_t J) const {
#ifdef GSE_STAMPS
        atomicAdd(&g_slow_hits, 1ull);
#endif
        const double c = x / W;
        J = min(max(J, -1), N - 1);
#pragma unroll
        for (int it = 0; it < 2; ++it)
            if (J + 1 < N && u_of(J + 1) <= c) ++J;
#pragma unroll
        for (int it = 0; it < 2; ++it)
            if (J >= 0 && u_of(J) > c) --J;
        return J;
    }
    __device__ __forceinline__ int32_t operator()(double x) const {
        bool near;
        int32_t J = candidate(x, near);
        if (near) J = settle(x, J);
        return J + 1;
    }
};

// Inclusive max-scan across the 64 lanes of a wavefront (unsigned), same structure as wave_inclusive_scan.
__device__ __forceinline__ uint32_t wave_inclusive_max(uint32_t v) {
    v = max(v, dpp_or_zero<kDppRowShr1, 0xf>(v));
    v = max(v, dpp_or_zero<kDppRowShr2, 0xf>(v));
    v = max(v, dpp_or_zero<kDppRowShr4, 0xf>(v));
    v = max(v, dpp_or_zero<kDppRowShr8, 0xf>(v));
    v = max(v, dpp_or_zero<kDppRowBcast15, 0xa>(v));
    v = max(v, dpp_or_zero<kDppRowBcast31, 0xc>(v));
    return v;
}

#ifndef GSE_SPAN_MAX
#define GSE_SPAN_MAX 2048
#endif
constexpr int kSpanMax = GSE_SPAN_MAX;    // cumulative weights staged in LDS per pass (8 B each)
static_assert(kSpanMax <= 4096 && kSpanMax % 1024 == 0, "the staged span holds whole tiles of 1024 keys");


// Lower bound in the staged span: out = number of keys below thr = first q with seg[q] >= thr.  The caller guarantees
// seg[len - 1] >= thr (the span ends at the ancestor of the tile's last output).  Branch-free bit-by-bit descent:
// the trip count ceil(log2 len) is uniform over the workgroup (a scalar loop), a probe beyond the span is clamped to
// its last key -- which is not below thr, so the step is (correctly) not taken -- and each probe costs five vector
// instructions.  Positions are kept as byte offsets.
__device__ __forceinline__ int span_steps(int len) { return len > 1 ? 32 - __clz(len - 1) : 0; }

// Four searches advanced together (the four outputs of a thread): the dependent LDS reads of one search are the
// latency the other three fill.
__device__ __forceinline__ void span_search4(const double* seg, int len, const double (&thr)[4], int (&out)[4]) {
    const char* base = reinterpret_cast<const char*>(seg);
    const uint32_t last = (uint32_t)(len - 1) * 8u;
    uint32_t pos[4] = {0u, 0u, 0u, 0u};
    for (uint32_t step = (4u << span_steps(len)); step >= 8u; step >>= 1) {      // bytes: 8 * 2^(bits-1) ... 8
        double v[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) v[p] = *reinterpret_cast<const double*>(base + min(pos[p] + step - 8u, last));
#pragma unroll
        for (int p = 0; p < 4; ++p) pos[p] = (v[p] < thr[p]) ? pos[p] + step : pos[p];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) out[p] = (int)(min(pos[p], last) >> 3);
}

// One search (same algorithm).
__device__ __forceinline__ int span_search(const double* seg, int len, double u) {
    const char* base = reinterpret_cast<const char*>(seg);
    const uint32_t last = (uint32_t)(len - 1) * 8u;
    uint32_t pos = 0u;
    for (uint32_t step = (4u << span_steps(len)); step >= 8u; step >>= 1) {
        const double v = *reinterpret_cast<const double*>(base + min(pos + step - 8u, last));
        pos = (v < u) ? pos + step : pos;
    }
    return (int)(min(pos, last) >> 3);
}

}  // namespace gse
