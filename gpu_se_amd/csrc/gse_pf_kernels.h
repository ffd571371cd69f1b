// gse_pf_kernels.h -- the device code of the particle filter (kernels and their helpers), split from its host side
// (gse_pf.hip) so that the SAME text can be compiled a second way: at run time, by hiprtc, for a user model (gse_rtc.hip;
// gpu_se_amd/usermodel.py).  Nothing here may need a host header.
//
// gse_pf.hip -- bootstrap particle filter on gfx950: predict / update / systematic resample / moments.
//
// One workgroup = 256 threads = 4 wavefronts owns one TILE of 1024 consecutive particles of one filter;
// every thread owns 4 consecutive particles, so each state row is read and written with one dwordx4 per
// lane (1 KiB per wave-instruction, fully coalesced) from the state-major layout x[b][k][i].
// Everything here is HBM-bound streaming work by the roofline (140 algorithmic bytes / particle / cycle), and at
// N = 2^20 one kernel is a single generation of workgroups (4 tiles per CU), so every launch costs a fixed
// 3.5-5 us of ramp-up / latency chain whatever it computes.  The design therefore minimises LAUNCHES and passes:
//
//   pf_step_kernel<gather, predict, injected, update, Nd, scan>   ONE fused pass over the particles that does any subset of
//        gather    the offspring's ancestors (resample, second half).  With SCAN_IN (the steady state) by the INVERSE MAP:
//                  the tile prefix of the weights is rebuilt in LDS, the one or two ancestor tiles of the workgroup's 1024
//                  outputs are found there, their keys are re-formed from the weights and the per-thread tile prefixes the
//                  previous update left (8 B per 4 particles) and every key marks the outputs it is the last ancestor
//                  below (LDS atomic max + a max-scan) -- no stored cumulative weights, no search; wide ancestor ranges take
//                  a census of the tiles first, and beyond GSE_SEARCH_FROM listed tiles every output looks its ancestor
//                  up (search_ancestors).  Without SCAN_IN (weights assigned by the caller, N > 2^21): the round-1 path --
//                  64-ary wavefront search + LDS bisection in the materialised cumulative weights of cumsum_kernel.
//                  Then an indexed read of the 5 state rows.
//        predict   ODE step (fp64 in registers) + Philox/Box-Muller mixture draw (no memory: hides the loads), or
//                  injected noise (the parity mode)
//        update    likelihood in fp64, w *= pdf, workgroup sum of the new weights -> one tile sum; with SCAN_OUT also the
//                  per-thread tile prefixes the next launch's gather resamples from
//   cumsum_kernel   tile prefix + fp64 scan of the weights -> normalised cumulative weights (resample, first half, where
//                   the single-launch cycle does not apply)
//
// The steady-state cycle  resample -> predict -> update  is ONE launch: step<gather, predict, update, scan in + out>
// (r 20 + 4..8 + 2..4 B, w 20 + 4 + 2 B per particle; 57 MB per launch at N = 2^20 by the PMC counters).  The classic
// one-method-one-kernel entry points are the same kernel with a single stage enabled.
#pragma once

#include "gse_host.h"
#include "gse_merge_device.h"

namespace gse {

// ------------------------------------------------------------------------------------------------
// the fused step
// ------------------------------------------------------------------------------------------------
struct StepArgs {
    FilterIOs io;          // u, z per filter
    ResampleArgs ra;       // r per filter
    DrawParams draw;       // state noise (predict)
    Pdf2 pdf;              // measurement noise (update)
    RngKey rng;
    double dt;
    const double* u_dev;   // GSE_OP_U_DEVICE / GSE_OP_Z_DEVICE: [B][2] in device memory instead of io (a closed loop
    const double* z_dev;   // whose controller and plant run on the device never brings them to the host)
    double* wscale;        // GSE_OP_WEIGHT_SCALE: [B] accumulated binary exponent of the stored weights (workspace), or null
    // the scale's sample particles, carried from launch to launch beside the scan slots (see scale_samples_load):
    const float* smp_in;   // [B][kScaleSamples][kSampleRow] of the slot this launch resamples from (GSE_OP_SAMPLES), or null
    float* smp_out;        // ... of the slot this launch's update fills, or null
    uint32_t smp_stride;   // N / kScaleSamples: sample l is particle l * smp_stride
    uint32_t smp_magic;    // ceil(2^32 / smp_stride)
    // ... and for launches that update WITHOUT a resample in front (the weights may be anything by then): the best exponent over
    // ALL particles, left by scale_exact_kernel just before this launch ([B], biased by kExactBias; null: the samples decide)
    const uint32_t* k_exact;
    uint32_t* wstatus;     // [B] sticky status words of the filters (workspace; GSE_WSTATUS_*), or null
};
constexpr int32_t kExactBias = 1 << 20;

#ifndef GSE_STEP_MIN_WAVES
#define GSE_STEP_MIN_WAVES 4     // waves per SIMD the register allocator must leave room for (tuning knob)
#endif

#ifndef GSE_STAGGER
#define GSE_STAGGER 1            // gather variants: half of the workgroups draw their noise before the memory phases
#endif
// Where a workgroup of a gather launch draws its noise (tuning knob, see the kernel comment): 0 = before anything,
// 1 = after the tile prefix, 2 = while the span is being staged, 3 = behind the state loads.
// GSE_STAGGER = 0: everybody at 3; 1: two groups (0 and 3); 2: four groups.
__device__ __forceinline__ int draw_position(unsigned bx) {
#if GSE_STAGGER == 0
    return 3;
#elif GSE_STAGGER == 2
    return (int)(((bx >> 8) ^ bx) & 3u);
#elif GSE_STAGGER == 3
    return (int)((((bx >> 8) ^ bx) & 3u) % 3u == 0 ? 0 : ((((bx >> 8) ^ bx) & 3u) % 3u == 1 ? 2 : 3));
#else
    return (((bx >> 8) ^ bx) & 1u) ? 0 : 3;
#endif
}
// In-kernel phase stamps (diagnostic build only, -DGSE_STAMPS): workgroup 0's thread 0 records s_memtime at the phase
// boundaries into a buffer of its own; read back with gse_debug_stamps().  No stamp executes in the normal build.
#ifdef GSE_STAMPS
__device__ unsigned long long g_block_times[2 * 4096];     // s_memrealtime (100 MHz) at entry / exit of every workgroup
#define GSE_BLOCK_TIME(which)                                                                      \
    do {                                                                                           \
        if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 4096) {                            \
            unsigned long long t_;                                                                 \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");         \
            g_block_times[2 * blockIdx.x + which] = t_;                                            \
        }                                                                                          \
    } while (0)
__device__ long long g_block_info[2 * 4096];               // (ancestor span length, draw position) per workgroup
__device__ unsigned long long g_stamps[32];
#define GSE_STAMP(i)                                                              \
    do {                                                                          \
        if (blockIdx.x == GSE_STAMPS && blockIdx.y == 0 && threadIdx.x == 0) {    \
            unsigned long long t_;                                                \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
            g_stamps[i] = t_;                                                     \
        }                                                                         \
    } while (0)
__device__ unsigned long long g_mstamps[32];              // ... of the resampled-moments kernel (workgroup GSE_STAMPS / the last one)
#define GSE_MSTAMP(i, cond)                                                       \
    do {                                                                          \
        if ((cond) && blockIdx.y == 0 && threadIdx.x == 0) {                      \
            unsigned long long t_;                                                \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
            g_mstamps[i] = t_;                                                    \
        }                                                                         \
    } while (0)
#else
#define GSE_STAMP(i)
#define GSE_BLOCK_TIME(which)
#define GSE_MSTAMP(i, cond)
#endif
#ifndef GSE_PRIO
#define GSE_PRIO 4               // gather variants: issue priority by phase.  0: none; 1: memory phases high, the noise draw
                                 // and the arithmetic tail low; 2: the reverse; 3: as 1 without touching the draw; 4 / 5: as
                                 // 1 / 3 with s_setprio as an asm statement (the builtin is a scheduling barrier for hipcc
                                 // and costs 11 VGPRs -> 28 B of scratch per lane, 11 MB of spill writes per launch)
#endif
#if GSE_PRIO == 4 || GSE_PRIO == 5
#define GSE_PRIO_MEM() asm volatile("s_setprio 3")
#define GSE_PRIO_ALU() asm volatile("s_setprio 0")
#elif GSE_PRIO == 1 || GSE_PRIO == 3
#define GSE_PRIO_MEM() __builtin_amdgcn_s_setprio(3)
#define GSE_PRIO_ALU() __builtin_amdgcn_s_setprio(0)
#elif GSE_PRIO == 2
#define GSE_PRIO_MEM() __builtin_amdgcn_s_setprio(0)
#define GSE_PRIO_ALU() __builtin_amdgcn_s_setprio(3)
#else
#define GSE_PRIO_MEM()
#define GSE_PRIO_ALU()
#endif
#if GSE_PRIO == 1 || GSE_PRIO == 2 || GSE_PRIO == 4       // (3, 5: the draw keeps the priority of the phase it sits in)
#define GSE_PRIO_DRAW_BEGIN() GSE_PRIO_ALU()
#define GSE_PRIO_DRAW_END() GSE_PRIO_MEM()
#else
#define GSE_PRIO_DRAW_BEGIN()
#define GSE_PRIO_DRAW_END()
#endif
#ifndef GSE_XCD_REMAP
#define GSE_XCD_REMAP 1
#endif
__device__ __forceinline__ int xcd_contiguous_tile(unsigned bx, int T) {
#if GSE_XCD_REMAP
    if ((T & 7) == 0) return (int)((bx & 7u) * (unsigned)(T >> 3) + (bx >> 3));
#endif
    return (int)bx;
}
#ifndef GSE_INVMAP
#define GSE_INVMAP 1             // SCAN_IN gather: 1 = inverse map from the weights (no stored scans, no search); 0 = the
                                 // round-1 path (tile-local scans stored by the update stage, staged in LDS and bisected)
#endif
#ifndef GSE_GATHER_STRIDED
#define GSE_GATHER_STRIDED 1     // gather variants: a thread's 4 outputs are 256 apart (1) or consecutive (0)
#endif

#ifndef GSE_WT_STORES
#define GSE_WT_STORES 7          // bit mask of step-kernel outputs stored write-through (sc1): 1 states, 2 weights, 4 tile prefixes
#endif
// Output stores of the step kernel.  A launch at N = 2^20 writes 27 MB; what stays dirty in the eight L2s is written back
// by the release at the END of the kernel, a serial tail (tools/micro/store_tail.hip: 32 MB of plain stores trail a
// launch by +2.9 us, write-through ones by +0.7).  Write-through stores drop their lines from the L2, so the next launch
// finds less of its input there: in round 1, when the gather stage re-read 16-24 B per particle of stored scans, the lost
// hits cost more than the tail (29.2 us against 27.7); with the inverse-map gather (4-8 B of weights + 2-4 B of tile
// prefixes per particle) the tail is the larger term: all outputs write-through 20.6 us against 21.9 (r02 A/B, knob kept).
// Relaxed agent-scope atomic stores are plain `global_store ... sc1` instructions, counted and padded by the compiler.
template <int kBit>
__device__ __forceinline__ void store_out(float* p, float v) {
    if constexpr ((GSE_WT_STORES & kBit) != 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <int kBit>
__device__ __forceinline__ void store_out2(double* p, double a_, double b_) {      // p 16-byte aligned
    if constexpr ((GSE_WT_STORES & kBit) != 0) {
        __hip_atomic_store(p, a_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(p + 1, b_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        *reinterpret_cast<double2*>(p) = make_double2(a_, b_);
    }
}
using v4f_t = float __attribute__((ext_vector_type(4)));
template <int kBit>
__device__ __forceinline__ void store_out4(float* p, v4f_t v) {                     // p 16-byte aligned
    if constexpr ((GSE_WT_STORES & kBit) != 0) {
        unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
        const unsigned long long lo = ((unsigned long long)__float_as_uint(v.y) << 32) | __float_as_uint(v.x);
        const unsigned long long hi = ((unsigned long long)__float_as_uint(v.w) << 32) | __float_as_uint(v.z);
        __hip_atomic_store(q, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        *reinterpret_cast<v4f_t*>(p) = v;
    }
}

#define GSE_LOAD_DWORD(dst, ptr) asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
// scalar base (uniform row pointer) + 32-bit byte offset per lane
// (hipcc pads nothing inside an asm statement: an SGPR written by the scalar unit just before needs 5 wait states
// before a vector-memory instruction may read it as its base -- the s_nop opens the string)
#define GSE_LOAD_DWORD_S(dst, voff, sbase) \
    asm volatile("s_nop 4\n\tglobal_load_dword %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory")
#define GSE_LOAD_DWORDX4(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")

// Particle <-> thread mapping.  Without gather a thread owns 4 CONSECUTIVE particles (one dwordx4 per state row).
// With gather the reads are indexed anyway, so a thread owns 4 outputs 256 apart: for each of them the 64 lanes of a
// wavefront then touch consecutive outputs and (near-)consecutive ancestors -- every dword load / store instruction
// covers whole cache lines, where the consecutive mapping would use a quarter of each line it touches.
//
// kScan: where the gather finds the cumulative weights and what the update leaves behind for the NEXT gather.
//   bit 0 (SCAN_IN)   c[k] is not read from a materialised array (cumsum_kernel) but formed on the fly as
//                     (Pfx[tile(k)] + L[k]) / W  from the tile sums and the tile-local inclusive scans L that the update
//                     stage of the PREVIOUS step wrote; the tile prefix Pfx (T + 1 values) is computed by every workgroup
//                     into LDS.  Same arithmetic, same bits as cumsum_kernel.
//   bit 1 (SCAN_OUT)  the update stage also writes L and the tile sums of the new weights (8 B / particle).
// With both bits the steady-state cycle is ONE launch: no cumsum kernel, no second pass over the weights.
// Inverse map, one step: the 1024 keys of ancestor tile tq -> marks.  Thread t holds keys 4t .. 4t+3, formed exactly as
// cumsum_kernel forms them -- x[k] = min(Pfx[tq] + (excl + s_p), Pfx[tq+1]), the tile's last key pinned to Pfx[tq+1] -- from the
// tile's weights (s_p: running sums of the thread's four) and the thread's exclusive prefix `excl` inside the tile, which the
// update stage that produced the weights left behind (8 bytes per 4 particles).  Every key whose slot e = clamp(F(x) - j0, 0,
// 1024) is below 1024 raises mark[e] to k + 1 (LDS atomic max; a thread's equal neighbours are folded first).  No barrier.
// kNeighbours: the lanes of the wavefront hold consecutive quads of one tile (t = threadIdx.x), so runs of equal slots are
// folded across lanes too; false: the lanes hold unrelated quads (the work list of a wide ancestor range).
template <bool kNeighbours, int kSlots = kTile>      // kSlots: outputs the calling workgroup owns (marks it keeps)
__device__ __forceinline__ void expand_quad(const OffspringCount& F, const double* pfx_s, uint32_t* mark, int64_t N,
                                            int32_t j0, int tq, int t, const float4 wv, double excl) {
    const int64_t k0 = (int64_t)tq * kTile + (int64_t)t * kPerThread;
    double sc[kPerThread];
    sc[0] = (k0 + 0 < N) ? (double)wv.x : 0.0;
    sc[1] = sc[0] + ((k0 + 1 < N) ? (double)wv.y : 0.0);
    sc[2] = sc[1] + ((k0 + 2 < N) ? (double)wv.z : 0.0);
    sc[3] = sc[2] + ((k0 + 3 < N) ? (double)wv.w : 0.0);
    const double P0 = pfx_s[tq], P1 = pfx_s[tq + 1];
    const int64_t tile_last = min((int64_t)(tq + 1) * kTile, N) - 1;
    // the four counts side by side (independent chains), the rare guard-band hits settled afterwards
    double xs[kPerThread];
    int32_t es[kPerThread];
    bool near[kPerThread];
#pragma unroll
    for (int p = 0; p < kPerThread; ++p) {
        xs[p] = (k0 + p >= tile_last) ? P1 : fmin(P0 + (excl + sc[p]), P1);
        es[p] = F.candidate(xs[p], near[p]);
    }
    if (near[0] || near[1] || near[2] || near[3]) {
#pragma unroll
        for (int p = 0; p < kPerThread; ++p)
            if (near[p]) es[p] = F.settle(xs[p], es[p]);
    }
#pragma unroll
    for (int p = 0; p < kPerThread; ++p) es[p] = min(max(es[p] + 1 - j0, 0), (int32_t)kSlots);
    // Only the LAST key of a run of equal slots needs to speak (it has the largest index).  Runs are folded inside the thread
    // and across the lanes of the wavefront (the next lane holds the next four keys): in a stretch of light particles
    // hundreds of consecutive keys share a slot, and their atomics on one LDS address would be served one by one.
    uint32_t next_first = 0u;                                                     // (0: nothing known -> speak)
    if (kNeighbours) next_first = wave_shift_down1((uint32_t)es[0] + 1u);         // (0 in lane 63)
#pragma unroll
    for (int p = 0; p < kPerThread; ++p) {
        const bool last_of_run = k0 + p == tile_last ||
                                 ((p == kPerThread - 1) ? (next_first != (uint32_t)es[p] + 1u) : (es[(p + 1) % kPerThread] != es[p]));
        if (k0 + p <= tile_last && es[p] < kSlots && last_of_run) atomicMax(&mark[es[p]], (uint32_t)(k0 + p) + 1u);
    }
}

// Inverse map, the other way round, for WIDE ancestor ranges (round 3).  When the 1024 outputs of a workgroup descend from
// particles scattered over many tiles -- a stretch of lineages whose weights are a few hundred times below the average: the
// first cycles after the prior, a measurement jump -- the census lists one tile per few outputs and expanding them one pair
// after the other costs 0.45 us each while the other workgroups of the launch are long done (55 us launches among 19 us
// ones in the first 300 cycles from the prior, 173 us in a long run).  Beyond GSE_SEARCH_FROM listed tiles every output
// looks its ancestor up instead: output j descends from the first key with fl(x[k] / W) >= u_j, i.e. x[k] >= tau_j (the
// smallest such x: x -> fl(x / W) is monotone) -- the comparison the offspring counts are DEFINED by, so the ancestors are
// the same.  Three steps, a thread's four outputs side by side: the tile in the LDS prefix (its last key is pinned to
// Pfx[t + 1]); the quad inside the tile by bisecting the per-thread prefixes the update stage left (8 dependent 8-byte
// loads: the quad's last key is P0 + prefix of the NEXT quad up to rounding); then the twelve keys of that quad and its
// two neighbours re-formed exactly as everywhere else, the ancestor being the key after the last one below tau.  Cost: about
// 10 us whatever the width.
#ifndef GSE_SEARCH_FROM
#define GSE_SEARCH_FROM 20
#endif
#ifndef GSE_SEARCH_QUADS_TOGETHER
#define GSE_SEARCH_QUADS_TOGETHER 1
#endif
#ifndef GSE_SEARCH_WINDOWS_TOGETHER
#define GSE_SEARCH_WINDOWS_TOGETHER 1   // outputs whose three-quad windows are loaded side by side (1, 2 or 4: no difference measured)
#endif
constexpr int kPrefixWindow = 1026;      // entries of the global tile prefix a workgroup stages (ranges of up to 1024 tiles)
__device__ __forceinline__ int32_t window_ancestor(const float* __restrict__ wrow, const double* __restrict__ qx, int32_t N,
                                                   int32_t ld, int tq, int uq, double P0, double P1, double tau, int tl0) {
    // the quads uq - 1, uq, uq + 1 of tile tq: 18 registers of loads in flight (what fits; see search_ancestors)
    const int u0 = max(uq - 1, 0);
    const int32_t tile_first = tq * kTile;
    const int32_t tile_last = min(tile_first + kTile, N) - 1;
    int32_t below = tile_first + u0 * kPerThread - 1;                      // (everything before the window is below tau)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int u = min(u0 + q, kBlock - 1);
        const int32_t k0 = tile_first + u * kPerThread;
        const float4 wv = *reinterpret_cast<const float4*>(wrow + min(k0, ld - kPerThread));
        const double ex = qx[tq * kBlock + u];
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

// The same from the quad alone where that is certain -- two loads for six: the key before the quad (P0 + the quad's own
// exclusive prefix, which it loads anyway; for the first quad the previous tile's pinned end, exactly) lies a clear margin
// below tau and the quad's last key a clear margin at or above it, margins 2^-49 against a last-bit wiggle of neighbouring
// keys that associate differently.  Anything else -- a threshold within that margin of a quad boundary -- takes the window.
// (The look-up path is bound by the texture unit's rate of scattered requests, not by their latency: profiles r03, stamps.)
__device__ __forceinline__ int32_t quad_ancestor(const float* __restrict__ wrow, const double* __restrict__ qx, int32_t N,
                                                 int32_t ld, int tq, int uq, double P0, double P1, double tau, int tl0) {
    const int32_t tile_first = tq * kTile;
    const int32_t tile_last = min(tile_first + kTile, N) - 1;
    const int32_t k0 = tile_first + uq * kPerThread;
    const float4 wv = *reinterpret_cast<const float4*>(wrow + min(k0, ld - kPerThread));
    const double ex = qx[tq * kBlock + uq];
    double sc[kPerThread];
    sc[0] = (k0 + 0 < N) ? (double)wv.x : 0.0;
    sc[1] = sc[0] + ((k0 + 1 < N) ? (double)wv.y : 0.0);
    sc[2] = sc[1] + ((k0 + 2 < N) ? (double)wv.z : 0.0);
    sc[3] = sc[2] + ((k0 + 3 < N) ? (double)wv.w : 0.0);
    int32_t below = k0 - 1;
    double x_last = 0.0;
#pragma unroll
    for (int i = 0; i < kPerThread; ++i) {
        const double x = (k0 + i >= tile_last) ? P1 : fmin(P0 + (ex + sc[i]), P1);
        if (k0 + i <= tile_last) {
            x_last = x;
            if (!(x >= tau)) below = max(below, k0 + i);
        }
    }
    const bool certain = k0 <= tile_last && fmin(P0 + ex, P1) < tau * (1.0 - 0x1p-49) && x_last >= tau * (1.0 + 0x1p-49);
    if (!certain) return window_ancestor(wrow, qx, N, ld, tq, uq, P0, P1, tau, tl0);
    return min(max(below + 1, tl0 * kTile), N - 1);
}

// Register discipline: the hot path of the kernel sits at 106 vector registers for four wavefronts per SIMD and the register
// allocator spreads the pressure of whatever is inlined here over the whole kernel (a first version -- four 16-ary searches side
// by side, all twelve window loads of the four outputs in flight -- cost 300-450 bytes of spills per lane and 1.6 us on EVERY
// launch, as a real function call too).  So: the four tile searches together (LDS, a few integers each), the four quads
// together in 4-ary rounds (quads_of4: 24 registers of loads in flight), then the windows one output at a time, scheduling
// barriers in between.  (quad_of, the 16-ary search of one output, is what quads_of4 replaced: 37 -> 35 us on the launches
// that take this path.)
__device__ __forceinline__ int32_t quad_of(const double* __restrict__ qrow, double P0, double P1, double tau) {
    // first quad u in [0, 255] whose last key reaches tau, judged by the NEXT quad's prefix (P1 for the last quad): two 16-ary
    // rounds of independent loads.  Counting the pivots below tau instead of bisecting shrugs off a last-bit wiggle between
    // neighbouring prefixes.
    int blk = 0;
#pragma unroll
    for (int m = 1; m < 16; ++m) blk += (fmin(P0 + qrow[16 * m], P1) < tau) ? 1 : 0;       // quads 16 blk .. 16 blk + 15
    int u = 16 * blk;
#pragma unroll
    for (int m = 1; m < 16; ++m) u += (fmin(P0 + qrow[16 * blk + m], P1) < tau) ? 1 : 0;
    return u;
}

// The quads of a thread's four outputs TOGETHER: four 4-ary rounds (pivots 64, 16, 4, 1 quads apart), the twelve loads of a
// round -- three per output -- in flight side by side, where quad_of runs two 16-ary rounds per output, one output after the
// other: four dependent round trips for eight, with 24 registers of loads in flight for 30.  P0 / P1 are re-read from LDS in
// every round instead of being kept (16 registers the kernel does not have).  The same count of pivots below tau, level by
// level; window_ancestor looks at the quad and both neighbours either way.
__device__ __forceinline__ void quads_of4(const double* __restrict__ qx, const double* pfx_s, const int (&tq)[kPerThread],
                                          const double (&tau)[kPerThread], int (&u)[kPerThread]) {
#pragma unroll
    for (int p = 0; p < kPerThread; ++p) u[p] = 0;
#pragma unroll
    for (int stride = 64; stride >= 1; stride >>= 2) {
        double v[kPerThread][3];
#pragma unroll
        for (int p = 0; p < kPerThread; ++p)
#pragma unroll
            for (int m = 1; m < 4; ++m) v[p][m - 1] = qx[(int64_t)tq[p] * kBlock + u[p] + m * stride];
#pragma unroll
        for (int p = 0; p < kPerThread; ++p) {
            const double P0 = pfx_s[tq[p]], P1 = pfx_s[tq[p] + 1];
            int below = 0;
#pragma unroll
            for (int m = 0; m < 3; ++m) below += (fmin(P0 + v[p][m], P1) < tau[p]) ? 1 : 0;
            u[p] += below * stride;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ void search_ancestors(const OffspringCount& F, const double* pfx_s, const float* __restrict__ wrow,
                                                 const double* __restrict__ qx, int64_t N, int64_t ld, int tl0, int tl1,
                                                 int32_t j0, int32_t j1, int tid, int32_t (&anc)[kPerThread]) {
    double tau[kPerThread];
    int tq[kPerThread];
    {
        int lo[kPerThread];
#pragma unroll
        for (int p = 0; p < kPerThread; ++p) {
            const int32_t j = min(j0 + tid + p * kBlock, j1 - 1);
            tau[p] = smallest_with_quotient_ge(F.u_of(j), F.W);
            lo[p] = tl0;
        }
        // 1. the tile: first t in [tl0, tl1] with Pfx[t + 1] >= tau (tl1 qualifies by construction of the range).  4-ary: the
        //    answer lies in [lo, lo + width); three pivots a round, counted (five dependent rounds of LDS reads for a range of
        //    a thousand tiles where a bisection takes ten)
        int width = 1;
        while (width < tl1 - tl0 + 1) width <<= 2;                       // (uniform)
        for (; width > 1; width >>= 2) {
            const int step = width >> 2;
#pragma unroll
            for (int p = 0; p < kPerThread; ++p) {
                int below = 0;
#pragma unroll
                for (int m = 1; m < 4; ++m) {
                    const int at = lo[p] + m * step - 1;
                    below += (at < tl1 && pfx_s[min(at, tl1) + 1] < tau[p]) ? 1 : 0;
                }
                lo[p] += below * step;
            }
        }
#pragma unroll
        for (int p = 0; p < kPerThread; ++p) tq[p] = min(lo[p], tl1);
    }
    // 2. + 3. per output: the quad, then the keys of that quad and its neighbours exactly as cumsum_kernel forms them;
    //         ancestor = 1 + last key below tau
#if GSE_SEARCH_QUADS_TOGETHER
    int uq[kPerThread];
    __builtin_amdgcn_sched_barrier(0);
    quads_of4(qx, pfx_s, tq, tau, uq);
#pragma unroll
    for (int p = 0; p < kPerThread; p += GSE_SEARCH_WINDOWS_TOGETHER) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = p; q < p + GSE_SEARCH_WINDOWS_TOGETHER; ++q) {
            const double P0 = pfx_s[tq[q]], P1 = pfx_s[tq[q] + 1];
            anc[q] = quad_ancestor(wrow, qx, (int32_t)N, (int32_t)ld, tq[q], uq[q], P0, P1, tau[q], tl0);
        }
    }
#else
#pragma unroll
    for (int p = 0; p < kPerThread; ++p) {
        __builtin_amdgcn_sched_barrier(0);
        const double P0 = pfx_s[tq[p]], P1 = pfx_s[tq[p] + 1];
        const int u = quad_of(qx + (int64_t)tq[p] * kBlock, P0, P1, tau[p]);
        anc[p] = window_ancestor(wrow, qx, (int32_t)N, (int32_t)ld, tq[p], u, P0, P1, tau[p], tl0);
    }
#endif
    __builtin_amdgcn_sched_barrier(0);
}

// GSE_OP_WEIGHT_SCALE: the launch's binary exponent (gse_device.h, likelihood_exponent), by ONE wavefront: its 64 lanes score
// the same 64 sample particles of x_in in every workgroup of the launch -- exponent of the old weight + exponent of the
// likelihood under this launch's measurement -- and the best one fixes k.  In two halves so that the loads can be issued at
// kernel entry and used much later (a wavefront that waits for them in one go keeps the rest of its workgroup at the next
// barrier for a memory round trip): scale_samples_load, then scale_exponent_of.  wrow: the filter's weight row, or null (a
// gather from materialised cumulative weights does not read the weights).
#ifndef GSE_SCALE_WAVE
#define GSE_SCALE_WAVE 3
#endif
#ifndef GSE_SCALE_FN_ATTR
#define GSE_SCALE_FN_ATTR __forceinline__
#endif
struct ScaleSample {
    float s[GSE_NX];
    float w;
};
__device__ __forceinline__ void scale_samples_load(ScaleSample& q, const float* __restrict__ xin_b, const float* __restrict__ wrow,
                                                   int64_t N, int64_t ld, int lane) {
    const int64_t is = (int64_t)lane * N / kScaleSamples;
#pragma unroll
    for (int k = 0; k < GSE_NX; ++k) q.s[k] = xin_b[(int64_t)k * ld + is];
    q.w = wrow ? wrow[is] : 1.f;
}
// The same samples CARRIED from the launch that wrote these particles (round 4).  Read where they lie they cost a
// wavefront 64 cache lines per row -- 384 line requests per workgroup, as many as the launch's whole particle traffic has
// (A/B on one box, profiles/r04_scale_ab.txt: +1.5 % at 2^20 ... 2^22) -- and the model step that follows sits on the
// workgroup's critical path (+2 %).  A launch whose update fills a scan slot therefore leaves the 64 sample particles of its
// OUTPUT beside the slot, one row of kSampleRow floats each, written by the threads that own them (at most one per tile:
// the launches that do this have N / 64 >= one tile): two 16-byte loads per lane of one wavefront bring them back.  For the
// bioreactor a row holds what the anchor needs of the particle and nothing else -- the two observed states and the two rates
// their increments are made of, which depend on the state alone -- so that the consumer's "model step" is four FMAs:
//     { Cg, r_g, Cfa, r_fa, -, w, -, - }          (a user model: { x0 .. x4, w, -, - } and its own exact step)
// GSE_OP_SAMPLES (include/gse.h) is the caller's word that x_in is still what that launch wrote.
constexpr int kSampleRow = 8;
constexpr int64_t kSamplesMinN = (int64_t)kScaleSamples * kTile;
__device__ __forceinline__ void scale_samples_carried(ScaleSample& q, const float* __restrict__ smp_b, int lane) {
    const float4 lo = *reinterpret_cast<const float4*>(smp_b + lane * kSampleRow);
    const float2 hi = *reinterpret_cast<const float2*>(smp_b + lane * kSampleRow + 4);
    q.s[0] = lo.x; q.s[1] = lo.y; q.s[2] = lo.z; q.s[3] = lo.w; q.s[4] = hi.x;
    q.w = hi.y;
}
// the state half of a row, by the thread that holds the sample's new state
__device__ __forceinline__ void scale_sample_store_state(float* __restrict__ row, const float (&s)[GSE_NX]) {
#ifdef GSE_USER_MODEL
    *reinterpret_cast<float4*>(row) = make_float4(s[0], s[1], s[2], s[3]);
    row[4] = s[4];
#else
    float r_g, r_fa;
    anchor_rates(s, r_g, r_fa);
    *reinterpret_cast<float4*>(row) = make_float4(s[0], r_g, s[2], r_fa);
#endif
}
// Which particle of tile [tile0, tile0 + kTile) is a sample, if any: l * stride with l = ceil(tile0 / stride) < 64.  Uniform
// over the workgroup (scalar arithmetic): the quotient by a multiplication with the host's reciprocal, put right by a comparison.
__device__ __forceinline__ bool scale_sample_of_tile(const StepArgs& a, uint32_t tile0, uint32_t& index, int& l) {
    const uint32_t stride = a.smp_stride;
    uint32_t q = __umulhi(tile0, a.smp_magic);           // floor(tile0 / stride) or one more (magic = ceil(2^32 / stride), tile0 < 2^31)
    if (q * stride > tile0) --q;
    if (q * stride != tile0) ++q;                        // ceil
    l = (int)q;
    index = q * stride;
    return q < (uint32_t)kScaleSamples && index - tile0 < (uint32_t)kTile;
}
template <bool kGather, bool kPredict, int kNd, bool kDevIO>
__device__ GSE_SCALE_FN_ATTR int scale_exponent_of(const ScaleSample& q, bool have_w, bool carried, const StepArgs& a, int64_t N, int b) {
    if (!kGather && a.k_exact) {                         // (uniform) the exact maximum is there: scale_exact_kernel
        const uint32_t stored = a.k_exact[b];
        if (stored == 0u) return 0;                       // (no particle with a positive weight)
        const int k = ((int)stored - kExactBias) + (N > 1 ? 64 - __clzll((unsigned long long)(N - 1)) : 0);
        return (k < -kWeightScaleFrom) ? max(k, kWeightScaleMin) : 0;
    }
    double fg0 = a.io.f[b].u0, fm0 = a.io.f[b].u1, za = a.io.f[b].z0, zb = a.io.f[b].z1;
    if constexpr (kDevIO) {
        if (a.u_dev) { fg0 = a.u_dev[2 * b]; fm0 = a.u_dev[2 * b + 1]; }
        if (a.z_dev) { za = a.z_dev[2 * b]; zb = a.z_dev[2 * b + 1]; }
    }
    // the sample as the update will find it: through this launch's model step (without its noise).  The benchmark protocol
    // moves the plant -- and with it the measurement -- by a hundred mg/L per cycle (dt = 1, random inputs,
    // results/pf_openloop/pf_run_seq.py:48): judged where it stood BEFORE the step every sample looks 2^-90 unlikely.
    float st[GSE_NX];
#pragma unroll
    for (int k = 0; k < GSE_NX; ++k) st[k] = q.s[k];
#ifndef GSE_USER_MODEL
    if (carried) {                                       // (uniform) a carried row: { Cg, r_g, Cfa, r_fa }
        const float r_g = st[1], r_fa = st[3];
        if (kPredict) anchor_advance(st[0], st[2], r_g, r_fa, fg0, fm0, a.dt);
    } else
#endif
    if (kPredict) model_step_anchor(st, fg0, fm0, a.dt);
    float o0[kObsKept], ya, yb;
    model_keep(st, o0);
    model_output_kept(o0, fg0, fm0, ya, yb);
    // exponents, clamped to 15 bits each: score = old weight x likelihood (which sample matters most), cand = what k is made of
    const int e_lik = min(max(likelihood_exponent<kNd>(a.pdf, za - (double)ya, zb - (double)yb), -16000), 2000);
    const int e_w = have_w ? ((q.w > 0.f) ? max(ilogbf(q.w), -300) : -16000) : 0;
    const int score = max(e_lik + e_w, -32000), cand = max(e_lik + (kGather ? 0 : e_w), -32000);
    // the best score, with its cand, by ONE max-reduction of a packed key over the wavefront (DPP: no LDS round trips)
    const uint32_t key = ((uint32_t)(score + 32768) << 16) | (uint32_t)(cand + 32768);
    const uint32_t best = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_max(key), 63);
    const int c = (int)(best & 0xffffu) - 32768;
    // gather: the offspring weigh lik / N, so k = the anchor's likelihood exponent; otherwise its new weight w lik 2^-k is
    // brought to about 1 / N  (ceil(log2 N) by a count of leading zeros)
    const int k = kGather ? c : c + (N > 1 ? 64 - __clzll((unsigned long long)(N - 1)) : 0);
    return (k < -kWeightScaleFrom) ? max(k, kWeightScaleMin) : 0;
}

// The anchor of a launch that updates WITHOUT a resample in front of it, exactly (round 4).  After a resample every weight is
// 1 / N and 64 samples say all there is to know about where the likelihood is large; after updates that were NOT followed by a
// resample the weights themselves span any number of binades, the heaviest particle need not stand anywhere near a sample, and
// a k taken from light samples would scale the heavy ones past the end of fp32.  Such launches are rare (the first cycle of a
// run; callers that skip resamples) and pay for certainty with a pass of their own: the maximum over ALL particles of
// (exponent of the weight + exponent of the likelihood after the launch's model step), one atomic per wavefront.
template <bool kPredict, bool kDevIO>
__global__ __launch_bounds__(kBlock) void scale_exact_kernel(const float* __restrict__ x_in, const float* __restrict__ w_in,
                                                             int64_t N, int64_t ld, uint32_t* __restrict__ k_out,
                                                             const StepArgs a) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    double fg0 = a.io.f[b].u0, fm0 = a.io.f[b].u1, za = a.io.f[b].z0, zb = a.io.f[b].z1;
    if constexpr (kDevIO) {
        if (a.u_dev) { fg0 = a.u_dev[2 * b]; fm0 = a.u_dev[2 * b + 1]; }
        if (a.z_dev) { za = a.z_dev[2 * b]; zb = a.z_dev[2 * b + 1]; }
    }
    const float* xb = x_in + (int64_t)b * GSE_NX * ld;
    const float* wb = w_in + (int64_t)b * ld;
    uint32_t best = 0u;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < N; i += (int64_t)gridDim.x * kBlock) {
        const float w = wb[i];
        float st[GSE_NX];
#pragma unroll
        for (int k = 0; k < GSE_NX; ++k) st[k] = xb[(int64_t)k * ld + i];
        if (!(w > 0.f)) continue;
        if (kPredict) model_step_anchor(st, fg0, fm0, a.dt);
        float o0[kObsKept], ya, yb;
        model_keep(st, o0);
        model_output_kept(o0, fg0, fm0, ya, yb);
        const int e_lik = min(max(likelihood_exponent<0>(a.pdf, za - (double)ya, zb - (double)yb), -16000), 2000);
        const int e = max(e_lik + max(ilogbf(w), -300), -32000);
        best = max(best, (uint32_t)(e + kExactBias));
    }
    best = (uint32_t)__builtin_amdgcn_readlane((int)wave_inclusive_max(best), 63);
    if (lane == 0 && best != 0u) atomicMax(k_out + b, best);
}

template <bool kGather, bool kPredict, bool kInjected, bool kUpdate, int kNd, int kScan>
__global__ __launch_bounds__(kBlock, GSE_STEP_MIN_WAVES) void pf_step_kernel(
    const float* __restrict__ x_in, float* __restrict__ x_out, const float* __restrict__ w_in, float* __restrict__ w_out,
    const float* __restrict__ noise, int32_t* __restrict__ idx_out, const double* __restrict__ c,
    const double* __restrict__ totals, double* __restrict__ tile_sums, const double* __restrict__ scan_in_sums,
    const double* __restrict__ scan_in, double* __restrict__ scan_out_sums, double* __restrict__ scan_out,
    const double* __restrict__ prefix_g, double* __restrict__ island, int64_t N, int64_t ld, int T, const StepArgs a) {
    constexpr bool kScanIn = (kScan & 1) != 0, kScanOut = (kScan & 2) != 0;
    // bit 3 (round 3): the tile prefix is not rebuilt by every workgroup from the tile sums but taken from prefix_g
    // ([B][T + 1], tile_scan_kernel ran on the slot's tile sums) -- beyond 2048 tiles rebuilding costs more than it saves
    // (every workgroup reads all T sums: 131 us against 102 at 2^22).  A workgroup finds its range of ancestor tiles in the
    // global prefix (three 32-ary rounds of the wavefront) and stages that WINDOW in LDS; everything after is the same code.
    constexpr bool kPrefixGlobal = (kScan & 8) != 0;
    static_assert(!kPrefixGlobal || kScanIn, "the global tile prefix belongs to SCAN_IN");
    constexpr bool kDraw = kPredict && !kInjected;      // noise from Philox (registers only) rather than from memory
    // strided particle <-> thread mapping (see below): whenever the launch gathers or draws -- the Philox blocks of the
    // draw are shared by the four particles a thread owns in THIS mapping (gse_device.h, rng_block_of_group)
    constexpr bool kStrided = GSE_GATHER_STRIDED && (kGather || kDraw);
    static_assert(!kDraw || kStrided, "the grouped draw needs the strided mapping");
    constexpr bool kDevIO = (kScan & 4) != 0;      // u / z may come from device memory (a closed loop on the device)
    static_assert(!kScanIn || kGather, "SCAN_IN belongs to the gather stage");
    static_assert(!kScanOut || kUpdate, "SCAN_OUT belongs to the update stage");
    __shared__ DrawParams dp;
    constexpr bool kInvMap = kScanIn && (GSE_INVMAP != 0);
    __shared__ double seg[(kGather && !kInvMap) ? kSpanMax : 2];
    __shared__ __attribute__((aligned(16))) uint32_t mark[kInvMap ? kTile : 4];     // inverse map: see the gather stage
    __shared__ int32_t act_list[kInvMap ? kTile : 1];
    __shared__ int32_t act_n;
    __shared__ __attribute__((aligned(16))) double wtile[(kStrided && kUpdate) ? kTile : 2];    // the new weights, fp64 (see the update)
    __shared__ int64_t span[2];
    __shared__ double red[4];
    __shared__ int kexp_s;                    // GSE_OP_WEIGHT_SCALE: this launch's binary exponent
    extern __shared__ double pfx_s[];         // [T + 1] with SCAN_IN (dynamic; statics above total a multiple of 16 B)

    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // Tile of this workgroup.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 shares an XCD); an output
    // tile reads ancestors from its own and the neighbouring tiles, which the previous launch wrote.  Giving each XCD a
    // CONTIGUOUS range of tiles keeps those neighbours -- and the lines they left in that XCD's L2 -- on one XCD
    // (speed only: nothing depends on the placement).
    const int tile_id = xcd_contiguous_tile(blockIdx.x, T);
    const int64_t tile0 = (int64_t)tile_id * kTile;
    const int64_t i0 = tile0 + (int64_t)tid * kPerThread;       // consecutive mapping: first of this thread's 4
    const double Nd = (double)N;
    const float w_uniform = (float)(1.0 / Nd);
    const float* xin_b = x_in + (int64_t)b * GSE_NX * ld;

    // index of this thread's p-th particle (N < 2^31 is checked by the host: 32-bit offsets from uniform row pointers
    // let the loads and stores use the scalar-base + 32-bit-offset form instead of a 64-bit address per access)
    uint32_t jp[kPerThread];
#pragma unroll
    for (int p = 0; p < kPerThread; ++p)
        jp[p] = (uint32_t)(kStrided ? tile0 + tid + (int64_t)p * kBlock : i0 + p);
    const uint32_t Nu = (uint32_t)N;

    if (kPredict && !kInjected) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&a.draw);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&dp);
        for (int i = tid; i < (int)(sizeof(DrawParams) / 4); i += kBlock) dst[i] = src[i];
    }

    // ---- noise: Philox + Box-Muller, registers only.  At N = 2^20 all workgroups of a launch are resident at once and
    //      march through the same phases together, so while they wait on memory (tile prefix, searches, staging,
    //      gather) no wavefront on the CU has arithmetic to do.  The draw is the one block of arithmetic that depends
    //      on nothing: half of the workgroups do it up front, the other half between issuing their state loads and
    //      using them, so that on every CU one group computes while the other waits.
    float e[kPerThread][GSE_NX];
    auto draw_all = [&]() {
        if (kGather) GSE_PRIO_DRAW_BEGIN();
        mixture_draw5x4(dp, a.rng, (uint32_t)tile_id * kBlock + (uint32_t)tid, (uint32_t)b, e);
        if (kGather) GSE_PRIO_DRAW_END();
    };
    const int draw_at = (kDraw && kGather) ? draw_position(blockIdx.x) : 3;      // workgroup-uniform
    GSE_STAMP(0);
    GSE_BLOCK_TIME(0);
    if (kGather) GSE_PRIO_MEM();
    // SCAN_IN: this thread's chunk of the tile sums (the scan modes run up to 2048 tiles, so at most 8), requested before
    // anything else -- also before an early noise draw, which then covers their latency
    constexpr int kMaxChunk = 8;
    double sv[kMaxChunk];
    if (kScanIn && !kPrefixGlobal) {
        const double* sums = scan_in_sums + (int64_t)b * T;
        const int m = (T + kBlock - 1) / kBlock;
        const int qlo = tid * m;
#pragma unroll
        for (int i = 0; i < kMaxChunk; ++i) sv[i] = (i < m && qlo + i < T) ? sums[qlo + i] : 0.0;
    }
    // ---- GSE_OP_WEIGHT_SCALE: the launch's binary exponent (scale_samples_load / scale_exponent_of above).  One wavefront (kScaleWave) of
    //      every workgroup requests the 64 samples HERE, with the tile sums, and turns them into k after the gather stage.
    ScaleSample ssm;
    constexpr int kScaleWave = GSE_SCALE_WAVE;       // which wavefront evaluates the anchor (wavefront 0 has thread 0's chores already)
    const bool scale_w = (!kGather || kScanIn) && w_in != nullptr;
    const bool carried = kScanIn && a.smp_in != nullptr;                   // (uniform)
    if (kUpdate && a.wscale && wv == kScaleWave) {
        if (carried) scale_samples_carried(ssm, a.smp_in + (int64_t)b * kScaleSamples * kSampleRow, lane);
        else if (kGather || !a.k_exact) scale_samples_load(ssm, xin_b, scale_w ? w_in + (int64_t)b * ld : nullptr, N, ld, lane);
    }
    if (kDraw) __syncthreads();                          // dp staged
    if (kDraw && draw_at == 0) draw_all();
    GSE_STAMP(1);

    // ---- gather: ancestors of this thread's 4 outputs -------------------------------------------------
    int32_t anc[kPerThread] = {0, 0, 0, 0};
    if (kStrided && !kGather) {
        // no resample in this launch: every particle is its own "ancestor" (clipped: lanes beyond N read a valid row)
#pragma unroll
        for (int p = 0; p < kPerThread; ++p) anc[p] = (int32_t)min(jp[p], Nu - 1u);
    }
    bool drawn2 = false;
    if (kGather) {
        double W;
        if constexpr (kPrefixGlobal) {
            if (kInvMap) {
                *reinterpret_cast<uint4*>(&mark[tid * kPerThread]) = make_uint4(0u, 0u, 0u, 0u);
                if (tid == 0) act_n = 0;
            }
            W = prefix_g[(int64_t)b * (T + 1) + T];
            if (island && tile_id == 0 && tid == 0) {      // shard bookkeeping (see gse_pf_resample)
                const double ws_b = a.wscale ? a.wscale[b] : 0.0;      // (stored weights = true weights * 2^-ws_b)
                island[b] += log(W) + ws_b * 0.69314718055994530942;
                island[gridDim.y + b] = ldexp(W, (int)ws_b);
            }
        } else if (kScanIn) {
            // tile prefix into LDS: thread j owns chunk [j*m, (j+1)*m) of the tile sums (as inline_tile_prefix); the sums were
            // loaded at kernel entry and stay in registers for the second sweep
            const int m = (T + kBlock - 1) / kBlock;
            const int qlo = tid * m;
            if (kInvMap) {
                *reinterpret_cast<uint4*>(&mark[tid * kPerThread]) = make_uint4(0u, 0u, 0u, 0u);
                if (tid == 0) act_n = 0;
            }
            double chunk = 0.0;
            double total;
            if (m <= kMaxChunk) {                                          // workgroup-uniform; up to 2048 tiles
#pragma unroll
                for (int i = 0; i < kMaxChunk; ++i) chunk += sv[i];        // (+ 0.0 beyond the chunk: exact)
                double run = block_exclusive_scan(chunk, red, total);
#pragma unroll
                for (int i = 0; i < kMaxChunk; ++i) {
                    if (i < m && qlo + i < T) pfx_s[qlo + i] = run;
                    run += sv[i];
                }
            } else {
                // longer chunks (2049 .. 4096 tiles: several generations of workgroups hide the second sweep's loads): the
                // sums are read again instead of kept -- same order, same bits
                const double* sums = scan_in_sums + (int64_t)b * T;
                const int qhi = min(qlo + m, T);
                for (int q = qlo; q < qhi; ++q) chunk += sums[q];
                double run = block_exclusive_scan(chunk, red, total);
                for (int q = qlo; q < qhi; ++q) {
                    pfx_s[q] = run;
                    run += sums[q];
                }
            }
            if (tid == 0) pfx_s[T] = total;
            __syncthreads();
            W = pfx_s[T];
    GSE_STAMP(2);
            if (island && tile_id == 0 && tid == 0) {      // shard bookkeeping (see gse_pf_resample)
                const double ws_b = a.wscale ? a.wscale[b] : 0.0;      // (stored weights = true weights * 2^-ws_b)
                island[b] += log(W) + ws_b * 0.69314718055994530942;
                island[gridDim.y + b] = ldexp(W, (int)ws_b);
            }
        } else {
            W = totals[b];
        }
        // a weight sum that is +inf or NaN (with GSE_OP_WEIGHT_SCALE: the stored weights overflowed): the resample below then
        // copies particle 0 everywhere, as the reference's does -- but never silently (include/gse.h, GSE_WSTATUS_SUM_NOT_FINITE)
        if (a.wstatus && tile_id == 0 && tid == 0 && !(W < INFINITY)) a.wstatus[b] = GSE_WSTATUS_SUM_NOT_FINITE;
        if (kDraw && draw_at == 1) draw_all();
        const double r = a.ra.r[b];
        // Weights that do not sum to a positive finite number: the reference's normalised cumsum is all-NaN and its
        // sweep never advances (filter/particle.py:98), so every offspring copies particle 0 (anc stays 0).
        if ((W > 0.0) && (W < INFINITY)) {
          if constexpr (kInvMap) {
            // ---- inverse map (gse_device.h, OffspringCount).  No stored scans, no search: the keys of an ancestor tile are
            //      re-formed from its WEIGHTS exactly as cumsum_kernel forms them -- x[k] = min(Pfx[t] + (excl + s_p),
            //      Pfx[t+1]), thread t of the workgroup holding keys 4t .. 4t+3 -- and every key with e = F(k) - j0 in
            //      [0, 1024) raises mark[e] to k + 1 (LDS atomic max; a thread's equal neighbours are folded first).  F is
            //      monotone, so the ancestor of output slot i is  1 + max{k : e_k <= i}  (the key after the last one whose
            //      offspring all lie before i) = an inclusive max-scan of the marks, floored at the first key of the first
            //      tile that can hold an ancestor.
            OffspringCount F;
            F.init(N, r, W);
            const int32_t j0 = (int32_t)tile0;
            const int32_t j1 = (int32_t)min(tile0 + kTile, N);
            auto slot_of = [&](double x) -> int32_t { return min(max(F(x) - j0, 0), (int32_t)kTile); };
            // which tiles can hold ancestors: conservative on both sides by 2^-50 (see the round-1 path below)
            const double p_lo = (F.u_of(j0) * W) * (1.0 - 0x1p-50);
            const double p_hi = (F.u_of(j1 - 1) * W) * (1.0 + 0x1p-50);
            int tl0, tl1;
            const double* pfx = pfx_s;                                             // tile prefix, indexed by tile (LDS)
            bool too_wide = false;
            if constexpr (kPrefixGlobal) {
                const double* Pg = prefix_g + (int64_t)b * (T + 1);
                wave_search_ge_pair([&](int q) { return Pg[q + 1]; }, T, p_lo, p_hi, lane, tl0, tl1);
                const int nwin = tl1 - tl0 + 2;                                    // Pfx[tl0] .. Pfx[tl1 + 1]
                too_wide = nwin > kPrefixWindow;                                   // workgroup-uniform
                if (!too_wide)
                    for (int i = tid; i < nwin; i += kBlock) pfx_s[i] = Pg[tl0 + i];
                __syncthreads();                                                   // (the window; also the zeroed marks)
                pfx = pfx_s - tl0;
            } else {
                wave_search_ge_pair([&](int q) { return pfx_s[q + 1]; }, T, p_lo, p_hi, lane, tl0, tl1);
            }
            const float* wrow = w_in + (int64_t)b * ld;
    GSE_STAMP(11);
            const double* qx = scan_in + (int64_t)b * T * kBlock;                  // [T][256] per-thread exclusive prefixes
            // Which tiles to expand.  One or two (the common case): both.  A wider range (a stretch of light particles):
            // census first -- all keys of tile q lie in [Pfx[q], Pfx[q+1]]; when both ends fall into the same slot the whole
            // tile does and its LAST key stands for it; only tiles that some output of this workgroup descends from are
            // listed.  The listed tiles are expanded two at a time, the loads of the next pair in flight meanwhile.
            // (A second census per quad for very wide ranges -- only the quads that contain a slot boundary get the full
            // treatment -- was measured and dropped: no faster than this loop once its loads really ran ahead.)
            const bool direct = tl1 - tl0 < 2;                                     // workgroup-uniform
            int n_act = tl1 - tl0 + 1;
            if (!direct && !too_wide) {
                for (int q0 = tl0; q0 <= tl1; q0 += kBlock) {                       // (whole wavefronts take part: DPP below)
                    const int q = q0 + tid;
                    const bool have = q <= tl1;
                    const int32_t e0 = have ? slot_of(pfx[q]) : 0, e1 = have ? slot_of(pfx[q + 1]) : -1;
                    // a run of tiles that all fall into one slot: the last of the run speaks for it
                    const uint32_t next_e1 = wave_shift_down1((uint32_t)(e1 + 2));     // (0 in lane 63 -> speak)
                    if (!have) continue;
                    if (e0 == e1) {
                        if (e1 < kTile && next_e1 != (uint32_t)(e1 + 2))
                            atomicMax(&mark[e1], (uint32_t)(min((int64_t)(q + 1) * kTile, N) - 1) + 1u);
                    } else {
                        act_list[atomicAdd(&act_n, 1)] = q;                        // at most 1024: the slots e0 < e1 are disjoint
                    }
                }
                __syncthreads();
                n_act = act_n;
            }
            auto tile_at = [&](int i) { return direct ? tl0 + i : act_list[i]; };
            constexpr bool kSearchPath = GSE_SEARCH_FROM < kTile;                  // (a knob >= 1024 compiles the path out)
            bool wide = false;
            if constexpr (kSearchPath) wide = !direct && n_act > GSE_SEARCH_FROM;  // workgroup-uniform
            if constexpr (kPrefixGlobal) wide = wide || too_wide;                  // (a range beyond the window: look up in
                                                                                   //  the global prefix itself)
            if ((kSearchPath || kPrefixGlobal) && wide) {
                if (kDraw && draw_at == 2) {
                    draw_all();
                    drawn2 = true;
                }
                if (kPrefixGlobal && too_wide)
                    search_ancestors(F, prefix_g + (int64_t)b * (T + 1), wrow, qx, N, ld, tl0, tl1, j0, j1, tid, anc);
                else
                    search_ancestors(F, pfx, wrow, qx, N, ld, tl0, tl1, j0, j1, tid, anc);
            } else if (n_act > 0) {                                                // (a census may list nothing)
                // (every load below is unconditional, its index clamped instead: with loads under conditions the compiler
                //  cannot count what is in flight and waits for ALL of it before the first use -- the next pair's loads
                //  included, which is the opposite of a prefetch)
                const int last = max(n_act - 1, 0);
                auto weights_nc = [&](int tq) {
                    const int64_t k0 = min((int64_t)tq * kTile + (int64_t)tid * kPerThread, ld - kPerThread);
                    return *reinterpret_cast<const float4*>(wrow + k0);          // (beyond N: masked where it is used)
                };
                int t0 = tile_at(0), t1 = tile_at(min(1, last));
                float4 c0 = weights_nc(t0), c1 = weights_nc(t1);
                double x0 = qx[t0 * kBlock + tid], x1 = qx[t1 * kBlock + tid];
                if (kDraw && draw_at == 2) {
                    draw_all();
                    drawn2 = true;
                }
                for (int i = 0; i < n_act; i += 2) {
                    const int u0 = tile_at(min(i + 2, last)), u1 = tile_at(min(i + 3, last));
                    const float4 n0 = weights_nc(u0), n1 = weights_nc(u1);
                    const double y0 = qx[u0 * kBlock + tid], y1 = qx[u1 * kBlock + tid];
                    expand_quad<true>(F, pfx, mark, N, j0, t0, tid, c0, x0);
                    if (i + 1 < n_act) expand_quad<true>(F, pfx, mark, N, j0, t1, tid, c1, x1);
                    c0 = n0; c1 = n1; x0 = y0; x1 = y1; t0 = u0; t1 = u1;
                }
            }
    GSE_STAMP(3);
#ifdef GSE_STAMPS
            if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 4096) {
                g_block_info[2 * blockIdx.x] = (long long)(tl1 - tl0 + 1) | ((long long)act_n << 20);
                g_block_info[2 * blockIdx.x + 1] = draw_at;
            }
#endif
            __syncthreads();
            // inclusive max-scan of the 1024 marks: thread t owns slots 4t .. 4t+3
            if (!wide) {
                const uint4 mk = *reinterpret_cast<const uint4*>(&mark[tid * kPerThread]);
                const uint32_t m0 = mk.x, m1 = max(m0, mk.y), m2 = max(m1, mk.z), m3 = max(m2, mk.w);
                const uint32_t inc = wave_inclusive_max(m3);
                uint32_t before = wave_shift_up1(inc);
                uint32_t* wmax = reinterpret_cast<uint32_t*>(red);                 // (free since the tile prefix)
                if (lane == 63) wmax[wv] = inc;
                __syncthreads();
                if (wv > 0) before = max(before, wmax[0]);
                if (wv > 1) before = max(before, wmax[1]);
                if (wv > 2) before = max(before, wmax[2]);
                // marks hold k + 1 = the key that FOLLOWS the last exhausted one: that is the ancestor itself
                const uint32_t floor_key = (uint32_t)tl0 * (uint32_t)kTile;        // first key of the range
                const uint32_t last_key = (uint32_t)N - 1u;
                uint4 av;
                av.x = min(max(max(before, m0), floor_key), last_key);
                av.y = min(max(max(before, m1), floor_key), last_key);
                av.z = min(max(max(before, m2), floor_key), last_key);
                av.w = min(max(max(before, m3), floor_key), last_key);
                *reinterpret_cast<uint4*>(&mark[tid * kPerThread]) = av;           // the slots this thread alone has read
                __syncthreads();
#pragma unroll
                for (int p = 0; p < kPerThread; ++p) anc[p] = (int32_t)mark[tid + p * kBlock];
            }
          } else {
            const double* cb = kScanIn ? scan_in + (int64_t)b * N : c + (int64_t)b * N;
            // The search key of ancestor k.
            //   materialised c[] : ckey(k) = c[k]                                   (normalised by cumsum_kernel)
            //   SCAN_IN          : xkey(k) = min(Pfx[t] + L[k], Pfx[t+1])           (un-normalised; tile's last: Pfx[t+1])
            //                      ckey(k) = fl(xkey(k) / W)                         -- the bits cumsum_kernel would store
            // Outputs compare ckey with u = (j + r) / N, the reference's comparison verbatim.  Where a division per probe
            // would sit on a dependent chain (the bracket search, the long-span fallback) the un-normalised key is
            // compared with tau(u) = the smallest fp64 x with fl(x / W) >= u instead: x -> fl(x / W) is monotone, so
            // fl(xkey / W) >= u  <=>  xkey >= tau(u) -- exactly the same decision, no division in the loop.
            auto xkey = [&](int64_t k) -> double {
                if (!kScanIn) return cb[k];
                const int t = (int)(k / kTile);
                const int64_t last = min((int64_t)(t + 1) * kTile, N) - 1;
                return (k == last) ? pfx_s[t + 1] : fmin(pfx_s[t] + cb[k], pfx_s[t + 1]);
            };
            // u_j = (j + r) / N as the reference forms it; for N a power of two the division is an exact scaling and
            // (j + r) * (1 / N) gives the same bits without a division
            const bool pow2 = (N & (N - 1)) == 0;
            const double invN = 1.0 / Nd;
            auto u_of = [&](uint32_t j) -> double {
                const double jr = (double)j + r;
                return pow2 ? jr * invN : jr / Nd;
            };
            auto tau_of = [&](double u) -> double { return kScanIn ? smallest_with_quotient_ge(u, W) : u; };
            const int64_t j0 = tile0;
            const int64_t j1 = min(j0 + kTile, N);
            double thr[kPerThread];                       // thresholds of this thread's four outputs
#pragma unroll
            for (int p = 0; p < kPerThread; ++p) thr[p] = u_of(jp[p]);
            // Where the ancestors of this tile's outputs live.  SCAN_IN, the common case: the span lies inside one or
            // two ancestor TILES, found in the LDS prefix alone (every wavefront for itself: two 64-ary rounds per end,
            // no memory access, no barrier) -- the whole tiles are staged and bisected, a superset of the exact span,
            // which changes nothing in the result.  The tile range is conservative by construction: a tile is excluded
            // at the low end only if its last key is below u_first W (1 - 2^-50), at the high end the first tile whose
            // last key reaches u_last W (1 + 2^-50) closes the range; the margins dwarf the roundings (2^-52) of the
            // products and of the quotient the exact comparison uses.
            int64_t klo = 0, khi = -1;
            bool whole_tiles = false;
            int tl0 = 0;
            if (kScanIn) {
                const double p_lo = (u_of((uint32_t)j0) * W) * (1.0 - 0x1p-50);
                const double p_hi = (u_of((uint32_t)(j1 - 1)) * W) * (1.0 + 0x1p-50);
                tl0 = (int)wave_search_ge([&](int64_t q) { return pfx_s[q + 1]; }, T, p_lo, lane);
                const int tl1 = (int)wave_search_ge([&](int64_t q) { return pfx_s[q + 1]; }, T, p_hi, lane);
                if (tl1 - tl0 < kSpanMax / kTile) {                            // workgroup-uniform
                    whole_tiles = true;
                    klo = (int64_t)tl0 * kTile;
                    khi = min((int64_t)(tl1 + 1) * kTile, N) - 1;
                }
            }
            if (!whole_tiles) {
                // exact bracket: wavefront 0 searches the ancestor of the first output, wavefront 1 that of the last
                if (wv < 2) {
                    const int64_t j = (wv == 0) ? j0 : j1 - 1;
                    const double t_e = tau_of(u_of((uint32_t)j));
                    int64_t lo_k = 0, hi_k = N - 1;
                    if (kScanIn) {
                        // narrow to one tile in the LDS prefix first (no memory access), then a 64-ary search inside it
                        int tl = 0, th = T - 1;                                    // first tile whose end key >= t_e
                        while (tl < th) {
                            const int mid = (tl + th) >> 1;
                            if (pfx_s[mid + 1] < t_e) tl = mid + 1; else th = mid;
                        }
                        lo_k = (int64_t)tl * kTile;
                        hi_k = min((int64_t)(tl + 1) * kTile, N) - 1;
                    }
                    const int64_t k =
                        lo_k + wave_search_ge([&](int64_t q) { return xkey(lo_k + q); }, hi_k - lo_k + 1, t_e, lane);
                    if (lane == 0) span[wv] = k;
                }
                __syncthreads();
                klo = span[0];
                khi = span[1];
            }
    GSE_STAMP(3);
            const int64_t len = khi - klo + 1;
#ifdef GSE_STAMPS
            if (tid == 0 && blockIdx.y == 0 && blockIdx.x < 4096) {
                g_block_info[2 * blockIdx.x] = len;
                g_block_info[2 * blockIdx.x + 1] = draw_at;
            }
#endif
            if (kDraw && draw_at == 2) {
                draw_all();
                drawn2 = true;
            }
            if (len <= kSpanMax) {
                // the span fits LDS (the common case: ~1000 ancestors at the median): stage the normalised keys (all
                // loads, then all divisions, in flight together) and bisect there against u
                {
                    constexpr int kPer = kSpanMax / kBlock;
                    double kv[kPer];
                    if (kScanIn && whole_tiles) {
                        // whole tiles: the tile of a round is uniform, so are its two prefix entries
#pragma unroll
                        for (int m = 0; m < kPer; ++m) {
                            if ((int64_t)m * kBlock < len) {                   // workgroup-uniform: whole rounds are skipped
                                const int tm = tl0 + (m * kBlock) / kTile;
                                const int64_t k = klo + m * kBlock + tid;
                                const int64_t tile_last = min((int64_t)(tm + 1) * kTile, N) - 1;
                                const double below = pfx_s[tm], top = pfx_s[tm + 1];
                                const double l = cb[min(k, tile_last)];
                                kv[m] = (k >= tile_last) ? top : fmin(below + l, top);
                            }
                        }
                    } else {
#pragma unroll
                        for (int m = 0; m < kPer; ++m) {
                            if ((int64_t)m * kBlock < len) {
                                const int q = tid + m * kBlock;
                                kv[m] = xkey(klo + min((int64_t)q, len - 1));
                            }
                        }
                    }
#pragma unroll
                    for (int m = 0; m < kPer; ++m) {
                        if ((int64_t)m * kBlock < len) {
                            const int q = tid + m * kBlock;
                            if (q < (int)len) seg[q] = kScanIn ? kv[m] / W : kv[m];
                        }
                    }
                }
                __syncthreads();
                int found[kPerThread];
                span_search4(seg, (int)len, thr, found);
#pragma unroll
                for (int p = 0; p < kPerThread; ++p) anc[p] = (int32_t)klo + found[p];
            } else {
                // A long span: the tile's ancestors are spread over a stretch of (nearly) weightless particles.  Two
                // levels: LDS gets every s-th key (the last of each group of s, so seg[g] = max of group g); an output
                // bisects the groups in LDS, then the <= s keys of its group in memory (log2 s probes inside a few
                // cache lines).  One staging pass and one short dependent chain whatever the length of the span.
                const int64_t s_ = (len + kSpanMax - 1) / kSpanMax;              // group size, >= 2
                const int groups = (int)((len + s_ - 1) / s_);                    // <= kSpanMax
                for (int q = tid; q < groups; q += kBlock) seg[q] = xkey(min(klo + (int64_t)(q + 1) * s_, khi + 1) - 1);
                __syncthreads();
#pragma unroll
                for (int p = 0; p < kPerThread; ++p) {
                    const double t_p = tau_of(thr[p]);                            // un-normalised keys here
                    const int g_ = span_search(seg, groups, t_p);
                    int64_t lo = klo + (int64_t)g_ * s_;
                    int64_t hi = min(lo + s_, khi + 1) - 1;
                    while (lo < hi) {
                        const int64_t mid = (lo + hi) >> 1;
                        if (xkey(mid) < t_p) lo = mid + 1; else hi = mid;
                    }
                    anc[p] = (int32_t)lo;
                }
            }
          }
        }
    }
    if (kDraw && draw_at == 2 && !drawn2) draw_all();      // (weights that do not sum to a positive number: no search)
    GSE_STAMP(4);
    // GSE_OP_WEIGHT_SCALE: the launch's binary exponent, from the samples wavefront kScaleWave requested at kernel entry.  The other
    // wavefronts must not WAIT for it here: a barrier at this point lines the four wavefronts up again just before the state
    // loads and costs the launch 2.7 % (A/B on one box: 67.5 us against 65.8 at 2^22, profiles/r04_scale_ab.txt; the arithmetic
    // itself is free).  In the strided variants -- every launch that gathers or draws -- the update stage exchanges the new
    // weights through LDS behind a barrier of its own: the likelihoods are formed UNSCALED, the fp64 products w pdf go through
    // that exchange, and the power of two is applied behind its barrier (the reference's own order: weights *= pdf in fp64,
    // particle.py:294 -- then one rounding to the stored fp32).  Variants without that exchange (a bare update, the
    // injected-noise predict + update) take a barrier here.
    int kexp = 0;
    constexpr bool kScaleBehindExchange = kStrided && kUpdate;
    if (kUpdate && a.wscale) {                               // (workgroup-uniform)
        if (wv == kScaleWave) {
            const int k = scale_exponent_of<kGather, kPredict, kNd, kDevIO>(ssm, scale_w, carried, a, N, b);
            if (lane == 0) kexp_s = k;
        }
        if (!kScaleBehindExchange) {
            __syncthreads();
            kexp = kexp_s;
        }
    }
    // a thread is live if it owns at least one particle (its first one)
    const bool live = jp[0] < Nu;
    float wnew[kPerThread] = {w_uniform, w_uniform, w_uniform, w_uniform};
    double wd[kPerThread] = {0.0, 0.0, 0.0, 0.0};            // strided update: the new weights before their rounding to fp32
    if (live) {
        // ---- issue every load now; the RNG arithmetic below needs no memory and hides their latency.  (Plain loads
        //      get sunk below the RNG by the compiler to save registers, which serialises "RNG, then wait for HBM";
        //      the asm statements pin the issue point and the wait is tied to the registers -- guide 5.7.)
        float g[2][GSE_NX];                   // indexed rows of the current half (strided mapping)
        v4f r0, r1, r2, r3, r4;               // contiguous rows (consecutive mapping)
        if (kStrided) {
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int k = 0; k < GSE_NX; ++k) GSE_LOAD_DWORD_S(g[q][k], (uint32_t)anc[q] * 4u, xin_b + k * ld);
        } else {
            const float* xb = xin_b + i0;
            GSE_LOAD_DWORDX4(r0, xb);
            GSE_LOAD_DWORDX4(r1, xb + ld);
            GSE_LOAD_DWORDX4(r2, xb + 2 * ld);
            GSE_LOAD_DWORDX4(r3, xb + 3 * ld);
            GSE_LOAD_DWORDX4(r4, xb + 4 * ld);
        }
        if (kUpdate && !kGather && !kStrided) {
            const float4 w4 = *reinterpret_cast<const float4*>(w_in + (int64_t)b * ld + i0);
            wnew[0] = w4.x; wnew[1] = w4.y; wnew[2] = w4.z; wnew[3] = w4.w;
        }
        float nz[kPerThread][GSE_NX];
        if (kPredict && kInjected) {
            const float* nb = noise + (int64_t)b * GSE_NX * ld;
            if (kStrided) {
#pragma unroll
                for (int p = 0; p < kPerThread; ++p)
#pragma unroll
                    for (int k = 0; k < GSE_NX; ++k) nz[p][k] = (jp[p] < Nu) ? (nb + k * ld)[jp[p]] : 0.f;
            } else {
#pragma unroll
                for (int k = 0; k < GSE_NX; ++k) {
                    const float4 t4 = *reinterpret_cast<const float4*>(nb + k * ld + i0);
                    nz[0][k] = t4.x; nz[1][k] = t4.y; nz[2][k] = t4.z; nz[3][k] = t4.w;
                }
            }
        }

        // ---- two halves of two particles each (keeps the live register set small): the noise of a half is drawn
        //      first -- registers only -- and the first half's draw is what hides the latency of the loads above.
        v4f row[GSE_NX];
        float gq[2][GSE_NX];
        float obs[kPerThread][kObsKept];              // what g reads of the (new) state (bioreactor: Cg, Cfa)
        // per-filter inputs: kernel arguments, or two uniform (scalar) loads each when they live on the device
        double fg = a.io.f[b].u0, fm = a.io.f[b].u1, z0 = a.io.f[b].z0, z1 = a.io.f[b].z1;
        if constexpr (kDevIO) {
            if (a.u_dev) { fg = a.u_dev[2 * b]; fm = a.u_dev[2 * b + 1]; }
            if (a.z_dev) { z0 = a.z_dev[2 * b]; z1 = a.z_dev[2 * b + 1]; }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 0) {
                GSE_STAMP(5);
                if (kDraw && draw_at == 3) draw_all();
                GSE_STAMP(6);
                if (kPredict && kInjected) {
#pragma unroll
                    for (int p = 0; p < kPerThread; ++p)
#pragma unroll
                        for (int k = 0; k < GSE_NX; ++k) e[p][k] = nz[p][k];
                }
            }
            // the state rows of this half must have landed now; in the strided mapping the other half's loads are issued
            // right behind the wait so that they fly during this half's ODE step
            if (kStrided) {
                asm volatile("s_waitcnt vmcnt(0)"
                             : "+v"(g[0][0]), "+v"(g[0][1]), "+v"(g[0][2]), "+v"(g[0][3]), "+v"(g[0][4]),
                               "+v"(g[1][0]), "+v"(g[1][1]), "+v"(g[1][2]), "+v"(g[1][3]), "+v"(g[1][4])::"memory");
                if (h == 0) GSE_STAMP(7);
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int k = 0; k < GSE_NX; ++k) gq[q][k] = g[q][k];
                if (h == 0) {
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int k = 0; k < GSE_NX; ++k)
                            GSE_LOAD_DWORD_S(g[q][k], (uint32_t)anc[2 + q] * 4u, xin_b + k * ld);
                    if (kGather) GSE_PRIO_ALU();
                }
            } else if (h == 0) {
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4)::"memory");
                row[0] = r0; row[1] = r1; row[2] = r2; row[3] = r3; row[4] = r4;
            }
            // predict: ODE step + noise.  In the strided mapping the particle's new state goes out at once (dword stores,
            // contiguous across the wavefront) and only the two observed states stay in registers for the update.
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int p = 2 * h + q;
                float s[GSE_NX];
#pragma unroll
                for (int k = 0; k < GSE_NX; ++k) s[k] = kStrided ? gq[q][k] : row[k][p];
                if (kPredict) {
                    model_step(s, fg, fm, a.dt);
#pragma unroll
                    for (int k = 0; k < GSE_NX; ++k) s[k] += e[p][k];
                }
                if (kStrided) {
                    if ((kGather || kPredict) && jp[p] < Nu) {
                        float* xb = x_out + (int64_t)b * GSE_NX * ld;
#pragma unroll
                        for (int k = 0; k < GSE_NX; ++k) store_out<1>(xb + k * ld + jp[p], s[k]);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < GSE_NX; ++k) row[k][p] = s[k];
                }
                model_keep(s, obs[p]);
            }
        }
        // the new state goes out now, ahead of the update phase (consecutive mapping: one dwordx4 per row)
        if (!kStrided && kPredict) {
            float* xb = x_out + (int64_t)b * GSE_NX * ld + i0;
#pragma unroll
            for (int k = 0; k < GSE_NX; ++k) store_out4<1>(xb + k * ld, row[k]);
        }
        // update as a phase of its own: the ODE and the likelihood each need ~20 fp64 constants; interleaving them per
        // particle keeps both sets live, overflows the scalar registers and spills ~100 VGPRs
        __builtin_amdgcn_sched_barrier(0);
        GSE_STAMP(8);
        if (kUpdate && !kGather && kStrided) {
            // (the old weights of a launch that draws but does not resample: fetched here, not ahead of the draw and the
            //  model step -- four more live registers there spill)
            const float* wb = w_in + (int64_t)b * ld;
#pragma unroll
            for (int p = 0; p < kPerThread; ++p) wnew[p] = (jp[p] < Nu) ? wb[jp[p]] : 0.f;
        }
        if (kUpdate) {
#pragma unroll
            for (int p = 0; p < kPerThread; ++p) {
                float y0, y1;
                model_output_kept(obs[p], fg, fm, y0, y1);
                const double lik = mixture_pdf2<kNd>(a.pdf, z0 - (double)y0, z1 - (double)y1);
                if (kScaleBehindExchange) wd[p] = (double)wnew[p] * lik;             // (rounded behind the exchange below)
                else wnew[p] = (float)ldexp((double)wnew[p] * lik, -kexp);            // (kexp = 0: the product itself)
            }
        } else if (kGather) {
            if (a.wscale && tile_id == 0 && tid == 0) a.wscale[b] = 0.0;     // (the offspring weigh 1 / N each)
        }

        // ---- ancestor indices; the uniform weights of a resample without update (an update stores its weights below,
        //      once they are back in consecutive order)
        if (kGather) {
            float* wb = w_out + (int64_t)b * ld;
            int32_t* ib = idx_out ? idx_out + (int64_t)b * ld : nullptr;
#pragma unroll
            for (int p = 0; p < kPerThread; ++p)
                if (jp[p] < Nu) {
                    if (!kUpdate) store_out<2>(wb + jp[p], wnew[p]);
                    if (ib) ib[jp[p]] = anc[p];
                }
        }
    }
    GSE_STAMP(9);
    if (kUpdate) {
        // tile sum of the new weights, always formed in the same order (4 consecutive per thread, then the workgroup
        // tree) so that fused and separate launches leave bit-identical sums; the strided mapping goes through LDS.
        if (kStrided) {
#pragma unroll
            for (int p = 0; p < kPerThread; ++p) wtile[tid + p * kBlock] = wd[p];
            __syncthreads();
            if (a.wscale) kexp = kexp_s;                    // (written by wavefront kScaleWave before it came to this barrier)
            const double2 t01 = *reinterpret_cast<const double2*>(&wtile[tid * kPerThread]);
            const double2 t23 = *reinterpret_cast<const double2*>(&wtile[tid * kPerThread + 2]);
            wnew[0] = (float)ldexp(t01.x, -kexp); wnew[1] = (float)ldexp(t01.y, -kexp);
            wnew[2] = (float)ldexp(t23.x, -kexp); wnew[3] = (float)ldexp(t23.y, -kexp);
        }
        // GSE_OP_WEIGHT_SCALE: the exponent the stored weights now carry (an update on top of an update accumulates)
        if (a.wscale && tile_id == 0 && tid == 0) a.wscale[b] = (kGather ? 0.0 : a.wscale[b]) + (double)kexp;
        // the new weights, four consecutive ones per thread: one dwordx4 store (rows are padded to a multiple of 4)
        if (i0 < N) {
            const v4f w4o = {wnew[0], wnew[1], wnew[2], wnew[3]};
            store_out4<2>(w_out + (int64_t)b * ld + i0, w4o);
        }
        // (the tile sum is DEFINED as the total of the workgroup scan of the per-thread sums -- tile_scan_total -- in
        // every kernel that forms one, so one reduction serves both the sum and the tile-local scan)
        double sc[kPerThread];
        sc[0] = (i0 + 0 < N) ? (double)wnew[0] : 0.0;
        sc[1] = sc[0] + ((i0 + 1 < N) ? (double)wnew[1] : 0.0);
        sc[2] = sc[1] + ((i0 + 2 < N) ? (double)wnew[2] : 0.0);
        sc[3] = sc[2] + ((i0 + 3 < N) ? (double)wnew[3] : 0.0);
        double total;
        const double excl = block_exclusive_scan(sc[3], red, total);
        if (tid == 0) tile_sums[(int64_t)b * T + tile_id] = total;
        if (kScanOut) {
            // the tile sums the NEXT launch's gather stage builds its prefix from (a slot of their own: this launch may
            // still be reading the previous ones)
            if (tid == 0) scan_out_sums[(int64_t)b * T + tile_id] = total;
        }
        if (kScanOut) {
            // ... and this thread's exclusive prefix inside the tile: with the four weights it gives back the tile-local
            // cumulative weights excl + s_p, the expression cumsum_kernel uses (8 bytes per 4 particles)
            double* qo = scan_out + ((int64_t)b * T + tile_id) * kBlock + tid;
            if constexpr ((GSE_WT_STORES & 4) != 0) __hip_atomic_store(qo, excl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *qo = excl;
        }
        if (kScanOut && a.smp_out) {
            // the scale's sample particle of this tile, if it has one (scale_samples_carried), by the thread that holds its new
            // weight.  Its new state is read back from x_out: another thread of this workgroup stored it, ahead of the
            // barriers above.  One scalar test in the tiles without a sample -- all but 64 of them.
            uint32_t index;
            int l;
            if (scale_sample_of_tile(a, (uint32_t)tile0, index, l) && (index >> 2) == (uint32_t)(i0 >> 2)) {
                const float* xs = ((kGather || kPredict) ? x_out : x_in) + (int64_t)b * GSE_NX * ld + index;
                float sn[GSE_NX];
#pragma unroll
                for (int k = 0; k < GSE_NX; ++k) sn[k] = xs[(int64_t)k * ld];
                float* row = a.smp_out + ((int64_t)b * kScaleSamples + l) * kSampleRow;
                scale_sample_store_state(row, sn);
                const int q = (int)(index & 3u);
                row[5] = q == 0 ? wnew[0] : q == 1 ? wnew[1] : q == 2 ? wnew[2] : wnew[3];
            }
        }
    }
    GSE_STAMP(10);
    GSE_BLOCK_TIME(1);
}

// A resample whose ancestors are GIVEN (the replay of a recorded run: tests/golden/closed_loop.npz holds the ancestors the
// reference's own resample chose): offspring j copies particle idx[j] and weighs 1 / N.  One thread per offspring.
__global__ __launch_bounds__(kBlock) void index_gather_kernel(const float* __restrict__ x_in, const int32_t* __restrict__ idx,
                                                              float* __restrict__ x_out, float* __restrict__ w_out, int64_t N,
                                                              int64_t ld) {
    const int b = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= N) return;
    const int64_t anc = min(max((int64_t)idx[(int64_t)b * ld + j], (int64_t)0), N - 1);
    const float* s = x_in + (int64_t)b * GSE_NX * ld + anc;
    float* d = x_out + (int64_t)b * GSE_NX * ld + j;
    float v[GSE_NX];
#pragma unroll
    for (int k = 0; k < GSE_NX; ++k) v[k] = s[k * ld];
#pragma unroll
    for (int k = 0; k < GSE_NX; ++k) d[k * ld] = v[k];
    w_out[(int64_t)b * ld + j] = (float)(1.0 / (double)N);
}

// Tile sums for a resample that does not directly follow an update (weights assigned by the caller).
__global__ __launch_bounds__(kBlock) void tile_sum_kernel(const float* __restrict__ w, double* __restrict__ tile_sums,
                                                          int64_t N, int64_t ldw, int T) {
    __shared__ double red[4];
    const int b = blockIdx.y;
    const int64_t i0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) * kPerThread;
    const float4 wv = load_weights4(w + (int64_t)b * ldw, i0, N, ldw);
    const double mine = masked_sum4(wv.x, wv.y, wv.z, wv.w, i0, N);
    double total;
    block_exclusive_scan(mine, red, total);
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
//   gather          one workgroup per tile of 1024 OUTPUTS: two 64-ary wavefront searches bracket the tile's
//   (stage of        ancestor span in c[], the span is staged in LDS, every output finds its ancestor
//   pf_step_kernel;  (first k with c[k] >= (j + r)/N, the reference's comparison verbatim) by binary search in
//   gather_kernel    LDS and copies it; stores are contiguous in j.           r 8 + 20 B, w 24 B per particle
//   for GS-UKF rows)
//
// Tile prefix computed inside the cumsum kernel (T <= 4096 tiles): every workgroup scans the T tile sums itself
// -- identical code on identical data, hence identical bits in every workgroup -- and picks out
//   Pfx(t), Pfx(t+1), Pfx(T) = total,   Pfx(q) = excl[q / m] + s[(q/m)*m] + ... + s[q-1]
// (thread j owns chunk [j*m, (j+1)*m); excl = workgroup exclusive scan of the chunk sums).  Larger T: a scan
// kernel of its own writes the prefix once.
constexpr int kInlinePrefixMaxT = 4096;
constexpr int kScanMaxT = 4096;              // scan modes of the step kernel: the tile prefix (T + 1 doubles) lives in LDS

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
// Shards of a sharded ensemble (exact global resampling, gse_pf_shard_cumsum): tile_src is the prefix over the tiles of
// ALL shards, this shard's tiles start at tile_offset and the total sits at total_at.
template <bool kInlinePrefix>
__global__ __launch_bounds__(kBlock) void cumsum_kernel(const float* __restrict__ w, double* __restrict__ c,
                                                        const double* __restrict__ tile_src,
                                                        double* __restrict__ totals, double* __restrict__ island,
                                                        int64_t N, int64_t ldw, int T, int64_t prefix_stride,
                                                        int64_t tile_offset, int64_t total_at,
                                                        const double* __restrict__ wscale) {
    __shared__ double red[4];
    __shared__ double pfx[3];
    const int b = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
    double P0, P1, W;
    if (kInlinePrefix) {
        inline_tile_prefix(tile_src + (int64_t)b * T, T, t, red, pfx);
        P0 = pfx[0]; P1 = pfx[1]; W = pfx[2];
    } else {
        const double* pf = tile_src + (int64_t)b * prefix_stride;
        P0 = pf[tile_offset + t]; P1 = pf[tile_offset + t + 1]; W = pf[total_at];
    }
    if (t == 0 && tid == 0 && totals) {
        totals[b] = W;
        if (island) {                          // shard bookkeeping: accumulated log-weight and this cycle's sum
            const double ws_b = wscale ? wscale[b] : 0.0;      // (GSE_RESAMPLE_WEIGHT_SCALE: stored = true * 2^-ws_b)
            island[b] += log(W) + ws_b * 0.69314718055994530942;
            island[gridDim.y + b] = ldexp(W, (int)ws_b);
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

// Outputs j_base .. j_base + count - 1 of an ensemble of n_total particles, whose ancestors all lie among the N keys
// of c (the whole ensemble: j_base = 0, count = n_total = N; one shard of a sharded ensemble: its own keys, the outputs
// that descend from it, and everything written relative to j_base).  w_out may be null.
template <class Payload>
__global__ __launch_bounds__(kBlock) void gather_kernel(const Payload pl, const double* __restrict__ c,
                                                        const double* __restrict__ totals, float* __restrict__ w_out,
                                                        int32_t* __restrict__ idx_out, int64_t N, int64_t ldw,
                                                        const ResampleArgs ra, int64_t j_base, int64_t count,
                                                        int64_t n_total) {
    __shared__ double seg[kSpanMax];
    __shared__ int64_t span[2];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t j0 = j_base + (int64_t)blockIdx.x * kTile;
    const int64_t j1 = min(j0 + kTile, j_base + count);
    const double r = ra.r[b];
    const double Nd = (double)n_total;
    const double W = totals ? totals[b] : 1.0;
    const double* cb = c + (int64_t)b * N;
    const float w_uniform = (float)(1.0 / Nd);
    float* wo = w_out ? w_out + (int64_t)b * ldw - j_base : nullptr;      // indexed by j below
    int32_t* io = idx_out ? idx_out + (int64_t)b * ldw - j_base : nullptr;

    // Weights that do not sum to a positive finite number: the reference's normalised cumsum is all-NaN and its
    // sweep never advances (filter/particle.py:98), so every offspring copies particle 0.
    if (!(W > 0.0) || !(W < INFINITY)) {
        for (int64_t j = j0 + tid; j < j1; j += kBlock) {
            pl.copy(b, 0, j - j_base);
            if (wo) wo[j] = w_uniform;
            if (io) io[j] = 0;
        }
        return;
    }

    // 1. bracket the ancestors of this output tile: wavefront 0 searches for the first output, wavefront 1 for the last
    if (wv < 2) {
        const int64_t j = (wv == 0) ? j0 : j1 - 1;
        const int64_t k = wave_search_ge([&](int64_t q) { return cb[q]; }, N, ((double)j + r) / Nd, lane);
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
            const int lo = span_search(seg, (int)len, u);
            const int64_t anc = klo + lo;
            pl.copy(b, anc, j - j_base);
            if (wo) wo[j] = w_uniform;
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
            pl.copy(b, lo, j - j_base);
            if (wo) wo[j] = w_uniform;
            if (io) io[j] = (int32_t)lo;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Banks small enough for ONE workgroup (N <= 4 tiles): tile sums, tile prefix, normalised cumulative weights and the
// ancestors of every output in a single launch -- what tile_sum_kernel + cumsum_kernel<true> + gather_kernel<NoPayload> do
// in three (each ~5 us of mostly latency at these sizes).  The cumulative weights never leave LDS (32 KB), and they
// are formed by the SAME arithmetic, association included: a workgroup of 1024 threads works as four blocks of 256, each
// doing for its tile what a workgroup of those kernels does (block_exclusive_scan's order of the four wavefront totals, the
// tile prefix as the wavefront scan of the tile sums, the clamp and the pinned last entry), so the ancestors are bit for bit
// those of the three-kernel path.  One workgroup per filter.
// ------------------------------------------------------------------------------------------------
constexpr int kSmallResampleTiles = 4;       // (measured: 21-23 us per GS-UKF cycle against 25-26 up to 4096 Gaussians; at 8192 the
                                             //  lone workgroup loses, 28 against 26.5)
constexpr int kSmallResampleBlock = 1024;

// block_exclusive_scan for a 256-thread quarter of a larger workgroup (lds: this quarter's 4 doubles)
__device__ __forceinline__ double quarter_exclusive_scan(double v, double* lds, double& total) {
    const int lane = threadIdx.x & 63, wq = (threadIdx.x >> 6) & 3;
    const double inc = wave_inclusive_scan(v, lane);
    __syncthreads();
    if (lane == 63) lds[wq] = inc;
    __syncthreads();
    double base = 0.0;
    const double w0 = lds[0], w1 = lds[1], w2 = lds[2], w3 = lds[3];
    if (wq > 0) base = w0;
    if (wq > 1) base += w1;
    if (wq > 2) base += w2;
    total = ((w0 + w1) + w2) + w3;
    return base + wave_shift_up1(inc);
}

template <int kUnused>
__global__ __launch_bounds__(kSmallResampleBlock) void small_resample_kernel(const float* __restrict__ w, float* __restrict__ w_out,
                                                                            int32_t* __restrict__ idx_out, int64_t N,
                                                                            const ResampleArgs ra) {
    __shared__ double c_s[kSmallResampleTiles * kTile];
    __shared__ double red[4 * 4];
    __shared__ double tsum[kSmallResampleTiles];
    __shared__ double pfx[kSmallResampleTiles + 1];
    const int b = blockIdx.x, tid = threadIdx.x, quarter = tid >> 8, vt = tid & 255;
    const int T = (int)((N + kTile - 1) / kTile);
    const float* wb = w + (int64_t)b * N;
    // 1. tile sums (tile_sum_kernel)
    for (int t0 = 0; t0 < T; t0 += 4) {
        const int t = t0 + quarter;
        const int64_t i0 = (int64_t)t * kTile + (int64_t)vt * kPerThread;
        const float4 wv = load_weights4(wb, i0, N, N);                 // (zeros beyond N, tiles beyond T included)
        const double mine = masked_sum4(wv.x, wv.y, wv.z, wv.w, i0, N);
        double total;
        quarter_exclusive_scan(mine, red + 4 * quarter, total);
        if (vt == 0 && t < T) tsum[t] = total;
    }
    __syncthreads();
    // 2. tile prefix (inline_tile_prefix with one tile sum per thread: everything happens in wavefront 0)
    if (tid < 64) {
        const double sv = tid < T ? tsum[tid] : 0.0;
        const double inc = wave_inclusive_scan(sv, tid);
        const double excl = wave_shift_up1(inc);
        if (tid < T) pfx[tid] = excl;
        if (tid == 63) pfx[T] = inc;
    }
    __syncthreads();
    const double W = pfx[T];
    const double Nd = (double)N;
    const float w_uniform = (float)(1.0 / Nd);
    float* wo = w_out + (int64_t)b * N;
    int32_t* io = idx_out + (int64_t)b * N;
    if (!(W > 0.0) || !(W < INFINITY)) {                               // (gather_kernel: every offspring copies Gaussian 0)
        for (int64_t j = tid; j < N; j += kSmallResampleBlock) {
            wo[j] = w_uniform;
            io[j] = 0;
        }
        return;
    }
    // 3. cumulative weights (cumsum_kernel) into LDS
    for (int t0 = 0; t0 < T; t0 += 4) {
        const int t = t0 + quarter;
        const int64_t i0 = (int64_t)t * kTile + (int64_t)vt * kPerThread;
        const float4 wv = load_weights4(wb, i0, N, N);
        double sc[kPerThread];
        sc[0] = (i0 + 0 < N) ? (double)wv.x : 0.0;
        sc[1] = sc[0] + ((i0 + 1 < N) ? (double)wv.y : 0.0);
        sc[2] = sc[1] + ((i0 + 2 < N) ? (double)wv.z : 0.0);
        sc[3] = sc[2] + ((i0 + 3 < N) ? (double)wv.w : 0.0);
        double tile_total;
        const double excl = quarter_exclusive_scan(sc[3], red + 4 * quarter, tile_total);
        if (t < T) {
            const double P0 = pfx[t], P1 = pfx[t + 1];
            const double top = P1 / W;
            const int64_t last = min((int64_t)(t + 1) * kTile, N) - 1;
#pragma unroll
            for (int p = 0; p < kPerThread; ++p)
                if (i0 + p < N) c_s[i0 + p] = (i0 + p == last) ? top : fmin((P0 + (excl + sc[p])) / W, top);
        }
    }
    __syncthreads();
    // 4. ancestors (gather_kernel): first k with c[k] >= (j + r) / N; four outputs of a thread searched side by side
    const double r = ra.r[b];
    const int n32 = (int)N;
    for (int j0 = tid; j0 < n32; j0 += 4 * kSmallResampleBlock) {
        double thr[4];
        int out[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) thr[p] = ((double)min(j0 + p * kSmallResampleBlock, n32 - 1) + r) / Nd;
        span_search4(c_s, n32, thr, out);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int j = j0 + p * kSmallResampleBlock;
            if (j < n32) {
                io[j] = out[p];
                wo[j] = w_uniform;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Larger banks: the ancestors straight from the weights and their tile sums, without materialising the cumulative weights --
// cumsum_kernel<true> + gather_kernel<NoPayload> in one launch (what is saved is a launch and 16 bytes per Gaussian; at these
// sizes mostly the launch).  One workgroup per tile of 1024 OUTPUTS:
//   1. the tile prefix of the whole filter in LDS, the arithmetic of inline_tile_prefix for every tile at once;
//   2. every output finds the TILE of its ancestor: the first t whose pinned last entry c = Pfx(t+1) / W reaches its threshold
//      (a search on products, settled on neighbours with the division the cumulative weights are formed with);
//   3. the tiles that hold an ancestor of this workgroup -- flagged in LDS, one or two as a rule -- are scanned one after the
//      other exactly as cumsum_kernel scans them (256 threads x 4 weights, the same clamp and pinned end), their 1024
//      cumulative weights staged in LDS, and the outputs that belong there finish with the reference's comparison.
// Same doubles, same comparisons: the ancestors of the two-kernel route bit for bit.  T <= 2048 (N <= 2^21 per filter).
// ------------------------------------------------------------------------------------------------
constexpr int kAncestorsMaxT = 2048;

// kQuarters = 1: tile_sums holds, per filter, the four QUARTER sums of every tile (gsukf_predict_update_lanes_kernel leaves them:
// what the four wavefronts of tile_sum_kernel would have found) instead of the tile sums; a tile's sum is ((q0 + q1) + q2) + q3,
// block_exclusive_scan's order.
template <int kQuarters>
__global__ __launch_bounds__(kBlock) void ancestors_kernel(const float* __restrict__ w, const double* __restrict__ tile_sums,
                                                           float* __restrict__ w_out, int32_t* __restrict__ idx_out, int64_t N, int T,
                                                           const ResampleArgs ra) {
    __shared__ double red[4];
    __shared__ double seg[kTile];
    __shared__ int32_t tile_lo_hi[2];
    extern __shared__ double anc_lds[];                // pfx [T + 1] doubles, then the flags [T] bytes
    double* pfx_s = anc_lds;
    unsigned char* flag = reinterpret_cast<unsigned char*>(anc_lds + (T + 1));
    const int b = blockIdx.y, tid = threadIdx.x;
    const float* wb = w + (int64_t)b * N;
    const int64_t j0 = (int64_t)blockIdx.x * kTile;
    const int64_t j1 = min(j0 + kTile, N);
    // 1. tile prefix: thread j owns chunk [j*m, (j+1)*m) of the tile sums (m <= 8)
    {
        const double* sums = tile_sums + (int64_t)b * T * (kQuarters ? 4 : 1);
        const int m = (T + kBlock - 1) / kBlock;
        const int qlo = tid * m;
        const int nq = (int)((N + kBlock - 1) / kBlock);               // quarters that exist (the rest of the last tile: 0)
        double sv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if constexpr (kQuarters) {
                double q4[4];
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const int at = 4 * (qlo + i) + h;
                    q4[h] = (i < m && qlo + i < T && at < nq) ? sums[at] : 0.0;
                }
                sv[i] = ((q4[0] + q4[1]) + q4[2]) + q4[3];
            } else {
                sv[i] = (i < m && qlo + i < T) ? sums[qlo + i] : 0.0;
            }
        }
        double chunk = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) chunk += sv[i];
        double total;
        double run = block_exclusive_scan(chunk, red, total);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < m && qlo + i < T) pfx_s[qlo + i] = run;
            run += sv[i];
        }
        if (tid == 0) pfx_s[T] = total;
        for (int q = tid; q < T; q += kBlock) flag[q] = 0;
        __syncthreads();
    }
    const double W = pfx_s[T];
    const double Nd = (double)N, r = ra.r[b];
    const float w_uniform = (float)(1.0 / Nd);
    float* wo = w_out + (int64_t)b * N;
    int32_t* io = idx_out + (int64_t)b * N;
    if (!(W > 0.0) || !(W < INFINITY)) {                               // (gather_kernel: every offspring copies element 0)
        for (int64_t j = j0 + tid; j < j1; j += kBlock) {
            wo[j] = w_uniform;
            io[j] = 0;
        }
        return;
    }
    // 2. the ancestor's tile of each of this thread's (up to) four outputs j0 + tid + 256 p
    double thr[kPerThread];
    int tj[kPerThread];
#pragma unroll
    for (int p = 0; p < kPerThread; ++p) {
        const int64_t j = j0 + tid + (int64_t)p * kBlock;
        thr[p] = ((double)min(j, j1 - 1) + r) / Nd;
        const double target = thr[p] * W;                              // candidate by products ...
        int lo = 0, hi = T - 1;                                        // first t with Pfx(t + 1) >= target
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (pfx_s[mid + 1] < target) lo = mid + 1; else hi = mid;
        }
        // ... settled with the quotient itself: tile t ends at c = Pfx(t + 1) / W; the ancestor's tile is the first with
        // c >= thr (the last tile always qualifies: W / W = 1 > thr)
        while (lo < T - 1 && pfx_s[lo + 1] / W < thr[p]) ++lo;
        while (lo > 0 && pfx_s[lo] / W >= thr[p]) --lo;
        tj[p] = lo;
        if (j < j1) flag[lo] = 1;
    }
    // outputs ascend, so do their tiles: the first belongs to output j0 (thread 0), the last to output j1 - 1
    if (tid == 0) tile_lo_hi[0] = tj[0];
    {
        const int last_p = (int)((j1 - 1 - j0) / kBlock);
        int t_of_last = tj[0];
#pragma unroll
        for (int p = 1; p < kPerThread; ++p) t_of_last = (p == last_p) ? tj[p] : t_of_last;
        if (tid == (int)((j1 - 1 - j0) % kBlock)) tile_lo_hi[1] = t_of_last;
    }
    __syncthreads();
    const int t_first = tile_lo_hi[0], t_last = tile_lo_hi[1];
    // 3. the flagged tiles, one after the other
    for (int t = t_first; t <= t_last; ++t) {
        if (!flag[t]) continue;                                        // (uniform over the workgroup)
        const int64_t i0 = (int64_t)t * kTile + (int64_t)tid * kPerThread;
        const float4 wv = load_weights4(wb, i0, N, N);
        double sc[kPerThread];
        sc[0] = (i0 + 0 < N) ? (double)wv.x : 0.0;
        sc[1] = sc[0] + ((i0 + 1 < N) ? (double)wv.y : 0.0);
        sc[2] = sc[1] + ((i0 + 2 < N) ? (double)wv.z : 0.0);
        sc[3] = sc[2] + ((i0 + 3 < N) ? (double)wv.w : 0.0);
        double tile_total;
        const double excl = block_exclusive_scan(sc[3], red, tile_total);
        const double P0 = pfx_s[t], P1 = pfx_s[t + 1];
        const double top = P1 / W;
        const int64_t last = min((int64_t)(t + 1) * kTile, N) - 1;
        const int len = (int)(last - (int64_t)t * kTile) + 1;
#pragma unroll
        for (int p = 0; p < kPerThread; ++p)
            if (i0 + p <= last) seg[tid * kPerThread + p] = (i0 + p == last) ? top : fmin((P0 + (excl + sc[p])) / W, top);
        __syncthreads();
#pragma unroll
        for (int p = 0; p < kPerThread; ++p) {
            const int64_t j = j0 + tid + (int64_t)p * kBlock;
            if (j < j1 && tj[p] == t) {
                const int k = span_search(seg, len, thr[p]);
                io[j] = (int32_t)((int64_t)t * kTile + k);
                wo[j] = w_uniform;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// The same ancestors by the INVERSE MAP (round 3): one workgroup per tile of 1024 WEIGHTS.  Instead of every output searching
// for its ancestor -- tile prefix, a bisection for the tile, then per flagged tile a load, a scan, four divisions and a
// bisection in LDS: ten dependent steps of 0.5-1 us on a kernel of 256 workgroups, 10.3 us at 2^18 Gaussians -- every KEY
// says which outputs it owns: x[k] formed exactly as above (and as cumsum_kernel forms it: P0 + (excl + s_p) clamped to the
// tile's pinned last key P1), F(k) = #{ j : u_j <= fl(x[k] / W) } by OffspringCount (gse_device.h: arithmetic, settled with the
// reference's own comparison inside the guard band), and key k writes itself to idx[F(k-1) .. F(k)).  Same keys, the
// comparison the search applies read the other way round: the same ancestors (tests/test_gpu_gsukf.py holds both against the
// oracle and against each other).  The tile's weights and the tile sums are loaded side by side and scanned with ONE
// exchange; what remains is load -> scan -> count -> store.  The stores of a key are its own thread's while they are few, sixteen
// lanes' up to 256, a wavefront's up to 4096 and the workgroup's beyond (a degenerate bank -- one Gaussian owning every output -- is 1 MB of
// stores by one workgroup at 2^18); profiles/r03_gsukf_ancestors_tail.txt times the adversarial patterns on both kernels.
// ------------------------------------------------------------------------------------------------
constexpr int kInlineOffspring = 16;     // a key with at most this many offspring is written by its own thread
constexpr int kGroupOffspring = 256;      // ... this many: by 16 lanes
constexpr int kWaveOffspring = 4096;      // ... this many: by one wavefront; beyond: by the whole workgroup

template <int kQuarters>
__global__ __launch_bounds__(kBlock) void ancestors_scatter_kernel(const float* __restrict__ w, const double* __restrict__ tile_sums,
                                                                   float* __restrict__ w_out, int32_t* __restrict__ idx_out,
                                                                   int64_t N, int T, const ResampleArgs ra) {
    __shared__ double red[8];
    __shared__ int32_t wave_last[4];
    __shared__ int32_t heavy_n, heavier_n;
    __shared__ int32_t heavy_lo[kTile], heavy_hi[kTile], heavy_k[kTile];
    extern __shared__ double anc_lds[];                // pfx [T + 1] doubles
    double* pfx_s = anc_lds;
    const int b = blockIdx.y, tid = threadIdx.x, tq = blockIdx.x;
    const int lane = tid & 63, wvf = tid >> 6;
    const float* wb = w + (int64_t)b * N;
    const int64_t k0 = (int64_t)tq * kTile + (int64_t)tid * kPerThread;
    // both loads in flight together
    const float4 wv = load_weights4(wb, k0, N, N);
    const double* sums = tile_sums + (int64_t)b * T * (kQuarters ? 4 : 1);
    const int m = (T + kBlock - 1) / kBlock;
    const int qlo = tid * m;
    const int nq = (int)((N + kBlock - 1) / kBlock);                   // quarters that exist (the rest of the last tile: 0)
    double sv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if constexpr (kQuarters) {
            double q4[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int at = 4 * (qlo + i) + h;
                q4[h] = (i < m && qlo + i < T && at < nq) ? sums[at] : 0.0;
            }
            sv[i] = ((q4[0] + q4[1]) + q4[2]) + q4[3];
        } else {
            sv[i] = (i < m && qlo + i < T) ? sums[qlo + i] : 0.0;
        }
    }
    double chunk = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) chunk += sv[i];
    double sc[kPerThread];
    sc[0] = (k0 + 0 < N) ? (double)wv.x : 0.0;
    sc[1] = sc[0] + ((k0 + 1 < N) ? (double)wv.y : 0.0);
    sc[2] = sc[1] + ((k0 + 2 < N) ? (double)wv.z : 0.0);
    sc[3] = sc[2] + ((k0 + 3 < N) ? (double)wv.w : 0.0);
    // the tile prefix of the filter and this tile's exclusive scan: block_exclusive_scan's arithmetic for each, one exchange
    double run, excl, W;
    {
        const double inc0 = wave_inclusive_scan(chunk, lane), inc1 = wave_inclusive_scan(sc[3], lane);
        if (lane == 63) {
            red[wvf] = inc0;
            red[4 + wvf] = inc1;
        }
        if (tid == 0) heavy_n = heavier_n = 0;
        __syncthreads();
        const double a0 = red[0], a1 = red[1], a2 = red[2], a3 = red[3];
        double base0 = 0.0, base1 = 0.0;
        if (wvf > 0) { base0 = a0; base1 = red[4]; }
        if (wvf > 1) { base0 += a1; base1 += red[5]; }
        if (wvf > 2) { base0 += a2; base1 += red[6]; }
        W = ((a0 + a1) + a2) + a3;
        run = base0 + wave_shift_up1(inc0);
        excl = base1 + wave_shift_up1(inc1);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (i < m && qlo + i < T) pfx_s[qlo + i] = run;
        run += sv[i];
    }
    if (tid == 0) pfx_s[T] = W;
    __syncthreads();
    const double Nd = (double)N, r = ra.r[b];
    const float w_uniform = (float)(1.0 / Nd);
    float* wo = w_out + (int64_t)b * N;
    int32_t* io = idx_out + (int64_t)b * N;
    const bool degenerate = !(W > 0.0) || !(W < INFINITY);             // (gather_kernel: every offspring copies element 0)
    // the new weights of this tile's outputs (and their ancestors when the weights do not sum to a positive finite number)
#pragma unroll
    for (int p = 0; p < kPerThread; ++p)
        if (k0 + p < N) {
            wo[k0 + p] = w_uniform;
            if (degenerate) io[k0 + p] = 0;
        }
    if (degenerate) return;
    const double P0 = pfx_s[tq], P1 = pfx_s[tq + 1];
    const int32_t tile_last = (int32_t)(min((int64_t)(tq + 1) * kTile, N) - 1);
    OffspringCount F;
    F.init(N, r, W);
    const int32_t n32 = (int32_t)N, kk = (int32_t)k0;
    double xs[kPerThread];
    int32_t fs[kPerThread];
    bool near[kPerThread];
#pragma unroll
    for (int p = 0; p < kPerThread; ++p) {
        xs[p] = (kk + p >= tile_last) ? P1 : fmin(P0 + (excl + sc[p]), P1);
        fs[p] = F.candidate(xs[p], near[p]);
    }
    if (near[0] || near[1] || near[2] || near[3]) {
#pragma unroll
        for (int p = 0; p < kPerThread; ++p)
            if (near[p]) fs[p] = F.settle(xs[p], fs[p]);
    }
#pragma unroll
    for (int p = 0; p < kPerThread; ++p) fs[p] = min(max(fs[p] + 1, 0), n32);
    // F of the key before this thread's first: the lane below, the wavefront below, the tile below (whose last key is P0)
    if (lane == 63) wave_last[wvf] = fs[kPerThread - 1];
    const int32_t tile_first_f = tq ? F(P0) : 0;
    __syncthreads();
    int32_t prev = (int32_t)wave_shift_up1((uint32_t)fs[kPerThread - 1]);
    if (lane == 0) prev = wvf ? wave_last[wvf - 1] : min(max(tile_first_f, 0), n32);
#pragma unroll
    for (int p = 0; p < kPerThread; ++p) {
        const int32_t lo = prev, hi = max(fs[p], prev);
        prev = hi;
        if (kk + p > tile_last || hi == lo) continue;
        if (hi - lo <= kInlineOffspring) {
            // (the common case: a handful; the first four without a loop)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (lo + q < hi) io[lo + q] = kk + p;
            for (int32_t j = lo + 4; j < hi; ++j) io[j] = kk + p;
        } else {
            // two lists in one array: up to kGroupOffspring from the bottom, beyond from the top
            const int slot = (hi - lo <= kGroupOffspring) ? atomicAdd(&heavy_n, 1) : kTile - 1 - atomicAdd(&heavier_n, 1);
            heavy_lo[slot] = lo;
            heavy_hi[slot] = hi;
            heavy_k[slot] = kk + p;
        }
    }
    __syncthreads();
    const int n_heavy = heavy_n, n_heavier = heavier_n;
    // keys with up to kGroupOffspring offspring: 16 lanes each, sixteen keys side by side ...
    for (int e = tid >> 4; e < n_heavy; e += kBlock / 16) {
        const int32_t lo = heavy_lo[e], hi = heavy_hi[e], k = heavy_k[e];
        for (int32_t j = lo + (tid & 15); j < hi; j += 16) io[j] = k;
    }
    // ... up to kWaveOffspring: one wavefront each ...
    for (int e = wvf; e < n_heavier; e += kBlock / 64) {
        const int32_t lo = heavy_lo[kTile - 1 - e], hi = heavy_hi[kTile - 1 - e], k = heavy_k[kTile - 1 - e];
        if (hi - lo > kWaveOffspring) continue;
        for (int32_t j = lo + lane; j < hi; j += 64) io[j] = k;
    }
    // ... and the few that own thousands: the whole workgroup, 16 bytes per store
    for (int e0 = 0; e0 < n_heavier; ++e0) {
        const int e = kTile - 1 - e0;
        const int32_t k = heavy_k[e], count = heavy_hi[e] - heavy_lo[e];
        if (count <= kWaveOffspring) continue;
        int32_t* p = io + heavy_lo[e];
        // up to three single entries to a 16-byte boundary, four entries per store from there, the rest singly
        const int head = min(count, (int)(((16u - (uint32_t)(reinterpret_cast<unsigned long long>(p) & 15ull)) >> 2) & 3u));
        if (tid < head) p[tid] = k;
        const int body4 = (count - head) >> 2;
        int4* q = reinterpret_cast<int4*>(p + head);
        for (int i = tid; i < body4; i += kBlock) q[i] = make_int4(k, k, k, k);
        const int done = head + 4 * body4;
        if (tid < count - done) p[done + tid] = k;
    }
}

// ------------------------------------------------------------------------------------------------
// moments
// ------------------------------------------------------------------------------------------------
// Per-workgroup partial sums of {w, w d, w d d^T} with d = x - pivot, fp64; blocks stride over the tiles.
__global__ __launch_bounds__(kBlock) void pf_moments_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            double* __restrict__ partials, int64_t N, int64_t ld) {
    __shared__ double red[4 * 21];
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
    block_sums<21>(acc, red, [&](int c, double total) { out[c] = total; });
}

// Moments of the resampled ensemble without resampling (gse_pf_moments_resampled): as pf_moments_kernel, but particle k
// weighs n_k * fp32(1 / N) with n_k = F(k) - F(k-1) its offspring count under the systematic resample with this r -- the
// inverse map of the gather stage (OffspringCount; keys re-formed from weights + per-thread tile prefixes exactly as there).
// A workgroup builds the tile prefix of its filter in LDS, then its wavefronts stride over the tiles.
// Round 2 tried to get this pass off its 188 vector registers (21 fp64 accumulators per lane: two wavefronts per SIMD) by
// COLLECTING the particles with offspring in a per-wavefront LDS list and working them off term-parallel (one accumulator
// per lane, 116 registers, four wavefronts per SIMD): 32 us against 40 when 5 % of the particles have offspring, but inside
// the closed loop -- where 3 % to 70 % of them do, 16 % on average -- 59 us against 53: pushing and flushing cost what the
// occupancy wins.  Fetching a step's state rows whole and coalesced instead of each such particle's own 20 bytes, and a
// software pipeline over the steps (inputs requested one step ahead, states used one step late) changed nothing either way.
// Round 3 kept the 21 accumulators per lane and compacted only the WORK: per step of 256 particles a wavefront lists the
// (particle, count) pairs with offspring in LDS -- ballots and population counts, so the order is the particles' own and the
// sums do not depend on timing -- and works the list off 64 at a time, every lane busy.  Taken where they stand, each of a
// lane's four particles costs the whole wavefront the 60 fp64 instructions of the accumulation whenever any lane has
// offspring there, i.e. always: 400 vector instructions a step against about 230 now, and with one particle's states in
// flight instead of four the kernel needs 124 registers for 204 (four wavefronts per SIMD for two): 55.6 -> 47.0 us per
// tick of 32 x 2^18 particles.  Not better: twice the workgroups (52.2), the next step's inputs requested a step ahead
// (48.7), full groups of 64 carried over the steps (48.5).
// Round 5: the four quarters of a tile no longer take their turns (weights and prefixes requested, counted, the states of the
// particles with offspring fetched, accumulated -- four dependent trips to memory per tile, which is what a filter of 2^20
// particles alone spent its 14.5 us on: 2048 wavefronts with one tile each); a wavefront now requests the weights and prefixes of
// the WHOLE tile at once, counts all sixteen particles of a lane, lists every particle with offspring of the tile, and works the
// list off 128 at a time with both halves' states in flight.  Fuller groups of 64 too (a tile has ~170 such particles inside
// a closed loop: three groups where the quarters made four).
// kQuarters: the slot is in QUARTER FORMAT (gse_pf_cluster_kernels.h): tile_sums = [B][T][4] quarter sums, qx_all the quads'
// exclusive prefixes inside their quarter; the tile sum and the tile-level prefix are formed as block_exclusive_scan forms them.
// kSplit (1 or 2): a wavefront's unit of work is 1 / kSplit of a tile.  A filter ALONE on the device (gse_pf_merge_begin: one shard
// of 2^20 particles) is a chain of dependent trips to memory with ONE wavefront per SIMD and tile; halving the unit doubles the
// wavefronts that wait side by side.  The half that does not begin a tile finds the count below its first key by re-forming
// the key before it (the last key of the previous quarter: that thread's weights and prefix, the same operations).
#ifndef GSE_MOMRES_PREFETCH
#define GSE_MOMRES_PREFETCH 1    // unit i + 1's weights and prefixes requested before unit i's list is worked off
#endif
template <bool kQuarters, int kSplit = 1>
__global__ __launch_bounds__(kBlock) void pf_moments_resampled_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                      const double* __restrict__ tile_sums,
                                                                      const double* __restrict__ qx_all,
                                                                      double* __restrict__ partials, int64_t N, int64_t ld,
                                                                      int T, const ResampleArgs ra,
                                                                      uint32_t* __restrict__ wstatus, const MergeTail tail,
                                                                      uint32_t* __restrict__ counter) {
    __shared__ double red[4 * 21];
    __shared__ int2 items[kBlock / 64][kTile / kSplit];          // per wavefront: (particle, offspring count) of one unit
    __shared__ int last_s;
    extern __shared__ double pfx_m[];              // [T + 1]
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* xb = x + (int64_t)b * GSE_NX * ld;
    const float* wb = w + (int64_t)b * ld;
    const double* qx = qx_all + (int64_t)b * T * kBlock;
    constexpr int kSubs = kTile / (64 * kPerThread) / kSplit;    // quarters of a tile per unit: 4 or 2
    static_assert(kSplit == 1 || kSplit == 2, "a unit is a tile or half a tile");
    // this wavefront's first unit: weights and prefixes requested before the tile prefix is built (they do not depend on it)
    const int unit0 = blockIdx.x * (kBlock / 64) + wv, units = T * kSplit;
    // (requests go into the `next` set: a wavefront with several units -- many filters sharing the launch -- asks for unit i + 1
    //  while it works unit i's list off, round 5)
    float4 wq[kSubs], wq_next[kSubs];
    double ex[kSubs], ex_next[kSubs];
    float4 wq_before = make_float4(0.f, 0.f, 0.f, 0.f);           // kSplit == 2, second half: the quad before the unit's first
    double ex_before = 0.0;
    auto request = [&](int unit) {
        const int tq = unit / kSplit, sub0 = (unit % kSplit) * kSubs;
#pragma unroll
        for (int sub = 0; sub < kSubs; ++sub) {
            const int t = (sub0 + sub) * 64 + lane;
            const int64_t k0 = (int64_t)tq * kTile + (int64_t)t * kPerThread;
            wq_next[sub] = load_weights4(wb, k0, N, ld);
            ex_next[sub] = qx[tq * kBlock + t];
        }
        if (kSplit > 1 && sub0 > 0) {                             // (uniform addresses: one request for the wavefront)
            const int t = sub0 * 64 - 1;
            wq_before = load_weights4(wb, (int64_t)tq * kTile + (int64_t)t * kPerThread, N, ld);
            ex_before = qx[tq * kBlock + t];
        }
    };
    GSE_MSTAMP(0, blockIdx.x == GSE_STAMPS);
    if (unit0 < units) request(unit0);
    // tile prefix (the arithmetic of the step kernel's gather stage)
    {
        const double* sums = tile_sums + (int64_t)b * T * (kQuarters ? 4 : 1);
        const int m = (T + kBlock - 1) / kBlock;
        const int qlo = tid * m;
        double sv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (kQuarters) {
                const double* q = sums + 4 * (qlo + i);
                sv[i] = (i < m && qlo + i < T) ? ((q[0] + q[1]) + q[2]) + q[3] : 0.0;
            } else {
                sv[i] = (i < m && qlo + i < T) ? sums[qlo + i] : 0.0;
            }
        }
        double chunk = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) chunk += sv[i];
        double total;
        double run = block_exclusive_scan(chunk, red, total);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (i < m && qlo + i < T) pfx_m[qlo + i] = run;
            run += sv[i];
        }
        if (tid == 0) pfx_m[T] = total;
        __syncthreads();
    }
    const double W = pfx_m[T];
    GSE_MSTAMP(1, blockIdx.x == GSE_STAMPS);
    if (blockIdx.x == 0 && tid == 0) {
        // the slot's weight sum beside the partials (column 21 of the first row: gse_pf_merge_begin forms the shard's
        // log-weight from it), and the filter's sticky status when it is not a number the resample can use
        partials[(int64_t)b * kMomentBlocks * kMomentWidth + 21] = W;
        if (!(W < INFINITY)) wstatus[b] = GSE_WSTATUS_SUM_NOT_FINITE;
    }
    double pv[GSE_NX];
#pragma unroll
    for (int k = 0; k < GSE_NX; ++k) pv[k] = (double)xb[k * ld];          // pivot = particle 0
    double acc[21];
#pragma unroll
    for (int c = 0; c < 21; ++c) acc[c] = 0.0;
    const bool valid = (W > 0.0) && (W < INFINITY);
    OffspringCount F;
    F.init(N, ra.r[b], valid ? W : 1.0);
    const double wu = (double)(float)(1.0 / (double)N);
    auto add_particle = [&](const float (&xv)[GSE_NX], int count) {
        const double wp = (double)count * wu;                            // (count 0: adds zeros)
        double d[GSE_NX];
#pragma unroll
        for (int k = 0; k < GSE_NX; ++k) d[k] = (double)xv[k] - pv[k];
        acc[0] += wp;
        int c = 6;
#pragma unroll
        for (int k = 0; k < GSE_NX; ++k) {
            const double wd = wp * d[k];
            acc[1 + k] += wd;
#pragma unroll
            for (int l = k; l < GSE_NX; ++l) acc[c++] += wd * d[l];
        }
    };
    // One WAVEFRONT per tile.  The count below a lane's first key comes from the lane before (DPP) and, for lane 0, from the
    // previous quarter's last lane (kept in a scalar) -- no barrier, and the four wavefronts of the workgroup are at different
    // tiles.  The states of a particle are fetched only if it has offspring: under the usual degenerate weights that is about
    // one particle in a hundred, so the pass moves ~6 bytes per particle (weights and tile prefixes) instead of 26.
    int2* mine = &items[wv][0];
    const int unit_stride = gridDim.x * (kBlock / 64);
    for (int unit = unit0; valid && unit < units; unit += unit_stride) {
#pragma unroll
        for (int sub = 0; sub < kSubs; ++sub) { wq[sub] = wq_next[sub]; ex[sub] = ex_next[sub]; }
        const int tq = unit / kSplit, sub0 = (unit % kSplit) * kSubs;
        const double P0 = pfx_m[tq], P1 = pfx_m[tq + 1];
        const int64_t tile_last = min((int64_t)(tq + 1) * kTile, N) - 1;
        // the quarters before quarter q, as block_exclusive_scan adds them (quarter format)
        auto quarter_base = [&](int q_at) {
            const double* q = tile_sums + ((int64_t)b * T + tq) * 4;
            double qb = 0.0;
            if (q_at > 0) qb = q[0];
            if (q_at > 1) qb += q[1];
            if (q_at > 2) qb += q[2];
            return qb;
        };
        double key_before = P0;                                          // (the previous tile's last key is pinned to P0)
        if (kSplit > 1 && sub0 > 0) {
            // the last key of the quarter before this unit, exactly as the wavefront that owns it forms it
            const int t = sub0 * 64 - 1;
            const int64_t k0 = (int64_t)tq * kTile + (int64_t)t * kPerThread;
            double exb = ex_before;
            if (kQuarters) exb = quarter_base(sub0 - 1) + exb;
            double sb = (k0 + 0 < N) ? (double)wq_before.x : 0.0;
            sb = sb + ((k0 + 1 < N) ? (double)wq_before.y : 0.0);
            sb = sb + ((k0 + 2 < N) ? (double)wq_before.z : 0.0);
            sb = sb + ((k0 + 3 < N) ? (double)wq_before.w : 0.0);
            key_before = (k0 + 3 >= tile_last) ? P1 : fmin(P0 + (exb + sb), P1);
        }
        int32_t below = F(key_before);
        int base = 0;
#pragma unroll
        for (int sub = 0; sub < kSubs; ++sub) {
            const int t = (sub0 + sub) * 64 + lane;
            const int64_t k0 = (int64_t)tq * kTile + (int64_t)t * kPerThread;
            double excl = ex[sub];
            if (kQuarters) excl = quarter_base(sub0 + sub) + excl;
            double sc[kPerThread];
            sc[0] = (k0 + 0 < N) ? (double)wq[sub].x : 0.0;
            sc[1] = sc[0] + ((k0 + 1 < N) ? (double)wq[sub].y : 0.0);
            sc[2] = sc[1] + ((k0 + 2 < N) ? (double)wq[sub].z : 0.0);
            sc[3] = sc[2] + ((k0 + 3 < N) ? (double)wq[sub].w : 0.0);
            double xs[kPerThread];
            int32_t fs[kPerThread];
            bool near[kPerThread];
#pragma unroll
            for (int p = 0; p < kPerThread; ++p) {
                xs[p] = (k0 + p >= tile_last) ? P1 : fmin(P0 + (excl + sc[p]), P1);
                fs[p] = F.candidate(xs[p], near[p]);
            }
            if (near[0] || near[1] || near[2] || near[3]) {
#pragma unroll
                for (int p = 0; p < kPerThread; ++p)
                    if (near[p]) fs[p] = F.settle(xs[p], fs[p]);
            }
#pragma unroll
            for (int p = 0; p < kPerThread; ++p) fs[p] += 1;             // counts
            int32_t prev = (int32_t)wave_shift_up1((uint32_t)fs[3]);
            if (lane == 0) prev = below;
            below = __builtin_amdgcn_readlane(fs[3], 63);
            // The particles with offspring -- one in six inside a closed loop -- are COMPACTED through the wavefront's list in
            // LDS (ballots and population counts: the order is the particles' own, the sums do not depend on timing): taken
            // where they stand, each of a lane's particles costs the whole wavefront the 60 fp64 instructions of the
            // accumulation if any lane has offspring there, i.e. always.
#pragma unroll
            for (int p = 0; p < kPerThread; ++p) {
                const int32_t n_off = (k0 + p <= tile_last) ? fs[p] - (p == 0 ? prev : fs[p - 1]) : 0;
                const unsigned long long m = __ballot(n_off > 0);
                if (n_off > 0) mine[base + __popcll(m & ((1ull << lane) - 1ull))] = make_int2((int)(k0 + p), n_off);
                base += __popcll(m);
            }
        }
        if (GSE_MOMRES_PREFETCH && kSplit == 1 && unit + unit_stride < units)
            request(unit + unit_stride);                                                 // (in flight while the list is worked off)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        GSE_MSTAMP(2, blockIdx.x == GSE_STAMPS);
#ifdef GSE_STAMPS
        if (blockIdx.x == GSE_STAMPS && blockIdx.y == 0 && threadIdx.x == 0) g_mstamps[16] = (unsigned long long)base;
#endif
        for (int at = 0; at < base; at += 128) {
            const bool two = at + 64 < base;                             // (uniform)
            const int2 it0 = (at + lane < base) ? mine[at + lane] : make_int2(0, 0);
            const int2 it1 = (at + 64 + lane < base) ? mine[at + 64 + lane] : make_int2(0, 0);
            float xv0[GSE_NX], xv1[GSE_NX];
#pragma unroll
            for (int k = 0; k < GSE_NX; ++k) xv0[k] = xb[k * ld + it0.x];
            if (two) {
#pragma unroll
                for (int k = 0; k < GSE_NX; ++k) xv1[k] = xb[k * ld + it1.x];
            }
            add_particle(xv0, it0.y);
            if (two) add_particle(xv1, it1.y);
        }
        __builtin_amdgcn_wave_barrier();                                  // (the list is rewritten for the next unit)
        if ((kSplit > 1 || !GSE_MOMRES_PREFETCH) && unit + unit_stride < units) request(unit + unit_stride);
    }
    GSE_MSTAMP(3, blockIdx.x == GSE_STAMPS);
    if (!valid && blockIdx.x == 0 && tid == 0) {
        // weights that do not sum to a positive finite number: every offspring copies particle 0 (see the gather stage)
        acc[0] = (double)N * wu;
    }
    double* out = partials + ((int64_t)b * kMomentBlocks + blockIdx.x) * kMomentWidth;
    if (!(tail.flags & kMergeTailOn)) {                                  // (uniform)
        block_sums_dpp<21>(acc, red, [&](int c, double total) { out[c] = total; });
        return;
    }
    // ---- gse_pf_merge_begin (round 5): the LAST workgroup to get here finishes the partial sums and runs the merge's tail --
    //      the shard's row of the merge buffer, and for a single shard the combination and the estimate -- instead of a second
    //      launch (a kernel boundary costs ~3 us of drain and start; the whole estimate is to fit beside an 18 us cycle).
    //      The partials cross XCDs (L2s that do not see each other): they are written and read as AGENT-scope atomics, which
    //      go through to memory, each workgroup waits for its own stores before it draws its ticket (agent-scope atomic add),
    //      and the last one therefore finds them all.  (Fences instead -- __threadfence on both sides -- write back and
    //      invalidate the L2 in every workgroup: 32 us for the launch where this takes 19.)  The sums are added in a FIXED
    //      order, whoever is last.  gridDim.x <= 256 kSplit (the host sees to it), B = 1.
    block_sums_dpp<21>(acc, red, [&](int c, double total) {
        __hip_atomic_store(out + c, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    });
    GSE_MSTAMP(4, blockIdx.x == GSE_STAMPS);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");               // (the stores have left: s_waitcnt, no cache maintenance)
    __syncthreads();
    GSE_MSTAMP(5, blockIdx.x == GSE_STAMPS);
    if (tid == 0)
        last_s = (__hip_atomic_fetch_add(counter + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u) ? 1 : 0;
    __syncthreads();
    GSE_MSTAMP(6, blockIdx.x == GSE_STAMPS);
    if (!last_s) return;
    GSE_MSTAMP(8, true);
    if (tid == 0) __hip_atomic_store(counter + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (ready for the next launch)
    double* part = reinterpret_cast<double*>(&items[0][0]);             // [8][32] (the lists are spent)
    double* sm = red;                                                    // [32] the shard's moments
    double* sx = red + 32;                                               // [32] its row
    {
        const int c = tid & 31, g = tid >> 5;                            // 8 groups x 32 columns (21 used)
        const double* pb = partials + (int64_t)b * kMomentBlocks * kMomentWidth;
        const int nblk = (int)gridDim.x;
        constexpr int kPer = 32 * kSplit;
        double v[kPer];
#pragma unroll
        for (int j = 0; j < kPer; ++j) {
            const int q = g + 8 * j;
            v[j] = (c < 21 && q < nblk) ? __hip_atomic_load(pb + (int64_t)q * kMomentWidth + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                        : 0.0;
        }
        // (the one transcendental of the tail -- the shard's log-weight absorbs log(W) -- while those loads are in flight)
        if (tid == 0) red[64] = (tail.flags & GSE_MERGE_PENDING) ? log(W) : 0.0;
#pragma unroll
        for (int span = 1; span < kPer; span *= 2)
#pragma unroll
            for (int j = 0; j < kPer; j += 2 * span) v[j] += v[j + span];
        part[g * 32 + c] = v[0];
    }
    GSE_MSTAMP(9, true);
    for (int i = tid; i < tail.P * 22; i += kBlock)
        if (i / 22 != tail.rank) tail.rows[i] = 0.0;
    __syncthreads();
    if (tid < 21) {
        const double t = ((part[tid] + part[32 + tid]) + (part[64 + tid] + part[96 + tid])) +
                         ((part[128 + tid] + part[160 + tid]) + (part[192 + tid] + part[224 + tid]));
        sm[tid] = t;
        tail.moments26[tid] = t;
    }
    if (tid >= 32 && tid < 32 + GSE_NX) {
        const double pvt = (double)xb[(tid - 32) * ld];
        sm[21 + tid - 32] = pvt;
        tail.moments26[21 + tid - 32] = pvt;
    }
    __syncthreads();
    GSE_MSTAMP(10, true);
    if (tid < 64) merge_tail_wave(tail, sm, sx, red[64], tid);
    GSE_MSTAMP(11, true);
}

// Fixed-order reduction of the per-workgroup partials -> out[b][0..width); appends pivot_count pivot values
// (moments_finish_block, gse_device.h: the closed loop's fused tail kernel runs the same arithmetic).
__global__ __launch_bounds__(64 * kFinishGroups) void moments_finish_kernel(const double* __restrict__ partials, int nblk,
                                                                           int width, double* __restrict__ out,
                                                                           int out_stride,
                                                                           const float* __restrict__ pivot_src,
                                                                           int64_t pivot_filter_stride,
                                                                           int64_t pivot_elem_stride, int pivot_at,
                                                                           int pivot_count) {
    __shared__ double part[kFinishGroups][64];
    moments_finish_block(partials + (int64_t)blockIdx.x * kMomentBlocks * kMomentWidth, kMomentWidth, nblk, width, out, out_stride,
                         pivot_src, pivot_filter_stride, pivot_elem_stride, pivot_at, pivot_count, blockIdx.x, part);
}

}  // namespace gse
