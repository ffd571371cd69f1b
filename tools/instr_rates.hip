// Issue cost of the vector instructions the GS-UKF lane kernel is made of, measured: W wavefronts per SIMD, each a stream
// of eight independent chains of one instruction.  Prints nanoseconds per instruction per SIMD and the ratio to v_fma_f32.
//   hipcc --offload-arch=gfx950 -O2 tools/instr_rates.hip -o /tmp/instr_rates && /tmp/instr_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int kIters = 2048, kChains = 8;

#define KERNEL_F32(name, text)                                                                                       \
    __global__ __launch_bounds__(64) void name(float* out, float seed) {                                             \
        float v[kChains];                                                                                            \
        for (int i = 0; i < kChains; ++i) v[i] = seed + threadIdx.x + i;                                             \
        float a = seed * 0.5f, b = seed * 0.25f;                                                                     \
        for (int it = 0; it < kIters; ++it) {                                                                        \
            _Pragma("unroll") for (int i = 0; i < kChains; ++i) asm volatile(text : "+v"(v[i]) : "v"(a), "v"(b));    \
        }                                                                                                            \
        float s = 0;                                                                                                 \
        for (int i = 0; i < kChains; ++i) s += v[i];                                                                 \
        out[blockIdx.x * 64 + threadIdx.x] = s;                                                                      \
    }
#define KERNEL_F64(name, text)                                                                                       \
    __global__ __launch_bounds__(64) void name(float* out, float seed) {                                             \
        double v[kChains];                                                                                           \
        for (int i = 0; i < kChains; ++i) v[i] = seed + threadIdx.x + i;                                             \
        double a = seed * 0.5, b = seed * 0.25;                                                                      \
        for (int it = 0; it < kIters; ++it) {                                                                        \
            _Pragma("unroll") for (int i = 0; i < kChains; ++i) asm volatile(text : "+v"(v[i]) : "v"(a), "v"(b));    \
        }                                                                                                            \
        double s = 0;                                                                                                \
        for (int i = 0; i < kChains; ++i) s += v[i];                                                                 \
        out[blockIdx.x * 64 + threadIdx.x] = (float)s;                                                               \
    }
#define KERNEL_U32(name, text, clob)                                                                                 \
    __global__ __launch_bounds__(64) void name(float* out, float seed) {                                             \
        uint32_t v[kChains];                                                                                         \
        for (int i = 0; i < kChains; ++i) v[i] = (uint32_t)seed + threadIdx.x + i;                                   \
        uint32_t a = (uint32_t)seed * 77u + 3u, b = (uint32_t)seed * 13u + 1u;                                       \
        for (int it = 0; it < kIters; ++it) {                                                                        \
            _Pragma("unroll") for (int i = 0; i < kChains; ++i) asm volatile(text : "+v"(v[i]) : "v"(a), "v"(b) : clob);   \
        }                                                                                                            \
        uint32_t s = 0;                                                                                              \
        for (int i = 0; i < kChains; ++i) s += v[i];                                                                 \
        out[blockIdx.x * 64 + threadIdx.x] = (float)s;                                                               \
    }
#define KERNEL_U64(name, text, clob)                                                                                 \
    __global__ __launch_bounds__(64) void name(float* out, float seed) {                                             \
        uint64_t v[kChains];                                                                                         \
        for (int i = 0; i < kChains; ++i) v[i] = (uint64_t)seed + threadIdx.x + i;                                   \
        uint32_t a = (uint32_t)seed * 77u + 3u, b = (uint32_t)seed * 13u + 1u;                                       \
        for (int it = 0; it < kIters; ++it) {                                                                        \
            _Pragma("unroll") for (int i = 0; i < kChains; ++i) asm volatile(text : "+v"(v[i]) : "v"(a), "v"(b) : clob);   \
        }                                                                                                            \
        uint64_t s = 0;                                                                                              \
        for (int i = 0; i < kChains; ++i) s += v[i];                                                                 \
        out[blockIdx.x * 64 + threadIdx.x] = (float)s;                                                               \
    }
// conversions: the chain alternates through a second register
__global__ __launch_bounds__(64) void k_cvt_f64_f32(float* out, float seed) {
    float v[kChains];
    double d[kChains];
    for (int i = 0; i < kChains; ++i) v[i] = seed + threadIdx.x + i;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < kChains; ++i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(v[i]));
    }
    double s = 0;
    for (int i = 0; i < kChains; ++i) s += d[i];
    out[blockIdx.x * 64 + threadIdx.x] = (float)s;
}
__global__ __launch_bounds__(64) void k_cvt_f32_f64(float* out, float seed) {
    float v[kChains];
    double d[kChains];
    for (int i = 0; i < kChains; ++i) d[i] = seed + threadIdx.x + i;
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < kChains; ++i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(v[i]) : "v"(d[i]));
    }
    float s = 0;
    for (int i = 0; i < kChains; ++i) s += v[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

KERNEL_F32(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL_F32(k_mul_f32, "v_mul_f32 %0, %0, %1")
KERNEL_F32(k_sqrt_f32, "v_sqrt_f32 %0, %0")
KERNEL_F32(k_rcp_f32, "v_rcp_f32 %0, %0")
KERNEL_F32(k_sin_f32, "v_sin_f32 %0, %0")
KERNEL_F32(k_log_f32, "v_log_f32 %0, %0")
KERNEL_F64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL_F64(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
KERNEL_F64(k_pk_add_f32, "v_pk_add_f32 %0, %0, %1")
KERNEL_F64(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
KERNEL_F64(k_mul_f64, "v_mul_f64 %0, %0, %1")
KERNEL_F64(k_add_f64, "v_add_f64 %0, %0, %1")
KERNEL_F64(k_max_f64, "v_max_f64 %0, %0, %1")
KERNEL_F64(k_rcp_f64, "v_rcp_f64 %0, %0")
KERNEL_U32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1", "memory")
KERNEL_U32(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1", "memory")
KERNEL_U32(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1", "memory")
KERNEL_U32(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2", "memory")
KERNEL_U32(k_bitop3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96", "memory")
KERNEL_U32(k_xor, "v_xor_b32 %0, %0, %1", "memory")
KERNEL_U32(k_add_u32, "v_add_u32 %0, %0, %1", "memory")
KERNEL_U32(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %1", "memory")
KERNEL_U64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0", "vcc")

struct Case { const char* name; void (*fn)(float*, float); };

int main() {
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, dev));
    const int simds = prop.multiProcessorCount * 4;
    printf("%s: %d CUs, %d SIMDs, clock %d kHz\n", prop.name, prop.multiProcessorCount, simds, prop.clockRate);
    float* out;
    CHECK(hipMalloc(&out, (size_t)simds * 8 * 64 * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::vector<Case> cases = {
        {"v_fma_f32", k_fma_f32}, {"v_mul_f32", k_mul_f32}, {"v_pk_fma_f32", k_pk_fma_f32}, {"v_pk_mul_f32", k_pk_mul_f32},
        {"v_pk_add_f32", k_pk_add_f32}, {"v_sqrt_f32", k_sqrt_f32}, {"v_rcp_f32", k_rcp_f32}, {"v_sin_f32", k_sin_f32},
        {"v_log_f32", k_log_f32}, {"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64},
        {"v_max_f64", k_max_f64}, {"v_rcp_f64", k_rcp_f64}, {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_cvt_f32_f64", k_cvt_f32_f64},
        {"v_mad_u64_u32", k_mad_u64_u32}, {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32},
        {"v_mul_u32_u24", k_mul_u32_u24}, {"v_mad_u32_u24", k_mad_u32_u24}, {"v_bitop3_b32", k_bitop3}, {"v_xor_b32", k_xor},
        {"v_add_u32", k_add_u32}, {"v_lshl_add_u32", k_lshl_add},
    };
    const int waves_per_simd[] = {1, 2, 4};
    double base[3] = {0, 0, 0};
    printf("%-16s %12s %12s %12s   (ns per instruction per SIMD; x = relative to v_fma_f32)\n", "instruction", "1 wave/SIMD", "2 waves/SIMD",
           "4 waves/SIMD");
    for (auto& c : cases) {
        printf("%-16s", c.name);
        for (int wi = 0; wi < 3; ++wi) {
            const int W = waves_per_simd[wi];
            const int blocks = simds * W;
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                hipLaunchKernelGGL(c.fn, dim3(blocks), dim3(64), 0, 0, out, 1.0f);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            const double ns = (double)best * 1e6 / ((double)kIters * kChains * W);
            if (base[wi] == 0) base[wi] = ns;
            printf("  %6.3f x%4.2f", ns, ns / base[wi]);
        }
        printf("\n");
        fflush(stdout);
    }
    CHECK(hipFree(out));
    return 0;
}
