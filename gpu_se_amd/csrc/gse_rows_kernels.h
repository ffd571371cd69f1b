// gse_rows_kernels.h -- the row-wise model kernels (f / g on plain [M][5] rows), in a header of their own for the same
// reason as gse_pf_kernels.h: they are also compiled at run time for user models.
#pragma once

#include "gse_host.h"

namespace gse {

// out[i] = f(x[i], u, dt): the Euler INCREMENT, stored fp32 (what f_vectorize returns, particle.py:176-191)
__global__ __launch_bounds__(kBlock) void model_f_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t M,
                                                         double fg, double fm, double dt) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= M) return;
    float s[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) s[k] = x[i * 5 + k];
    double dx[5];
    model_increment(s, fg, fm, dt, dx);
#pragma unroll
    for (int k = 0; k < 5; ++k) out[i * 5 + k] = (float)dx[k];
}

__global__ __launch_bounds__(kBlock) void model_g_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t M,
                                                         double u0, double u1) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= M) return;
    float s[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) s[k] = x[i * 5 + k];
    float y0, y1;
    model_output(s, u0, u1, y0, y1);
    out[i * 2 + 0] = y0;
    out[i * 2 + 1] = y1;
}

}  // namespace gse
