// Microbenchmark: what does a launch that leaves ~33 MB of freshly written lines behind cost, by store flavour?
// build: hipcc --offload-arch=gfx950 -O3 -o /tmp/store_tail tools/micro/store_tail.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
using v4f = float __attribute__((ext_vector_type(4)));

template <int kMode>   // 0 empty, 1 plain dwordx4, 2 sc1 dwordx4, 3 plain dword (wave-contiguous), 4 sc1 dword, 5 nt dwordx4
__global__ __launch_bounds__(256) void k(float* out, int rows, long ld, int spin) {
    const long i0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    float acc = (float)threadIdx.x;
    for (int s = 0; s < spin; ++s) acc = acc * 1.0001f + 0.5f;        // some arithmetic in front of the stores
    if (kMode == 0) { if (acc == 12345.f) out[0] = acc; return; }
    for (int r = 0; r < rows; ++r) {
        float* p = out + r * ld + i0;
        v4f v = {acc, acc + 1, acc + 2, acc + 3};
        if (kMode == 1) *reinterpret_cast<v4f*>(p) = v;
        if (kMode == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
        if (kMode == 5) __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p));
        if (kMode == 3 || kMode == 4) {
            float* q = out + r * ld + (long)blockIdx.x * 1024 + threadIdx.x;
            for (int j = 0; j < 4; ++j) {
                if (kMode == 3) q[j * 256] = v[j];
                else asm volatile("global_store_dword %0, %1, off sc1" ::"v"(q + j * 256), "v"(v[j]) : "memory");
            }
        }
    }
}

template <int kMode>
void run(const char* name, float* buf, int rows, long ld, int spin) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<kMode>, dim3(1024), dim3(256), 0, 0, buf, rows, ld, spin);
    hipEventRecord(a, 0);
    const int reps = 200;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k<kMode>, dim3(1024), dim3(256), 0, 0, buf, rows, ld, spin);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%-34s rows %d spin %5d: %7.2f us per launch\n", name, rows, spin, ms * 1000.f / reps);
}

int main() {
    const long ld = 1 << 20;
    float* buf; hipMalloc(&buf, ld * 4 * 16);
    for (int spin : {0, 4000}) {
        run<0>("empty", buf, 8, ld, spin);
        run<1>("plain dwordx4, 32 MB", buf, 8, ld, spin);
        run<2>("sc1 dwordx4, 32 MB", buf, 8, ld, spin);
        run<5>("nt dwordx4, 32 MB", buf, 8, ld, spin);
        run<3>("plain dword, 32 MB", buf, 8, ld, spin);
        run<4>("sc1 dword, 32 MB", buf, 8, ld, spin);
        run<1>("plain dwordx4, 8 MB", buf, 2, ld, spin);
        run<2>("sc1 dwordx4, 8 MB", buf, 2, ld, spin);
    }
    return 0;
}
