// MFMA issue-rate microbenchmark for the weight-gradient kernel's accumulator layout: 20 independent 32x32x2 fp32 MFMAs per step on 320
// accumulators (15 tiles in AGPRs by the builtin, 5 pinned to VGPRs by inline asm), 120 steps, one wave per SIMD, 256 workgroups.
//   hipcc --offload-arch=gfx950 -O3 tests/microbench/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>   // 0: all builtin (compiler keeps what it can in AGPRs); 1: tile 3 of each group through inline asm "+v" with s_nop 1
__global__ __launch_bounds__(256, 1) void k(float *out, int steps, float seed) {
    f32x16 acc[5][4];
    for (int j = 0; j < 5; ++j) for (int f = 0; f < 4; ++f) for (int e = 0; e < 16; ++e) acc[j][f][e] = 0.0f;
    float g[4], v = seed + threadIdx.x;
    for (int m = 0; m < 4; ++m) g[m] = seed * (m + 1);
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                if (MODE == 1 && f == 3) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[j][f]) : "v"(g[f]), "v"(v));
                else acc[j][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[f], v, acc[j][f], 0, 0, 0);
            }
    }
    float r = 0.0f;
    for (int j = 0; j < 5; ++j) for (int f = 0; f < 4; ++f) for (int e = 0; e < 16; ++e) r += acc[j][f][e];
    if (r == 12345.678f) out[threadIdx.x] = r;
}
template <int MODE> void run(const char *name) {
    float *d; hipMalloc(&d, 4096);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, d, 120, 1.0f);
    hipEventRecord(a); for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, d, 120, 1.0f); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("%s: %.1f us per launch of 256 workgroups x 120 steps x 20 MFMAs (ideal at 2.3 GHz: 66.8 us)\n", name, ms / 20 * 1e3);
}
int main() { run<0>("all builtin"); run<1>("five tiles through inline asm"); run<0>("all builtin"); run<1>("five tiles through inline asm"); return 0; }
