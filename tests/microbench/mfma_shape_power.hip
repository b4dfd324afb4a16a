// Microbenchmark (gfx950): sustained fp32 MFMA rate of v_mfma_f32_32x32x2_f32 vs v_mfma_f32_16x16x4_f32 on RANDOM operands,
// 160 accumulator registers per wave, two waves per SIMD, every CU busy, seconds-long runs -- the boards are power-capped, and
// the accumulator traffic per FLOP of the 16x16x4 form is half that of the 32x32x2 form (K = 4 instead of 2 per C read/D write).
// Build: hipcc --offload-arch=gfx950 -O3 mfma_shape_power.hip -o mfma_shape_power ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>   // 0: 32x32x2 (10 tiles of 16 regs), 1: 16x16x4 (40 tiles of 4 regs)
__global__ __launch_bounds__(256, 2) void k(const float *src, float *out, int iters) {
    const int tid = threadIdx.x + blockIdx.x * 256;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = src[(tid * 8 + i) & 0xFFFFF]; b[i] = src[(tid * 8 + i + 77777) & 0xFFFFF]; }
    float s = 0.f;
    if (SHAPE == 0) {
        f32x16 acc[10];
        for (int i = 0; i < 10; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 40; ++i) acc[i % 10] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i & 7], b[(i * 3) & 7], acc[i % 10], 0, 0, 0);
        }
        for (int i = 0; i < 10; ++i) for (int e = 0; e < 16; ++e) s += acc[i][e];
    } else {
        f32x4 acc[40];
        for (int i = 0; i < 40; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 80; ++i) acc[i % 40] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 7], b[(i * 3) & 7], acc[i % 40], 0, 0, 0);
        }
        for (int i = 0; i < 40; ++i) for (int e = 0; e < 4; ++e) s += acc[i][e];
    }
    if (s == 1234.5f) out[tid] = s;
}

int main() {
    const int N = 1 << 20;
    std::vector<float> h(N);
    srand(1);
    for (auto &x : h) x = (rand() / (float)RAND_MAX) * 2.f - 1.f;
    float *src, *out;
    hipMalloc(&src, N * 4); hipMalloc(&out, 512 * 256 * 4);
    hipMemcpy(src, h.data(), N * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;   // per launch: 512 blocks x 4 waves x iters x 40 (or 80) MFMAs
    for (int rep = 0; rep < 3; ++rep)
        for (int shape = 0; shape < 2; ++shape) {
            const int launches = 12;
            hipEventRecord(e0);
            for (int l = 0; l < launches; ++l) {
                if (shape == 0) hipLaunchKernelGGL(k<0>, dim3(512), dim3(256), 0, 0, src, out, iters);
                else hipLaunchKernelGGL(k<1>, dim3(512), dim3(256), 0, 0, src, out, iters);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = 512.0 * 4 * iters * 40 * 4096.0 * launches;   // 40 x 4096 == 80 x 2048
            printf("rep %d %s  %8.1f ms  %7.1f TFLOP/s\n", rep, shape == 0 ? "32x32x2" : "16x16x4", ms, flops / ms / 1e9);
            fflush(stdout);
        }
    return 0;
}
