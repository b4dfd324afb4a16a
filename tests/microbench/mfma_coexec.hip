// Microbenchmark (gfx950): does v_mfma_f32_32x32x2_f32 co-execute with VALU / LDS work of the SAME wave or of the
// SIMD PARTNER wave?  Build: hipcc --offload-arch=gfx950 -O3 mfma_coexec.hip -o mfma_coexec ; run on the GPU box.
// One workgroup of 512 threads per CU (8 waves: waves w and w+4 share a SIMD), 156 KB dynamic LDS to force 1 WG/CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode: what waves 0-3 do / what waves 4-7 do / per-iteration filler inside the MFMA waves
// A: 0 = 32 MFMA per iter, 1 = nothing
// B (partner): 0 = same as A, 1 = idle (exit), 2 = VALU only (64 v_fma), 3 = LDS reads only (16 ds_read_b128), 4 = LDS writes (8 b128)
// F (filler inside MFMA waves, interleaved one per MFMA): 0 none, 1 = 1 VALU per MFMA, 2 = 1 ds_read_b128 per 2 MFMA, 3 = 2 VALU per MFMA
template <int B, int F>
__global__ __launch_bounds__(512, 2) void k(float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = tid * 0.001f, b = 1.0f + lane * 0.01f;
    f32x4 v = {a, b, a, b}, t = {0, 0, 0, 0};
    const char *lp = lds + lane * 16 + wave * 1024;
    bool mf = wave < 4 || B == 0;
    if (!mf && B == 1) return;
    for (int it = 0; it < iters; ++it) {
        if (mf) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                acc[i & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i & 7], 0, 0, 0);
                if (F == 1) { v.x = v.x * b + a; }
                if (F == 3) { v.x = v.x * b + a; v.y = v.y * b + a; }
                if (F == 2 && (i & 1)) { t += *(const f32x4 *)(lp + (i >> 1) * 8192 % 65536); }
            }
        } else if (B == 2) {
#pragma unroll
            for (int i = 0; i < 64; ++i) { v.x = v.x * b + a; v.y = v.y * b + a; v.z = v.z * b + a; v.w = v.w * b + a; }
        } else if (B == 3) {
#pragma unroll
            for (int i = 0; i < 16; ++i) t += *(const f32x4 *)(lp + i * 8192 % 131072);
        } else if (B == 4) {
#pragma unroll
            for (int i = 0; i < 8; ++i) *(f32x4 *)(lds + lane * 16 + wave * 1024 + i * 8192) = v + t;
        }
    }
    float s = v.x + v.y + v.z + v.w + t.x + t.y + t.z + t.w;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][7];
    if (s == 1234.5f) out[tid] = s;
}

// MIX bits: 1 = 8 conflict-free ds_read_b128, 2 = 8 four-way-conflicted ds_read_b128, 4 = 4 ds_write_b128,
// 8 = 4 global_load_dwordx4 (L2-resident), 16 = 32 VALU, 32 = __syncthreads per iteration, 64 = 3 more global loads + 3 ds_write (staging)
template <int MIX>
__global__ __launch_bounds__(512, 2) void kmix(float *out, const float *g, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = tid * 0.001f, b = 1.0f + lane * 0.01f;
    f32x4 v = {a, b, a, b}, t = {0, 0, 0, 0}, u[4] = {v, v, v, v}, xs[3] = {v, v, v}, un[4] = {v, v, v, v};
    const char *lp = lds + lane * 16 + wave * 1024;
    const char *lc = lds + 65536 + (lane & 31) * 256 + (lane >> 5) * 16 + wave * 32;   // 256-B stride: 16 lanes of a b128 group on 1 bank quad... 4+ way
    char *lw = lds + 131072 - 8192 + lane * 16 + wave * 1024;
    const float *gp = g + (size_t)blockIdx.x % 8 * 262144 + wave * 4096 + lane * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            acc[i & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[i & 3].x + a, b, acc[i & 7], 0, 0, 0);
            if ((MIX & 1) && (i & 3) == 0) t += *(const f32x4 *)(lp + (i >> 2) * 8192);
            if ((MIX & 2) && (i & 3) == 1) t += *(const f32x4 *)(lc + (i >> 2) * 4096 % 32768);
            if ((MIX & 4) && (i & 7) == 6) *(f32x4 *)(lw + (i >> 3) * 65536 % 8192) = t;
            if ((MIX & 8) && (i & 7) == 2) u[i >> 3] = *(const f32x4 *)(gp + ((it * 4 + (i >> 3)) & 63) * 1024);
            if ((MIX & 64) && (i & 7) == 5 && i < 24) xs[i >> 3] = *(const f32x4 *)(gp + 65536 + ((it * 4 + (i >> 3)) & 63) * 1024);
            if ((MIX & 64) && (i & 7) == 7 && i < 24) *(f32x4 *)(lw - 16384 + (i >> 3) * 1024) = xs[i >> 3];
            if ((MIX & 128) && (i & 1) && i < 8) un[i >> 1] = *(const f32x4 *)(gp + ((it * 4 + (i >> 1)) & 63) * 1024);
            if (MIX & 16) v.x = v.x * b + a;
        }
        if (MIX & 32) __syncthreads();
        if (MIX & 128) { u[0] = un[0]; u[1] = un[1]; u[2] = un[2]; u[3] = un[3]; }
    }
    float s = v.x + v.y + v.z + v.w + t.x + t.y + t.z + t.w + u[0].x + u[1].y + u[2].z + u[3].w + xs[0].x + xs[1].x + xs[2].x;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][7];
    if (s == 1234.5f) out[tid] = s;
}

// 16x16x4 twin of kmix (two MFMAs of half the size per slot: same FLOPs, half the accumulator write-back)
// MIX bits: 1 = 8 conflict-free ds_read_b128, 2 = 8 four-way-conflicted ds_read_b128, 4 = 4 ds_write_b128,
// 8 = 4 global_load_dwordx4 (L2-resident), 16 = 32 VALU, 32 = __syncthreads per iteration, 64 = 3 more global loads + 3 ds_write (staging)
template <int MIX>
__global__ __launch_bounds__(512, 2) void kmix16(float *out, const float *g, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    f32x4 acc[32];
    for (int i = 0; i < 32; ++i) for (int e = 0; e < 4; ++e) acc[i][e] = 0.f;
    float a = tid * 0.001f, b = 1.0f + lane * 0.01f;
    f32x4 v = {a, b, a, b}, t = {0, 0, 0, 0}, u[4] = {v, v, v, v}, xs[3] = {v, v, v};
    const char *lp = lds + lane * 16 + wave * 1024;
    const char *lc = lds + 65536 + (lane & 31) * 256 + (lane >> 5) * 16 + wave * 32;   // 256-B stride: 16 lanes of a b128 group on 1 bank quad... 4+ way
    char *lw = lds + 131072 - 8192 + lane * 16 + wave * 1024;
    const float *gp = g + (size_t)blockIdx.x % 8 * 262144 + wave * 4096 + lane * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[i & 3].x + a, b, acc[i], 0, 0, 0);
            acc[(i + 16) & 31] = __builtin_amdgcn_mfma_f32_16x16x4f32(u[i & 3].y + a, b, acc[(i + 16) & 31], 0, 0, 0);
            if ((MIX & 1) && (i & 3) == 0) t += *(const f32x4 *)(lp + (i >> 2) * 8192);
            if ((MIX & 2) && (i & 3) == 1) t += *(const f32x4 *)(lc + (i >> 2) * 4096 % 32768);
            if ((MIX & 4) && (i & 7) == 6) *(f32x4 *)(lw + (i >> 3) * 65536 % 8192) = t;
            if ((MIX & 8) && (i & 7) == 2) u[i >> 3] = *(const f32x4 *)(gp + ((it * 4 + (i >> 3)) & 63) * 1024);
            if ((MIX & 64) && (i & 7) == 5 && i < 24) xs[i >> 3] = *(const f32x4 *)(gp + 65536 + ((it * 4 + (i >> 3)) & 63) * 1024);
            if ((MIX & 64) && (i & 7) == 7 && i < 24) *(f32x4 *)(lw - 16384 + (i >> 3) * 1024) = xs[i >> 3];
            if (MIX & 16) v.x = v.x * b + a;
        }
        if (MIX & 32) __syncthreads();
    }
    float s = v.x + v.y + v.z + v.w + t.x + t.y + t.z + t.w + u[0].x + u[1].y + u[2].z + u[3].w + xs[0].x + xs[1].x + xs[2].x;
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][3];
    if (s == 1234.5f) out[tid] = s;
}


// One wave per SIMD: 4 waves per workgroup, 16 accumulator tiles (256 registers, AGPRs) per wave, 64 MFMAs per iteration --
// the same MFMA work per SIMD and iteration as kmix with both partner waves, every other instruction issued by the
// MFMA wave itself.  MIX bits as kmix but scaled to keep the per-SIMD totals: 1 = 8 conflict-free ds_read_b128 (A
// fragments: one wave reads them for both N halves), 2 = 16 conflicted ds_read_b128, 4 = 8 ds_write_b128,
// 8 = 8 global_load_dwordx4, 16 = 64 VALU, 32 = barrier, 64 = 6 global loads + 6 ds_write (staging)
template <int MIX>
__global__ __launch_bounds__(256) void kmix1(float *out, const float *g, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = tid * 0.001f, b = 1.0f + lane * 0.01f;
    f32x4 v = {a, b, a, b}, t = {0, 0, 0, 0}, u[8] = {v, v, v, v, v, v, v, v}, xs[6] = {v, v, v, v, v, v};
    const char *lp = lds + lane * 16 + wave * 1024;
    const char *lc = lds + 65536 + (lane & 31) * 256 + (lane >> 5) * 16 + wave * 32;
    char *lw = lds + 131072 - 8192 + lane * 16 + wave * 1024;
    const float *gp = g + (size_t)blockIdx.x % 8 * 262144 + wave * 8192 + lane * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            acc[i & 15] = __builtin_amdgcn_mfma_f32_32x32x2f32(u[i & 7].x + a, b, acc[i & 15], 0, 0, 0);
            if ((MIX & 1) && (i & 7) == 0) t += *(const f32x4 *)(lp + (i >> 3) * 8192);
            if ((MIX & 2) && (i & 3) == 1) t += *(const f32x4 *)(lc + (i >> 2) * 4096 % 32768);
            if ((MIX & 4) && (i & 7) == 6) *(f32x4 *)(lw + (i >> 3) * 65536 % 8192) = t;
            if ((MIX & 8) && (i & 7) == 2) u[i >> 3] = *(const f32x4 *)(gp + ((it * 8 + (i >> 3)) & 63) * 1024);
            if ((MIX & 64) && (i & 7) == 5 && i < 48) xs[i >> 3] = *(const f32x4 *)(gp + 65536 + ((it * 8 + (i >> 3)) & 63) * 1024);
            if ((MIX & 64) && (i & 7) == 7 && i < 48) *(f32x4 *)(lw - 16384 + (i >> 3) * 1024) = xs[i >> 3];
            if (MIX & 16) v.x = v.x * b + a;
        }
        if (MIX & 32) __syncthreads();
    }
    float s = v.x + v.y + v.z + v.w + t.x + t.y + t.z + t.w;
    for (int i = 0; i < 8; ++i) s += u[i].x;
    for (int i = 0; i < 6; ++i) s += xs[i].x;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][7];
    if (s == 1234.5f) out[tid] = s;
}

template <int MIX>
void runmix1(const char *name, float *d, const float *g) {
    hipFuncSetAttribute((const void *)kmix1<MIX>, hipFuncAttributeMaxDynamicSharedMemorySize, 159744);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kmix1<MIX><<<256, 256, 159744>>>(d, g, 10);
    hipEventRecord(e0);
    kmix1<MIX><<<256, 256, 159744>>>(d, g, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double tf = (double)256 * 4 * iters * 64 * 4096 / (ms * 1e-3) / 1e12;
    printf("1 wave/SIMD mix %3d %-46s %8.3f ms  %6.1f TFLOP/s  cycles/iter/SIMD@2.4GHz %.0f\n", MIX, name, ms, tf, ms * 1e-3 * 2.4e9 / iters);
}


// Dependent-accumulator distance: 32 MFMAs per iteration where MFMA i accumulates into acc[(i / RUN) % 8], i.e. RUN
// back-to-back MFMAs on the same accumulator before moving on (RUN = 1: distance 8 as in k<>, RUN = 4: chains of four).
template <int RUN>
__global__ __launch_bounds__(512, 2) void kdep(float *out, int iters) {
    const int tid = threadIdx.x, lane = tid & 63;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = tid * 0.001f, b = 1.0f + lane * 0.01f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[(i / RUN) & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[(i / RUN) & 7], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][7];
    if (s == 1234.5f) out[tid] = s;
}
template <int RUN>
void rundep(float *d) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kdep<RUN><<<256, 512>>>(d, 10);
    hipEventRecord(e0);
    kdep<RUN><<<256, 512>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double tf = (double)256 * 8 * iters * 32 * 4096 / (ms * 1e-3) / 1e12;
    printf("dependent chains of %d MFMAs on one accumulator (2 waves/SIMD): %8.3f ms  %6.1f TFLOP/s\n", RUN, ms, tf);
}

template <int MIX>
void runmix16(const char *name, float *d, const float *g) {
    hipFuncSetAttribute((const void *)kmix16<MIX>, hipFuncAttributeMaxDynamicSharedMemorySize, 159744);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kmix16<MIX><<<256, 512, 159744>>>(d, g, 10);
    hipEventRecord(e0);
    kmix16<MIX><<<256, 512, 159744>>>(d, g, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double tf = (double)256 * 8 * iters * 64 * 2048 / (ms * 1e-3) / 1e12;
    printf("16x16x4 mix %3d %-42s %8.3f ms  %6.1f TFLOP/s  cycles/iter/SIMD@2.4GHz %.0f\n", MIX, name, ms, tf, ms * 1e-3 * 2.4e9 / iters);
}

template <int MIX>
void runmix(const char *name, float *d, const float *g) {
    hipFuncSetAttribute((const void *)kmix<MIX>, hipFuncAttributeMaxDynamicSharedMemorySize, 159744);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kmix<MIX><<<256, 512, 159744>>>(d, g, 10);
    hipEventRecord(e0);
    kmix<MIX><<<256, 512, 159744>>>(d, g, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double tf = (double)256 * 8 * iters * 32 * 4096 / (ms * 1e-3) / 1e12;
    printf("mix %3d %-50s %8.3f ms  %6.1f TFLOP/s  cycles/iter/SIMD@2.4GHz %.0f\n", MIX, name, ms, tf, ms * 1e-3 * 2.4e9 / iters);
}

template <int B, int F>
void run(const char *name, float *d, int waves_mfma) {
    hipFuncSetAttribute((const void *)k<B, F>, hipFuncAttributeMaxDynamicSharedMemorySize, 159744);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<B, F><<<256, 512, 159744>>>(d, 10);
    hipEventRecord(e0);
    k<B, F><<<256, 512, 159744>>>(d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mfma = (double)256 * waves_mfma * iters * 32;
    double tf = mfma * 4096 / (ms * 1e-3) / 1e12;
    printf("%-58s %8.3f ms  %6.1f TFLOP/s (%.1f%% of 157.3)  cycles/iter/SIMD@2.4GHz %.0f\n", name, ms, tf, tf / 157.3 * 100, ms * 1e-3 * 2.4e9 / iters);
}

int main() {
    float *d; hipMalloc(&d, 4096);
    run<1, 0>("MFMA x32, one wave per SIMD (partner exits)", d, 4);
    run<0, 0>("MFMA x32 on both partner waves", d, 8);
    run<2, 0>("MFMA wave + partner 256 VALU/iter", d, 4);
    run<3, 0>("MFMA wave + partner 16 ds_read_b128/iter", d, 4);
    run<4, 0>("MFMA wave + partner 8 ds_write_b128/iter", d, 4);
    run<1, 1>("one wave/SIMD: MFMA + 1 VALU per MFMA", d, 4);
    run<1, 3>("one wave/SIMD: MFMA + 2 VALU per MFMA", d, 4);
    run<1, 2>("one wave/SIMD: MFMA + 1 ds_read_b128 per 2 MFMA", d, 4);
    run<0, 1>("both waves: MFMA + 1 VALU per MFMA", d, 8);
    run<0, 3>("both waves: MFMA + 2 VALU per MFMA", d, 8);
    run<0, 2>("both waves: MFMA + 1 ds_read_b128 per 2 MFMA", d, 8);
    float *g; hipMalloc(&g, 8 * 262144 * 4 + 1048576); hipMemset(g, 0, 8 * 262144 * 4 + 1048576);
    runmix<0>("MFMA only (both waves)", d, g);
    runmix<1>("+8 ds_read_b128 (conflict-free)", d, g);
    runmix<2>("+8 ds_read_b128 (bank-conflicted)", d, g);
    runmix<4>("+4 ds_write_b128", d, g);
    runmix<8>("+4 global_load_dwordx4", d, g);
    runmix<16>("+32 VALU", d, g);
    runmix<32>("+barrier per iteration", d, g);
    runmix<64>("+3 global loads + 3 ds_write (staging)", d, g);
    runmix<1 + 2 + 4 + 16>("LDS+VALU mix, no barrier", d, g);
    runmix<1 + 2 + 4 + 16 + 32>("LDS+VALU mix + barrier", d, g);
    runmix<1 + 2 + 4 + 8 + 16 + 32>("full chunk mix (no staging)", d, g);
    runmix<1 + 2 + 4 + 8 + 16 + 32 + 64>("full chunk mix + staging", d, g);
    runmix16<0>("MFMA only (both waves)", d, g);
    runmix16<1>("+8 ds_read_b128", d, g);
    runmix16<8>("+4 global_load_dwordx4", d, g);
    runmix16<16>("+32 VALU", d, g);
    runmix16<1 + 2 + 4 + 16>("LDS+VALU mix, no barrier", d, g);
    runmix16<1 + 4 + 8 + 16 + 32>("kernel-like mix: 8 reads, 4 writes, 4 loads, 32 VALU, barrier", d, g);
    runmix<1 + 4 + 8 + 16 + 32>("(32x32x2) kernel-like mix: same", d, g);
    runmix<128>("+4 global loads consumed an iteration later", d, g);
    runmix<128 + 32>("+4 latency-hidden global loads + barrier", d, g);
    runmix<128 + 1 + 4 + 16>("hidden loads + 8 reads + 4 writes + 32 VALU", d, g);
    runmix<128 + 1 + 4 + 16 + 32>("hidden loads + 8 reads + 4 writes + 32 VALU + barrier", d, g);
    runmix<1 + 4 + 16 + 32>("8 reads + 4 writes + 32 VALU + barrier", d, g);
    runmix1<0>("MFMA only", d, g);
    runmix1<1>("+8 ds_read_b128", d, g);
    runmix1<2>("+16 ds_read_b128 (bank-conflicted)", d, g);
    runmix1<4>("+8 ds_write_b128", d, g);
    runmix1<8>("+8 global_load_dwordx4", d, g);
    runmix1<16>("+64 VALU", d, g);
    runmix1<32>("+barrier", d, g);
    runmix1<1 + 2 + 4 + 16>("LDS+VALU mix, no barrier", d, g);
    runmix1<1 + 2 + 4 + 8 + 16 + 32>("full chunk mix (no staging)", d, g);
    runmix1<1 + 2 + 4 + 8 + 16 + 32 + 64>("full chunk mix + staging", d, g);
    rundep<1>(d); rundep<2>(d); rundep<4>(d); rundep<32>(d);
    return 0;
}
