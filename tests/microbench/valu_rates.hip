// Microbenchmark (gfx950): issue cost of the instruction kinds the conv epilogue is made of, ONE wave per SIMD (256-thread
// workgroup, one per CU, as k_wino_conv<4> runs).  Cycles per instruction from s_memtime around 512 back-to-back independent
// instructions of one kind.   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int KIND>
__global__ __launch_bounds__(256, 1) void k(unsigned long long *out, float *sink, float *gbuf, int active_mod) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x % active_mod != 0) { if (lane == 0) out[blockIdx.x * 4 + wave] = 0; return; }   // only every active_mod-th CU works
    float v[8];
    f32x2 p[8];
    f32x4 q[8];
    for (int i = 0; i < 8; ++i) { v[i] = tid * 0.5f + i; p[i] = f32x2{v[i], v[i] + 1.0f}; q[i] = f32x4{v[i], 1.0f, 2.0f, 3.0f}; }
    float a0 = tid, a1 = tid + 1.f;
    f32x2 c2 = {1.0001f, 0.9999f};
    const unsigned lp = wave * 36864 + lane * 16;      // dynamic LDS starts at offset 0 (the kernel has no static LDS)
    for (int o = tid * 16; o < 147456; o += 4096) *(f32x4 *)(lds + o) = q[0];
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0;
    // accumulators in AGPRs for the accvgpr kinds
    float acc0, acc1, acc2, acc3, acc4, acc5, acc6, acc7;
    asm volatile("v_accvgpr_write_b32 %0, %8\n v_accvgpr_write_b32 %1, %8\n v_accvgpr_write_b32 %2, %8\n v_accvgpr_write_b32 %3, %8\n"
                 "v_accvgpr_write_b32 %4, %8\n v_accvgpr_write_b32 %5, %8\n v_accvgpr_write_b32 %6, %8\n v_accvgpr_write_b32 %7, %8\n"
                 : "=a"(acc0), "=a"(acc1), "=a"(acc2), "=a"(acc3), "=a"(acc4), "=a"(acc5), "=a"(acc6), "=a"(acc7) : "v"(a0));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)gbuf, 0, 512 * 1024 * 1024, 0x00020000);
    const unsigned goff = (blockIdx.x * 256 + tid) * 16;
    for (int rep = 0; rep < 3; ++rep) {
        __builtin_amdgcn_s_barrier();
        t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
        for (int it = 0; it < 8; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (KIND == 0) {
#define X(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a1));
                    REP8(X)
#undef X
                } else if (KIND == 1) {
#define X(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
                    REP8(X)
#undef X
                } else if (KIND == 2) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(c2));
                    REP8(X)
#undef X
                } else if (KIND == 3) {
#define X(i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(a1));
                    REP8(X)
#undef X
                } else if (KIND == 4) {
                    asm volatile("v_accvgpr_read_b32 %0, %8\n v_accvgpr_read_b32 %1, %9\n v_accvgpr_read_b32 %2, %10\n v_accvgpr_read_b32 %3, %11\n"
                                 "v_accvgpr_read_b32 %4, %12\n v_accvgpr_read_b32 %5, %13\n v_accvgpr_read_b32 %6, %14\n v_accvgpr_read_b32 %7, %15\n"
                                 : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]), "=v"(v[6]), "=v"(v[7])
                                 : "a"(acc0), "a"(acc1), "a"(acc2), "a"(acc3), "a"(acc4), "a"(acc5), "a"(acc6), "a"(acc7));
                } else if (KIND == 5) {
#define X(i) asm volatile("v_max_f32 %0, 0, %0" : "+v"(v[i]));
                    REP8(X)
#undef X
                } else if (KIND == 6) {       // 8 ds_write_b128, lane-linear
#define X(i) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(lp), "v"(q[i]), "n"(i * 1024) : "memory");
                    REP8(X)
#undef X
                } else if (KIND == 7) {       // 8 ds_read_b128, lane-linear
#define X(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i]) : "v"(lp), "n"(i * 1024) : "memory");
                    REP8(X)
#undef X
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                } else if (KIND == 8) {       // pk_add on values just read from AGPRs (dependent pairs as in the epilogue)
                    asm volatile("v_accvgpr_read_b32 %0, %2\n v_accvgpr_read_b32 %1, %3" : "=v"(v[0]), "=v"(v[1]) : "a"(acc0), "a"(acc1));
                    asm volatile("v_accvgpr_read_b32 %0, %2\n v_accvgpr_read_b32 %1, %3" : "=v"(v[2]), "=v"(v[3]) : "a"(acc2), "a"(acc3));
                    p[0] = f32x2{v[0], v[1]}; p[1] = f32x2{v[2], v[3]};
                    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[2]) : "v"(p[0]), "v"(p[1]));
                    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[3]) : "v"(p[2]), "v"(p[1]));
                    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[4]) : "v"(p[3]), "v"(p[0]));
                    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(p[5]) : "v"(p[4]), "v"(p[2]));
                } else if (KIND == 9) {       // 8 buffer_store_dwordx4 (16 B per lane, lane-linear 1 KB per instruction)
#define X(i) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen offset:%3" ::"v"(q[i]), "v"(goff), "s"(rs), "n"(i * 16 * 0) : "memory");
                    REP8(X)
#undef X
                } else if (KIND >= 11 && KIND <= 14) {
                    // the conv epilogue's two global access shapes, fresh lines every instruction (activations [pos][256 ch] f32):
                    // 11 / 13: thread (tile = lane >> 3, quad = lane & 7): 8 lanes cover one 128-byte line (round-2 epilogue);
                    // 12 / 14: lane (tile = lane & 31, half = lane >> 5): 32 tiles x 32 bytes per instruction (MFMA layout).
                    // 11, 12 stores; 13, 14 loads.  Rows are 1 KB apart (C = 256), tiles 3 rows apart.
                    const unsigned pp = (unsigned)(it * 8 + u) * 4u;      // + i: instruction number 0..255, every one on fresh lines
                    const bool lines = KIND == 11 || KIND == 13;
                    const unsigned trow = lines ? (unsigned)(wave * 8 + (lane >> 3)) : (unsigned)(lane & 31);
                    const unsigned base = lines
                        ? (((unsigned)blockIdx.x * 32u + trow) * 64u + pp / 8u) * 1024u + (pp % 8u) * 128u + (unsigned)(lane & 7) * 16u
                        : (((unsigned)blockIdx.x * 32u + trow) * 64u + pp / 4u) * 1024u + (unsigned)wave * 128u + (unsigned)(lane >> 5) * 16u;
                    if (KIND <= 12) {
#define X(i) asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen offset:%3" ::"v"(q[i]), "v"(base), "s"(rs), "n"((KIND == 11 ? 128 : 32) * i) : "memory");
                        X(0) X(1) X(2) X(3)
#undef X
                    } else {
#define X(i) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(q[i]) : "v"(base), "s"(rs), "n"((KIND == 13 ? 128 : 32) * i) : "memory");
                        X(0) X(1) X(2) X(3)
#undef X
                    }
                } else if (KIND == 10) {      // 8 v_mov_b32
#define X(i) asm volatile("v_mov_b32 %0, %1" : "=v"(v[i]) : "v"(a1));
                    REP8(X)
#undef X
                }
            }
        }
        if (KIND == 6 || KIND == 9 || KIND >= 11) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        t1 = __builtin_amdgcn_s_memtime();
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i] + p[i].x + p[i].y + q[i].x + q[i].w;
    if (s == 1234.5f) sink[tid] = s;
    if (lane == 0) out[blockIdx.x * 4 + wave] = t1 - t0;
}

// fp32 MFMA stream (one wave per SIMD, 32 MFMAs per iteration on 8 accumulators) with R ds_read_b128 per iteration issued by
// the waves selected by `readers` (bit w = wave w reads): what does an LDS read cost the MFMA stream of the wave that issues it,
// alone and with the other three waves reading at the same time?
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int R>
__global__ __launch_bounds__(256, 1) void kmf(unsigned long long *out, float *sink, int readers, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = tid * 0.001f, b = 1.0f + lane * 0.01f;
    f32x4 t = {0, 0, 0, 0};
    for (int o = tid * 16; o < 147456; o += 4096) *(f32x4 *)(lds + o) = f32x4{a, b, a, b};
    __syncthreads();
    const bool rd = (readers >> wave) & 1;
    const char *lp = lds + wave * 36864 + lane * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            acc[i & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i & 7], 0, 0, 0);
            if (R > 0 && rd && (i % (32 / R)) == 0) t += *(const f32x4 *)(lp + (i / (32 / R)) * 1024);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = t.x + t.y + t.z + t.w;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][5];
    if (s == 1234.5f) sink[tid] = s;
    if (lane == 0) out[blockIdx.x * 4 + wave] = (t1 - t0) / iters;
}
// the same MFMA stream with V buffer_load_dwordx4 per iteration from an L2-resident region (results consumed one iteration later)
template <int V>
__global__ __launch_bounds__(256, 1) void kmv(unsigned long long *out, float *sink, const float *g, int iters) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    float a = tid * 0.001f, b = 1.0f + lane * 0.01f;
    f32x4 u[V > 0 ? V : 1];
    for (int i = 0; i < (V > 0 ? V : 1); ++i) u[i] = f32x4{a, b, a, b};
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)g, 0, 1 << 22, 0x00020000);
    const unsigned voff = (blockIdx.x % 8) * 262144 + wave * 16384 + lane * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            acc[i & 7] = __builtin_amdgcn_mfma_f32_32x32x2f32(V > 0 ? u[i % (V > 0 ? V : 1)].x : a, b, acc[i & 7], 0, 0, 0);
            if (V > 0 && (i % (32 / (V > 0 ? V : 1))) == (32 / (V > 0 ? V : 1)) - 1)
                u[i / (32 / V)] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, voff, ((it * V + i / (32 / V)) & 15) * 1024, 0));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][5];
    if (s == 1234.5f) sink[tid] = s;
    if (lane == 0) out[blockIdx.x * 4 + wave] = (t1 - t0) / iters;
}
template <int V>
static void run_mv(const char *name) {
    unsigned long long *d;
    float *sink, *g;
    hipMalloc(&d, 256 * 4 * 8);
    hipMalloc(&sink, 4096);
    hipMalloc(&g, 1 << 22);
    hipLaunchKernelGGL(kmv<V>, dim3(256), dim3(256), 0, 0, d, sink, g, 200);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(1024);
    hipMemcpy(h.data(), d, 1024 * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-70s median %6llu  cycles per 32 MFMAs (2048 = MFMA only)\n", name, h[512]);
    hipFree(d); hipFree(sink); hipFree(g);
}

template <int R>
static void run_mf(const char *name, int readers) {
    unsigned long long *d;
    float *sink;
    hipMalloc(&d, 256 * 4 * 8);
    hipMalloc(&sink, 4096);
    hipFuncSetAttribute((const void *)kmf<R>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipLaunchKernelGGL(kmf<R>, dim3(256), dim3(256), 150 * 1024, 0, d, sink, readers, 200);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(1024);
    hipMemcpy(h.data(), d, 1024 * 8, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> w0, w3;
    for (int b = 0; b < 256; ++b) { w0.push_back(h[b * 4]); w3.push_back(h[b * 4 + 3]); }
    std::sort(w0.begin(), w0.end()); std::sort(w3.begin(), w3.end());
    printf("%-70s wave 0: %6llu  wave 3: %6llu  cycles per 32 MFMAs (2048 = MFMA only)\n", name, w0[128], w3[128]);
    hipFree(d); hipFree(sink);
}

template <int KIND>
static void run(const char *name, int per_iter, int active_mod = 1) {
    unsigned long long *d;
    float *sink, *gbuf;
    hipMalloc(&d, 256 * 4 * 8);
    hipMalloc(&sink, 4096);
    hipMalloc(&gbuf, 512u * 1024 * 1024);
    hipFuncSetAttribute((const void *)k<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(256), 150 * 1024, 0, d, sink, gbuf, active_mod);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(1024);
    hipMemcpy(h.data(), d, 1024 * 8, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> nz;
    for (auto x : h) if (x) nz.push_back(x);
    std::sort(nz.begin(), nz.end());
    const double n = 8.0 * 8.0 * per_iter;
    printf("%-58s median %7.2f  min %7.2f  cycles per instruction (s_memtime ticks / %d; %d waves)\n", name, nz[nz.size() / 2] / n, nz[0] / n, (int)n, (int)nz.size());
    hipFree(d); hipFree(sink); hipFree(gbuf);
}

int main() {
    run<0>("v_add_f32 (8 independent chains)", 8);
    run<3>("v_fma_f32", 8);
    run<1>("v_pk_add_f32", 8);
    run<2>("v_pk_fma_f32", 8);
    run<5>("v_max_f32", 8);
    run<10>("v_mov_b32", 8);
    run<4>("v_accvgpr_read_b32", 8);
    run<8>("2x2 accvgpr_read + 4 dependent v_pk_add (per instruction of 8)", 8);
    run<6>("ds_write_b128 lane-linear (incl. final drain)", 8);
    run<7>("ds_read_b128 lane-linear, lgkmcnt(0) per 8", 8);
    run<9>("buffer_store_dwordx4 1 KB/instr (incl. final drain)", 8);
    run<11>("store dwordx4, 8 lanes per 128-B line (incl. drain)", 4);
    run<12>("store dwordx4, 32 tiles x 32 B per instr (incl. drain)", 4);
    run<13>("load  dwordx4, 8 lanes per 128-B line (incl. drain)", 4);
    run<14>("load  dwordx4, 32 tiles x 32 B per instr (incl. drain)", 4);
    run<11>("store dwordx4, 8 lanes per line, 1 CU in 8 active", 4, 8);
    run<11>("store dwordx4, 8 lanes per line, 1 CU in 32 active", 4, 32);
    run<13>("load  dwordx4, 8 lanes per line, 1 CU in 8 active", 4, 8);
    run<6>("ds_write_b128 lane-linear, 1 CU in 8 active", 8, 8);
    run_mv<0>("MFMA x32 per iteration, no loads");
    run_mv<4>("MFMA x32 + 4 buffer_load_dwordx4 (L2-resident) per iteration");
    run_mv<8>("MFMA x32 + 8 buffer_load_dwordx4 (L2-resident) per iteration");
    run_mv<16>("MFMA x32 + 16 buffer_load_dwordx4 (L2-resident) per iteration");
    run_mf<0>("MFMA x32 per iteration, no LDS reads", 0);
    run_mf<8>("MFMA x32 + 8 ds_read_b128 per iteration, ALL four waves read", 15);
    run_mf<8>("MFMA x32 + 8 ds_read_b128 per iteration, ONLY wave 0 reads", 1);
    run_mf<4>("MFMA x32 + 4 ds_read_b128 per iteration, ALL four waves read", 15);
    run_mf<4>("MFMA x32 + 4 ds_read_b128 per iteration, ONLY wave 0 reads", 1);
    run_mf<16>("MFMA x32 + 16 ds_read_b128 per iteration, ALL four waves read", 15);
    run_mf<16>("MFMA x32 + 16 ds_read_b128 per iteration, ONLY wave 0 reads", 1);
    return 0;
}
