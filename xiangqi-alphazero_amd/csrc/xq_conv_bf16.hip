// xq_conv_bf16.hip -- the REDUCED-PRECISION throughput mode of the residual tower's 3x3 convolution (model.py:20-36): the same fused
// Winograd F(2x3, 3x3) decomposition as xq_conv.hip's wide variant, with the 20 per-frequency products on the bf16 MFMA
// (v_mfma_f32_32x32x16_bf16, fp32 accumulation).  NOT the parity path: activations stay float32 in HBM and the transforms are float32,
// but both MFMA operands are rounded to bf16 (8 significand bits), so outputs miss the 1e-5 contract by orders of magnitude
// (tests/test_nn_fullsize.py states the measured deviation).  bench.py reports it only as the labelled second object.
//
// What changes against the fp32 kernel, and why it is a different loop rather than another MFMA in the same one:
//   * the bf16 MFMA takes K = 16 per instruction: a chunk is 16 input channels, a lane transforms EIGHT channels of its tile (two
//     16-byte units per patch position) and rounds the five transformed vectors to bf16 (v_cvt_pk_bf16_f32);
//   * one MFMA (32 cycles) does what 8 fp32 MFMAs (512 cycles) did, so the kernel is no longer MFMA-bound but bound by its weight
//     stream L2 -> CU: 1.3 MB of bf16 weights per workgroup (32 tiles x 128 channels).  By Little's law ~7 KB per wave in flight
//     cover 34 bytes/clock/CU at ~800 cycles of L2 latency: a pool of 16 fragment registers (4 VGPRs each) is more than enough;
//   * the bf16 MFMA runs on the matrix core, not on the vector ALU: the transform's vector instructions co-issue with it.
// Layouts:  X, Y, R float[B][90][C] as in xq_conv.hip;
//   Ub: bf16 [C/128][C/16][20][2][128][8] = U[cog][chunk16][xi = 5 p + j][h][co][k] for input channel 16 chunk + 8 h + k  (host:
//   hip.wino_transform_weights_bf16), 40 C^2 bytes.
// Epilogue: the fp32 kernel's pipelined one (accumulators have the same lane layout), see xq_conv.hip.
#include <type_traits>

#include "xq_common.h"

#pragma clang fp contract(off)

// In-situ ablation (tests/microbench; never shipped): 1 no weight loads in the loop, 2 no transform, 4 no staging, 8 no epilogue
#ifndef XQ_BABL
#define XQ_BABL 0
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int TILES = 32;
constexpr int NT = 4, NCO = 128, NF = 20, POOL = 16;               // as the fp32 wide variant: slot of fragment f of chunk c = (20 c + f) % 16
constexpr int XSTRIDE = 16;
constexpr int XU_ODD = 10, XU_PAIR = 25, XU_B = 157;                // staging layout of xq_conv.hip (bank-conflict-free tile bases)
constexpr int XPLANE = (4 * XU_B) * 16;
constexpr int XDUMP = 4 * XU_B - 1;
constexpr int XRAW = 4 * XPLANE;                                    // FOUR planes: 16 channels per chunk
constexpr int EPAD = 32, ESTR_R = 64 + EPAD;                        // exchange planes of a 64-channel round (conflict-free stride)
constexpr int LDS_BYTES = 4 * 3 * TILES * ESTR_R * 4;               // 147 456 >= 2 * XRAW (80 384)
constexpr int UCHUNK_BYTES = 20 * 2 * NCO * 16;                     // one 16-channel chunk of bf16 weights: 81 920

__device__ __forceinline__ f32x4 ld4(const char *p) { return *(const f32x4 *)p; }
__device__ __forceinline__ f32x4 buf_ld4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void buf_st4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
    asm volatile("s_nop 1" ::: "memory");          // store-data hazard fence (see xq_conv.hip: buf_st4)
}
__device__ __forceinline__ bf16x8 to_bf16(f32x4 lo, f32x4 hi) {
    bf16x8 r;
    r[0] = (__bf16)lo.x; r[1] = (__bf16)lo.y; r[2] = (__bf16)lo.z; r[3] = (__bf16)lo.w;
    r[4] = (__bf16)hi.x; r[5] = (__bf16)hi.y; r[6] = (__bf16)hi.z; r[7] = (__bf16)hi.w;
    return r;
}

__global__ __launch_bounds__(256, 1) void k_wino_conv_bf16(const float *__restrict__ X, const void *__restrict__ Ub,
                                                             const float *__restrict__ bias, const float *__restrict__ R,
                                                             float *__restrict__ Y, int B, int C, int flags, int n_groups) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *Xr = lds;
    const int tid = threadIdx.x, lane = tid & 63, wp = tid >> 6;
    const int NG = C / NCO, per = 8 / NG;
    const int xcd = blockIdx.x & 7, rr = blockIdx.x >> 3;
    const int cog = xcd % NG;
    const int tg_fwd = rr * per + xcd / NG;
    const int tg = (flags & 2) ? n_groups - 1 - tg_fwd : tg_fwd;
    if (tg_fwd >= n_groups) return;
    const int relu = flags & 1;
    const int T = B * 15;
    const int t0 = tg * TILES;
    const int b_lo = t0 / 15;
    const int NCH = C / 16;
    const int h = lane >> 5, l31 = lane & 31;

    // transform / MFMA role: tile l31, channels 8 h .. 8 h + 7 of the chunk (planes 2 h and 2 h + 1), Winograd row wp
    const int gt = t0 + l31 < T ? t0 + l31 : T - 1;
    const int tb = gt / 15, tt = gt - tb * 15, ty = tt / 3, tx = tt - ty * 3;
    const int tbase = ((tb - b_lo) * XU_B + ty * XU_PAIR + 3 * tx) * XSTRIDE + 2 * h * XPLANE;
    const int tb1 = tbase + (wp == 0 ? 0 : XU_ODD) * XSTRIDE;
    const int tb2 = tbase + (wp == 3 ? XU_PAIR + XU_ODD : XU_PAIR) * XSTRIDE;
    const float sg = wp == 1 ? 1.0f : -1.0f;

    // staging role: position pos_first + (tid >> 2) + 64 k, channel quad tid & 3 of the chunk
    const int tl = (t0 + TILES - 1 < T ? t0 + TILES - 1 : T - 1);
    const int b_hi = tl / 15;
    const int y_min = 2 * ((t0 - b_lo * 15) / 3) - 1, y_max = 2 * ((tl - b_hi * 15) / 3) + 2;
    const int pos_first = (y_min > 0 ? y_min : 0) * 9;
    const int pos_last = (b_hi - b_lo) * 90 + ((y_max < 9 ? y_max : 9) + 1) * 9 - 1;
    const int spos = pos_first + (tid >> 2), spart = tid & 3;
    unsigned xgk[4];
    int xl[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int pos = spos + 64 * k;
        const int bi = pos / 90, rem = pos - bi * 90, y = rem / 9, x = rem - y * 9;
        const bool ok = pos <= pos_last;
        xgk[k] = ok ? (unsigned)(((long long)b_lo * 90 + pos) * C + spart * 4) * 4u : 0xFFFFFFF0u;
        xl[k] = (ok ? bi * XU_B + ((y + 1) >> 1) * XU_PAIR + ((y + 1) & 1) * XU_ODD + x + 1 : XDUMP) * XSTRIDE + spart * XPLANE;
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)X, 0, (int)((unsigned)B * 90u * (unsigned)C * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)Ub + (size_t)cog * NCH * UCHUNK_BYTES), 0,
                                                                         NCH * UCHUNK_BYTES, 0x00020000);
    // B fragment (q, nt): lane (h, n) needs U[5 wp + q][16 chunk + 8 h + k][128 cog + 32 nt + n], k = 0..7: 16 bytes
    const unsigned ul = ((wp * 5 * 2 + h) * NCO + l31) * 16;

    f32x16 acc[5][NT];
    f32x4 xreg[4];
    bf16x8 a[2][5], ub[POOL];                        // a[k & 1]: the transformed input of chunk k (double-buffered: chunk c + 1 is
                                                     // transformed while chunk c multiplies)
    auto load_x = [&](int chunk) __attribute__((always_inline)) {
        const unsigned kill = chunk < NCH ? 0u : 0xFFFFFFF0u;          // past the last chunk: out-of-range offsets, no memory traffic
#pragma unroll
        for (int k = 0; k < 4; ++k) xreg[k] = buf_ld4(xrs, xgk[k] | kill, chunk * 64);
    };
    auto store_x = [&](int chunk) __attribute__((always_inline)) {
        char *dst = Xr + (chunk & 1) * XRAW;
#pragma unroll
        for (int k = 0; k < 4; ++k) *(f32x4 *)(dst + xl[k]) = xreg[k];
    };
    auto load_frag = [&](int chunk, int f, int slot) __attribute__((always_inline)) {
        const int q = (f / NT) == 0 ? 1 : (f / NT) == 1 ? 2 : (f / NT) == 2 ? 3 : (f / NT) == 3 ? 0 : 4;
        ub[slot] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(urs, ul, (unsigned)chunk * UCHUNK_BYTES + q * (2 * NCO * 16) + (f % NT) * (32 * 16), 0));
    };
    // one 4-channel half of a lane's 8 channels: float32 transform (formulas of xq_conv.hip), rounded to bf16 into elements
    // 4 half .. 4 half + 3 of the five A operands
    auto transform_half = [&](const char *base, bf16x8 (&dst)[5], auto half_tag) __attribute__((always_inline)) {
        constexpr int H4 = 4 * decltype(half_tag)::value;
        f32x4 w[5], v[5];
#pragma unroll
        for (int c = 0; c < 5; ++c) w[c] = ld4(base + tb1 + c * XSTRIDE) + sg * ld4(base + tb2 + c * XSTRIDE);
        const f32x4 t = w[3] - w[1];
        v[0] = 2.0f * (w[0] - w[2]) + t;
        v[1] = 2.0f * w[1] - w[3] + w[2];
        v[2] = 3.0f * w[2] - (2.0f * w[1] + w[3]);
        v[3] = t;
        v[4] = (w[4] - w[2]) - 2.0f * t;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            dst[q][H4 + 0] = (__bf16)v[q].x; dst[q][H4 + 1] = (__bf16)v[q].y; dst[q][H4 + 2] = (__bf16)v[q].z; dst[q][H4 + 3] = (__bf16)v[q].w;
        }
    };
    // one chunk: 20 MFMAs of chunk c (fragment 16 positions further down the stream replaces the one just used), the transform of
    // chunk c + 1 in two halves between them (the bf16 MFMA runs on the matrix core: the vector instructions fill its shadow)
    auto chunk_mfma = [&](int nchunk_u, int lchunk, auto first_tag, auto ph_tag) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        constexpr int PH = decltype(ph_tag)::value;
        constexpr int P = PH & 1;
        const char *nxt = Xr + (P ^ 1) * XRAW;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int g = f / NT, nt = f % NT;
            const int q = g == 0 ? 1 : g == 1 ? 2 : g == 2 ? 3 : g == 3 ? 0 : 4;
            const int slot = (NF * PH + f) % POOL;
            if (FIRST) {
                const f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                acc[q][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[P][q], ub[slot], zero, 0, 0, 0);
            } else if (nt == 3) {
                // 20 accumulator tiles do not fit the 256 AGPRs: N-tile 3 is pinned to VGPRs (as in xq_conv.hip)
                // s_nop 1: a vector write of a source register must be 2 wait states ahead of an MFMA; the compiler pads its own MFMAs, it cannot
                // see this one (tools/check_asm_mfma_hazards.py found a v_cvt_pk_bf16_f32 right in front of it)
                asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[q][nt]) : "v"(a[P][q]), "v"(ub[slot]));
            } else {
                acc[q][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[P][q], ub[slot], acc[q][nt], 0, 0, 0);
            }
            if (!(XQ_BABL & 1)) { if (f + POOL < NF) load_frag(lchunk, f + POOL, slot); else load_frag(nchunk_u, f + POOL - NF, slot); }
            if (!(XQ_BABL & 2)) {
                if (f == 3) transform_half(nxt, a[P ^ 1], std::integral_constant<int, 0>{});
                if (f == 11) transform_half(nxt + XPLANE, a[P ^ 1], std::integral_constant<int, 1>{});
            }
        }
    };

    // ---- prologue: input of chunks 0 and 1, the first 16 weight fragments, zero fill under their latency
    f32x4 x1[4];
    load_x(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) x1[k] = buf_ld4(xrs, xgk[k], 64);
#pragma unroll
    for (int f = 0; f < POOL; ++f) load_frag(0, f, f);
    {
        const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int o = tid * 16; o < 2 * XRAW; o += 256 * 16) *(f32x4 *)(Xr + o) = z;
    }
    __syncthreads();
    store_x(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) *(f32x4 *)(Xr + XRAW + xl[k]) = x1[k];
    load_x(2);
    __syncthreads();
    transform_half(Xr, a[0], std::integral_constant<int, 0>{});
    transform_half(Xr + XPLANE, a[0], std::integral_constant<int, 1>{});
    __syncthreads();

    // ---- main loop, ONE barrier per chunk.  Step c: chunk c + 2 (fetched during step c - 1) is stored into buffer c & 1 -- last read by
    // the transform of chunk c in step c - 1, before that step's barrier -- and chunk c + 3 fetched; chunk c multiplies while chunk
    // c + 1 (buffer (c + 1) & 1, stored in step c - 1) is transformed.  Branch-free: past the end the loads are out of range, the
    // stores and the transform work on dead buffers.
    auto step = [&](int c, auto first_tag, auto ph_tag) __attribute__((always_inline)) {
        if (!(XQ_BABL & 4)) { store_x(c + 2); load_x(c + 3); }
        chunk_mfma(c + 1 < NCH ? c + 1 : c, c, first_tag, ph_tag);
        __syncthreads();
    };
    step(0, std::true_type{}, std::integral_constant<int, 0>{});
    step(1, std::false_type{}, std::integral_constant<int, 1>{});
    step(2, std::false_type{}, std::integral_constant<int, 2>{});
    step(3, std::false_type{}, std::integral_constant<int, 3>{});
    for (int c = 4; c < NCH; c += 4) {
        step(c, std::false_type{}, std::integral_constant<int, 0>{});
        step(c + 1, std::false_type{}, std::integral_constant<int, 1>{});
        step(c + 2, std::false_type{}, std::integral_constant<int, 2>{});
        step(c + 3, std::false_type{}, std::integral_constant<int, 3>{});
    }

    if (XQ_BABL & 8) {
        float sacc = 0.0f;
        for (int q = 0; q < 5; ++q) for (int n = 0; n < NT; ++n) for (int e = 0; e < 16; ++e) sacc += acc[q][n][e];
        if (sacc == 1234.5f) Y[tid] = sacc;
        return;
    }
    // ---- epilogue: xq_conv.hip's pipelined one (two rounds of 64 channels through the exchange planes) -----------------------------
    // an MFMA's D -> any non-MFMA reader: the asm MFMAs are invisible to the hazard recogniser (8 passes: 12 wait states suffice; 16 here)
    asm volatile("s_nop 15" : "+v"(acc[4][3]));
    const f32x2 two = {2.0f, 2.0f}, four = {4.0f, 4.0f};
    const int etile = tid >> 3, co = (tid & 7) * 4;
    const int eg = t0 + etile, egc = eg < T ? eg : T - 1;
    const int ebd = egc / 15, et2 = egc - ebd * 15, ety = et2 / 3, etx = et2 - ety * 3;
    const int Cs = __builtin_amdgcn_readfirstlane(C);
    const unsigned ooff_in = (unsigned)(((ebd * 90 + 2 * ety * 9 + 3 * etx) * Cs + cog * NCO + co) * 4);
    const unsigned ooff = eg < T ? ooff_in : 0xFFFFFFF0u;
    const unsigned nbytes = (unsigned)B * 90u * (unsigned)C * 4u;
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)Y, 0, (int)nbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc((void *)(R != nullptr ? R : X), 0, R != nullptr ? (int)nbytes : 0, 0x00020000);
    const float rlo = relu ? 0.0f : -__builtin_inff();
    auto goff = [&](int n, int it) __attribute__((always_inline)) { return (unsigned)(32 * n + ((it / 3) * 9 + it % 3) * Cs) * 4u; };
    f32x4 resv[2][6], bv[4], pend[12];
#pragma unroll
    for (int n = 0; n < 4; ++n) bv[n] = *(const f32x4 *)(bias + cog * NCO + 32 * n + co);
    float *E = (float *)lds;
    float *ew = E + ((wp * 3) * TILES + 4 * h) * ESTR_R + l31;
    const float *er = E + etile * ESTR_R + co;
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int n = rd * 2 + s;
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const f32x2 m0 = {acc[0][n][e], acc[0][n][e + 1]}, m1 = {acc[1][n][e], acc[1][n][e + 1]};
                const f32x2 m2 = {acc[2][n][e], acc[2][n][e + 1]}, m3 = {acc[3][n][e], acc[3][n][e + 1]};
                const f32x2 m4 = {acc[4][n][e], acc[4][n][e + 1]};
                const f32x2 s12 = m1 + m2;
                const f32x2 y0 = (m0 + m3) + s12;
                const f32x2 y1 = m3 * two + (m1 - m2);
                const f32x2 y2 = (m3 * four + s12) + m4;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int tile = ((e + k) & 3) + 8 * ((e + k) >> 2);
                    ew[(0 * TILES + tile) * ESTR_R + 32 * s] = y0[k];
                    ew[(1 * TILES + tile) * ESTR_R + 32 * s] = y1[k];
                    ew[(2 * TILES + tile) * ESTR_R + 32 * s] = y2[k];
                }
                const int seg = s * 8 + e / 2;
                if (rd == 1 && seg < 12) buf_st4(yrs, ooff, goff(seg / 6, seg % 6), pend[seg]);
                if (rd == 0) {
#pragma unroll
                    for (int j = seg * 2; j < (seg + 1) * 2 && j < 12; ++j) resv[j / 6][j % 6] = buf_ld4(rrs, ooff, goff(j / 6, j % 6));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int n = rd * 2 + s;
#pragma unroll
            for (int it = 0; it < 6; ++it) {
                const int ya = it / 3, yb = it % 3;
                const float *e0 = er + yb * TILES * ESTR_R + 32 * s;
                const int pstride = 3 * TILES * ESTR_R;
                f32x4 y;
                if (ya == 0) y = *(const f32x4 *)(e0) + *(const f32x4 *)(e0 + pstride) + *(const f32x4 *)(e0 + 2 * pstride);
                else y = *(const f32x4 *)(e0 + pstride) - *(const f32x4 *)(e0 + 2 * pstride) - *(const f32x4 *)(e0 + 3 * pstride);
                y = y + bv[n] + resv[s][it];
                y.x = __builtin_fmaxf(rlo, y.x); y.y = __builtin_fmaxf(rlo, y.y); y.z = __builtin_fmaxf(rlo, y.z); y.w = __builtin_fmaxf(rlo, y.w);
                if (rd == 0) {
                    pend[s * 6 + it] = y;
                    resv[s][it] = buf_ld4(rrs, ooff, goff(2 + s, it));
                } else {
                    buf_st4(yrs, ooff, goff(n, it), y);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (rd == 0) __syncthreads();
    }
}

}  // namespace

extern "C" {

/* bytes of the bf16 pre-transformed weight tensor for C channels: 20 * C * C bf16 values */
size_t xq_wino_weight_bytes_bf16(int channels) { return (size_t)20 * channels * channels * 2; }

int xq_wino_conv3x3_bf16(const float *dev_x, const void *dev_u_bf16, const float *dev_bias, const float *dev_residual, float *dev_y,
                         int batch, int channels, int flags, void *stream) {
    if (!dev_x || !dev_u_bf16 || !dev_bias || !dev_y || batch <= 0) return XQ_ERR_ARG;
    if (channels < NCO || channels % NCO || 8 % (channels / NCO) || (channels / 16) % 4) return XQ_ERR_ARG;
    if (dev_x == dev_y || dev_residual == dev_y) return XQ_ERR_ARG;
    if (((uintptr_t)dev_x | (uintptr_t)dev_u_bf16 | (uintptr_t)dev_bias | (uintptr_t)dev_residual | (uintptr_t)dev_y) & 15) return XQ_ERR_ARG;
    if ((unsigned long long)batch * 90ull * (unsigned)channels * 4ull >= (1ull << 32)) return XQ_ERR_ARG;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        XQ_TRY(hipFuncSetAttribute((const void *)k_wino_conv_bf16, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set = true;
    }
    const int n_groups = (batch * 15 + TILES - 1) / TILES;
    const int per = 8 / (channels / NCO);
    const int rows = (n_groups + per - 1) / per;
    hipLaunchKernelGGL(k_wino_conv_bf16, dim3(rows * 8), dim3(256), LDS_BYTES, (hipStream_t)stream, dev_x, dev_u_bf16, dev_bias,
                       dev_residual, dev_y, batch, channels, flags, n_groups);
    return xq::launch_status();
}

}  // extern "C"
