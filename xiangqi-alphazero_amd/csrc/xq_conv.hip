// xq_conv.hip -- 3x3 convolution of the residual tower (model.py:20-36) as a fused Winograd F(2x2,3x3) kernel
// on the fp32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
// Why Winograd: the contract is fp32 (1e-5 against the fp32 reference; gfx950 has no xf32), and the direct
// implicit-GEMM form is already at ~88 % of the 157 TFLOP/s fp32 MFMA peak in the ROCm library.  F(2x2,3x3) needs
// 16 multiplies per 2x2 output tile instead of 36: with the 10x9 board padded to 5x5 tiles the tower does
// 2.03x fewer MFMA flops, all arithmetic still fp32.  The transformed input (4x the activation bytes) and the 16
// per-frequency products never leave the CU -- an unfused Winograd would be HBM-bound and lose the gain.
//
// Layouts (C = channels in = channels out, C % 64 == 0):
//   X, Y, R : float[B][90][C]   (NHWC, position-major)                       activations / residual
//   Ug      : float[C/64][C/8][16][2][64][4]  = U[cog][chunk][xi][quad][co][j], U_xi = (G g G^T)_xi[ci][co] with the
//             four frequencies of Winograd row 2 negated, ci = 8*chunk + 4*quad + j     pre-transformed weights (host)
// Work decomposition: workgroup = 32 tiles (1.28 boards) x 64 output channels, 4 waves, TWO workgroups per CU.
// Wave p owns Winograd row p (frequencies xi = 4p..4p+3) for all 64 channels: 4 xi x 1 M-tile x 2 N-tiles = 8
// accumulator tiles of 32x32 = 128 VGPRs.  The MFMA A operand of lane (h, m) -- V[xi][tile m][ci = 4h..4h+3] -- is
// exactly what the input transform of (tile m, channel quad h, row p) produces, so every lane transforms what it
// multiplies: the transformed input never goes through LDS, and the only shared data is the raw input of the <= 3
// boards a tile group touches, staged once per 16 channels (double-buffered, one barrier per 16 channels).  Every
// weight element is used by exactly one wave: the B operand goes global (L2) -> registers through buffer loads
// (wave-uniform descriptor + scalar offset, one address VGPR), issued a whole chunk ahead of its use.
// The two workgroups of a CU are not synchronised with each other: one's prologue, barrier waits and epilogue are
// covered by the other's MFMAs.
// Epilogue: the column half of A^T M A in registers, the row half across the 4 waves through LDS (the staging
// buffers are reused), then bias + residual + ReLU and coalesced NHWC stores.
// Blocks are dealt so that each XCD works on one 64-channel slice of U at a time (1 MB at C=256: L2-resident).
#include <type_traits>

#include "xq_common.h"

#pragma clang fp contract(off)

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NCO = 64;              // output channels per workgroup
constexpr int KC = 8;                // input channels per chunk
constexpr int UBUF_BYTES = 16 * 2 * NCO * 16;     // one chunk of pre-transformed weights for 64 channels: 32 KB
constexpr int XSTRIDE = 80;          // bytes per staged position (16 channels + 16 B pad)

__device__ __forceinline__ f32x4 ld4(const char *p) { return *(const f32x4 *)p; }

// Staged input: boards with a zero halo, position P(b, y, x) = (11 b + y + 1) * 10 + x + 1 for y in [-1, 10], x in
// [-1, 9] (row 10 of a board is row -1 of the next, column 9 of a row is column -1 of the next: all zero and never
// written), 80 bytes per position (16 channels + 16 B pad).  A tile's 4x4 patch is then base + (10 r + c) * 80: one
// address register and immediates, no bounds logic.
constexpr int TILES = 32;                       // tiles per workgroup
constexpr int XPOS = (3 * 11 + 1) * 10 + 1; // 3 boards with halo: 341 positions
constexpr int XRAW = (XPOS + 1) * XSTRIDE; // 27360 B per staging buffer (+ one dump position)
constexpr int E_BYTES = 4 * 2 * TILES * NCO * 4;          // epilogue exchange [row p][b][tile][co] (64 KB)
constexpr int LDS_BYTES = 2 * XRAW > E_BYTES ? 2 * XRAW : E_BYTES;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// a - b / a + b on four floats as two packed instructions (v_pk_add_f32 has per-operand negation; the compiler only
// emits the packed form for additions).  Exactly the IEEE results of the scalar forms.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 pk_sub4(f32x4 a, f32x4 b) {
    f32x2 lo, hi;
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(lo) : "v"(f32x2{a.x, a.y}), "v"(f32x2{b.x, b.y}));
    asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(hi) : "v"(f32x2{a.z, a.w}), "v"(f32x2{b.z, b.w}));
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}
__device__ __forceinline__ f32x4 pk_add4(f32x4 a, f32x4 b) {
    f32x2 lo, hi;
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(lo) : "v"(f32x2{a.x, a.y}), "v"(f32x2{b.x, b.y}));
    asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(hi) : "v"(f32x2{a.z, a.w}), "v"(f32x2{b.z, b.w}));
    return f32x4{lo.x, lo.y, hi.x, hi.y};
}

__device__ __forceinline__ f32x4 buf_ld4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

__global__ __launch_bounds__(256, 2) void k_wino_conv(const float *__restrict__ X, const float *__restrict__ Ug,
                                                       const float *__restrict__ bias, const float *__restrict__ R,
                                                       float *__restrict__ Y, int B, int C, int relu, int n_groups) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *Xr = lds;                                   // [2][341][80 B]

    const int tid = threadIdx.x, lane = tid & 63, wp = tid >> 6;
    const int NG = C / NCO;                           // channel groups; divides 8
    const int per = 8 / NG;
    const int xcd = blockIdx.x & 7, rr = blockIdx.x >> 3;
    const int cog = xcd % NG;
    const int tg = rr * per + xcd / NG;
    if (tg >= n_groups) return;
    const int T = B * 25;
    const int t0 = tg * TILES;
    const int b_lo = t0 / 25;
    const int NCH = C / KC;
    const int h = lane >> 5, l31 = lane & 31;

    // ---- transform / MFMA role: tile l31, channel quad h, Winograd row wp --------------------------------
    const int gt = t0 + l31 < T ? t0 + l31 : T - 1;   // tiles past the end recompute the last one (never stored)
    const int tb = gt / 25, tt = gt - tb * 25, ty = tt / 5, tx = tt - ty * 5;
    const int tbase = (((tb - b_lo) * 11 + 2 * ty) * 10 + 2 * tx) * XSTRIDE + h * 16;   // P(tb, 2ty-1, 2tx-1)
    // Row p of B^T d is d[r1] + s d[r2] with (r1, r2, s) = (0,2,-), (1,2,+), (1,2,-), (1,3,-); row 2 is therefore the
    // NEGATIVE of the textbook d2 - d1, and the pre-transformed weights of its four frequencies carry the other minus
    // sign (include/xq_hip.h).  One code path serves all four waves: the row is data (two base addresses and a sign).
    const int tb1 = tbase + (wp == 0 ? 0 : 10) * XSTRIDE, tb2 = tbase + (wp == 3 ? 30 : 20) * XSTRIDE;
    const float sg = wp == 1 ? 1.0f : -1.0f;
    const f32x2 sgn = {sg, sg};

    // ---- staging role: only the rows some tile of this group reads are fetched (from the first tile's halo in the first
    // board to the last tile's in the last): a contiguous run of at most 162 positions = 648 float4 per 16-channel
    // superchunk, 3 slots per thread
    const int tl = (t0 + TILES - 1 < T ? t0 + TILES - 1 : T - 1);
    const int b_hi = tl / 25;
    const int y_min = 2 * ((t0 - b_lo * 25) / 5) - 1, y_max = 2 * ((tl - b_hi * 25) / 5) + 2;
    const int pos_first = (y_min > 0 ? y_min : 0) * 9;
    const int pos_last = (b_hi - b_lo) * 90 + ((y_max < 9 ? y_max : 9) + 1) * 9 - 1;
    const int spos = pos_first + (tid >> 2), spart = tid & 3;
    const unsigned xgo = (unsigned)(((long long)b_lo * 90 + spos) * C + spart * 4) * 4u;      // byte offset of slot 0
    const unsigned xstep = 64u * (unsigned)C * 4u;                                            // 64 positions further
    // Slots past the run load nothing (offset beyond the buffer's range: the load returns zeros without a fetch) and
    // store into a dump position behind the staged image, so the staging code has no divergent control flow.
    unsigned xgk[3];
    int xl[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int pos = spos + 64 * k;
        const int bi = pos / 90, rem = pos - bi * 90, y = rem / 9, x = rem - y * 9;
        const bool ok = pos <= pos_last;
        xgk[k] = ok ? xgo + k * xstep : 0xFFFFFFF0u;
        xl[k] = (ok ? (bi * 11 + y + 1) * 10 + x + 1 : XPOS) * XSTRIDE + spart * 16;
    }
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)X, 0, (int)((unsigned)B * 90u * (unsigned)C * 4u), 0x00020000);
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc((void *)(Ug + (size_t)cog * NCH * (UBUF_BYTES / 4)), 0,
                                                                         NCH * UBUF_BYTES, 0x00020000);
    // B operand: lane (h, n) needs U[xi][8*chunk + 4h + j][64*cog + 32*nt + n], j = 0..3
    const unsigned ul = (h * (NCO * 4) + l31 * 4 + (wp * 4) * (2 * NCO * 4)) * 4;   // byte offset in a chunk

    f32x16 acc[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[q][n][e] = 0.0f;

    f32x4 xreg[3];
    auto load_x = [&](int super) __attribute__((always_inline)) {
#pragma unroll
        for (int k = 0; k < 3; ++k) xreg[k] = buf_ld4(xrs, xgk[k], super * 64);
    };
    auto store_x = [&](int super) __attribute__((always_inline)) {
        char *dst = Xr + (super & 1) * XRAW;
#pragma unroll
        for (int k = 0; k < 3; ++k) *(f32x4 *)(dst + xl[k]) = xreg[k];
    };
    f32x4 a[4], u[4][2];
    auto rowpair = [&](f32x4 d1, f32x4 d2) __attribute__((always_inline)) {      // d1 + sgn * d2, two packed FMAs
        f32x2 lo, hi;                                 // fma(d2, +-1, d1) rounds once: exactly d1 +- d2
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(lo) : "v"(f32x2{d2.x, d2.y}), "v"(sgn), "v"(f32x2{d1.x, d1.y}));
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(hi) : "v"(f32x2{d2.z, d2.w}), "v"(sgn), "v"(f32x2{d1.z, d1.w}));
        return f32x4{lo.x, lo.y, hi.x, hi.y};
    };
    // prologue transform of chunk 0: rows r1, r2 of B^T d for each column, then the column transform
    auto transform0 = [&]() __attribute__((always_inline)) {
        f32x4 w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = rowpair(ld4(Xr + tb1 + q * XSTRIDE), ld4(Xr + tb2 + q * XSTRIDE));
        a[0] = w[0] - w[2]; a[1] = w[1] + w[2]; a[2] = w[2] - w[1]; a[3] = w[1] - w[3];
    };
    auto load_u = [&](int chunk, int q) __attribute__((always_inline)) {
        const unsigned so = (unsigned)chunk * UBUF_BYTES + q * (2 * NCO * 16);      // wave-uniform
        u[q][0] = buf_ld4(urs, ul, so);
        u[q][1] = buf_ld4(urs, ul, so + 32 * 16);
    };
#define XQ_PIN(v) asm volatile("" : "+v"(v))
    // One chunk: 32 MFMAs on (a, u); the transform of the NEXT chunk (raw data at LDS offset XO, a compile-time
    // constant: staging buffer and chunk parity) and its weights replace a and u as they retire.  The instruction order
    // is pinned by hand (a sched_barrier fence every two MFMAs, empty asm pins on the VALU results): the raw columns are
    // read in the order 0, 2, 1, 3 so that the next chunk's first A fragment (w0 - w2) is ready early, each column's two
    // LDS reads sit two MFMA pairs ahead of the VALU that consumes them, at most four VALU share a fence with an
    // MFMA pair, and the two weight loads of a frequency are issued right behind its last MFMA -- a whole chunk ahead
    // of their use.
    auto chunk_body = [&](int uchunk, auto xo_tag, int stage) __attribute__((always_inline)) {
        constexpr int XO = decltype(xo_tag)::value;
        f32x4 w[4], d1, d2;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = q == 0 ? 0 : q == 1 ? 2 : q == 2 ? 1 : 3;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                if (jj == 0) { d1 = ld4(Xr + tb1 + XO + col * XSTRIDE); d2 = ld4(Xr + tb2 + XO + col * XSTRIDE); }
                acc[q][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][jj], u[q][0][jj], acc[q][0], 0, 0, 0);
                acc[q][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][jj], u[q][1][jj], acc[q][1], 0, 0, 0);
                if (jj == 0 && q == 3) a[2] = pk_sub4(w[2], w[1]);
                if (jj == 2) w[col] = rowpair(d1, d2);
                if (jj == 1 && q == 3 && stage >= 0) { store_x(stage); load_x(stage + 1); }
                if (jj == 3) {
                    load_u(uchunk, q);
                    if (q == 1) a[0] = pk_sub4(w[0], w[2]);
                    if (q == 2) a[1] = pk_add4(w[1], w[2]);
                    if (q == 3) a[3] = pk_sub4(w[1], w[3]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // ---- prologue ------------------------------------------------------------------------------------------
    load_x(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) load_u(0, q);
    {                                                 // zero both staging buffers (the halo stays zero from here on)
        f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int o = tid * 16; o < 2 * XRAW; o += 256 * 16) *(f32x4 *)(Xr + o) = z;
    }
    __syncthreads();
    store_x(0);
    __syncthreads();
    load_x(1);
    transform0();

    // ---- main loop: two 16-channel superchunks (four chunks) per trip, one barrier per superchunk ----------------
    // Chunk c multiplies with (a, u) of chunk c while transforming chunk c+1; superchunk s is staged in buffer s & 1.
    for (int c = 0; c < NCH; c += 4) {
        const int sup = c >> 1;                       // even
        // chunk c stores superchunk sup+1 into buffer 1 (last read during chunk c-2, before the previous barrier) and
        // fetches sup+2 into the staging registers; chunk c+2 does the same one further.  Both sit late in their chunk,
        // behind the waits for that chunk's weights, so the in-order load counter does not make the store wait for
        // younger weight loads.  Past the last superchunk the stores put stale registers into a free buffer and the
        // loads fetch channels of the next positions or zeros: never used.
        chunk_body(c + 1, std::integral_constant<int, 32>{}, sup + 1);              // transforms chunk c+1: buffer 0, upper half
        __syncthreads();
        chunk_body(c + 2, std::integral_constant<int, XRAW>{}, -1);                // chunk c+2: buffer 1, lower half
        chunk_body(c + 3, std::integral_constant<int, XRAW + 32>{}, sup + 2);      // chunk c+3: buffer 1, upper half
        __syncthreads();
        // the last trip transforms stale data for a chunk that does not exist; its weights are re-read from chunk c+3
        chunk_body(c + 4 < NCH ? c + 4 : c + 3, std::integral_constant<int, 0>{}, -1);   // chunk c+4: buffer 0, lower half
    }
#undef XQ_PIN
    __syncthreads();                                  // staging buffers become the exchange planes

    // ---- epilogue: Y = A^T M A, bias, residual, ReLU --------------------------------------------------------
    const int c4 = tid & 15;
    const int co = c4 * 4;
    const f32x4 bv = *(const f32x4 *)(bias + cog * NCO + co);
    size_t oaddr[8];
    f32x4 resv[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int pidx = it * 16 + (tid >> 4);
        const int tile = pidx >> 2, ya = (pidx >> 1) & 1, yb = pidx & 1;
        const int g = t0 + tile;
        const int bd = g / 25, t2 = g - bd * 25, ty2 = t2 / 5, tx2 = t2 - ty2 * 5;
        const int oy = 2 * ty2 + ya, ox = 2 * tx2 + yb;
        const bool ok = g < T && ox < 9;
        oaddr[it] = ok ? ((size_t)bd * 90 + oy * 9 + ox) * C + cog * NCO + co : (size_t)-1;
        f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        resv[it] = (ok && R) ? *(const f32x4 *)(R + oaddr[it]) : z;
    }
    // column half in registers: b=0: M0+M1+M2, b=1: M1-M2-M3 (A^T = [[1,1,1,0],[0,1,-1,-1]])
    float *E = (float *)lds;                          // [4 rows p][2 b][32 tiles][64 co]
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const f32x16 y0 = acc[0][n] + acc[1][n] + acc[2][n];
        const f32x16 y1 = acc[1][n] - acc[2][n] - acc[3][n];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int tile = (e & 3) + 8 * (e >> 2) + 4 * h;
            E[((wp * 2 + 0) * TILES + tile) * NCO + 32 * n + l31] = y0[e];
            E[((wp * 2 + 1) * TILES + tile) * NCO + 32 * n + l31] = y1[e];
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int pidx = it * 16 + (tid >> 4);
        const int tile = pidx >> 2, ya = (pidx >> 1) & 1, yb = pidx & 1;
        if (oaddr[it] == (size_t)-1) continue;
        const float *e0 = E + (yb * TILES + tile) * NCO + co;
        const int pstride = 2 * TILES * NCO;             // next Winograd row p
        f32x4 y;
        if (ya == 0) y = *(const f32x4 *)(e0) + *(const f32x4 *)(e0 + pstride) + *(const f32x4 *)(e0 + 2 * pstride);
        else y = *(const f32x4 *)(e0 + pstride) - *(const f32x4 *)(e0 + 2 * pstride) - *(const f32x4 *)(e0 + 3 * pstride);
        y = y + bv + resv[it];
        if (relu) { y.x = fmaxf(y.x, 0.0f); y.y = fmaxf(y.y, 0.0f); y.z = fmaxf(y.z, 0.0f); y.w = fmaxf(y.w, 0.0f); }
        *(f32x4 *)(Y + oaddr[it]) = y;
    }
}

}  // namespace

extern "C" {

/* bytes of the pre-transformed weight tensor Ug for C channels: 16 * C * C floats */
size_t xq_wino_weight_bytes(int channels) { return (size_t)16 * channels * channels * sizeof(float); }

int xq_wino_conv3x3(const float *dev_x, const float *dev_u, const float *dev_bias, const float *dev_residual, float *dev_y,
                    int batch, int channels, int relu, void *stream) {
    if (!dev_x || !dev_u || !dev_bias || !dev_y || batch <= 0) return XQ_ERR_ARG;
    if (channels < 64 || channels % 64 || 8 % (channels / 64)) return XQ_ERR_ARG;   // 64, 128, 256, 512
    if (dev_x == dev_y || dev_residual == dev_y) return XQ_ERR_ARG;                 // not in place
    if (((uintptr_t)dev_x | (uintptr_t)dev_u | (uintptr_t)dev_bias | (uintptr_t)dev_residual | (uintptr_t)dev_y) & 15) return XQ_ERR_ARG;
    if ((unsigned long long)batch * 90ull * (unsigned)channels * 4ull >= (1ull << 32)) return XQ_ERR_ARG;   // 32-bit buffer offsets
    static thread_local bool attr_set = false;
    if (!attr_set) {
        XQ_TRY(hipFuncSetAttribute((const void *)k_wino_conv, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set = true;
    }
    const int n_groups = (batch * 25 + TILES - 1) / TILES;
    const int per = 8 / (channels / NCO);
    const int rows = (n_groups + per - 1) / per;
    hipLaunchKernelGGL(k_wino_conv, dim3(rows * 8), dim3(256), LDS_BYTES, (hipStream_t)stream, dev_x, dev_u, dev_bias,
                       dev_residual, dev_y, batch, channels, relu, n_groups);
    return xq::launch_status();
}

}  // extern "C"
