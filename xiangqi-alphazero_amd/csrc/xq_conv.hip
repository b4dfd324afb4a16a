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
//   Ug      : float[C/64][C/8][16][2][64][4]  = U[cog][chunk][xi][quad][co][j], U_xi = (G g G^T)_xi[ci][co],
//             ci = 8*chunk + 4*quad + j                                        pre-transformed weights (host)
// Work decomposition: workgroup = 64 tiles (2.56 boards) x 64 output channels, 8 waves.  Wave w owns Winograd
// row p = w&3 (frequencies xi = 4p..4p+3) for the channel half w>>2: 4 xi x 2 M-tiles x 1 N-tile = 8 accumulator
// tiles of 32x32 = 128 VGPRs.  Per 8-channel chunk: U arrives by LDS-DMA (global_load_lds, 16 B/lane, linear
// image), the raw input of the 4 boards a tile group can touch is staged once per 16 channels (80-byte position
// stride: 2-way instead of 8-way bank conflicts on the strided tile reads), every thread turns one (tile, channel
// quad, Winograd row) into four B^T d B values for the next chunk while the MFMAs of the current chunk run.
// Epilogue: the column half of A^T M A in registers, the row half across the 4 waves of a channel half through
// LDS (the 128 KB of chunk buffers are reused), then bias + residual + ReLU and coalesced NHWC stores.
// Blocks are dealt so that each XCD works on one 64-channel slice of U at a time (1 MB at C=256: L2-resident).
#include "xq_common.h"

#pragma clang fp contract(off)

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TILES = 64;            // tiles per workgroup
constexpr int NCO = 64;              // output channels per workgroup
constexpr int KC = 8;                // input channels per chunk
constexpr int VBUF_BYTES = 16 * 2 * TILES * 16;   // 32 KB
constexpr int UBUF_BYTES = 16 * 2 * NCO * 16;     // 32 KB
constexpr int XPOS = 360;            // 4 boards
constexpr int XSTRIDE = 80;          // bytes per staged position (16 channels + 16 B pad)
constexpr int XRAW_BYTES = XPOS * XSTRIDE;
constexpr int LDS_BYTES = 2 * VBUF_BYTES + 2 * UBUF_BYTES + XRAW_BYTES;

__device__ __forceinline__ f32x4 ld4(const char *p) { return *(const f32x4 *)p; }

__global__ __launch_bounds__(512, 2) void k_wino_conv(const float *__restrict__ X, const float *__restrict__ Ug,
                                                      const float *__restrict__ bias, const float *__restrict__ R,
                                                      float *__restrict__ Y, int B, int C, int relu, int n_groups) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *Vb = lds;                                   // [2][16][2][64] float4
    char *Ub = lds + 2 * VBUF_BYTES;                  // [2][16][2][64] float4
    char *Xr = lds + 2 * VBUF_BYTES + 2 * UBUF_BYTES; // [360][80 B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

    // ---- roles -------------------------------------------------------------------------------------------
    // transform role: one (tile, quad, Winograd row) per thread
    const int tr_tile = tid & 63, tr_qd = (tid >> 6) & 1, tr_i = tid >> 7;
    const int gt = t0 + tr_tile;
    const bool tile_ok = gt < T;
    const int tb = gt / 25, tt = gt - tb * 25, ty = tt / 5, tx = tt - ty * 5;
    const int r1 = tr_i == 0 ? 0 : 1, r2 = tr_i == 3 ? 3 : 2;
    const float s1 = tr_i == 2 ? -1.0f : 1.0f, s2 = (tr_i == 0 || tr_i == 3) ? -1.0f : 1.0f;
    int xoff[2][4];                                   // byte offsets into Xr, or -1 when out of the board
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int y = 2 * ty - 1 + (k == 0 ? r1 : r2);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int x = 2 * tx - 1 + c;
            const bool ok = tile_ok && (unsigned)y < 10u && (unsigned)x < 9u;
            xoff[k][c] = ok ? ((tb - b_lo) * 90 + y * 9 + x) * XSTRIDE : -1;
        }
    }
    // staging role: 1440 float4 per 16-channel superchunk, 3 per thread
    const float *xg[3];
    int xl[3];
    bool xv[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int idx = tid + 512 * k;
        const int pos = idx >> 2, part = idx & 3;
        const long long gpos = (long long)b_lo * 90 + pos;
        xv[k] = idx < XPOS * 4 && gpos < (long long)B * 90;
        xg[k] = X + (xv[k] ? gpos : 0) * C + part * 4;
        xl[k] = pos * XSTRIDE + part * 16;
    }
    // MFMA role
    const int wp = wave & 3, wch = wave >> 2;
    const int h = lane >> 5, l31 = lane & 31;
    const float *ug = Ug + (size_t)cog * NCH * (UBUF_BYTES / 4);

    f32x16 acc[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[q][m][e] = 0.0f;

    auto dma_u = [&](int chunk, int buf) {
        const float *src = ug + (size_t)chunk * (UBUF_BYTES / 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int kb = wave + 8 * k;              // KB index inside the 32 KB chunk
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + kb * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void *)(Ub + buf * UBUF_BYTES + kb * 1024),
                                             16, 0, 0);
        }
    };
    f32x4 xreg[3];
    auto load_x = [&](int super) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
            xreg[k] = xv[k] ? *(const f32x4 *)(xg[k] + super * 16) : z;
        }
    };
    auto store_x = [&]() {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (tid + 512 * k < XPOS * 4) *(f32x4 *)(Xr + xl[k]) = xreg[k];
    };
    auto transform = [&](int chunk, int buf) {
        const int sub = ((chunk & 1) * 2 + tr_qd) * 16;
        f32x4 w[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
            const f32x4 d1 = xoff[0][c] >= 0 ? ld4(Xr + xoff[0][c] + sub) : z;
            const f32x4 d2 = xoff[1][c] >= 0 ? ld4(Xr + xoff[1][c] + sub) : z;
            w[c] = s1 * d1 + s2 * d2;
        }
        char *dst = Vb + buf * VBUF_BYTES + ((tr_i * 4) * 2 + tr_qd) * (TILES * 16) + tr_tile * 16;
        *(f32x4 *)(dst + 0 * 2 * TILES * 16) = w[0] - w[2];
        *(f32x4 *)(dst + 1 * 2 * TILES * 16) = w[1] + w[2];
        *(f32x4 *)(dst + 2 * 2 * TILES * 16) = w[2] - w[1];
        *(f32x4 *)(dst + 3 * 2 * TILES * 16) = w[1] - w[3];
    };

    // ---- prologue ------------------------------------------------------------------------------------------
    load_x(0);
    dma_u(0, 0);
    store_x();
    __syncthreads();
    if (NCH > 2) load_x(1);
    transform(0, 0);
    __syncthreads();

    // ---- main loop over 8-channel chunks -------------------------------------------------------------------
    for (int c = 0; c < NCH; ++c) {
        const int buf = c & 1;
        const bool more = c + 1 < NCH;
        const bool new_super = more && (c & 1);       // chunk c+1 starts a 16-channel superchunk
        if (more) dma_u(c + 1, buf ^ 1);
        if (new_super) store_x();                     // last readers of Xr finished before the previous barrier

        const char *vb = Vb + buf * VBUF_BYTES + h * (TILES * 16) + l31 * 16;
        const char *ub = Ub + buf * UBUF_BYTES + h * (NCO * 16) + (wch * 32 + l31) * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int xi = wp * 4 + q;
            const f32x4 a0 = ld4(vb + xi * (2 * TILES * 16));
            const f32x4 a1 = ld4(vb + xi * (2 * TILES * 16) + 32 * 16);
            const f32x4 bb = ld4(ub + xi * (2 * NCO * 16));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[q][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], bb[j], acc[q][0], 0, 0, 0);
                acc[q][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], bb[j], acc[q][1], 0, 0, 0);
            }
        }
        if (new_super) {
            __syncthreads();                          // staged input of the new superchunk visible
            if (c + 3 < NCH) load_x((c + 3) >> 1);
        }
        if (more) transform(c + 1, buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: Y = A^T M A, bias, residual, ReLU --------------------------------------------------------
    // column half in registers: b=0: M0+M1+M2, b=1: M1-M2-M3 (A^T = [[1,1,1,0],[0,1,-1,-1]])
    float *E = (float *)lds;                          // [16 planes][64 tiles][32 co], plane = (wch*4 + wp)*2 + b
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const f32x16 y0 = acc[0][m] + acc[1][m] + acc[2][m];
        const f32x16 y1 = acc[1][m] - acc[2][m] - acc[3][m];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int tile = 32 * m + (e & 3) + 8 * (e >> 2) + 4 * h;
            E[(((wch * 4 + wp) * 2 + 0) * TILES + tile) * 32 + l31] = y0[e];
            E[(((wch * 4 + wp) * 2 + 1) * TILES + tile) * 32 + l31] = y1[e];
        }
    }
    __syncthreads();
    const int c4 = tid & 15;
    const int co = c4 * 4, ech = co >> 5, ecol = co & 31;
    const f32x4 bv = *(const f32x4 *)(bias + cog * NCO + co);
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int pidx = it * 32 + (tid >> 4);
        const int tile = pidx >> 2, a = (pidx >> 1) & 1, b = pidx & 1;
        const int g = t0 + tile;
        const int bd = g / 25, t2 = g - bd * 25, ty2 = t2 / 5, tx2 = t2 - ty2 * 5;
        const int oy = 2 * ty2 + a, ox = 2 * tx2 + b;
        if (g >= T || ox >= 9) continue;
        const float *e0 = E + (((ech * 4) * 2 + b) * TILES + tile) * 32 + ecol;
        const int pstride = 2 * TILES * 32;           // next Winograd row p
        f32x4 y;
        if (a == 0) y = *(const f32x4 *)(e0) + *(const f32x4 *)(e0 + pstride) + *(const f32x4 *)(e0 + 2 * pstride);
        else y = *(const f32x4 *)(e0 + pstride) - *(const f32x4 *)(e0 + 2 * pstride) - *(const f32x4 *)(e0 + 3 * pstride);
        const size_t o = ((size_t)bd * 90 + oy * 9 + ox) * C + cog * NCO + co;
        y = y + bv;
        if (R) y = y + *(const f32x4 *)(R + o);
        if (relu) { y.x = fmaxf(y.x, 0.0f); y.y = fmaxf(y.y, 0.0f); y.z = fmaxf(y.z, 0.0f); y.w = fmaxf(y.w, 0.0f); }
        *(f32x4 *)(Y + o) = y;
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
    static thread_local bool attr_set = false;
    if (!attr_set) {
        XQ_TRY(hipFuncSetAttribute((const void *)k_wino_conv, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set = true;
    }
    const int n_groups = (batch * 25 + TILES - 1) / TILES;
    const int ng = channels / NCO;
    const int per = 8 / ng;
    const int rows = (n_groups + per - 1) / per;
    hipLaunchKernelGGL(k_wino_conv, dim3(rows * 8), dim3(512), LDS_BYTES, (hipStream_t)stream, dev_x, dev_u, dev_bias,
                       dev_residual, dev_y, batch, channels, relu, n_groups);
    return xq::launch_status();
}

}  // extern "C"
