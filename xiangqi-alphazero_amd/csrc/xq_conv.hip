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
// tiles of 32x32 = 128 VGPRs.  Per 8-channel chunk: every U element is used by exactly one wave, so the B operand
// goes global (L2) -> registers, one 16 B load per lane and frequency, issued one chunk ahead -- no LDS and no
// barrier for it; the raw input of the 4 boards a tile group can touch is staged in LDS once per 16 channels
// (double-buffered; 80-byte position stride: 2-way instead of 8-way bank conflicts on the strided tile reads);
// every thread turns one (tile, channel quad, Winograd row) into four B^T d B values for the next chunk.
// Epilogue: the column half of A^T M A in registers, the row half across the 4 waves of a channel half through
// LDS (the 128 KB of chunk buffers are reused), then bias + residual + ReLU and coalesced NHWC stores.
// Blocks are dealt so that each XCD works on one 64-channel slice of U at a time (1 MB at C=256: L2-resident).
#include <cstdlib>
#include <type_traits>

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
constexpr int XZERO = XPOS * XSTRIDE;        // a zeroed 16-byte slot behind the staged positions (out-of-board reads)
constexpr int XRAW_BYTES = (XPOS + 1) * XSTRIDE;
constexpr int E_BYTES = 16 * TILES * 32 * 4;       // epilogue exchange planes (128 KB) reuse the whole image
constexpr int LDS_BYTES = (2 * VBUF_BYTES + 2 * XRAW_BYTES) > E_BYTES ? (2 * VBUF_BYTES + 2 * XRAW_BYTES) : E_BYTES;

__device__ __forceinline__ f32x4 ld4(const char *p) { return *(const f32x4 *)p; }

__global__ __launch_bounds__(512, 2) void k_wino_conv(const float *__restrict__ X, const float *__restrict__ Ug,
                                                      const float *__restrict__ bias, const float *__restrict__ R,
                                                      float *__restrict__ Y, int B, int C, int relu, int n_groups,
                                                      unsigned long long *__restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *Vb = lds;                                   // [2][16][2][64] float4
    char *Xr = lds + 2 * VBUF_BYTES;                  // [2][360][80 B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int NG = C / NCO;                           // channel groups; divides 8
    const int per = 8 / NG;
    const int xcd = blockIdx.x & 7, rr = blockIdx.x >> 3;
    const int cog = xcd % NG;
    const int tg = rr * per + xcd / NG;
    if (tg >= n_groups) return;
    unsigned long long st0 = 0, st1 = 0, st2 = 0;
    if (stamps) st0 = __builtin_amdgcn_s_memrealtime();
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
    int xoff[2][4];                                   // byte offsets into Xr (the zero slot when out of the board)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int y = 2 * ty - 1 + (k == 0 ? r1 : r2);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int x = 2 * tx - 1 + c;
            const bool ok = tile_ok && (unsigned)y < 10u && (unsigned)x < 9u;
            xoff[k][c] = ok ? ((tb - b_lo) * 90 + y * 9 + x) * XSTRIDE : XZERO;
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

    // B operand: lane (h, n) of wave (p, half) needs U[xi][8*chunk + 4h + j][64*cog + 32*half + n], j = 0..3
    const unsigned ul = (h * (NCO * 4) + (wch * 32 + l31) * 4 + (wp * 4) * (2 * NCO * 4)) * 4;   // byte offset in a chunk
    auto load_u = [&](int chunk, f32x4 *dst) {
        const char *src = (const char *)(ug + (size_t)chunk * (UBUF_BYTES / 4));                  // wave-uniform
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q] = *(const f32x4 *)(src + ul + q * (2 * NCO * 16));
    };
    f32x4 xreg[3];
    auto load_x = [&](int super) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
            xreg[k] = xv[k] ? *(const f32x4 *)(xg[k] + super * 16) : z;
        }
    };
    auto store_x = [&](int super) {
        char *dst = Xr + (super & 1) * XRAW_BYTES;
#pragma unroll
        for (int k = 0; k < 3; ++k)
            if (tid + 512 * k < XPOS * 4) *(f32x4 *)(dst + xl[k]) = xreg[k];
    };
    auto transform = [&](int chunk, int buf) {
        const char *xr = Xr + ((chunk >> 1) & 1) * XRAW_BYTES + ((chunk & 1) * 2 + tr_qd) * 16;
        f32x4 w[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            w[c] = s1 * ld4(xr + xoff[0][c]) + s2 * ld4(xr + xoff[1][c]);
        }
        char *dst = Vb + buf * VBUF_BYTES + ((tr_i * 4) * 2 + tr_qd) * (TILES * 16) + tr_tile * 16;
        *(f32x4 *)(dst + 0 * 2 * TILES * 16) = w[0] - w[2];
        *(f32x4 *)(dst + 1 * 2 * TILES * 16) = w[1] + w[2];
        *(f32x4 *)(dst + 2 * 2 * TILES * 16) = w[2] - w[1];
        *(f32x4 *)(dst + 3 * 2 * TILES * 16) = w[1] - w[3];
    };
    // One chunk of work for a wave: 32 MFMAs on V[buf] x u, with the B^T d B transform of chunk `nchunk` into V[buf^1]
    // threaded between them (sched_group_barrier pins the interleave).  Measured on gfx950 (tests/microbench): the fp32
    // MFMA does not co-execute with other vector work the way the bf16 matrix core does -- every LDS read, VALU op and
    // above all every global load a SIMD issues adds its issue time to the MFMA stream, whichever of the two resident
    // waves issues it -- so the lever is the COUNT of non-MFMA instructions per MFMA.  The transform is therefore
    // specialised per Winograd row (ROW is wave-uniform: waves 2r, 2r+1 own row r) so that the +-1 coefficients of B^T
    // are operand order, not multiplies, and the weight operand comes straight from L2 into registers.
    auto chunk_body = [&](int buf, const f32x4 *u, int nchunk, auto row_tag) {
        constexpr int ROW = decltype(row_tag)::value;
        const char *vb = Vb + buf * VBUF_BYTES + h * (TILES * 16) + l31 * 16 + (wp * 4) * (2 * TILES * 16);
        const char *xr = Xr + ((nchunk >> 1) & 1) * XRAW_BYTES + ((nchunk & 1) * 2 + tr_qd) * 16;
        char *dst = Vb + (buf ^ 1) * VBUF_BYTES + ((ROW * 4) * 2 + tr_qd) * (TILES * 16) + tr_tile * 16;
        f32x4 w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 a0 = ld4(vb + q * (2 * TILES * 16));
            const f32x4 a1 = ld4(vb + q * (2 * TILES * 16) + 32 * 16);
            const f32x4 d1 = ld4(xr + xoff[0][q]), d2 = ld4(xr + xoff[1][q]);
            w[q] = ROW == 0 ? d1 - d2 : ROW == 1 ? d1 + d2 : ROW == 2 ? d2 - d1 : d1 - d2;   // rows of B^T d
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[q][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[j], u[q][j], acc[q][0], 0, 0, 0);
                acc[q][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[j], u[q][j], acc[q][1], 0, 0, 0);
            }
        }
        *(f32x4 *)(dst + 0 * 2 * TILES * 16) = w[0] - w[2];
        *(f32x4 *)(dst + 1 * 2 * TILES * 16) = w[1] + w[2];
        *(f32x4 *)(dst + 2 * 2 * TILES * 16) = w[2] - w[1];
        *(f32x4 *)(dst + 3 * 2 * TILES * 16) = w[1] - w[3];
#pragma unroll
        for (int i = 0; i < 32; ++i) {                // pin the interleave: per MFMA one LDS read and a little VALU
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
            if (i >= 28) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
    };

    // ---- prologue ------------------------------------------------------------------------------------------
    const int NSUP = NCH / 2;
    f32x4 uA[4], uB[4];
    load_x(0);
    load_u(0, uA);
    if (tid < 8) {                                    // zero slots of both staging buffers (4 x 16 B cover the sub offsets)
        f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        *(f32x4 *)(Xr + (tid >> 2) * XRAW_BYTES + XZERO + (tid & 3) * 16) = z;
    }
    store_x(0);
    __syncthreads();
    if (NSUP > 1) load_x(1);
    transform(0, 0);
    __syncthreads();

    if (stamps) st1 = __builtin_amdgcn_s_memrealtime();
    // ---- main loop, two 8-channel chunks per trip (U register sets alternate) ------------------------------
    auto run_chunk = [&](int buf, const f32x4 *u, int nchunk) {
        switch (tr_i) {                               // wave-uniform
        case 0: chunk_body(buf, u, nchunk, std::integral_constant<int, 0>{}); break;
        case 1: chunk_body(buf, u, nchunk, std::integral_constant<int, 1>{}); break;
        case 2: chunk_body(buf, u, nchunk, std::integral_constant<int, 2>{}); break;
        default: chunk_body(buf, u, nchunk, std::integral_constant<int, 3>{}); break;
        }
    };
    for (int c = 0; c < NCH; c += 2) {
        const int sup = c >> 1;
        // even chunk c: V[0], U set A; next chunk's U and the staging of superchunk sup+1 are fetched first
        load_u(c + 1, uB);
        if (sup + 1 < NSUP) {
            store_x(sup + 1);                         // this buffer was last read two chunks ago
            if (sup + 2 < NSUP) load_x(sup + 2);
        }
        run_chunk(0, uA, c + 1);
        __syncthreads();
        // odd chunk c+1: V[1], U set B.  On the last trip the prefetch/transform targets are clamped (harmless
        // redundant work into buffers nobody reads) so the body stays one straight-line scheduling region.
        const int nc = c + 2 < NCH ? c + 2 : c;
        load_u(nc, uA);
        run_chunk(1, uB, nc);
        __syncthreads();
    }
    if (stamps) st2 = __builtin_amdgcn_s_memrealtime();
    // ---- epilogue: Y = A^T M A, bias, residual, ReLU --------------------------------------------------------
    // Output coordinates and the residual loads come first: their HBM latency then hides behind the register
    // reduction and the LDS exchange (the prefetch registers of the main loop are dead by now).
    const int c4 = tid & 15;
    const int co = c4 * 4, ech = co >> 5, ecol = co & 31;
    const f32x4 bv = *(const f32x4 *)(bias + cog * NCO + co);
    size_t oaddr[8];
    f32x4 resv[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int pidx = it * 32 + (tid >> 4);
        const int tile = pidx >> 2, a = (pidx >> 1) & 1, b = pidx & 1;
        const int g = t0 + tile;
        const int bd = g / 25, t2 = g - bd * 25, ty2 = t2 / 5, tx2 = t2 - ty2 * 5;
        const int oy = 2 * ty2 + a, ox = 2 * tx2 + b;
        const bool ok = g < T && ox < 9;
        oaddr[it] = ok ? ((size_t)bd * 90 + oy * 9 + ox) * C + cog * NCO + co : (size_t)-1;
        f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        resv[it] = (ok && R) ? *(const f32x4 *)(R + oaddr[it]) : z;
    }
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
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int pidx = it * 32 + (tid >> 4);
        const int tile = pidx >> 2, a = (pidx >> 1) & 1, b = pidx & 1;
        if (oaddr[it] == (size_t)-1) continue;
        const float *e0 = E + (((ech * 4) * 2 + b) * TILES + tile) * 32 + ecol;
        const int pstride = 2 * TILES * 32;           // next Winograd row p
        f32x4 y;
        if (a == 0) y = *(const f32x4 *)(e0) + *(const f32x4 *)(e0 + pstride) + *(const f32x4 *)(e0 + 2 * pstride);
        else y = *(const f32x4 *)(e0 + pstride) - *(const f32x4 *)(e0 + 2 * pstride) - *(const f32x4 *)(e0 + 3 * pstride);
        y = y + bv + resv[it];
        if (relu) { y.x = fmaxf(y.x, 0.0f); y.y = fmaxf(y.y, 0.0f); y.z = fmaxf(y.z, 0.0f); y.w = fmaxf(y.w, 0.0f); }
        *(f32x4 *)(Y + oaddr[it]) = y;
    }
    if (stamps && lane == 0) {                        // diagnostic path only (xq_wino_conv3x3_dbg)
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long *d = stamps + (size_t)blockIdx.x * 16;
        if (wave == 0) { d[0] = st0; d[1] = st1; d[2] = st2; d[3] = __builtin_amdgcn_s_memrealtime(); d[4] = xcc; }
        d[8 + wave] = hwid;
    }
}


// ---------------------------------------------------------------------------------------------------------
// Second decomposition: workgroup = 32 tiles x 64 output channels, 4 waves, two workgroups resident per CU.
// Wave p owns Winograd row p (xi = 4p..4p+3) for all 64 channels: 4 xi x 1 M-tile x 2 N-tiles = 8 accumulator tiles.
// Lane (h, m) of wave p needs, as its MFMA A operand, V[xi][tile m][ci = 4h..4h+3] -- exactly the four values the
// transform of (tile m, channel quad h, row p) produces, so every lane transforms what it multiplies: no V buffers in
// LDS, no A-fragment reads, and only the raw-input staging needs a barrier (one per 16 channels).  The two workgroups
// of a CU run unsynchronised, so one's prologue, barrier waits and epilogue are covered by the other's MFMAs.
//
// Staged input: boards with a zero halo, position P(b, y, x) = (11 b + y + 1) * 10 + x + 1 for y in [-1, 10], x in
// [-1, 9] (row 10 of a board is row -1 of the next, column 9 of a row is column -1 of the next: all zero and never
// written), 80 bytes per position (16 channels + 16 B pad).  A tile's 4x4 patch is then base + (10 r + c) * 80: one
// address register and immediates, no bounds logic.
constexpr int T2 = 32;                       // tiles per workgroup
constexpr int XPOS2 = (3 * 11 + 1) * 10 + 1; // 3 boards with halo: 341 positions
constexpr int XRAW2 = (XPOS2 + 1) * XSTRIDE; // 27360 B per staging buffer (+ one dump position)
constexpr int E2_BYTES = 4 * 2 * T2 * NCO * 4;          // epilogue exchange [row p][b][tile][co] (64 KB)
constexpr int LDS2_BYTES = 2 * XRAW2 > E2_BYTES ? 2 * XRAW2 : E2_BYTES;
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

__global__ __launch_bounds__(256, 2) void k_wino_conv2(const float *__restrict__ X, const float *__restrict__ Ug,
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
    const int t0 = tg * T2;
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
    const int tl = (t0 + T2 - 1 < T ? t0 + T2 - 1 : T - 1);
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
        xl[k] = (ok ? (bi * 11 + y + 1) * 10 + x + 1 : XPOS2) * XSTRIDE + spart * 16;
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
        char *dst = Xr + (super & 1) * XRAW2;
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
    const int NSUP = NCH / 2;
    load_x(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) load_u(0, q);
    {                                                 // zero both staging buffers (the halo stays zero from here on)
        f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int o = tid * 16; o < 2 * XRAW2; o += 256 * 16) *(f32x4 *)(Xr + o) = z;
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
        chunk_body(c + 2, std::integral_constant<int, XRAW2>{}, -1);                // chunk c+2: buffer 1, lower half
        chunk_body(c + 3, std::integral_constant<int, XRAW2 + 32>{}, sup + 2);      // chunk c+3: buffer 1, upper half
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
            E[((wp * 2 + 0) * T2 + tile) * NCO + 32 * n + l31] = y0[e];
            E[((wp * 2 + 1) * T2 + tile) * NCO + 32 * n + l31] = y1[e];
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int pidx = it * 16 + (tid >> 4);
        const int tile = pidx >> 2, ya = (pidx >> 1) & 1, yb = pidx & 1;
        if (oaddr[it] == (size_t)-1) continue;
        const float *e0 = E + (yb * T2 + tile) * NCO + co;
        const int pstride = 2 * T2 * NCO;             // next Winograd row p
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
    static thread_local bool attr_set = false;
    if (!attr_set) {
        XQ_TRY(hipFuncSetAttribute((const void *)k_wino_conv, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        attr_set = true;
    }
    static const bool v1 = getenv("XQ_WINO_V1") != nullptr;
    if (!v1) {
        if ((unsigned long long)batch * 90ull * (unsigned)channels * 4ull >= (1ull << 32)) return XQ_ERR_ARG;   // 32-bit staging offsets
        static thread_local bool attr2_set = false;
        if (!attr2_set) {
            XQ_TRY(hipFuncSetAttribute((const void *)k_wino_conv2, hipFuncAttributeMaxDynamicSharedMemorySize, LDS2_BYTES));
            attr2_set = true;
        }
        const int n_groups = (batch * 25 + T2 - 1) / T2;
        const int per = 8 / (channels / NCO);
        const int rows = (n_groups + per - 1) / per;
        hipLaunchKernelGGL(k_wino_conv2, dim3(rows * 8), dim3(256), LDS2_BYTES, (hipStream_t)stream, dev_x, dev_u, dev_bias,
                           dev_residual, dev_y, batch, channels, relu, n_groups);
        return xq::launch_status();
    }
    const int n_groups = (batch * 25 + TILES - 1) / TILES;
    const int ng = channels / NCO;
    const int per = 8 / ng;
    const int rows = (n_groups + per - 1) / per;
    hipLaunchKernelGGL(k_wino_conv, dim3(rows * 8), dim3(512), LDS_BYTES, (hipStream_t)stream, dev_x, dev_u, dev_bias,
                       dev_residual, dev_y, batch, channels, relu, n_groups, (unsigned long long *)nullptr);
    return xq::launch_status();
}

/* diagnostic twin: per-block 100 MHz timestamps {start, after prologue, after main loop, end, HW_ID, XCC_ID} into
 * dev_stamps[grid][16] (grid = 8 * ceil(ceil(batch*25/64) / (8 / (channels/64)))).  Not part of the product path. */
int xq_wino_conv3x3_dbg(const float *dev_x, const float *dev_u, const float *dev_bias, const float *dev_residual, float *dev_y,
                        int batch, int channels, int relu, unsigned long long *dev_stamps, void *stream) {
    if (!dev_x || !dev_u || !dev_bias || !dev_y || batch <= 0 || !dev_stamps) return XQ_ERR_ARG;
    if (channels < 64 || channels % 64 || 8 % (channels / 64)) return XQ_ERR_ARG;
    XQ_TRY(hipFuncSetAttribute((const void *)k_wino_conv, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    const int n_groups = (batch * 25 + TILES - 1) / TILES;
    const int per = 8 / (channels / NCO);
    const int rows = (n_groups + per - 1) / per;
    hipLaunchKernelGGL(k_wino_conv, dim3(rows * 8), dim3(512), LDS_BYTES, (hipStream_t)stream, dev_x, dev_u, dev_bias,
                       dev_residual, dev_y, batch, channels, relu, n_groups, dev_stamps);
    return xq::launch_status();
}

}  // extern "C"
