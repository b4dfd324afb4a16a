// xq_train.hip -- train-step kernels around the hand-written convolution (the reference trains through
// torch.nn.BatchNorm2d + ReLU inside ResBlock, training/model.py:20-36 under training/train.py:376-447).
//
// BatchNorm in TRAINING mode on NHWC activations x[rows][C] (rows = batch * 90), fused with what surrounds it in a ResBlock:
//   forward :  y = act( (x - mean_c) * invstd_c * gamma_c + beta_c  (+ residual) ),   batch statistics over the rows (biased variance),
//              running_mean / running_var updated as torch does (momentum, unbiased variance), save_mean / save_invstd kept for backward;
//   backward:  g = dy * (y > 0) when act = ReLU;  dbeta = sum g;  dgamma = sum g * xhat;
//              dx = gamma * invstd * (g - dbeta / rows - xhat * dgamma / rows);  d_residual = g.
// All of it is HBM/cache-bound streaming over a [rows][C] float32 tensor (23.6 MB at batch 256, C = 256): a thread owns four consecutive
// channels (16-byte accesses, a wave covers 1 KB of a row), the rows are cut into NSEG contiguous segments (one workgroup each), per-segment
// sums are float64 and reduced in a fixed order by a small finalize kernel -- deterministic, no atomics.  Algorithmic bytes:
// forward 3 passes (+1 with a residual), backward 7 (+1), of rows * C * 4 bytes.
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "xq_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NSEG = 512;                         // row segments = workgroups of the reduction kernels (two per CU)

// Per-segment partial sums.  MODE 0 (forward): a = sum x, b = sum x^2.  MODE 1 (backward): a = sum g, b = sum g * xhat.
template <int MODE>
__global__ __launch_bounds__(256) void k_bn_partial(const float *__restrict__ X, const float *__restrict__ DY, const float *__restrict__ Yout,
                                                    const float *__restrict__ mean, const float *__restrict__ invstd, long long rows, int C,
                                                    int relu, double *__restrict__ part) {
    __shared__ double red[256][8];
    const int tpr = C >> 2, rp = 256 / tpr;        // threads per row, rows per pass
    const int tc = threadIdx.x % tpr, tr = threadIdx.x / tpr;
    const long long per = (rows + NSEG - 1) / NSEG;
    const long long lo = (long long)blockIdx.x * per, hi = lo + per < rows ? lo + per : rows;
    double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
    f32x4 mu = {0.0f, 0.0f, 0.0f, 0.0f}, is = {0.0f, 0.0f, 0.0f, 0.0f};
    if (MODE == 1) {
        mu = *(const f32x4 *)(mean + 4 * tc);
        is = *(const f32x4 *)(invstd + 4 * tc);
    }
#pragma unroll 4
    for (long long r = lo + tr; r < hi; r += rp) {
        const size_t o = (size_t)r * C + 4 * tc;
        const f32x4 x = *(const f32x4 *)(X + o);
        if (MODE == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a[k] += (double)x[k];
                b[k] += (double)x[k] * (double)x[k];
            }
        } else {
            f32x4 g = *(const f32x4 *)(DY + o);
            if (relu) {
                const f32x4 y = *(const f32x4 *)(Yout + o);
#pragma unroll
                for (int k = 0; k < 4; ++k) g[k] = y[k] > 0.0f ? g[k] : 0.0f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float xh = (x[k] - mu[k]) * is[k];
                a[k] += (double)g[k];
                b[k] += (double)g[k] * (double)xh;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        red[threadIdx.x][k] = a[k];
        red[threadIdx.x][4 + k] = b[k];
    }
    __syncthreads();
    if (tr == 0) {
        for (int j = 1; j < rp; ++j)
#pragma unroll
            for (int k = 0; k < 8; ++k) red[tc][k] += red[j * tpr + tc][k];
        double *pa = part + ((size_t)blockIdx.x * C + 4 * tc), *pb = part + ((size_t)(NSEG + blockIdx.x) * C + 4 * tc);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            pa[k] = red[tc][k];
            pb[k] = red[tc][4 + k];
        }
    }
}

// Sum of the NSEG partials of 4 channels by one workgroup (C / 4 workgroups): thread (j, c) adds segments j, j + 64, j + 128, j + 192, then a
// fixed binary tree over the 64 j through LDS -- a fixed summation order, so the result does not depend on scheduling.
__device__ __forceinline__ void reduce_partials(const double *__restrict__ part, int C, double &s, double &q) {
    __shared__ double rs[64][4], rq[64][4];
    const int c = threadIdx.x & 3, j = threadIdx.x >> 2, ch = blockIdx.x * 4 + c;
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int k = j; k < NSEG; k += 64) {                            // NSEG / 64 partials per thread
        a += part[(size_t)k * C + ch];
        b += part[(size_t)(NSEG + k) * C + ch];
    }
    rs[j][c] = a;
    rq[j][c] = b;
    __syncthreads();
    for (int w = 32; w >= 1; w >>= 1) {
        if (j < w) {
            rs[j][c] += rs[j + w][c];
            rq[j][c] += rq[j + w][c];
        }
        __syncthreads();
    }
    s = rs[0][c];
    q = rq[0][c];
}

// forward finalize: grid C / 4 workgroups of 256 threads
__global__ __launch_bounds__(256) void k_bn_fwd_finalize(const double *__restrict__ part, long long rows, int C, float momentum, float eps,
                                                          float *__restrict__ run_mean, float *__restrict__ run_var,
                                                          float *__restrict__ save_mean, float *__restrict__ save_invstd,
                                                          long long *__restrict__ batches_tracked) {
    double s, q;
    reduce_partials(part, C, s, q);
    if (batches_tracked != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *batches_tracked += 1;    // BatchNorm2d.num_batches_tracked
    if (threadIdx.x >= 4) return;
    const int c = blockIdx.x * 4 + threadIdx.x;
    const double n = (double)rows, m = s / n;
    double var = q / n - m * m;
    if (var < 0.0) var = 0.0;
    save_mean[c] = (float)m;
    save_invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean != nullptr) {
        const double unb = rows > 1 ? var * n / (n - 1.0) : var;
        run_mean[c] = (float)((1.0 - (double)momentum) * (double)run_mean[c] + (double)momentum * m);
        run_var[c] = (float)((1.0 - (double)momentum) * (double)run_var[c] + (double)momentum * unb);
    }
}

__global__ __launch_bounds__(256) void k_bn_bwd_finalize(const double *__restrict__ part, int C, float *__restrict__ dgamma,
                                                          float *__restrict__ dbeta) {
    double s, q;
    reduce_partials(part, C, s, q);
    if (threadIdx.x >= 4) return;
    const int c = blockIdx.x * 4 + threadIdx.x;
    dbeta[c] = (float)s;
    dgamma[c] = (float)q;
}

__global__ __launch_bounds__(256) void k_bn_apply(const float *__restrict__ X, const float *__restrict__ R, const float *__restrict__ gamma,
                                                  const float *__restrict__ beta, const float *__restrict__ mean,
                                                  const float *__restrict__ invstd, long long quads, int C, int relu, float *__restrict__ Y) {
    const long long stride = (long long)gridDim.x * 256;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < quads; i += stride) {
        const int c = (int)((i * 4) % C);
        const f32x4 x = *(const f32x4 *)(X + i * 4);
        const f32x4 mu = *(const f32x4 *)(mean + c), is = *(const f32x4 *)(invstd + c), ga = *(const f32x4 *)(gamma + c),
                    be = *(const f32x4 *)(beta + c);
        f32x4 y;
#pragma unroll
        for (int k = 0; k < 4; ++k) y[k] = (x[k] - mu[k]) * is[k] * ga[k] + be[k];
        if (R != nullptr) {
            const f32x4 r = *(const f32x4 *)(R + i * 4);
            y = y + r;
        }
        if (relu) {
#pragma unroll
            for (int k = 0; k < 4; ++k) y[k] = y[k] > 0.0f ? y[k] : 0.0f;
        }
        *(f32x4 *)(Y + i * 4) = y;
    }
}

__global__ __launch_bounds__(256) void k_bn_dx(const float *__restrict__ DY, const float *__restrict__ X, const float *__restrict__ Yout,
                                               const float *__restrict__ gamma, const float *__restrict__ mean,
                                               const float *__restrict__ invstd, const float *__restrict__ dgamma,
                                               const float *__restrict__ dbeta, long long quads, long long rows, int C, int relu,
                                               float *__restrict__ DX, float *__restrict__ DR) {
    const long long stride = (long long)gridDim.x * 256;
    const float inv_n = (float)(1.0 / (double)rows);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < quads; i += stride) {
        const int c = (int)((i * 4) % C);
        f32x4 g = *(const f32x4 *)(DY + i * 4);
        if (relu) {
            const f32x4 y = *(const f32x4 *)(Yout + i * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) g[k] = y[k] > 0.0f ? g[k] : 0.0f;
        }
        const f32x4 x = *(const f32x4 *)(X + i * 4);
        const f32x4 mu = *(const f32x4 *)(mean + c), is = *(const f32x4 *)(invstd + c), ga = *(const f32x4 *)(gamma + c),
                    dg = *(const f32x4 *)(dgamma + c), db = *(const f32x4 *)(dbeta + c);
        f32x4 dx;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xh = (x[k] - mu[k]) * is[k];
            dx[k] = ga[k] * is[k] * (g[k] - db[k] * inv_n - xh * dg[k] * inv_n);
        }
        *(f32x4 *)(DX + i * 4) = dx;
        if (DR != nullptr) *(f32x4 *)(DR + i * 4) = g;
    }
}


// ------------------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the 3x3 convolution in the Winograd domain (the transpose of xq_conv.hip's algorithm):
//   dW = G_r'^T [ sum over tiles  (A_r dY A_c^T)  (.)  (B_r'^T d B_c') ] G_c'
// with the SAME matrices as the forward kernel: per output tile (2 x 3) the gradient tile is expanded to the 4 x 5 frequencies (A), the input
// patch (4 x 5) is transformed as in the forward pass (B'), and for each of the 20 frequencies the products are summed over tiles by the fp32
// MFMA: M = 32 output channels, N = 32 input channels, K = 2 tiles per instruction -- 300 instead of 810 multiplies per tile and channel pair.
// A workgroup owns a 128 x 32 (or 64 x 64) block of channel pairs and one contiguous range of tiles (split-K); wave p owns Winograd row p:
// 5 frequencies x 4 MFMA tiles = 320 accumulators (15 tiles in AGPRs, 5 pinned to VGPRs, as xq_conv.hip).  Operands come straight from global
// memory (NHWC: channels contiguous), one tile per half-wave, two steps in flight, transformed in registers; board edges are out-of-range buffer
// offsets (zeros, no traffic).  No LDS and no barrier in the main loop.  Epilogue: the column half of G'^T . G' in registers, the row half
// across the four waves through LDS, 9 values per channel pair into this split's partial; k_wgrad_reduce adds the splits in order.
#ifndef XQ_WGRAD_STAGES
#define XQ_WGRAD_STAGES 2
#endif
constexpr int WG_ESTR = 72;                                         // LDS row stride of the exchange (floats): half-waves 32 banks apart
constexpr int WG_LDS_BYTES = 4 * 3 * 32 * WG_ESTR * 4;               // [p][s][32 rows][72] = 110 592

typedef float f32x16 __attribute__((ext_vector_type(16)));

// f(integral_constant<int, 0>), f(integral_constant<int, 1>), ... : a compile-time unrolled loop whose index is a constant expression
template <int... I, class F>
__device__ __forceinline__ void for_seq(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, I>{}), ...);
}

// Epilogue of the weight-gradient kernels: dW[r][s] = sum_p G_r'[p][r] sum_j G_c'[j][s] acc[j] -- column half in registers, row half across the
// four waves through LDS -- 9 values per channel pair into this split's partial.
template <int MT>
__device__ __forceinline__ void wgrad_epilogue(f32x16 (&acc)[5][4], char *lds, float *__restrict__ part, int tid, int p, int n, int kp,
                                               int split, int cb, int nb, int C) {
    constexpr int NT = 4 / MT, BCO = 32 * MT, BCI = 32 * NT;
    // ---- epilogue: dW[r][s] = sum_p G_r'[p][r] sum_j G_c'[j][s] acc[j]; column half here, row half across the waves through LDS, two
    // accumulator tiles (f = 2 h, 2 h + 1) per round: exchange columns 0-31 / 32-63
    float *E = (float *)lds;
    const float k6 = 1.0f / 6.0f;
    const int rrow = tid >> 3, rcol = (tid & 7) * 8, ru = rcol >> 5;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int f = 2 * h + u;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float m0 = acc[0][f][e], m1 = acc[1][f][e], m2 = acc[2][f][e], m3 = acc[3][f][e], m4 = acc[4][f][e];
                const float h1 = 0.5f * m1;
                const float c0 = (0.5f * m0 + h1) + k6 * (m2 + m3);
                const float c1 = h1 + k6 * (2.0f * m3 - m2);
                const float c2 = (h1 + k6 * (m2 + 4.0f * m3)) + m4;
                const int row = (e & 3) + 8 * (e >> 2) + 4 * kp;
                float *dst = E + ((p * 3) * 32 + row) * WG_ESTR + u * 32 + n;
                dst[0] = c0;
                dst[32 * WG_ESTR] = c1;
                dst[2 * 32 * WG_ESTR] = c2;
            }
        }
        __syncthreads();
        // reader: exchange row rrow, columns rcol .. rcol + 7 -> channel pair of the block
        //   MT = 2 (f = (mt = h, nt = u)): co = 32 h + rrow,          ci = rcol .. (32 u + n)
        //   MT = 4 (f = mt = 2 h + u):     co = 4 rrow + 2 h + ru,    ci = (rcol & 31) ..
        const int co_l = MT == 4 ? 4 * rrow + 2 * h + ru : 32 * h + rrow, ci_l = MT == 4 ? (rcol & 31) : rcol;
        float *out = part + (((size_t)split * 9) * C + (size_t)(cb * BCO + co_l)) * C + nb * BCI + ci_l;
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3) {
            f32x4 q[4][2];
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {
                const float *src = E + ((pp * 3 + s3) * 32 + rrow) * WG_ESTR + rcol;
                q[pp][0] = *(const f32x4 *)src;
                q[pp][1] = *(const f32x4 *)(src + 4);
            }
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                const f32x4 r0 = q[0][hh] + 0.5f * (q[1][hh] - q[2][hh]);
                const f32x4 r1 = 0.5f * (q[1][hh] + q[2][hh]);
                const f32x4 r2 = 0.5f * (q[1][hh] - q[2][hh]) + q[3][hh];
                *(f32x4 *)(out + ((size_t)(0 * 3 + s3) * C) * C + 4 * hh) = r0;
                *(f32x4 *)(out + ((size_t)(1 * 3 + s3) * C) * C + 4 * hh) = r1;
                *(f32x4 *)(out + ((size_t)(2 * 3 + s3) * C) * C + 4 * hh) = r2;
            }
        }
        if (h == 0) __syncthreads();
    }
}

// MT = 4: block of 128 output x 32 input channels (C % 128 == 0): the gradient tile comes by 16-byte loads (lane n: output channels 4 n .. 4 n + 3,
//         so MFMA tile mt holds the channels = mt mod 4), the input by dword loads: 16 vector-memory instructions per step and wave -- the
//         texture addresser takes ~16 cycles per wave instruction whatever its width, and with dword loads only (32 per step) it was the bound.
// MT = 2: block of 64 x 64 channels, dword loads (C = 64).
template <int MT>
__global__ __launch_bounds__(256, 1) void k_wino_wgrad(const float *__restrict__ X, const float *__restrict__ DY, float *__restrict__ part,
                                                        int B, int C, int n_split) {
    constexpr int NT = 4 / MT, BCO = 32 * MT, BCI = 32 * NT;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, kp = lane >> 5;
    const int p = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform, and KNOWN uniform: the row offsets below go into SGPRs
                                                                     // (as a VGPR value every load grew a readfirstlane waterfall loop)
    const int NBI = C / BCI, nblk = (C / BCO) * NBI;
    const int total = n_split * nblk;
    const int L = (int)(blockIdx.x & 7) * (total >> 3) + (int)(blockIdx.x >> 3);       // consecutive logical ids share an XCD (and its L2)
    const int split = L / nblk, cb = (L % nblk) / NBI, nb = L % NBI;
    const int T = B * 15, pairs = (T + 1) >> 1, pps = (pairs + n_split - 1) / n_split;
    const int t_lo = 2 * split * pps, t_hi = t_lo + 2 * pps < T ? t_lo + 2 * pps : T;
    const unsigned C4 = (unsigned)C * 4u;
    const unsigned nbytes = (unsigned)B * 90u * C4;
    // input descriptor starts 10 positions BEFORE the tensor: patch element (rho, c) of the tile at position pos0 is at offset
    // (pos0 + 9 rho + c) * C4 >= 0; elements before the tensor are exactly the masked ones (never fetched)
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)X - 10 * (size_t)C4), 0, (int)(nbytes + 10u * C4), 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)DY, 0, (int)nbytes, 0x00020000);
    const unsigned OOB = 0xFFFFF000u;                                // out of range for every tensor the host check admits
    const float sg = p == 1 ? 1.0f : -1.0f;
    const int rho_a = p == 0 ? 0 : 1, rho_b = p == 3 ? 3 : 2;
    const unsigned xch = (unsigned)(nb * BCI + n) * 4u, ych = (unsigned)(cb * BCO + (MT == 4 ? 4 * n : n)) * 4u;

    f32x16 acc[5][4];                                                // [frequency j][tile f = mt * NT + nt]
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][f][e] = 0.0f;

    // operands of TWO steps in flight (stage = step parity): one step of MFMAs (1 280 cycles) does not cover a loaded L2 / Infinity-Cache
    // round trip
    constexpr int NSTG = XQ_WGRAD_STAGES;
    float xa_[NSTG][NT][5], xb_[NSTG][NT][5], ya_[NSTG][MT][3], yb_[NSTG][MT][3];
    int r15_[NSTG];                                                     // tile-in-board (0..14) of the stage's NEXT issue, this lane's tile
    unsigned bbase_[NSTG];                                              // 90 * board of it
#pragma unroll
    for (int sgi = 0; sgi < NSTG; ++sgi) {
        const int tau0 = t_lo + 2 * sgi + kp, b0 = tau0 / 15;
        r15_[sgi] = tau0 - 15 * b0;
        bbase_[sgi] = 90u * (unsigned)b0;
    }
    auto issue = [&](auto stage_tag, int t) __attribute__((always_inline)) {
        constexpr int SG = decltype(stage_tag)::value;
        auto &xa = xa_[SG]; auto &xb = xb_[SG]; auto &ya = ya_[SG]; auto &yb = yb_[SG];
        const int tau = t + kp;
        // tile -> (board, ty, tx): each stage advances by 4 tiles per call, so board and tile-in-board are carried, not divided out
        int &r = r15_[SG];
        unsigned &bb = bbase_[SG];
        const int ty = (r * 11) >> 5, tx = r - 3 * ty;               // r / 3 for r < 32
        const unsigned pos0 = bb + (unsigned)(18 * ty + 3 * tx);
        const bool ok = tau < t_hi;
        r += 2 * NSTG;
        if (r >= 15) {
            r -= 15;
            bb += 90u;
        }
        const unsigned xv = ok ? pos0 * C4 + xch : OOB, yv = ok ? pos0 * C4 + ych : OOB;
        const unsigned xv0 = tx > 0 ? xv : OOB, xv4 = tx < 2 ? xv : OOB;
        const bool ma = p == 0 && ty == 0, mb = p == 3 && ty == 4;
        const unsigned a1 = ma ? OOB : xv, a0 = ma ? OOB : xv0, a4 = ma ? OOB : xv4;
        const unsigned b1 = mb ? OOB : xv, b0 = mb ? OOB : xv0, b4 = mb ? OOB : xv4;
        const unsigned ya_v = p == 3 ? OOB : yv, yb_v = p == 0 ? OOB : yv;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                xa[nt][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, c == 0 ? a0 : c == 4 ? a4 : a1, (unsigned)(rho_a * 9 + c) * C4 + 128u * nt, 0));
                xb[nt][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, c == 0 ? b0 : c == 4 ? b4 : b1, (unsigned)(rho_b * 9 + c) * C4 + 128u * nt, 0));
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if constexpr (MT == 4) {
                // (the 8-byte builtin, __builtin_amdgcn_raw_buffer_load_b64, is lowered to a single-dword load by this hipcc: 16 bytes it is)
                const f32x4 qa = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yrs, ya_v, (unsigned)c * C4, 0));
                const f32x4 qb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(yrs, yb_v, (unsigned)(9 + c) * C4, 0));
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    ya[mt][c] = qa[mt];
                    yb[mt][c] = qb[mt];
                }
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    ya[mt][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrs, ya_v, (unsigned)c * C4 + 128u * mt, 0));
                    yb[mt][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yrs, yb_v, (unsigned)(9 + c) * C4 + 128u * mt, 0));
                }
            }
        }
    };

    auto step = [&](auto stage_tag, int t) __attribute__((always_inline)) {
        constexpr int SG = decltype(stage_tag)::value;
        auto &xa = xa_[SG]; auto &xb = xb_[SG]; auto &ya = ya_[SG]; auto &yb = yb_[SG];
        float v[NT][5], g[MT][5];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {                           // B_c'^T (B_r'^T d): the forward kernel's formulas
            float w[5];
#pragma unroll
            // products by +-1, 2 and 4 are exact, so these fused multiply-adds round exactly as the separate multiply and add of the
            // forward kernel's formulas do (one vector instruction instead of two; the fp32 MFMA shares the vector ALU)
            for (int c = 0; c < 5; ++c) w[c] = __builtin_fmaf(sg, xb[nt][c], xa[nt][c]);
            const float tt = w[3] - w[1];
            v[nt][0] = __builtin_fmaf(2.0f, w[0] - w[2], tt);
            v[nt][1] = __builtin_fmaf(2.0f, w[1], -w[3]) + w[2];
            v[nt][2] = 3.0f * w[2] - __builtin_fmaf(2.0f, w[1], w[3]);
            v[nt][3] = tt;
            v[nt][4] = __builtin_fmaf(-2.0f, tt, w[4] - w[2]);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {                           // A_c (A_r dY): rows (1,0) (1,1) (1,-1) (0,-1); columns at 0, 1, -1, 2, inf
            const float a0 = __builtin_fmaf(sg, yb[mt][0], ya[mt][0]), a1 = __builtin_fmaf(sg, yb[mt][1], ya[mt][1]),
                        a2 = __builtin_fmaf(sg, yb[mt][2], ya[mt][2]);
            const float s02 = a0 + a2;
            g[mt][0] = a0;
            g[mt][1] = s02 + a1;
            g[mt][2] = s02 - a1;
            g[mt][3] = __builtin_fmaf(2.0f, a1, __builtin_fmaf(4.0f, a2, a0));
            g[mt][4] = a2;
        }
        issue(stage_tag, t + 2 * NSTG);                                    // this stage's registers are free: fetch the step after next (past the
                                                                    // range: out-of-range offsets, no traffic)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
#pragma unroll
            for (int fo = 0; fo < 4; ++fo) {
                // the pinned tile's MFMA is inline asm, invisible to the compiler's hazard recogniser.  (a) A vector instruction that writes
                // one of its source registers must be 2 wait states ahead (the compiler puts s_nop 1 before its own MFMAs; found here as a
                // deterministic wrong tile): the s_nop travels inside the asm.  (b) It goes FIRST of its group, so three compiler-known
                // MFMAs separate it from the next step's vector writes to its source registers.
                const int f = (fo + 3) & 3;
                const int mt = f / NT, nt = f % NT;
                if (f == 3)
                    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[j][f]) : "v"(g[mt][j]), "v"(v[nt][j]));
                else
                    acc[j][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(g[mt][j], v[nt][j], acc[j][f], 0, 0, 0);
            }
        }
    };
    issue(std::integral_constant<int, 0>{}, t_lo);
    issue(std::integral_constant<int, 1>{}, t_lo + 2);
    if constexpr (NSTG == 3) issue(std::integral_constant<int, 2>{}, t_lo + 4);
    for (int t = t_lo; t < t_hi; t += 2 * NSTG) {                   // a step past t_hi multiplies zeros (every offset out of range)
        step(std::integral_constant<int, 0>{}, t);
        step(std::integral_constant<int, 1>{}, t + 2);
        if constexpr (NSTG == 3) step(std::integral_constant<int, 2>{}, t + 4);
    }
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[4][3]));        // asm MFMA results -> VALU readers (invisible to the hazard recogniser)

    wgrad_epilogue<MT>(acc, lds, part, tid, p, n, kp, split, cb, nb, C);
}

// ------------------------------------------------------------------------------------------------------------------------------------
// The same weight gradient with its operands staged ONCE per workgroup through LDS.  k_wino_wgrad above is bound by vector-memory issue:
// its four waves fetch overlapping rows of the same two tiles, 64 wave instructions per K-step and CU at ~20 cycles of the texture addresser
// each.  Here a K-step's operands -- one tile position (ty, tx) of two consecutive boards: 2 x 4 x 5 input positions x BCI channels and
// 2 x 2 x 3 gradient positions x BCO channels -- are fetched by 16-byte loads spread over the 256 threads (11 wave instructions per step at
// MT = 4, 13 at MT = 2), written to one of two LDS buffers, and read back by every wave in its MFMA operand layout (ds_read_b32 / _b128,
// conflict-free: consecutive lanes = consecutive channels).  Board edges: per-lane out-of-range offsets at the (few) global loads, so the LDS
// holds zeros there; a wave whose Winograd row does not use a gradient row reads a zeroed LDS page instead.  One barrier per step.
constexpr int WGL_ZERO_BYTES = 2048;

template <int MT>
__global__ __launch_bounds__(256, 1) void k_wino_wgrad_lds(const float *__restrict__ X, const float *__restrict__ DY, float *__restrict__ part,
                                                            int B, int C, int n_split) {
    constexpr int NT = 4 / MT, BCO = 32 * MT, BCI = 32 * NT;
    constexpr int XLP = BCI / 4, XPI = 64 / XLP, NIX = 40 / XPI;      // input: lanes per position, positions per wave instruction, instructions
    constexpr int YLP = BCO / 4, YPI = 64 / YLP, NIY = 12 / YPI;      // gradient: likewise
    constexpr int NI = NIX + NIY, IPW = (NI + 3) / 4;                 // instructions per step; per wave (wave w takes ids w, w + 4, ...)
    constexpr int XBYTES = 40 * BCI * 4, YBYTES = 12 * BCO * 4, BUF = XBYTES + YBYTES;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, n = lane & 31, kp = lane >> 5;
    const int p = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NBI = C / BCI, nblk = (C / BCO) * NBI;
    const int total = n_split * nblk;
    const int L = (int)(blockIdx.x & 7) * (total >> 3) + (int)(blockIdx.x >> 3);
    const int split = L / nblk, cb = (L % nblk) / NBI, nb = L % NBI;
    const int steps_all = ((B + 1) >> 1) * 15, pps = (steps_all + n_split - 1) / n_split;
    const int t_lo = split * pps, t_hi = t_lo + pps < steps_all ? t_lo + pps : steps_all;
    const unsigned C4 = (unsigned)C * 4u;
    const unsigned nbytes = (unsigned)B * 90u * C4;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)X - 10 * (size_t)C4), 0, (int)(nbytes + 10u * C4), 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void *)DY, 0, (int)nbytes, 0x00020000);
    const unsigned OOB = 0xFFFFF000u;
    const float sg = p == 1 ? 1.0f : -1.0f;
    const int rho_a = p == 0 ? 0 : 1, rho_b = p == 3 ? 3 : 2;

    // ---- loader role: this lane's part of the wave's instructions (constant over the steps)
    unsigned ld_off[IPW], ld_lds[IPW], ld_flags[IPW];                 // global offset without the step's base; LDS byte offset; edge flags
    bool ld_isx[IPW];
#pragma unroll
    for (int j = 0; j < IPW; ++j) {
        const int id = p + 4 * j;                                     // wave-uniform
        ld_isx[j] = id < NIX;
        if (id < NIX) {
            const int px = id * XPI + lane / XLP, ch = (lane % XLP) * 4;
            const int k = px / 20, rho = (px % 20) / 5, c = px % 5;
            ld_off[j] = (unsigned)(k * 90 + 9 * rho + c) * C4 + (unsigned)(nb * BCI + ch) * 4u;
            ld_lds[j] = (unsigned)(px * BCI + ch) * 4u;
            ld_flags[j] = (rho == 0 ? 1u : 0u) | (rho == 3 ? 2u : 0u) | (c == 0 ? 4u : 0u) | (c == 4 ? 8u : 0u) | (k == 1 ? 16u : 0u);
        } else {
            const int iy = id - NIX;
            const int py = iy * YPI + lane / YLP, ch = (lane % YLP) * 4;
            const int k = py / 6, row = (py % 6) / 3, c = py % 3;
            ld_off[j] = (unsigned)(k * 90 + 9 * row + c) * C4 + (unsigned)(cb * BCO + ch) * 4u;
            ld_lds[j] = (unsigned)XBYTES + (unsigned)(py * BCO + ch) * 4u;
            ld_flags[j] = k == 1 ? 16u : 0u;
        }
    }
    // ---- reader role: LDS byte offsets of this lane's operands inside a buffer
    const unsigned rx_a = (unsigned)((kp * 20 + rho_a * 5) * BCI + n) * 4u, rx_b = (unsigned)((kp * 20 + rho_b * 5) * BCI + n) * 4u;
    const unsigned ry_lane = (unsigned)(MT == 4 ? 4 * n : n) * 4u;
    const unsigned ZERO = 2u * BUF;                                   // the zero page behind the two buffers
    // gradient row 0 / row 1 of tile kp; a wave that does not use a row (p = 3: row 0, p = 0: row 1) reads zeros
    const unsigned ry_a = (unsigned)XBYTES + (unsigned)((kp * 6 + 0) * BCO) * 4u + ry_lane, ry_b = (unsigned)XBYTES + (unsigned)((kp * 6 + 3) * BCO) * 4u + ry_lane;

    f32x16 acc[5][4];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int f = 0; f < 4; ++f)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][f][e] = 0.0f;

    for (int o = tid * 16; o < 2 * BUF + WGL_ZERO_BYTES; o += 256 * 16) *(f32x4 *)(lds + o) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    f32x4 pre[2][IPW];                                               // two steps of global loads in flight (stage = step parity)
    auto fetch = [&](auto stage_tag, int t) __attribute__((always_inline)) {
        constexpr int SG = decltype(stage_tag)::value;
        const int bp = t / 15, r = t - 15 * bp, ty = r / 3, tx = r - 3 * ty;           // scalar
        const unsigned base = (unsigned)(2 * bp * 90 + 18 * ty + 3 * tx) * C4;
        const unsigned edge = (ty == 0 ? 1u : 0u) | (ty == 4 ? 2u : 0u) | (tx == 0 ? 4u : 0u) | (tx == 2 ? 8u : 0u) | (2 * bp + 1 >= B ? 16u : 0u);
        const bool live = t < t_hi;
#pragma unroll
        for (int j = 0; j < IPW; ++j) {
            if (p + 4 * j < NI) {                                    // wave-uniform
                const unsigned off = (live && (ld_flags[j] & edge) == 0u) ? base + ld_off[j] : OOB;
                pre[SG][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ld_isx[j] ? xrs : yrs, off, 0, 0));
            }
        }
    };
    auto stash = [&](auto stage_tag, int buf) __attribute__((always_inline)) {        // the fetched step -> LDS buffer `buf`
        constexpr int SG = decltype(stage_tag)::value;
#pragma unroll
        for (int j = 0; j < IPW; ++j)
            if (p + 4 * j < NI) *(f32x4 *)(lds + buf * BUF + ld_lds[j]) = pre[SG][j];
    };
    // Software pipeline over the steps: while step t's 20 MFMAs run from registers (vg[t & 1]), step t + 1's operands are read from LDS and
    // transformed into vg[(t + 1) & 1] -- the transform's ~50 vector instructions and its LDS latency sit in the MFMAs' shadow instead of in
    // front of them (one wave per SIMD: nothing else would fill that time).
    float vv[2][NT][5], gg[2][MT][5];
    auto transform = [&](auto dst_tag, int buf) __attribute__((always_inline)) {
        constexpr int D = decltype(dst_tag)::value;
        const char *bb = lds + buf * BUF;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float w[5];
#pragma unroll
            for (int c = 0; c < 5; ++c) {
                const float xa = *(const float *)(bb + rx_a + (c * BCI + nt * 32) * 4), xb = *(const float *)(bb + rx_b + (c * BCI + nt * 32) * 4);
                w[c] = __builtin_fmaf(sg, xb, xa);
            }
            const float tt = w[3] - w[1];
            vv[D][nt][0] = __builtin_fmaf(2.0f, w[0] - w[2], tt);
            vv[D][nt][1] = __builtin_fmaf(2.0f, w[1], -w[3]) + w[2];
            vv[D][nt][2] = 3.0f * w[2] - __builtin_fmaf(2.0f, w[1], w[3]);
            vv[D][nt][3] = tt;
            vv[D][nt][4] = __builtin_fmaf(-2.0f, tt, w[4] - w[2]);
        }
        const char *ya_p = p == 3 ? lds + ZERO + ry_lane : bb + ry_a, *yb_p = p == 0 ? lds + ZERO + ry_lane : bb + ry_b;
        float a[MT][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if constexpr (MT == 4) {
                const int zo = c * 512;                                // inside the zero page: any offset below 2 KB - 512
                const f32x4 qa = *(const f32x4 *)(ya_p + (p == 3 ? zo : c * BCO * 4)), qb = *(const f32x4 *)(yb_p + (p == 0 ? zo : c * BCO * 4));
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) a[mt][c] = __builtin_fmaf(sg, qb[mt], qa[mt]);
            } else {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const int zo = (c * MT + mt) * 128;
                    const float qa = *(const float *)(ya_p + (p == 3 ? zo : (c * BCO + 32 * mt) * 4)), qb = *(const float *)(yb_p + (p == 0 ? zo : (c * BCO + 32 * mt) * 4));
                    a[mt][c] = __builtin_fmaf(sg, qb, qa);
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const float s02 = a[mt][0] + a[mt][2];
            gg[D][mt][0] = a[mt][0];
            gg[D][mt][1] = s02 + a[mt][1];
            gg[D][mt][2] = s02 - a[mt][1];
            gg[D][mt][3] = __builtin_fmaf(2.0f, a[mt][1], __builtin_fmaf(4.0f, a[mt][2], a[mt][0]));
            gg[D][mt][4] = a[mt][2];
        }
    };
    auto mfmas = [&](auto src_tag) __attribute__((always_inline)) {
        constexpr int S = decltype(src_tag)::value;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
#pragma unroll
            for (int fo = 0; fo < 4; ++fo) {
                const int f = (fo + 3) & 3;                           // the inline-asm MFMA first of its group, with its own wait states (see above)
                const int mt = f / NT, nt = f % NT;
                if (f == 3)
                    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[j][f]) : "v"(gg[S][mt][j]), "v"(vv[S][nt][j]));
                else
                    acc[j][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(gg[S][mt][j], vv[S][nt][j], acc[j][f], 0, 0, 0);
            }
        }
    };

    // One step, hand-interleaved (MT = 4): behind each of the 20 MFMAs of step t (from vg[S]) goes one slice of the transform of step t + 1
    // (LDS buffer `buf` -> vg[D]); an LDS read is issued one slice before its first use.  One wave per SIMD executes in order: a transform
    // in FRONT of the MFMAs costs its full dependent-chain latency (~8 cycles per instruction, nothing to hide it), behind an MFMA it costs
    // ~5 cycles of a pipe that is busy 64.  sched_barrier(0) pins the order.
    auto fused_step = [&](auto src_tag, auto dst_tag, int buf) __attribute__((always_inline)) {
        constexpr int S = decltype(src_tag)::value, D = decltype(dst_tag)::value;
        if constexpr (MT != 4) {
            transform(dst_tag, buf);
            mfmas(src_tag);
        } else {
            const char *bb = lds + buf * BUF;
            const char *ya_p = p == 3 ? lds + ZERO + ry_lane : bb + ry_a, *yb_p = p == 0 ? lds + ZERO + ry_lane : bb + ry_b;
            float xa[5], xb[5], w[5], tt = 0.0f, a[4][3];
            f32x4 qa[3], qb[3];
            auto rdx = [&](int c) __attribute__((always_inline)) {
                xa[c] = *(const float *)(bb + rx_a + c * BCI * 4);
                xb[c] = *(const float *)(bb + rx_b + c * BCI * 4);
            };
            auto rdy = [&](int c) __attribute__((always_inline)) {
                qa[c] = *(const f32x4 *)(ya_p + (p == 3 ? c * 512 : c * BCO * 4));
                qb[c] = *(const f32x4 *)(yb_p + (p == 0 ? c * 512 : c * BCO * 4));
            };
            rdx(0); rdx(1);
            for_seq(std::make_integer_sequence<int, 20>{}, [&](auto i_tag) __attribute__((always_inline)) {
                constexpr int I = decltype(i_tag)::value;
                constexpr int j = I / 4, fo = I % 4;
                constexpr int f = (fo + 3) & 3;                       // the inline-asm MFMA first of its group, with its own wait states
                if (f == 3)
                    asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[j][f]) : "v"(gg[S][f][j]), "v"(vv[S][0][j]));
                else
                    acc[j][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(gg[S][f][j], vv[S][0][j], acc[j][f], 0, 0, 0);
                if constexpr (I == 0) { rdx(2); rdx(3); }
                if constexpr (I == 1) { w[0] = __builtin_fmaf(sg, xb[0], xa[0]); w[1] = __builtin_fmaf(sg, xb[1], xa[1]); rdx(4); }
                if constexpr (I == 2) { w[2] = __builtin_fmaf(sg, xb[2], xa[2]); w[3] = __builtin_fmaf(sg, xb[3], xa[3]); rdy(0); }
                if constexpr (I == 3) { w[4] = __builtin_fmaf(sg, xb[4], xa[4]); tt = w[3] - w[1]; rdy(1); }
                if constexpr (I == 4) { vv[D][0][0] = __builtin_fmaf(2.0f, w[0] - w[2], tt); vv[D][0][3] = tt; rdy(2); }
                if constexpr (I == 5) vv[D][0][1] = __builtin_fmaf(2.0f, w[1], -w[3]) + w[2];
                if constexpr (I == 6) vv[D][0][2] = 3.0f * w[2] - __builtin_fmaf(2.0f, w[1], w[3]);
                if constexpr (I == 7) vv[D][0][4] = __builtin_fmaf(-2.0f, tt, w[4] - w[2]);
                if constexpr (I >= 8 && I <= 10) {
                    constexpr int c = I - 8;
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) a[mt][c] = __builtin_fmaf(sg, qb[c][mt], qa[c][mt]);
                }
                if constexpr (I >= 11 && I <= 14) {
                    constexpr int mt = I - 11;
                    const float s02 = a[mt][0] + a[mt][2];
                    gg[D][mt][0] = a[mt][0];
                    gg[D][mt][1] = s02 + a[mt][1];
                    gg[D][mt][2] = s02 - a[mt][1];
                    gg[D][mt][3] = __builtin_fmaf(2.0f, a[mt][1], __builtin_fmaf(4.0f, a[mt][2], a[mt][0]));
                    gg[D][mt][4] = a[mt][2];
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        }
    };

    // step t's operands: fetched (global -> registers) during step t - 4 / t - 3, stashed into LDS buffer t & 1 during step t - 2,
    // transformed during step t - 1, multiplied in step t
    fetch(std::integral_constant<int, 0>{}, t_lo);
    fetch(std::integral_constant<int, 1>{}, t_lo + 1);
    __syncthreads();                                                  // zero fill done
    stash(std::integral_constant<int, 0>{}, 0);
    stash(std::integral_constant<int, 1>{}, 1);
    fetch(std::integral_constant<int, 0>{}, t_lo + 2);
    fetch(std::integral_constant<int, 1>{}, t_lo + 3);
    __syncthreads();
    transform(std::integral_constant<int, 0>{}, 0);
    __syncthreads();                                                  // buffer 0 is free for step t_lo + 2
    for (int t = t_lo; t < t_hi; t += 2) {
        stash(std::integral_constant<int, 0>{}, 0);                  // step t + 2 -> buffer 0
        fetch(std::integral_constant<int, 0>{}, t + 4);
        __builtin_amdgcn_sched_barrier(0);
        fused_step(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, 1);   // step t's MFMAs from vg[0]; step t + 1 (buffer 1) -> vg[1]
        __syncthreads();
        stash(std::integral_constant<int, 1>{}, 1);                  // step t + 3 -> buffer 1
        fetch(std::integral_constant<int, 1>{}, t + 5);
        __builtin_amdgcn_sched_barrier(0);
        fused_step(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, 0);   // step t + 1 from vg[1]; step t + 2 (buffer 0) -> vg[0]
        __syncthreads();
    }
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[4][3]));
    wgrad_epilogue<MT>(acc, lds, part, tid, p, n, kp, split, cb, nb, C);
}

// dW[co][ci][r][s] = sum over splits, in order, of part[split][3 r + s][co][ci]
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float *__restrict__ part, float *__restrict__ DW, int C, int n_split) {
    const int i = blockIdx.x * 256 + threadIdx.x;                    // channel pair co * C + ci
    if (i >= C * C) return;
    float o[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k] = 0.0f;
    for (int sp = 0; sp < n_split; ++sp)
#pragma unroll
        for (int k = 0; k < 9; ++k) o[k] += part[((size_t)sp * 9 + k) * C * C + i];
#pragma unroll
    for (int k = 0; k < 9; ++k) DW[(size_t)i * 9 + k] = o[k];
}

int wgrad_blocks(int channels) { return channels % 128 == 0 ? (channels / 128) * (channels / 32) : (channels / 64) * (channels / 64); }

#ifndef XQ_WGRAD_LDS
#define XQ_WGRAD_LDS 1                                               // 1: k_wino_wgrad_lds (operands staged through LDS); 0: k_wino_wgrad
#endif

int wgrad_splits(int batch, int channels) {
    const int nblk = wgrad_blocks(channels);
    // K-steps of a launch: two consecutive tiles each (k_wino_wgrad) / one tile position of two consecutive boards each (k_wino_wgrad_lds)
    const int pairs = XQ_WGRAD_LDS ? ((batch + 1) / 2) * 15 : (batch * 15 + 1) / 2;
    int n = 256 / nblk;                                              // one round of the 256 CUs
    if (const char *e = getenv("XQ_WGRAD_SPLITS")) n = atoi(e) > 0 ? atoi(e) : n;   // experiments only
    if (n < 1) n = 1;
    if (n > pairs) n = pairs;
    while ((n * nblk) % 8) ++n;                                      // the XCD mapping wants a multiple of 8 workgroups
    return n;
}

bool bn_args_ok(long long rows, int C) { return rows > 0 && C >= 64 && C <= 1024 && C % 64 == 0 && 256 % (C / 4) == 0; }

}  // namespace

extern "C" {

size_t xq_bn_scratch_bytes(int channels) { return (size_t)2 * NSEG * channels * sizeof(double); }

int xq_bn_train_forward(const float *dev_x, const float *dev_residual, const float *dev_gamma, const float *dev_beta,
                        float *dev_running_mean, float *dev_running_var, float momentum, float eps, long long rows, int channels,
                        int relu, float *dev_y, float *dev_save_mean, float *dev_save_invstd, long long *dev_batches_tracked,
                        void *dev_scratch, void *stream) {
    if (!dev_x || !dev_gamma || !dev_beta || !dev_y || !dev_save_mean || !dev_save_invstd || !dev_scratch) return XQ_ERR_ARG;
    if (!bn_args_ok(rows, channels) || (dev_running_mean == nullptr) != (dev_running_var == nullptr)) return XQ_ERR_ARG;
    if (((uintptr_t)dev_x | (uintptr_t)dev_residual | (uintptr_t)dev_y | (uintptr_t)dev_gamma | (uintptr_t)dev_beta |
         (uintptr_t)dev_save_mean | (uintptr_t)dev_save_invstd | (uintptr_t)dev_scratch) & 15)
        return XQ_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    double *part = (double *)dev_scratch;
    hipLaunchKernelGGL(k_bn_partial<0>, dim3(NSEG), dim3(256), 0, s, dev_x, nullptr, nullptr, nullptr, nullptr, rows, channels, 0, part);
    hipLaunchKernelGGL(k_bn_fwd_finalize, dim3(channels / 4), dim3(256), 0, s, part, rows, channels, momentum, eps, dev_running_mean,
                       dev_running_var, dev_save_mean, dev_save_invstd, dev_batches_tracked);
    const long long quads = rows * channels / 4;
    const int grid = (int)((quads + 255) / 256 < 4096 ? (quads + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_bn_apply, dim3(grid), dim3(256), 0, s, dev_x, dev_residual, dev_gamma, dev_beta, dev_save_mean, dev_save_invstd,
                       quads, channels, relu, dev_y);
    return xq::launch_status();
}

int xq_bn_train_backward(const float *dev_dy, const float *dev_x, const float *dev_y, const float *dev_gamma, const float *dev_save_mean,
                         const float *dev_save_invstd, long long rows, int channels, int relu, float *dev_dx, float *dev_dresidual,
                         float *dev_dgamma, float *dev_dbeta, void *dev_scratch, void *stream) {
    if (!dev_dy || !dev_x || !dev_gamma || !dev_save_mean || !dev_save_invstd || !dev_dx || !dev_dgamma || !dev_dbeta || !dev_scratch)
        return XQ_ERR_ARG;
    if (!bn_args_ok(rows, channels) || (relu && !dev_y)) return XQ_ERR_ARG;
    if (((uintptr_t)dev_dy | (uintptr_t)dev_x | (uintptr_t)dev_y | (uintptr_t)dev_dx | (uintptr_t)dev_dresidual | (uintptr_t)dev_gamma |
         (uintptr_t)dev_save_mean | (uintptr_t)dev_save_invstd | (uintptr_t)dev_dgamma | (uintptr_t)dev_dbeta | (uintptr_t)dev_scratch) & 15)
        return XQ_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    double *part = (double *)dev_scratch;
    hipLaunchKernelGGL(k_bn_partial<1>, dim3(NSEG), dim3(256), 0, s, dev_x, dev_dy, dev_y, dev_save_mean, dev_save_invstd, rows, channels,
                       relu, part);
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3(channels / 4), dim3(256), 0, s, part, channels, dev_dgamma, dev_dbeta);
    const long long quads = rows * channels / 4;
    const int grid = (int)((quads + 255) / 256 < 4096 ? (quads + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_bn_dx, dim3(grid), dim3(256), 0, s, dev_dy, dev_x, dev_y, dev_gamma, dev_save_mean, dev_save_invstd, dev_dgamma,
                       dev_dbeta, quads, rows, channels, relu, dev_dx, dev_dresidual);
    return xq::launch_status();
}

size_t xq_wino_wgrad_scratch_bytes(int batch, int channels) {
    if (batch <= 0 || channels < 64 || channels % 64) return 0;
    return (size_t)wgrad_splits(batch, channels) * 9 * channels * channels * sizeof(float);
}

int xq_wino_wgrad(const float *dev_x, const float *dev_dy, float *dev_dw, void *dev_scratch, int batch, int channels, void *stream) {
    if (!dev_x || !dev_dy || !dev_dw || !dev_scratch || batch <= 0) return XQ_ERR_ARG;
    if (channels < 64 || channels % 64 || 8 % (channels / 64)) return XQ_ERR_ARG;
    if (((uintptr_t)dev_x | (uintptr_t)dev_dy | (uintptr_t)dev_dw | (uintptr_t)dev_scratch) & 15) return XQ_ERR_ARG;
    if (((unsigned long long)batch * 90ull + 64ull) * (unsigned)channels * 4ull >= 0xFFF00000ull) return XQ_ERR_ARG;
    static thread_local bool attr_set = false;
    if (!attr_set) {
        XQ_TRY(hipFuncSetAttribute((const void *)k_wino_wgrad<2>, hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS_BYTES));
        XQ_TRY(hipFuncSetAttribute((const void *)k_wino_wgrad<4>, hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS_BYTES));
        XQ_TRY(hipFuncSetAttribute((const void *)k_wino_wgrad_lds<2>, hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS_BYTES));
        XQ_TRY(hipFuncSetAttribute((const void *)k_wino_wgrad_lds<4>, hipFuncAttributeMaxDynamicSharedMemorySize, WG_LDS_BYTES));
        attr_set = true;
    }
    const int n_split = wgrad_splits(batch, channels);
    const int nblk = wgrad_blocks(channels);
    const dim3 grid(n_split * nblk), block(256);
    hipStream_t st = (hipStream_t)stream;
    float *part = (float *)dev_scratch;
    if (XQ_WGRAD_LDS) {
        if (channels % 128 == 0)
            hipLaunchKernelGGL(k_wino_wgrad_lds<4>, grid, block, WG_LDS_BYTES, st, dev_x, dev_dy, part, batch, channels, n_split);
        else
            hipLaunchKernelGGL(k_wino_wgrad_lds<2>, grid, block, WG_LDS_BYTES, st, dev_x, dev_dy, part, batch, channels, n_split);
    } else {
        if (channels % 128 == 0)
            hipLaunchKernelGGL(k_wino_wgrad<4>, grid, block, WG_LDS_BYTES, st, dev_x, dev_dy, part, batch, channels, n_split);
        else
            hipLaunchKernelGGL(k_wino_wgrad<2>, grid, block, WG_LDS_BYTES, st, dev_x, dev_dy, part, batch, channels, n_split);
    }
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((channels * channels + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const float *)dev_scratch, dev_dw, channels, n_split);
    return xq::launch_status();
}

}  // extern "C"
