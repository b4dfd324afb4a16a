// xq_nn.hip -- evaluator-side kernels (B2).  Round 1: fused epilogue for NHWC convolution outputs.
//
// k_bias_act: y = act(y + bias[c] (+ residual)) in place over a channels-last tensor [rows][C], 16 bytes per lane,
// grid-stride.  HBM-bound streaming kernel: 4*C*rows bytes read (+ residual) and written once; replaces the
// separate bias-add / clamp / residual-add launches of the eager graph (model.py:30-36 folded BatchNorm + ReLU + skip).
#include "xq_common.h"

namespace {

// Training-batch materialisation (train.py:114-151 SelfPlayDataset + augment_data, parallel_selfplay.py:137-151):
// compact 640-byte samples -> dense float32 planes [15][10][9], dense float32 pi[8100] (visits^(1/T) normalised in
// float64, then cast, as torch.FloatTensor(policy) does) and z; `flip` mirrors the columns of the board and of both
// squares of every action.  One wavefront per output sample; HBM-bound (37.8 KB written per sample).
__global__ __launch_bounds__(64) void k_samples_to_batch(const xq_sample *__restrict__ rec, const int32_t *__restrict__ idx,
                                                         const uint8_t *__restrict__ flip, int n, double late_temperature,
                                                         float *__restrict__ states, float *__restrict__ pi,
                                                         float *__restrict__ z) {
    const int o = blockIdx.x;
    if (o >= n) return;
    const int lane = threadIdx.x;
    const xq_sample *s = rec + idx[o];
    const bool fl = flip[o] != 0;
    const int side = s->side;
    float *st = states + (size_t)o * XQ_STATE_FLOATS;
    for (int e = lane; e < XQ_STATE_FLOATS; e += 64) {
        const int plane = e / 90, sq = e - plane * 90;
        float v;
        if (plane == 14) {
            v = side == 1 ? 1.0f : 0.0f;
        } else {
            const int r = sq / 9, c = sq - r * 9;
            const int p = s->board[fl ? r * 9 + (8 - c) : sq];
            const int want = (plane < 7 ? plane + 1 : plane - 6) * (plane < 7 ? side : -side);
            v = p == want ? 1.0f : 0.0f;
        }
        st[e] = v;
    }
    float4 *p4 = (float4 *)(pi + (size_t)o * XQ_ACTION_SPACE);
    for (int i = lane; i < XQ_ACTION_SPACE / 4; i += 64) p4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();                                  // one wave per block: orders the zero fill before the scatter
    const int m = s->n_moves;
    const double inv_t = s->late_temp ? 1.0 / late_temperature : 1.0;
    double total = 0.0;
    for (int i = 0; i < m; ++i) {
        const double c = (double)s->visits[i];
        total += s->late_temp ? (c > 0.0 ? pow(c, inv_t) : 0.0) : c;
    }
    for (int i = lane; i < m; i += 64) {
        const double c = (double)s->visits[i];
        const double w = s->late_temp ? (c > 0.0 ? pow(c, inv_t) : 0.0) : c;
        int a = s->actions[i];
        if (fl) {
            const int from = a / 90, to = a - from * 90;
            const int fr = from / 9, fc = from - fr * 9, tr = to / 9, tc = to - tr * 9;
            a = (fr * 9 + (8 - fc)) * 90 + tr * 9 + (8 - tc);
        }
        pi[(size_t)o * XQ_ACTION_SPACE + a] = total > 0.0 ? (float)(w / total) : 0.0f;
    }
    if (lane == 0) z[o] = (float)s->z;
}

__global__ __launch_bounds__(256) void k_bias_act(float4 *__restrict__ y, const float4 *__restrict__ bias,
                                                  const float4 *__restrict__ res, long long n4, int c4, int relu) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = y[i];
        const float4 b = bias[(int)(i % c4)];
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        if (res) {
            const float4 r = res[i];
            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
        }
        if (relu) {
            v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f);
        }
        y[i] = v;
    }
}


// Both heads' 1x1 convolutions in ONE pass over the tower output (model.py:43-62: policy Conv2d(C,32,1) and value
// Conv2d(C,4,1), BatchNorm folded, ReLU): out[r][o] = relu(b[o] + sum_c h[r][c] W[o][c]) for the 36 output channels.
// HBM-bound (reads 4 C bytes per position row once instead of twice, writes 144); the library path was two GEMMs +
// two epilogue launches.  A wave takes 16 rows at a time: lane (r = lane >> 2, q = lane & 3) owns channels
// {16 k + 4 q + j} of row r (four lanes read 64 contiguous bytes), multiplies them with the weights broadcast from LDS,
// and the four lanes of a row are summed with two butterfly steps.
constexpr int HEAD_OUT = 36;
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_heads_1x1(const float *__restrict__ H, const float *__restrict__ W,
                                                   const float *__restrict__ bias, float *__restrict__ P,
                                                   float *__restrict__ V, long long rows, int C) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Ws = (float *)smem;                         // [36][C]
    for (int i = threadIdx.x * 4; i < HEAD_OUT * C; i += 256 * 4) *(float4 *)(Ws + i) = *(const float4 *)(W + i);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane >> 2, q = lane & 3;
    const int nk = C / 16;
    const long long n_batches = (rows + 15) / 16;
    for (long long bt = (long long)blockIdx.x * 4 + wave; bt < n_batches; bt += (long long)gridDim.x * 4) {
        const long long row = bt * 16 + r;
        const bool ok = row < rows;
        const float *hp = H + (ok ? row : 0) * C + 4 * q;
        f32x2 acc2[HEAD_OUT];                           // even / odd channel partial sums: packed FMAs
#pragma unroll
        for (int o = 0; o < HEAD_OUT; ++o) acc2[o] = f32x2{0.0f, 0.0f};
        float4 hv = *(const float4 *)hp;
        for (int k = 0; k < nk; ++k) {
            const f32x2 c01 = {hv.x, hv.y}, c23 = {hv.z, hv.w};
            if (k + 1 < nk) hv = *(const float4 *)(hp + 16 * (k + 1));
            const float *wk = Ws + 16 * k + 4 * q;
#pragma unroll
            for (int o = 0; o < HEAD_OUT; ++o) {
                const float4 w = *(const float4 *)(wk + o * C);
                acc2[o] = __builtin_elementwise_fma(c23, f32x2{w.z, w.w}, __builtin_elementwise_fma(c01, f32x2{w.x, w.y}, acc2[o]));
            }
        }
        float acc[HEAD_OUT];
#pragma unroll
        for (int o = 0; o < HEAD_OUT; ++o) {
            float v = acc2[o].x + acc2[o].y;
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            acc[o] = fmaxf(v + bias[o], 0.0f);
        }
        if (ok) {                                      // lane q of a row stores outputs 9q .. 9q+8
#pragma unroll
            for (int o = 0; o < HEAD_OUT; ++o) {
                if (o / 9 == q) {
                    if (o < 32) P[row * 32 + o] = acc[o];
                    else V[row * 4 + (o - 32)] = acc[o];
                }
            }
        }
    }
}


// Input convolution (model.py:87-93: Conv2d(15, C, 3, padding=1), BatchNorm folded, ReLU) over the encoder's planes
// float[G][15][90], output NHWC float[G][90][C].  The planes are one-hot piece maps plus a constant side plane, so of the
// 135 (plane, tap) products of a position about a dozen are non-zero: a wave takes one position, its lanes test the 135
// inputs (three ballots), and a scalar loop over the set bits accumulates x * w[entry][4 channels per lane] -- exact for
// ANY input (zeros contribute nothing), ~10x fewer multiply-adds than the dense form, and the kernel is bound by the
// 4 C bytes it writes per position.  Weights: float[135][C], entry = plane * 9 + (dy + 1) * 3 + (dx + 1).
__global__ __launch_bounds__(256) void k_stem_conv(const float *__restrict__ X, const float *__restrict__ Wt,
                                                   const float *__restrict__ bias, float *__restrict__ Y, int C) {
    __shared__ float xs[15 * 90];
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 15 * 90; i += 256) xs[i] = X[(size_t)g * (15 * 90) + i];
    __syncthreads();
    for (int pos = wave; pos < 90; pos += 4) {
        const int y = pos / 9, x = pos - y * 9;
        float xv[3];
        unsigned long long m[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int e = r * 64 + lane;
            float v = 0.0f;
            if (e < 135) {
                const int plane = e / 9, t = e - plane * 9, ty = t / 3, yy = y + ty - 1, xx = x + (t - ty * 3) - 1;
                if ((unsigned)yy < 10u && (unsigned)xx < 9u) v = xs[plane * 90 + yy * 9 + xx];
            }
            xv[r] = v;
            m[r] = __ballot(v != 0.0f);
        }
        for (int c0 = lane * 4; c0 < C; c0 += 256) {
            float4 acc = *(const float4 *)(bias + c0);
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                unsigned long long mm = m[r];                         // wave-uniform: a scalar loop over the set bits
                while (mm) {
                    const int b = __builtin_ctzll(mm);
                    mm &= mm - 1;
                    const float xe = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, xv[r]), b));
                    const float4 w = *(const float4 *)(Wt + (size_t)(r * 64 + b) * C + c0);
                    acc.x = fmaf(xe, w.x, acc.x); acc.y = fmaf(xe, w.y, acc.y); acc.z = fmaf(xe, w.z, acc.z); acc.w = fmaf(xe, w.w, acc.w);
                }
            }
            acc.x = fmaxf(acc.x, 0.0f); acc.y = fmaxf(acc.y, 0.0f); acc.z = fmaxf(acc.z, 0.0f); acc.w = fmaxf(acc.w, 0.0f);
            *(float4 *)(Y + ((size_t)g * 90 + pos) * C + c0) = acc;
        }
    }
}


// Policy head's fully connected layer ONLY where the search needs it (model.py:64-71 Linear(2880, 8100) restricted to
// the ordered legal moves of each pending evaluation, mcts.py:176-188): out[g][m] = b[a] + <feat[g], W[a]>, a = moves[g][m],
// m < counts[g].  The reference computes all 8 100 logits per position (23.3 M multiply-adds) and keeps ~40; here a game
// costs ~40 rows x 2 880 -- the dense [G, 8100] logits row (32 KB per position) never exists.  One wavefront per game:
// the position's 2 880 features stay in registers (12 float4 per lane), each legal move streams its 11.5 KB weight row
// (coalesced 1 KB per load instruction, served by L2 / Infinity Cache: the rows of the ~2 000 actions that occur are
// re-read by many games), two rows in flight.  Bound by L2/MALL row traffic (games x legal x 11 520 B), not by FLOPs.
constexpr int PF4 = 2880 / 4;             // float4 per feature / weight row
__global__ __launch_bounds__(256) void k_policy_legal(const float *__restrict__ feat, const float *__restrict__ W,
                                                      const float *__restrict__ bias, const uint16_t *__restrict__ moves,
                                                      const int32_t *__restrict__ counts, int games, float *__restrict__ out) {
    const int g = blockIdx.x * 4 + (int)(threadIdx.x >> 6);
    if (g >= games) return;
    const int lane = threadIdx.x & 63;
    int cnt = counts[g];
    cnt = cnt < 0 ? 0 : (cnt > XQ_MAXM ? XQ_MAXM : cnt);
    cnt = __builtin_amdgcn_readfirstlane(cnt);
    if (cnt == 0) return;
    const float4 *f4 = (const float4 *)(feat + (size_t)g * 2880);
    float4 f[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) f[i] = (lane + 64 * i < PF4) ? f4[lane + 64 * i] : make_float4(0.f, 0.f, 0.f, 0.f);
    const uint16_t *mv = moves + (size_t)g * XQ_MAXM;
    float r0 = 0.0f, r1 = 0.0f;                        // results of moves lane and 64 + lane
    auto row_dot = [&](int a) __attribute__((always_inline)) {
        const float4 *w4 = (const float4 *)(W + (size_t)a * 2880);
        float4 w[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) w[i] = (lane + 64 * i < PF4) ? w4[lane + 64 * i] : make_float4(0.f, 0.f, 0.f, 0.f);
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            s = fmaf(f[i].x, w[i].x, s); s = fmaf(f[i].y, w[i].y, s); s = fmaf(f[i].z, w[i].z, s); s = fmaf(f[i].w, w[i].w, s);
        }
        return s;
    };
    for (int m = 0; m < cnt; m += 2) {
        const int a0 = __builtin_amdgcn_readfirstlane((int)mv[m]);
        const int a1 = __builtin_amdgcn_readfirstlane((int)mv[m + 1 < cnt ? m + 1 : m]);
        float s0 = row_dot(a0), s1 = row_dot(a1);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_xor(s0, off); s1 += __shfl_xor(s1, off); }
        s0 += bias[a0]; s1 += bias[a1];
        if (m < 64) { if (lane == m) r0 = s0; if (lane == m + 1) r0 = s1; }
        else        { if (lane == m - 64) r1 = s0; if (lane == m - 63) r1 = s1; }
    }
    float *o = out + (size_t)g * XQ_MAXM;
    if (lane < cnt) o[lane] = r0;
    if (64 + lane < cnt) o[64 + lane] = r1;
}

// Value head's fully connected layers (model.py:73-85: Linear(360, 128) + ReLU + Linear(128, 1) + tanh) over the value
// features float[G][90][4] that k_heads_1x1 wrote.  VGB games per 128-thread block: the games' features sit in LDS,
// thread j owns hidden unit j for the block's games and streams its weight column (w1t[k][j]: coalesced, L2-resident 184 KB),
// the 128 -> 1 layer is a wave reduction.  A 0.75 GFLOP problem at G = 8192: microseconds; it exists so that the
// evaluator makes no library call.
constexpr int VGB = 4;        // games per block: 4 keeps >= 256 blocks in flight from G = 1024 up (16: 47 us at G = 1024)
__global__ __launch_bounds__(128) void k_value_head(const float *__restrict__ vf, const float *__restrict__ w1t,
                                                    const float *__restrict__ b1, const float *__restrict__ w2,
                                                    const float *__restrict__ b2, int games, float *__restrict__ value) {
    __shared__ __attribute__((aligned(16))) float vs[VGB][360];
    __shared__ float part[2][VGB];
    const int g0 = blockIdx.x * VGB, j = threadIdx.x;
    for (int i = j; i < VGB * 360; i += 128) {
        const int g = i / 360;
        vs[g][i - g * 360] = g0 + g < games ? vf[(size_t)(g0 + g) * 360 + (i - g * 360)] : 0.0f;
    }
    __syncthreads();
    float acc[VGB];
#pragma unroll
    for (int g = 0; g < VGB; ++g) acc[g] = 0.0f;
    for (int k = 0; k < 360; k += 4) {
        const float wa = w1t[(k + 0) * 128 + j], wb = w1t[(k + 1) * 128 + j], wc = w1t[(k + 2) * 128 + j], wd = w1t[(k + 3) * 128 + j];
#pragma unroll
        for (int g = 0; g < VGB; ++g) {
            const float4 v = *(const float4 *)&vs[g][k];
            acc[g] = fmaf(v.w, wd, fmaf(v.z, wc, fmaf(v.y, wb, fmaf(v.x, wa, acc[g]))));
        }
    }
    const float bj = b1[j], wj = w2[j];
#pragma unroll
    for (int g = 0; g < VGB; ++g) {
        float h = fmaxf(acc[g] + bj, 0.0f) * wj;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) h += __shfl_xor(h, off);
        if ((j & 63) == 0) part[j >> 6][g] = h;
    }
    __syncthreads();
    if (j < VGB && g0 + j < games) value[g0 + j] = tanhf(part[0][j] + part[1][j] + b2[0]);
}

}  // namespace

extern "C" int xq_bias_act(float *dev_y, const float *dev_bias, const float *dev_residual, long long rows, int channels,
                           int relu, void *stream) {
    if (!dev_y || !dev_bias || rows < 0 || channels <= 0 || (channels & 3)) return XQ_ERR_ARG;
    if (((uintptr_t)dev_y | (uintptr_t)dev_bias | (uintptr_t)dev_residual) & 15) return XQ_ERR_ARG;
    const long long n4 = rows * (long long)(channels / 4);
    if (n4 == 0) return XQ_OK;
    long long blocks = (n4 + 255) / 256;
    if (blocks > 2048 * 4) blocks = 2048 * 4;
    hipLaunchKernelGGL(k_bias_act, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float4 *)dev_y,
                       (const float4 *)dev_bias, (const float4 *)dev_residual, n4, channels / 4, relu);
    return xq::launch_status();
}

extern "C" int xq_samples_to_batch(const void *dev_samples, const int32_t *dev_index, const uint8_t *dev_flip, int n,
                                   double late_temperature, float *dev_states, float *dev_pi, float *dev_z, void *stream) {
    if (n < 0 || (n > 0 && (!dev_samples || !dev_index || !dev_flip || !dev_states || !dev_pi || !dev_z))) return XQ_ERR_ARG;
    if (late_temperature <= 0.0 || ((uintptr_t)dev_pi & 15)) return XQ_ERR_ARG;
    if (n == 0) return XQ_OK;
    hipLaunchKernelGGL(k_samples_to_batch, dim3(n), dim3(64), 0, (hipStream_t)stream, (const xq_sample *)dev_samples, dev_index,
                       dev_flip, n, late_temperature, dev_states, dev_pi, dev_z);
    return xq::launch_status();
}

extern "C" int xq_heads_1x1(const float *dev_h, const float *dev_w, const float *dev_bias, float *dev_p, float *dev_v,
                            long long rows, int channels, void *stream) {
    if (!dev_h || !dev_w || !dev_bias || !dev_p || !dev_v || rows < 0 || channels < 16 || channels % 16 || channels > 1024) return XQ_ERR_ARG;
    if (((uintptr_t)dev_h | (uintptr_t)dev_w) & 15) return XQ_ERR_ARG;
    if (rows == 0) return XQ_OK;
    const int lds = HEAD_OUT * channels * 4;
    static thread_local int attr_lds = 0;
    if (lds > attr_lds) {
        XQ_TRY(hipFuncSetAttribute((const void *)k_heads_1x1, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_lds = lds;
    }
    long long blocks = ((rows + 15) / 16 + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(k_heads_1x1, dim3((unsigned)blocks), dim3(256), lds, (hipStream_t)stream, dev_h, dev_w, dev_bias, dev_p,
                       dev_v, rows, channels);
    return xq::launch_status();
}

extern "C" int xq_stem_conv(const float *dev_planes, const float *dev_wt, const float *dev_bias, float *dev_y, int games,
                            int channels, void *stream) {
    if (!dev_planes || !dev_wt || !dev_bias || !dev_y || games < 0 || channels < 4 || channels % 4) return XQ_ERR_ARG;
    if (((uintptr_t)dev_wt | (uintptr_t)dev_bias | (uintptr_t)dev_y) & 15) return XQ_ERR_ARG;
    if (games == 0) return XQ_OK;
    hipLaunchKernelGGL(k_stem_conv, dim3(games), dim3(256), 0, (hipStream_t)stream, dev_planes, dev_wt, dev_bias, dev_y, channels);
    return xq::launch_status();
}

extern "C" int xq_policy_head_legal(const float *dev_feat, const float *dev_w, const float *dev_bias, const uint16_t *dev_moves,
                                    const int32_t *dev_counts, int games, float *dev_out, void *stream) {
    if (games < 0 || (games > 0 && (!dev_feat || !dev_w || !dev_bias || !dev_moves || !dev_counts || !dev_out))) return XQ_ERR_ARG;
    if (((uintptr_t)dev_feat | (uintptr_t)dev_w) & 15) return XQ_ERR_ARG;
    if (games == 0) return XQ_OK;
    hipLaunchKernelGGL(k_policy_legal, dim3((games + 3) / 4), dim3(256), 0, (hipStream_t)stream, dev_feat, dev_w, dev_bias, dev_moves,
                       dev_counts, games, dev_out);
    return xq::launch_status();
}

extern "C" int xq_value_head(const float *dev_vfeat, const float *dev_w1t, const float *dev_b1, const float *dev_w2,
                             const float *dev_b2, int games, float *dev_value, void *stream) {
    if (games < 0 || (games > 0 && (!dev_vfeat || !dev_w1t || !dev_b1 || !dev_w2 || !dev_b2 || !dev_value))) return XQ_ERR_ARG;
    if (games == 0) return XQ_OK;
    hipLaunchKernelGGL(k_value_head, dim3((games + VGB - 1) / VGB), dim3(128), 0, (hipStream_t)stream, dev_vfeat, dev_w1t, dev_b1,
                       dev_w2, dev_b2, games, dev_value);
    return xq::launch_status();
}
