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
