// xq_nn.hip -- evaluator-side kernels (B2).  Round 1: fused epilogue for NHWC convolution outputs.
//
// k_bias_act: y = act(y + bias[c] (+ residual)) in place over a channels-last tensor [rows][C], 16 bytes per lane,
// grid-stride.  HBM-bound streaming kernel: 4*C*rows bytes read (+ residual) and written once; replaces the
// separate bias-add / clamp / residual-add launches of the eager graph (model.py:30-36 folded BatchNorm + ReLU + skip).
#include "xq_common.h"

namespace {

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
