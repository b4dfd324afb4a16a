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
// sums are float64 and reduced in a fixed order by a one-workgroup finalize kernel -- deterministic, no atomics.  Algorithmic bytes:
// forward 3 passes (+1 with a residual), backward 7 (+1), of rows * C * 4 bytes.
#include "xq_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NSEG = 256;                         // row segments = workgroups of the reduction kernels = one round of the 256 CUs

struct D4 {
    double v[4];
};

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

// forward finalize: one thread per channel sums the NSEG partials in order
__global__ void k_bn_fwd_finalize(const double *__restrict__ part, long long rows, int C, float momentum, float eps,
                                  float *__restrict__ run_mean, float *__restrict__ run_var, float *__restrict__ save_mean,
                                  float *__restrict__ save_invstd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int j = 0; j < NSEG; ++j) {
        s += part[(size_t)j * C + c];
        q += part[(size_t)(NSEG + j) * C + c];
    }
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

__global__ void k_bn_bwd_finalize(const double *__restrict__ part, int C, float *__restrict__ dgamma, float *__restrict__ dbeta) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int j = 0; j < NSEG; ++j) {
        s += part[(size_t)j * C + c];
        q += part[(size_t)(NSEG + j) * C + c];
    }
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

bool bn_args_ok(long long rows, int C) { return rows > 0 && C >= 64 && C <= 1024 && C % 64 == 0 && 256 % (C / 4) == 0; }

}  // namespace

extern "C" {

size_t xq_bn_scratch_bytes(int channels) { return (size_t)2 * NSEG * channels * sizeof(double); }

int xq_bn_train_forward(const float *dev_x, const float *dev_residual, const float *dev_gamma, const float *dev_beta,
                        float *dev_running_mean, float *dev_running_var, float momentum, float eps, long long rows, int channels,
                        int relu, float *dev_y, float *dev_save_mean, float *dev_save_invstd, void *dev_scratch, void *stream) {
    if (!dev_x || !dev_gamma || !dev_beta || !dev_y || !dev_save_mean || !dev_save_invstd || !dev_scratch) return XQ_ERR_ARG;
    if (!bn_args_ok(rows, channels) || (dev_running_mean == nullptr) != (dev_running_var == nullptr)) return XQ_ERR_ARG;
    if (((uintptr_t)dev_x | (uintptr_t)dev_residual | (uintptr_t)dev_y | (uintptr_t)dev_gamma | (uintptr_t)dev_beta |
         (uintptr_t)dev_save_mean | (uintptr_t)dev_save_invstd | (uintptr_t)dev_scratch) & 15)
        return XQ_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    double *part = (double *)dev_scratch;
    hipLaunchKernelGGL(k_bn_partial<0>, dim3(NSEG), dim3(256), 0, s, dev_x, nullptr, nullptr, nullptr, nullptr, rows, channels, 0, part);
    hipLaunchKernelGGL(k_bn_fwd_finalize, dim3((channels + 63) / 64), dim3(64), 0, s, part, rows, channels, momentum, eps, dev_running_mean,
                       dev_running_var, dev_save_mean, dev_save_invstd);
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
    hipLaunchKernelGGL(k_bn_bwd_finalize, dim3((channels + 63) / 64), dim3(64), 0, s, part, channels, dev_dgamma, dev_dbeta);
    const long long quads = rows * channels / 4;
    const int grid = (int)((quads + 255) / 256 < 4096 ? (quads + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_bn_dx, dim3(grid), dim3(256), 0, s, dev_dy, dev_x, dev_y, dev_gamma, dev_save_mean, dev_save_invstd, dev_dgamma,
                       dev_dbeta, quads, rows, channels, relu, dev_dx, dev_dresidual);
    return xq::launch_status();
}

}  // extern "C"
