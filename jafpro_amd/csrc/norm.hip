// Normalisation layers of the stage-4 networks.
//  * CRN LayerNorm (src/crn_model.py:67-87): per-sample statistics over C*H*W, Bessel-corrected
//    std, eps added to the std, per-channel affine, followed by LeakyReLU(0.01) (:100).
//  * BatchNorm2d in training mode (propagater and discriminators), SURVEY F9.
// Reductions run in fp64 (sum and sum of squares of up to 16.7 M fp32 values per sample).
#include "jaf_common.h"

// ------------------------------------------------------------------ LayerNorm
__global__ void ln_partial_kernel(const float* x, long chw, double* ws) {
    const int n = blockIdx.y;
    const float* p = x + (long)n * chw;
    double s = 0.0, ss = 0.0;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < chw; i += stride) {
        const double v = (double)p[i];
        s += v;
        ss += v * v;
    }
    __shared__ double rs[4], rss[4];
    s = jaf_wave_sum(s);
    ss = jaf_wave_sum(ss);
    if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = s; rss[threadIdx.x >> 6] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&ws[2 * n], rs[0] + rs[1] + rs[2] + rs[3]);
        atomicAdd(&ws[2 * n + 1], rss[0] + rss[1] + rss[2] + rss[3]);
    }
}

__global__ void ln_finalize_kernel(const double* ws, int N, long chw, float eps, float* stats) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const double cnt = (double)chw;
    const double mean = ws[2 * n] / cnt;
    double var = (ws[2 * n + 1] - ws[2 * n] * mean) / (cnt > 1.0 ? cnt - 1.0 : 1.0);
    if (var < 0.0) var = 0.0;
    const float stdv = (float)sqrt(var);
    stats[2 * n] = (float)mean;
    stats[2 * n + 1] = 1.0f / (stdv + eps);
}

extern "C" int jaf_layernorm_stats(jaf_stream_t s_, const float* x, int32_t N, int64_t chw, float eps,
                                   double* workspace, float* stats) {
    JAF_REQUIRE(x && workspace && stats && N >= 1 && chw >= 1);
    hipStream_t s = (hipStream_t)s_;
    hipError_t e = hipMemsetAsync(workspace, 0, sizeof(double) * 2 * N, s);
    if (e != hipSuccess) return (int)e;
    int gx = jaf_ew_grid(chw, 8);
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(ln_partial_kernel, dim3(gx, N), dim3(256), 0, s, x, (long)chw, workspace);
    hipLaunchKernelGGL(ln_finalize_kernel, dim3(jaf_cdiv(N, 64)), dim3(64), 0, s, workspace, N, (long)chw, eps, stats);
    return jaf_launch_status();
}

__global__ void ln_lrelu_fwd_kernel(const float* x, const float* stats, const float* gamma, const float* beta,
                                    float* y, int N, int C, int HW, float slope) {
    const long total = (long)N * C * HW;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long nc = e / HW;
        const int c = (int)(nc % C);
        const int n = (int)(nc / C);
        const float xh = (x[e] - stats[2 * n]) * stats[2 * n + 1];
        const float z = xh * gamma[c] + beta[c];
        y[e] = z > 0.f ? z : z * slope;
    }
}

extern "C" int jaf_layernorm_lrelu_fwd(jaf_stream_t s, const float* x, const float* stats, const float* gamma,
                                       const float* beta, float* y, int32_t N, int32_t C, int32_t HW, float slope) {
    JAF_REQUIRE(x && stats && gamma && beta && y && N >= 1 && C >= 1 && HW >= 1);
    hipLaunchKernelGGL(ln_lrelu_fwd_kernel, dim3(jaf_ew_grid((long)N * C * HW)), dim3(256), 0, (hipStream_t)s,
                       x, stats, gamma, beta, y, N, C, HW, slope);
    return jaf_launch_status();
}

// pass A: per (c, n) block: a = sum dz, b = sum dz*xhat ; dbeta[c] += a, dgamma[c] += b,
// ws[2n] += gamma_c*a (S1), ws[2n+1] += gamma_c*b (S2).
__global__ void ln_bwd_reduce_kernel(const float* dy, const float* x, const float* stats, const float* gamma,
                                     const float* beta, float* dgamma, float* dbeta, double* ws, int C, int HW,
                                     float slope) {
    const int c = blockIdx.x;
    const int n = blockIdx.y;
    const long base = ((long)n * C + c) * HW;
    const float mean = stats[2 * n], r = stats[2 * n + 1];
    const float g = gamma[c], b = beta[c];
    double sa = 0.0, sb = 0.0;
    for (int i = threadIdx.x; i < HW; i += blockDim.x) {
        const float xh = (x[base + i] - mean) * r;
        const float z = xh * g + b;
        const float dz = dy[base + i] * (z > 0.f ? 1.f : slope);
        sa += (double)dz;
        sb += (double)dz * (double)xh;
    }
    __shared__ double ra[4], rb[4];
    sa = jaf_wave_sum(sa);
    sb = jaf_wave_sum(sb);
    if ((threadIdx.x & 63) == 0) { ra[threadIdx.x >> 6] = sa; rb[threadIdx.x >> 6] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double a = ra[0] + ra[1] + ra[2] + ra[3];
        const double bb = rb[0] + rb[1] + rb[2] + rb[3];
        atomicAdd(&dbeta[c], (float)a);
        atomicAdd(&dgamma[c], (float)bb);
        atomicAdd(&ws[2 * n], (double)g * a);
        atomicAdd(&ws[2 * n + 1], (double)g * bb);
    }
}

__global__ void ln_bwd_apply_kernel(const float* dy, const float* x, const float* stats, const float* gamma,
                                    const float* beta, const double* ws, float* dx, int N, int C, int HW,
                                    float slope, float eps) {
    const long total = (long)N * C * HW;
    const long stride = (long)gridDim.x * blockDim.x;
    const double M = (double)C * (double)HW;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const long nc = e / HW;
        const int c = (int)(nc % C);
        const int n = (int)(nc / C);
        const float mean = stats[2 * n], r = stats[2 * n + 1];
        const float sigma = 1.0f / r - eps;
        const float m1 = (float)(ws[2 * n] / M);
        // S2 / ((M-1) * sigma * r)
        const float k = (sigma > 0.f) ? (float)(ws[2 * n + 1] / ((M - 1.0) * (double)sigma * (double)r)) : 0.f;
        const float xh = (x[e] - mean) * r;
        const float z = xh * gamma[c] + beta[c];
        const float dxh = dy[e] * (z > 0.f ? 1.f : slope) * gamma[c];
        dx[e] = r * (dxh - m1 - xh * k);
    }
}

extern "C" int jaf_layernorm_lrelu_bwd(jaf_stream_t s_, const float* dy, const float* x, const float* stats,
                                       const float* gamma, const float* beta, float* dx, float* dgamma,
                                       float* dbeta, double* workspace, int32_t N, int32_t C, int32_t HW,
                                       float slope, float eps) {
    JAF_REQUIRE(dy && x && stats && gamma && beta && dx && dgamma && dbeta && workspace);
    JAF_REQUIRE(N >= 1 && C >= 1 && HW >= 1 && N <= 65535);
    hipStream_t s = (hipStream_t)s_;
    hipError_t e = hipMemsetAsync(workspace, 0, sizeof(double) * 2 * N, s);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(ln_bwd_reduce_kernel, dim3(C, N), dim3(256), 0, s, dy, x, stats, gamma, beta, dgamma, dbeta,
                       workspace, C, HW, slope);
    hipLaunchKernelGGL(ln_bwd_apply_kernel, dim3(jaf_ew_grid((long)N * C * HW)), dim3(256), 0, s, dy, x, stats, gamma,
                       beta, workspace, dx, N, C, HW, slope, eps);
    return jaf_launch_status();
}

// ------------------------------------------------------------------ BatchNorm2d
__global__ void bn_stats_kernel(const float* x, int N, int C, int HW, float eps, float momentum,
                                float* running_mean, float* running_var, float* stats, int training) {
    const int c = blockIdx.x;
    if (!training) {
        if (threadIdx.x == 0) {
            stats[c] = running_mean[c];
            stats[C + c] = 1.0f / sqrtf(running_var[c] + eps);
        }
        return;
    }
    double s = 0.0, ss = 0.0;
    for (int n = 0; n < N; ++n) {
        const float* p = x + ((long)n * C + c) * HW;
        for (int i = threadIdx.x; i < HW; i += blockDim.x) {
            const double v = (double)p[i];
            s += v;
            ss += v * v;
        }
    }
    __shared__ double rs[4], rss[4];
    s = jaf_wave_sum(s);
    ss = jaf_wave_sum(ss);
    if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = s; rss[threadIdx.x >> 6] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double cnt = (double)N * (double)HW;
        const double S = rs[0] + rs[1] + rs[2] + rs[3];
        const double SS = rss[0] + rss[1] + rss[2] + rss[3];
        const double mean = S / cnt;
        double var = SS / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[c] = (float)mean;
        stats[C + c] = (float)(1.0 / sqrt(var + (double)eps));
        if (running_mean) {
            const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
}

extern "C" int jaf_batchnorm_stats(jaf_stream_t s, const float* x, int32_t N, int32_t C, int32_t HW, float eps,
                                   float momentum, float* running_mean, float* running_var, float* stats,
                                   int training) {
    JAF_REQUIRE(x && stats && N >= 1 && C >= 1 && HW >= 1);
    JAF_REQUIRE(training || (running_mean && running_var));
    hipLaunchKernelGGL(bn_stats_kernel, dim3(C), dim3(256), 0, (hipStream_t)s, x, N, C, HW, eps, momentum, running_mean,
                       running_var, stats, training);
    return jaf_launch_status();
}

__global__ void bn_act_fwd_kernel(const float* x, const float* stats, const float* w, const float* b,
                                  const float* residual, float* y, int N, int C, int HW, int act, float slope) {
    const long total = (long)N * C * HW;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int c = (int)((e / HW) % C);
        float v = (x[e] - stats[c]) * stats[C + c] * w[c] + b[c];
        v = jaf_act(v, act, slope);
        if (residual) v += residual[e];
        y[e] = v;
    }
}

extern "C" int jaf_batchnorm_act_fwd(jaf_stream_t s, const float* x, const float* stats, const float* weight,
                                     const float* bias, const float* residual, float* y, int32_t N, int32_t C,
                                     int32_t HW, int act, float slope) {
    JAF_REQUIRE(x && stats && weight && bias && y && N >= 1 && C >= 1 && HW >= 1);
    JAF_REQUIRE(!residual || act == JAF_ACT_NONE);
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(jaf_ew_grid((long)N * C * HW)), dim3(256), 0, (hipStream_t)s, x, stats,
                       weight, bias, residual, y, N, C, HW, act, slope);
    return jaf_launch_status();
}

__device__ __forceinline__ float bn_dz(float dy, float y, int act, float slope) {
    switch (act) {
        case JAF_ACT_LRELU: return dy * (y > 0.f ? 1.f : slope);
        case JAF_ACT_RELU: return dy * (y > 0.f ? 1.f : 0.f);
        case JAF_ACT_SIGMOID: return dy * y * (1.f - y);
        default: return dy;
    }
}

__global__ void bn_bwd_reduce_kernel(const float* dy, const float* x, const float* y, const float* stats,
                                     float* dweight, float* dbias, int N, int C, int HW, int act, float slope) {
    const int c = blockIdx.x;
    const float mean = stats[c], r = stats[C + c];
    double sa = 0.0, sb = 0.0;
    for (int n = 0; n < N; ++n) {
        const long base = ((long)n * C + c) * HW;
        for (int i = threadIdx.x; i < HW; i += blockDim.x) {
            const float dz = bn_dz(dy[base + i], y[base + i], act, slope);
            sa += (double)dz;
            sb += (double)dz * (double)((x[base + i] - mean) * r);
        }
    }
    __shared__ double ra[4], rb[4];
    sa = jaf_wave_sum(sa);
    sb = jaf_wave_sum(sb);
    if ((threadIdx.x & 63) == 0) { ra[threadIdx.x >> 6] = sa; rb[threadIdx.x >> 6] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        dbias[c] = (float)(ra[0] + ra[1] + ra[2] + ra[3]);
        dweight[c] = (float)(rb[0] + rb[1] + rb[2] + rb[3]);
    }
}

__global__ void bn_bwd_apply_kernel(const float* dy, const float* x, const float* y, const float* stats,
                                    const float* w, const float* dweight, const float* dbias, float* dx, int N,
                                    int C, int HW, int act, float slope, int training) {
    const long total = (long)N * C * HW;
    const long stride = (long)gridDim.x * blockDim.x;
    const float inv_cnt = 1.0f / ((float)N * (float)HW);
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int c = (int)((e / HW) % C);
        const float r = stats[C + c];
        const float dz = bn_dz(dy[e], y[e], act, slope);
        if (training) {
            const float xh = (x[e] - stats[c]) * r;
            dx[e] = w[c] * r * (dz - dbias[c] * inv_cnt - xh * dweight[c] * inv_cnt);
        } else {
            dx[e] = w[c] * r * dz;
        }
    }
}

extern "C" int jaf_batchnorm_act_bwd(jaf_stream_t s_, const float* dy, const float* x, const float* y,
                                     const float* stats, const float* weight, float* dx, float* dweight,
                                     float* dbias, int32_t N, int32_t C, int32_t HW, int act, float slope,
                                     int training) {
    JAF_REQUIRE(dy && x && y && stats && weight && dx && dweight && dbias && N >= 1 && C >= 1 && HW >= 1);
    hipStream_t s = (hipStream_t)s_;
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(C), dim3(256), 0, s, dy, x, y, stats, dweight, dbias, N, C, HW, act, slope);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(jaf_ew_grid((long)N * C * HW)), dim3(256), 0, s, dy, x, y, stats, weight,
                       dweight, dbias, dx, N, C, HW, act, slope, training);
    return jaf_launch_status();
}
