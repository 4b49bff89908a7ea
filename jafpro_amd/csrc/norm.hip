// Normalisation layers of the stage-4 networks.
//  * CRN LayerNorm (src/crn_model.py:67-87): per-sample statistics over C*H*W, Bessel-corrected
//    std, eps added to the std, per-channel affine, followed by LeakyReLU(0.01) (:100).
//  * BatchNorm2d in training mode (propagater and discriminators), SURVEY F9.
// All kernels are HBM-bound: 16 bytes per lane whenever H*W % 4 == 0, (pixel block, channel,
// image) grids so that no lane divides, reductions split over enough workgroups to fill 256 CUs
// and finished with fp64 atomics (sums of up to 16.7 M fp32 values per sample).
#include "jaf_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline bool al16(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr) {
    return ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d)) & 15) == 0;
}

__device__ __forceinline__ void block_sum2_atomic(double s, double ss, double* out_s, double* out_ss) {
    __shared__ double rs[4], rss[4];
    s = jaf_wave_sum(s);
    ss = jaf_wave_sum(ss);
    if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = s; rss[threadIdx.x >> 6] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(out_s, rs[0] + rs[1] + rs[2] + rs[3]);
        atomicAdd(out_ss, rss[0] + rss[1] + rss[2] + rss[3]);
    }
}

// ------------------------------------------------------------------ LayerNorm
template <int V>
__global__ void ln_partial_kernel(const float* x, long chw, double* ws) {
    const int n = blockIdx.y;
    const float* p = x + (long)n * chw;
    double s = 0.0, ss = 0.0;
    const long stride = (long)gridDim.x * blockDim.x;
    if (V == 4) {
        const f32x4* p4 = (const f32x4*)p;
        const long n4 = chw >> 2;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
            const f32x4 v = p4[i];
            // pairwise in fp32 first (4 values), then fp64: one conversion per 16 bytes
            const float a = v[0] + v[1], b = v[2] + v[3];
            const float qa = v[0] * v[0] + v[1] * v[1], qb = v[2] * v[2] + v[3] * v[3];
            s += (double)a + (double)b;
            ss += (double)qa + (double)qb;
        }
    } else {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < chw; i += stride) {
            const double v = (double)p[i];
            s += v;
            ss += v * v;
        }
    }
    block_sum2_atomic(s, ss, &ws[2 * n], &ws[2 * n + 1]);
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

__global__ void ln_finalize_slots_kernel(double* ws, int N, int slots, long chw, float eps, float* stats) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double s = 0.0, ss = 0.0;
    for (int k = 0; k < slots; ++k) {
        double* w = ws + ((long)n * slots + k) * 2;
        s += w[0];
        ss += w[1];
        w[0] = 0.0;          // left clean for the next accumulation: the caller never has to zero the buffer again
        w[1] = 0.0;
    }
    const double cnt = (double)chw;
    const double mean = s / cnt;
    double var = (ss - s * mean) / (cnt > 1.0 ? cnt - 1.0 : 1.0);
    if (var < 0.0) var = 0.0;
    const float stdv = (float)sqrt(var);
    stats[2 * n] = (float)mean;
    stats[2 * n + 1] = 1.0f / (stdv + eps);
}

extern "C" int jaf_layernorm_finalize(jaf_stream_t s, double* sums, int32_t N, int32_t slots, int64_t chw, float eps,
                                      float* stats) {
    JAF_REQUIRE(sums && stats && N >= 1 && slots >= 1 && chw >= 1);
    hipLaunchKernelGGL(ln_finalize_slots_kernel, dim3(jaf_cdiv(N, 64)), dim3(64), 0, (hipStream_t)s, sums, N, slots, (long)chw, eps, stats);
    return jaf_launch_status();
}

extern "C" int jaf_layernorm_stats(jaf_stream_t s_, const float* x, int32_t N, int64_t chw, float eps,
                                   double* workspace, float* stats) {
    JAF_REQUIRE(x && workspace && stats && N >= 1 && chw >= 1 && N <= 65535);
    hipStream_t s = (hipStream_t)s_;
    hipError_t e = hipMemsetAsync(workspace, 0, sizeof(double) * 2 * N, s);
    if (e != hipSuccess) return (int)e;
    const bool v4 = (chw % 4 == 0) && al16(x);
    int gx = jaf_ew_grid(v4 ? chw / 4 : chw, 4);
    const int cap = 2048 / N < 8 ? 8 : 2048 / N;
    if (gx > cap) gx = cap;
    if (v4) hipLaunchKernelGGL(ln_partial_kernel<4>, dim3(gx, N), dim3(256), 0, s, x, (long)chw, workspace);
    else hipLaunchKernelGGL(ln_partial_kernel<1>, dim3(gx, N), dim3(256), 0, s, x, (long)chw, workspace);
    hipLaunchKernelGGL(ln_finalize_kernel, dim3(jaf_cdiv(N, 64)), dim3(64), 0, s, workspace, N, (long)chw, eps, stats);
    return jaf_launch_status();
}

// grid (pixel blocks, C, N).  XT: element type of x (fp32, or bf16: the pre-LayerNorm convolution output under bf16 storage).
template <int V, typename XT>
__global__ void ln_lrelu_fwd_kernel(const XT* x, const float* stats, const float* gamma, const float* beta,
                                    float* y, int C, int HW, float slope) {
    const int c = blockIdx.y, n = blockIdx.z;
    const int pix = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (pix >= HW) return;
    const long e = ((long)n * C + c) * HW + pix;
    const float mean = stats[2 * n], r = stats[2 * n + 1], g = gamma[c], b = beta[c];
    float xv[V], o[V];
    jaf_ldv<V, XT>(x + e, xv);
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const float z = (xv[k] - mean) * r * g + b;
        o[k] = z > 0.f ? z : z * slope;
    }
    jaf_stv<V, float>(y + e, o);
}

extern "C" int jaf_layernorm_lrelu_fwd(jaf_stream_t s, const float* x, const float* stats, const float* gamma,
                                       const float* beta, float* y, int32_t N, int32_t C, int32_t HW, float slope) {
    return jaf_layernorm_lrelu_fwd_dt(s, x, 0, stats, gamma, beta, y, N, C, HW, slope);
}

extern "C" int jaf_layernorm_lrelu_fwd_dt(jaf_stream_t s, const void* x, int x_bf16, const float* stats, const float* gamma,
                                          const float* beta, float* y, int32_t N, int32_t C, int32_t HW, float slope) {
    JAF_REQUIRE(x && stats && gamma && beta && y && N >= 1 && C >= 1 && HW >= 1 && C <= 65535 && N <= 65535);
    const bool v4 = (HW % 4 == 0) && al16(x, y);
#define JAF_LNF(V_, T_) hipLaunchKernelGGL((ln_lrelu_fwd_kernel<V_, T_>), dim3(jaf_cdiv(HW / V_, 256), C, N), dim3(256), 0, (hipStream_t)s, \
                                           (const T_*)x, stats, gamma, beta, y, C, HW, slope)
    if (x_bf16) { if (v4) JAF_LNF(4, __bf16); else JAF_LNF(1, __bf16); }
    else { if (v4) JAF_LNF(4, float); else JAF_LNF(1, float); }
#undef JAF_LNF
    return jaf_launch_status();
}

// Same, and ALSO written as the consumer convolution's packed bf16 image [n][ng8][HW][8] (jaf_packed_io): a lane
// owns 8 channels x V pixels so that every 16-byte item it stores is complete.  grid (pixel blocks, ceil(C/8), N).
template <int V, typename XT>
__global__ __launch_bounds__(256) void ln_lrelu_fwd_packed_kernel(const XT* __restrict__ x, const float* __restrict__ stats,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  float* __restrict__ y, unsigned char* __restrict__ dst, int dst_ng8,
                                                                  int dst_cg0, int C, int HW, float slope, int split) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int cg = blockIdx.y, n = blockIdx.z;
    const int pix = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (pix >= HW) return;
    const float mean = stats[2 * n], r = stats[2 * n + 1];
    float o[8][V];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cg * 8 + j;
#pragma unroll
        for (int k = 0; k < V; ++k) o[j][k] = 0.f;
        if (c < C) {
            const long e = ((long)n * C + c) * HW + pix;
            const float g = gamma[c], b = beta[c];
            float xv[V];
            jaf_ldv<V, XT>(x + e, xv);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float z = (xv[k] - mean) * r * g + b;
                o[j][k] = z > 0.f ? z : z * slope;
            }
            if (y) jaf_stv<V, float>(y + e, o[j]);
        }
    }
    // split-bf16 destination: group cg's hi plane at 2 cg, the residual plane (v - bf16(v)) right behind it
    unsigned char* op = dst + ((((long)n * dst_ng8 + dst_cg0 + cg) * (split ? 2 : 1)) * (long)HW + pix) * 16;
#pragma unroll
    for (int k = 0; k < V; ++k) {
        u32x4 w;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f32x2 v2 = {o[2 * u][k], o[2 * u + 1][k]};
            w[u] = __builtin_bit_cast(unsigned int, __builtin_convertvector(v2, bf16x2));
        }
        *(u32x4*)(op + k * 16) = w;
        if (split) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float a0 = o[2 * u][k], a1 = o[2 * u + 1][k];
                const f32x2 v2 = {a0 - (float)(__bf16)a0, a1 - (float)(__bf16)a1};
                w[u] = __builtin_bit_cast(unsigned int, __builtin_convertvector(v2, bf16x2));
            }
            *(u32x4*)(op + (long)HW * 16 + k * 16) = w;
        }
    }
}

extern "C" int jaf_layernorm_lrelu_fwd_packed(jaf_stream_t s, const float* x, const float* stats, const float* gamma,
                                              const float* beta, float* y, void* dst, int32_t dst_ng8_tot, int32_t dst_coff,
                                              int32_t N, int32_t C, int32_t HW, float slope) {
    return jaf_layernorm_lrelu_fwd_packed_prec(s, x, stats, gamma, beta, y, dst, dst_ng8_tot, dst_coff, N, C, HW, slope, JAF_PREC_BF16);
}

extern "C" int jaf_layernorm_lrelu_fwd_packed_prec(jaf_stream_t s, const float* x, const float* stats, const float* gamma,
                                              const float* beta, float* y, void* dst, int32_t dst_ng8_tot, int32_t dst_coff,
                                              int32_t N, int32_t C, int32_t HW, float slope, int precision) {
    return jaf_layernorm_lrelu_fwd_packed_dt(s, x, 0, stats, gamma, beta, y, dst, dst_ng8_tot, dst_coff, N, C, HW, slope, precision);
}

extern "C" int jaf_layernorm_lrelu_fwd_packed_dt(jaf_stream_t s, const void* x, int x_bf16, const float* stats, const float* gamma,
                                                 const float* beta, float* y, void* dst, int32_t dst_ng8_tot, int32_t dst_coff,
                                                 int32_t N, int32_t C, int32_t HW, float slope, int precision) {
    JAF_REQUIRE(x && stats && gamma && beta && dst && N >= 1 && C >= 1 && HW >= 1 && N <= 65535);
    JAF_REQUIRE(precision == JAF_PREC_BF16 || precision == JAF_PREC_BF16X3);
    JAF_REQUIRE(!x_bf16 || precision == JAF_PREC_BF16);
    const int split = precision == JAF_PREC_BF16X3 ? 1 : 0;
    JAF_REQUIRE(dst_coff >= 0 && (dst_coff & 7) == 0 && dst_coff / 8 + jaf_cdiv(C, 8) <= dst_ng8_tot && jaf_cdiv(C, 8) <= 65535);
    const dim3 block(256);
    const bool v4 = (HW % 4 == 0) && al16(x, y);
#define JAF_LNP(V_, T_) hipLaunchKernelGGL((ln_lrelu_fwd_packed_kernel<V_, T_>), dim3(jaf_cdiv(HW / V_, 256), jaf_cdiv(C, 8), N), block, 0, \
                                           (hipStream_t)s, (const T_*)x, stats, gamma, beta, y, (unsigned char*)dst, dst_ng8_tot,          \
                                           dst_coff / 8, C, HW, slope, split)
    // (bf16 x with 8 pixels per lane -- 16-byte loads -- measured SLOWER than 4: 0.97 vs 0.83 ms per step, round 5: the lane then
    // holds 64 values and fewer lanes are in flight)
    if (x_bf16) { if (v4) JAF_LNP(4, __bf16); else JAF_LNP(1, __bf16); }
    else { if (v4) JAF_LNP(4, float); else JAF_LNP(1, float); }
#undef JAF_LNP
    return jaf_launch_status();
}

// pass A: per (c, n) block: a = sum dz, b = sum dz*xhat ; dbeta[c] += a, dgamma[c] += b,
// ws[n][c % 16][0] += gamma_c*a (S1), ws[n][c % 16][1] += gamma_c*b (S2): the C workgroups of one image spread their
// fp64 atomics over 16 slots (one address per image cost a ~21 us floor per launch), folded by ln_bwd_fold_kernel.
#define LN_BWD_SLOTS 16
template <int V, typename DT, typename XT>
__global__ void ln_bwd_reduce_kernel(const DT* dy, const XT* x, const float* stats, const float* gamma,
                                     const float* beta, float* dgamma, float* dbeta, double* ws, int C, int HW,
                                     float slope, float* cn /* nullable: [N][C][2] = (gamma_c * sum dz, sum xhat) */) {
    const int c = blockIdx.x;
    const int n = blockIdx.y;
    const long base = ((long)n * C + c) * HW;
    const float mean = stats[2 * n], r = stats[2 * n + 1];
    const float g = gamma[c], b = beta[c];
    double sa = 0.0, sb = 0.0, sx = 0.0;
    if (V > 1) {
        for (int i = threadIdx.x; i < HW / V; i += blockDim.x) {
            float xv[V], dv[V];
            jaf_ldv<V, XT>(x + base + V * i, xv);
            jaf_ldv<V, DT>(dy + base + V * i, dv);
            float pa = 0.f, pb = 0.f, px = 0.f;
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float xh = (xv[k] - mean) * r;
                const float z = xh * g + b;
                const float dz = dv[k] * (z > 0.f ? 1.f : slope);
                pa += dz;
                pb += dz * xh;
                px += xh;
            }
            sa += (double)pa;
            sb += (double)pb;
            sx += (double)px;
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += blockDim.x) {
            const float xh = ((float)x[base + i] - mean) * r;
            const float z = xh * g + b;
            const float dz = (float)dy[base + i] * (z > 0.f ? 1.f : slope);
            sa += (double)dz;
            sb += (double)dz * (double)xh;
            sx += (double)xh;
        }
    }
    __shared__ double ra[4], rb[4], rx[4];
    sa = jaf_wave_sum(sa);
    sb = jaf_wave_sum(sb);
    if (cn) sx = jaf_wave_sum(sx);
    if ((threadIdx.x & 63) == 0) { ra[threadIdx.x >> 6] = sa; rb[threadIdx.x >> 6] = sb; rx[threadIdx.x >> 6] = sx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double a = ra[0] + ra[1] + ra[2] + ra[3];
        const double bb = rb[0] + rb[1] + rb[2] + rb[3];
        if (cn) {
            cn[((long)n * C + c) * 2] = (float)((double)g * a);
            cn[((long)n * C + c) * 2 + 1] = (float)(rx[0] + rx[1] + rx[2] + rx[3]);
        }
        atomicAdd(&dbeta[c], (float)a);
        atomicAdd(&dgamma[c], (float)bb);
        double* w = ws + ((long)n * LN_BWD_SLOTS + (c & (LN_BWD_SLOTS - 1))) * 2;
        atomicAdd(w, (double)g * a);
        atomicAdd(w + 1, (double)g * bb);
    }
}

// ws[n][0] = sum over the slots (in place): what ln_bwd_apply_kernel reads
__global__ void ln_bwd_fold_kernel(double* ws, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double* w = ws + (long)n * LN_BWD_SLOTS * 2;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < LN_BWD_SLOTS; ++k) { s1 += w[2 * k]; s2 += w[2 * k + 1]; }
    w[0] = s1;
    w[1] = s2;
}

// grid (pixel blocks, C, N).  dx takes x's element type.
template <int V, typename DT, typename XT>
__global__ void ln_bwd_apply_kernel(const DT* dy, const XT* x, const float* stats, const float* gamma,
                                    const float* beta, const double* ws, XT* dx, int C, int HW, float slope,
                                    float eps) {
    const int c = blockIdx.y, n = blockIdx.z;
    const int pix = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (pix >= HW) return;
    const long e = ((long)n * C + c) * HW + pix;
    const double M = (double)C * (double)HW;
    const float mean = stats[2 * n], r = stats[2 * n + 1];
    const float sigma = 1.0f / r - eps;
    const double* w = ws + (long)n * LN_BWD_SLOTS * 2;        // folded sums (ln_bwd_fold_kernel)
    const float m1 = (float)(w[0] / M);
    // S2 / ((M-1) * sigma * r)
    const float kk = (sigma > 0.f) ? (float)(w[1] / ((M - 1.0) * (double)sigma * (double)r)) : 0.f;
    const float g = gamma[c], b = beta[c];
    float xv[V], dv[V], o[V];
    jaf_ldv<V, XT>(x + e, xv);
    jaf_ldv<V, DT>(dy + e, dv);
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const float xh = (xv[k] - mean) * r;
        const float z = xh * g + b;
        const float dxh = dv[k] * (z > 0.f ? 1.f : slope) * g;
        o[k] = r * (dxh - m1 - xh * kk);
    }
    jaf_stv<V, XT>(dx + e, o);
}

extern "C" int jaf_layernorm_lrelu_bwd(jaf_stream_t s_, const float* dy, const float* x, const float* stats,
                                       const float* gamma, const float* beta, float* dx, float* dgamma,
                                       float* dbeta, double* workspace, int32_t N, int32_t C, int32_t HW,
                                       float slope, float eps) {
    return jaf_layernorm_lrelu_bwd_dt(s_, dy, 0, x, 0, stats, gamma, beta, dx, dgamma, dbeta, workspace, N, C, HW, slope, eps);
}

// LN_DISPATCH(F): F(V, DT, XT) for the element types of dy and x given at run time
#define LN_DISPATCH(F, V_)                                                       \
    do {                                                                         \
        if (dy_bf16) { if (x_bf16) F(V_, __bf16, __bf16); else F(V_, __bf16, float); } \
        else { if (x_bf16) F(V_, float, __bf16); else F(V_, float, float); }     \
    } while (0)

extern "C" int jaf_layernorm_lrelu_bwd_dt(jaf_stream_t s_, const void* dy, int dy_bf16, const void* x, int x_bf16, const float* stats,
                                          const float* gamma, const float* beta, void* dx, float* dgamma,
                                          float* dbeta, double* workspace, int32_t N, int32_t C, int32_t HW,
                                          float slope, float eps) {
    JAF_REQUIRE(dy && x && stats && gamma && beta && dx && dgamma && dbeta && workspace);
    JAF_REQUIRE(N >= 1 && C >= 1 && HW >= 1 && N <= 65535 && C <= 65535);
    hipStream_t s = (hipStream_t)s_;
    hipError_t e = hipMemsetAsync(workspace, 0, sizeof(double) * 2 * LN_BWD_SLOTS * N, s);
    if (e != hipSuccess) return (int)e;
    const bool v4 = (HW % 4 == 0) && al16(dy, x, dx);
#define LN_RED(V_, D_, X_) hipLaunchKernelGGL((ln_bwd_reduce_kernel<V_, D_, X_>), dim3(C, N), dim3(256), 0, s, (const D_*)dy, (const X_*)x, stats, \
                                              gamma, beta, dgamma, dbeta, workspace, C, HW, slope, (float*)nullptr)
#define LN_APP(V_, D_, X_) hipLaunchKernelGGL((ln_bwd_apply_kernel<V_, D_, X_>), dim3(jaf_cdiv(HW / V_, 256), C, N), dim3(256), 0, s,           \
                                              (const D_*)dy, (const X_*)x, stats, gamma, beta, workspace, (X_*)dx, C, HW, slope, eps)
    if (v4) LN_DISPATCH(LN_RED, 4); else LN_DISPATCH(LN_RED, 1);
    hipLaunchKernelGGL(ln_bwd_fold_kernel, dim3(jaf_cdiv(N, 64)), dim3(64), 0, s, workspace, N);
    if (v4) LN_DISPATCH(LN_APP, 4); else LN_DISPATCH(LN_APP, 1);
#undef LN_RED
#undef LN_APP
    return jaf_launch_status();
}

// The same backward with the result handed to the PRODUCING convolution (act NONE, groups 1: the CRN's conv -> LayerNorm
// -> LeakyReLU blocks, src/crn_model.py:90-106) in the form its data / weight gradient kernels read: dx as a packed bf16
// image [n][ceil(C/8)][HW][8] -- a lane owns 8 channels x V pixels, as in ln_lrelu_fwd_packed_kernel -- instead of an fp32
// tensor that jaf_conv2d_pack_dz would read back (4 + 4 + 2 bytes per element become 2).  grid (pixel blocks, ceil(C/8), N).
template <int V, typename DT, typename XT>
__global__ __launch_bounds__(256) void ln_bwd_apply_packed_kernel(const DT* __restrict__ dy, const XT* __restrict__ x,
                                                                  const float* __restrict__ stats, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, const double* __restrict__ ws,
                                                                  unsigned char* __restrict__ dst, int C, int HW, float slope, float eps, int split) {
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int cg = blockIdx.y, n = blockIdx.z;
    const int pix = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (pix >= HW) return;
    const double M = (double)C * (double)HW;
    const float mean = stats[2 * n], r = stats[2 * n + 1];
    const float sigma = 1.0f / r - eps;
    const double* w = ws + (long)n * LN_BWD_SLOTS * 2;
    const float m1 = (float)(w[0] / M);
    const float kk = (sigma > 0.f) ? (float)(w[1] / ((M - 1.0) * (double)sigma * (double)r)) : 0.f;
    float o[8][V];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cg * 8 + j;
#pragma unroll
        for (int k = 0; k < V; ++k) o[j][k] = 0.f;
        if (c < C) {
            const long e = ((long)n * C + c) * HW + pix;
            const float g = gamma[c], b = beta[c];
            float xv[V], dv[V];
            jaf_ldv<V, XT>(x + e, xv);
            jaf_ldv<V, DT>(dy + e, dv);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float xh = (xv[k] - mean) * r;
                const float z = xh * g + b;
                const float dxh = dv[k] * (z > 0.f ? 1.f : slope) * g;
                o[j][k] = r * (dxh - m1 - xh * kk);
            }
        }
    }
    // (split-bf16: hi plane of the group at 2 cg, the residual plane right behind it)
    unsigned char* op = dst + ((((long)n * gridDim.y + cg) * (split ? 2 : 1)) * (long)HW + pix) * 16;
#pragma unroll
    for (int k = 0; k < V; ++k) {
        u32x4 wv;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f32x2 v2 = {o[2 * u][k], o[2 * u + 1][k]};
            wv[u] = __builtin_bit_cast(unsigned int, __builtin_convertvector(v2, bf16x2));
        }
        *(u32x4*)(op + k * 16) = wv;
        if (split) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float a0 = o[2 * u][k], a1 = o[2 * u + 1][k];
                const f32x2 v2 = {a0 - (float)(__bf16)a0, a1 - (float)(__bf16)a1};
                wv[u] = __builtin_bit_cast(unsigned int, __builtin_convertvector(v2, bf16x2));
            }
            *(u32x4*)(op + (long)HW * 16 + k * 16) = wv;
        }
    }
}

// The producing convolution's bias gradient, sum over images and pixels of dx, from the per-(image, channel) sums the
// reduce pass already took: sum_p dx[n][c] = r_n (gamma_c sum dz - HW m1_n - kk_n sum xhat).  One thread per channel.
__global__ void ln_bwd_conv_bias_kernel(const float* cn, const float* stats, const double* ws, float* dbias, int N, int C, int HW,
                                        float eps, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double M = (double)C * (double)HW;
    double acc = 0.0;
    for (int n = 0; n < N; ++n) {
        const float r = stats[2 * n + 1];
        const float sigma = 1.0f / r - eps;
        const double* w = ws + (long)n * LN_BWD_SLOTS * 2;
        const float m1 = (float)(w[0] / M);
        const float kk = (sigma > 0.f) ? (float)(w[1] / ((M - 1.0) * (double)sigma * (double)r)) : 0.f;
        acc += (double)r * ((double)cn[((long)n * C + c) * 2] - (double)HW * (double)m1 - (double)kk * (double)cn[((long)n * C + c) * 2 + 1]);
    }
    dbias[c] = (accumulate ? dbias[c] : 0.f) + (float)acc;
}

extern "C" int jaf_layernorm_lrelu_bwd_packed(jaf_stream_t s_, const float* dy, const float* x, const float* stats,
                                              const float* gamma, const float* beta, void* packed_dx, float* dgamma,
                                              float* dbeta, double* workspace, float* scratch, float* conv_dbias,
                                              int accumulate_dbias, int32_t N, int32_t C, int32_t HW, float slope, float eps) {
    return jaf_layernorm_lrelu_bwd_packed_prec(s_, dy, x, stats, gamma, beta, packed_dx, dgamma, dbeta, workspace, scratch, conv_dbias,
                                               accumulate_dbias, N, C, HW, slope, eps, JAF_PREC_BF16);
}

extern "C" int jaf_layernorm_lrelu_bwd_packed_prec(jaf_stream_t s_, const float* dy, const float* x, const float* stats,
                                                   const float* gamma, const float* beta, void* packed_dx, float* dgamma,
                                                   float* dbeta, double* workspace, float* scratch, float* conv_dbias,
                                                   int accumulate_dbias, int32_t N, int32_t C, int32_t HW, float slope, float eps,
                                                   int precision) {
    return jaf_layernorm_lrelu_bwd_packed_dt(s_, dy, 0, x, 0, stats, gamma, beta, packed_dx, dgamma, dbeta, workspace, scratch, conv_dbias,
                                             accumulate_dbias, N, C, HW, slope, eps, precision);
}

extern "C" int jaf_layernorm_lrelu_bwd_packed_dt(jaf_stream_t s_, const void* dy, int dy_bf16, const void* x, int x_bf16,
                                                 const float* stats, const float* gamma, const float* beta, void* packed_dx,
                                                 float* dgamma, float* dbeta, double* workspace, float* scratch, float* conv_dbias,
                                                 int accumulate_dbias, int32_t N, int32_t C, int32_t HW, float slope, float eps,
                                                 int precision) {
    JAF_REQUIRE(dy && x && stats && gamma && beta && packed_dx && dgamma && dbeta && workspace && scratch);
    JAF_REQUIRE(precision == JAF_PREC_BF16 || precision == JAF_PREC_BF16X3);
    JAF_REQUIRE(!(dy_bf16 || x_bf16) || precision == JAF_PREC_BF16);
    const int split = precision == JAF_PREC_BF16X3 ? 1 : 0;
    JAF_REQUIRE(N >= 1 && C >= 1 && HW >= 1 && N <= 65535 && C <= 65535);
    hipStream_t s = (hipStream_t)s_;
    hipError_t e = hipMemsetAsync(workspace, 0, sizeof(double) * 2 * LN_BWD_SLOTS * N, s);
    if (e != hipSuccess) return (int)e;
    const bool v4 = (HW % 4 == 0) && al16(dy, x);
#define LN_RED(V_, D_, X_) hipLaunchKernelGGL((ln_bwd_reduce_kernel<V_, D_, X_>), dim3(C, N), dim3(256), 0, s, (const D_*)dy, (const X_*)x, stats, \
                                              gamma, beta, dgamma, dbeta, workspace, C, HW, slope, scratch)
#define LN_APP(V_, D_, X_) hipLaunchKernelGGL((ln_bwd_apply_packed_kernel<V_, D_, X_>), dim3(jaf_cdiv(HW / V_, 256), jaf_cdiv(C, 8), N), dim3(256), \
                                              0, s, (const D_*)dy, (const X_*)x, stats, gamma, beta, workspace, (unsigned char*)packed_dx, C, HW,  \
                                              slope, eps, split)
    if (v4) LN_DISPATCH(LN_RED, 4); else LN_DISPATCH(LN_RED, 1);
    hipLaunchKernelGGL(ln_bwd_fold_kernel, dim3(jaf_cdiv(N, 64)), dim3(64), 0, s, workspace, N);
    if (v4) LN_DISPATCH(LN_APP, 4); else LN_DISPATCH(LN_APP, 1);
#undef LN_RED
#undef LN_APP
    if (conv_dbias)
        hipLaunchKernelGGL(ln_bwd_conv_bias_kernel, dim3(jaf_cdiv(C, 64)), dim3(64), 0, s, scratch, stats, workspace, conv_dbias, N, C,
                           HW, eps, accumulate_dbias ? 1 : 0);
    return jaf_launch_status();
}

// ------------------------------------------------------------------ BatchNorm2d
// Per-channel reductions: grid (C, nsplit); a workgroup walks items = (image, 4096-element chunk of
// the plane) in strides of nsplit and adds its partial sums to ws[2c], ws[2c+1] (fp64 atomics).
#define BN_CHUNK 4096

static int bn_nsplit(int C, int N, int HW) {
    const long items = (long)N * jaf_cdiv(HW, BN_CHUNK);
    long ns = (1024 + C - 1) / C;
    if (ns > items) ns = items;
    if (ns < 1) ns = 1;
    return (int)ns;
}

template <int V>
__global__ void bn_stats_partial_kernel(const float* x, int N, int C, int HW, double* ws) {
    const int c = blockIdx.x;
    const int chunks = (HW + BN_CHUNK - 1) / BN_CHUNK;
    const int items = N * chunks;
    double s = 0.0, ss = 0.0;
    for (int item = blockIdx.y; item < items; item += gridDim.y) {
        const int n = item / chunks;
        const int ch = item - n * chunks;
        const float* p = x + ((long)n * C + c) * HW;
        const int lo = ch * BN_CHUNK;
        const int hi = lo + BN_CHUNK < HW ? lo + BN_CHUNK : HW;
        if (V == 4) {
            for (int i = lo + threadIdx.x * 4; i < hi; i += blockDim.x * 4) {
                const f32x4 v = *(const f32x4*)(p + i);
                s += (double)((v[0] + v[1]) + (v[2] + v[3]));
                ss += (double)((v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]));
            }
        } else {
            for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
                const double v = (double)p[i];
                s += v;
                ss += v * v;
            }
        }
    }
    block_sum2_atomic(s, ss, &ws[2 * c], &ws[2 * c + 1]);
}

__global__ void bn_stats_finalize_kernel(const double* ws, int N, int C, int HW, float eps, float momentum,
                                         float* running_mean, float* running_var, float* stats, int training) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (!training) {
        stats[c] = running_mean[c];
        stats[C + c] = 1.0f / sqrtf(running_var[c] + eps);
        return;
    }
    const double cnt = (double)N * (double)HW;
    const double mean = ws[2 * c] / cnt;
    double var = ws[2 * c + 1] / cnt - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[c] = (float)mean;
    stats[C + c] = (float)(1.0 / sqrt(var + (double)eps));
    if (running_mean) {
        const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

extern "C" int jaf_batchnorm_stats(jaf_stream_t s_, const float* x, int32_t N, int32_t C, int32_t HW, float eps,
                                   float momentum, float* running_mean, float* running_var, float* stats,
                                   int training, double* workspace) {
    JAF_REQUIRE(x && stats && workspace && N >= 1 && C >= 1 && HW >= 1);
    JAF_REQUIRE(training || (running_mean && running_var));
    hipStream_t s = (hipStream_t)s_;
    if (training) {
        hipError_t e = hipMemsetAsync(workspace, 0, sizeof(double) * 2 * C, s);
        if (e != hipSuccess) return (int)e;
        const int ns = bn_nsplit(C, N, HW);
        if ((HW % 4 == 0) && al16(x))
            hipLaunchKernelGGL(bn_stats_partial_kernel<4>, dim3(C, ns), dim3(256), 0, s, x, N, C, HW, workspace);
        else
            hipLaunchKernelGGL(bn_stats_partial_kernel<1>, dim3(C, ns), dim3(256), 0, s, x, N, C, HW, workspace);
    }
    hipLaunchKernelGGL(bn_stats_finalize_kernel, dim3(jaf_cdiv(C, 64)), dim3(64), 0, s, workspace, N, C, HW, eps, momentum,
                       running_mean, running_var, stats, training);
    return jaf_launch_status();
}

// grid (pixel blocks, C, N)
template <int V>
__global__ void bn_act_fwd_kernel(const float* x, const float* stats, const float* w, const float* b,
                                  const float* residual, float* y, int C, int HW, int act, float slope) {
    const int c = blockIdx.y, n = blockIdx.z;
    const int pix = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (pix >= HW) return;
    const long e = ((long)n * C + c) * HW + pix;
    const float mean = stats[c], sc = stats[C + c] * w[c], bb = b[c];
    if (V == 4) {
        const f32x4 xv = *(const f32x4*)(x + e);
        f32x4 rv = {0.f, 0.f, 0.f, 0.f};
        if (residual) rv = *(const f32x4*)(residual + e);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = jaf_act((xv[k] - mean) * sc + bb, act, slope) + rv[k];
        *(f32x4*)(y + e) = o;
    } else {
        float v = jaf_act((x[e] - mean) * sc + bb, act, slope);
        if (residual) v += residual[e];
        y[e] = v;
    }
}

extern "C" int jaf_batchnorm_act_fwd(jaf_stream_t s, const float* x, const float* stats, const float* weight,
                                     const float* bias, const float* residual, float* y, int32_t N, int32_t C,
                                     int32_t HW, int act, float slope) {
    JAF_REQUIRE(x && stats && weight && bias && y && N >= 1 && C >= 1 && HW >= 1 && C <= 65535 && N <= 65535);
    JAF_REQUIRE(!residual || act == JAF_ACT_NONE);
    if ((HW % 4 == 0) && al16(x, y, residual))
        hipLaunchKernelGGL(bn_act_fwd_kernel<4>, dim3(jaf_cdiv(HW / 4, 256), C, N), dim3(256), 0, (hipStream_t)s, x, stats,
                           weight, bias, residual, y, C, HW, act, slope);
    else
        hipLaunchKernelGGL(bn_act_fwd_kernel<1>, dim3(jaf_cdiv(HW, 256), C, N), dim3(256), 0, (hipStream_t)s, x, stats,
                           weight, bias, residual, y, C, HW, act, slope);
    return jaf_launch_status();
}

// Small tensors (the discriminators' 4 x 4 .. 64 x 64 feature maps: 55 BatchNorm layers per train step, each a chain of
// memset + partial sums + finalize + apply launches of 3-7 us that the next layer waits for): ONE workgroup per channel
// takes the statistics and applies them, the second pass over its <= 256 KB coming from L2.  Same fp64 sums and the same
// finalisation arithmetic as the three-kernel path.
#define BN_SMALL_MAX 65536      // elements per channel (N * HW)
// The 16-byte walk of the small kernels: a channel's n x (HW / 4) float4 of a chunk as ONE index space (image = q / hw4), thread t takes
// q = t, t + 256, ... and keeps FOUR loads in flight -- with one workgroup per channel the kernels are latency chains (the per-image loop
// they replace left 3/4 of the lanes idle on the 8 x 8 .. 16 x 16 maps and issued one load at a time on the 64 x 64 ones).  Every small
// kernel walks this way, so the one-launch split form and the per-chunk form stay bit-identical.
struct BnWalk { int hw4, tot4; float inv; };
__device__ __forceinline__ BnWalk bn_walk(int n, int HW) { BnWalk w; w.hw4 = HW >> 2; w.tot4 = n * w.hw4; w.inv = 1.0f / (float)w.hw4; return w; }
__device__ __forceinline__ long bn_off4(const BnWalk& w, int q, int C, int HW) {
    const int i = (int)(((float)q + 0.5f) * w.inv);
    return ((long)i * C) * HW + (long)(q - i * w.hw4) * 4;
}
template <int V>
__global__ __launch_bounds__(256) void bn_fwd_small_kernel(const float* x, int N, int C, int HW, float eps, float momentum,
                                                           float* running_mean, float* running_var, float* stats,
                                                           const float* w, const float* b, const float* residual, float* y,
                                                           int act, float slope) {
    const int c = blockIdx.x;
    double s = 0.0, ss = 0.0;
    if (V == 4) {
        const BnWalk wk = bn_walk(N, HW);
        const float* p0 = x + (long)c * HW;
        for (int q0 = threadIdx.x; q0 < wk.tot4; q0 += 1024) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (q0 + 256 * u < wk.tot4) v[u] = *(const f32x4*)(p0 + bn_off4(wk, q0 + 256 * u, C, HW));
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (q0 + 256 * u < wk.tot4) {
                    s += (double)((v[u][0] + v[u][1]) + (v[u][2] + v[u][3]));
                    ss += (double)((v[u][0] * v[u][0] + v[u][1] * v[u][1]) + (v[u][2] * v[u][2] + v[u][3] * v[u][3]));
                }
        }
    } else {
        for (int n = 0; n < N; ++n) {
            const float* p = x + ((long)n * C + c) * HW;
            for (int i = threadIdx.x; i < HW; i += 256) {
                const double v = (double)p[i];
                s += v;
                ss += v * v;
            }
        }
    }
    __shared__ double rs[4], rss[4];
    __shared__ float sh[2];
    s = jaf_wave_sum(s);
    ss = jaf_wave_sum(ss);
    if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = s; rss[threadIdx.x >> 6] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double cnt = (double)N * (double)HW;
        const double mean = ((rs[0] + rs[1]) + (rs[2] + rs[3])) / cnt;
        double var = ((rss[0] + rss[1]) + (rss[2] + rss[3])) / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        sh[0] = (float)mean;
        sh[1] = (float)(1.0 / sqrt(var + (double)eps));
        stats[c] = sh[0];
        stats[C + c] = sh[1];
        if (running_mean) {
            const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
        }
    }
    __syncthreads();
    const float mean = sh[0], sc = sh[1] * w[c], bb = b[c];
    if (V == 4) {
        const BnWalk wk = bn_walk(N, HW);
        const long c0 = (long)c * HW;
        for (int q0 = threadIdx.x; q0 < wk.tot4; q0 += 1024) {
            f32x4 xv[4], rv[4];
            long off[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                rv[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (q0 + 256 * u < wk.tot4) {
                    off[u] = c0 + bn_off4(wk, q0 + 256 * u, C, HW);
                    xv[u] = *(const f32x4*)(x + off[u]);
                    if (residual) rv[u] = *(const f32x4*)(residual + off[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (q0 + 256 * u < wk.tot4) {
                    f32x4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = jaf_act((xv[u][k] - mean) * sc + bb, act, slope) + rv[u][k];
                    *(f32x4*)(y + off[u]) = o;
                }
        }
    } else {
        for (int n = 0; n < N; ++n) {
            const long base = ((long)n * C + c) * HW;
            for (int i = threadIdx.x; i < HW; i += 256) {
                float v = jaf_act((x[base + i] - mean) * sc + bb, act, slope);
                if (residual) v += residual[base + i];
                y[base + i] = v;
            }
        }
    }
}

// `parts` equal chunks of the batch, each normalised with ITS OWN batch statistics, the running statistics updated chunk after chunk
// (what `parts` successive calls on the chunks compute: the discriminators' real / generated halves, ops._SplitBatchNormActFn), in ONE
// launch: 256 threads per chunk do exactly what bn_fwd_small_kernel's workgroup does on it -- same per-thread strides, same reduction
// order, the chunks finalised in order by thread 0 -- so the results are bit-identical to the per-chunk launches.
template <int V>
__global__ __launch_bounds__(1024) void bn_fwd_small_parts_kernel(const float* x, int n, int C, int HW, float eps, float momentum,
                                                                  float* running_mean, float* running_var, float* stats, const float* w,
                                                                  const float* b, float* y, int act, float slope, int parts) {
    const int c = blockIdx.x;
    const int part = threadIdx.x >> 8, tid = threadIdx.x & 255;
    const long poff = (long)part * n * C * HW;
    double s = 0.0, ss = 0.0;
    if (V == 4) {
        const BnWalk wk = bn_walk(n, HW);
        const float* p0 = x + poff + (long)c * HW;
        for (int q0 = tid; q0 < wk.tot4; q0 += 1024) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (q0 + 256 * u < wk.tot4) v[u] = *(const f32x4*)(p0 + bn_off4(wk, q0 + 256 * u, C, HW));
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (q0 + 256 * u < wk.tot4) {
                    s += (double)((v[u][0] + v[u][1]) + (v[u][2] + v[u][3]));
                    ss += (double)((v[u][0] * v[u][0] + v[u][1] * v[u][1]) + (v[u][2] * v[u][2] + v[u][3] * v[u][3]));
                }
        }
    } else {
        for (int i = 0; i < n; ++i) {
            const float* p = x + poff + ((long)i * C + c) * HW;
            for (int e = tid; e < HW; e += 256) {
                const double v = (double)p[e];
                s += v;
                ss += v * v;
            }
        }
    }
    __shared__ double rs[4][4], rss[4][4];
    __shared__ float sh[4][2];
    s = jaf_wave_sum(s);
    ss = jaf_wave_sum(ss);
    if ((tid & 63) == 0) { rs[part][tid >> 6] = s; rss[part][tid >> 6] = ss; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 0; k < parts; ++k) {
            const double cnt = (double)n * (double)HW;
            const double mean = ((rs[k][0] + rs[k][1]) + (rs[k][2] + rs[k][3])) / cnt;
            double var = ((rss[k][0] + rss[k][1]) + (rss[k][2] + rss[k][3])) / cnt - mean * mean;
            if (var < 0.0) var = 0.0;
            sh[k][0] = (float)mean;
            sh[k][1] = (float)(1.0 / sqrt(var + (double)eps));
            stats[(long)k * 2 * C + c] = sh[k][0];
            stats[(long)k * 2 * C + C + c] = sh[k][1];
            if (running_mean) {
                const double unbiased = cnt > 1.0 ? var * cnt / (cnt - 1.0) : var;
                running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
                running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
            }
        }
    }
    __syncthreads();
    const float mean = sh[part][0], sc = sh[part][1] * w[c], bb = b[c];
    if (V == 4) {
        const BnWalk wk = bn_walk(n, HW);
        const long c0 = poff + (long)c * HW;
        for (int q0 = tid; q0 < wk.tot4; q0 += 1024) {
            f32x4 xv[4];
            long off[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (q0 + 256 * u < wk.tot4) {
                    off[u] = c0 + bn_off4(wk, q0 + 256 * u, C, HW);
                    xv[u] = *(const f32x4*)(x + off[u]);
                }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (q0 + 256 * u < wk.tot4) {
                    f32x4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = jaf_act((xv[u][k] - mean) * sc + bb, act, slope) + 0.f;
                    *(f32x4*)(y + off[u]) = o;
                }
        }
    } else {
        for (int i = 0; i < n; ++i) {
            const long base = poff + ((long)i * C + c) * HW;
            for (int e = tid; e < HW; e += 256) y[base + e] = jaf_act((x[base + e] - mean) * sc + bb, act, slope);
        }
    }
}

extern "C" int jaf_batchnorm_act_fwd_split(jaf_stream_t s_, const float* x, int32_t N, int32_t C, int32_t HW, float eps, float momentum,
                                           float* running_mean, float* running_var, float* stats, const float* weight,
                                           const float* bias, float* y, int act, float slope, int32_t parts) {
    JAF_REQUIRE(x && stats && weight && bias && y && N >= 1 && C >= 1 && HW >= 1 && C <= 65535 && parts >= 1 && parts <= 4 && N % parts == 0);
    const int n = N / parts;
    if ((long)n * HW > BN_SMALL_MAX) return JAF_EUNSUPPORTED;
    if ((HW % 4 == 0) && al16(x, y))
        hipLaunchKernelGGL(bn_fwd_small_parts_kernel<4>, dim3(C), dim3(256 * parts), 0, (hipStream_t)s_, x, n, C, HW, eps, momentum,
                           running_mean, running_var, stats, weight, bias, y, act, slope, parts);
    else
        hipLaunchKernelGGL(bn_fwd_small_parts_kernel<1>, dim3(C), dim3(256 * parts), 0, (hipStream_t)s_, x, n, C, HW, eps, momentum,
                           running_mean, running_var, stats, weight, bias, y, act, slope, parts);
    return jaf_launch_status();
}

extern "C" int jaf_batchnorm_act_fwd_fused(jaf_stream_t s_, const float* x, int32_t N, int32_t C, int32_t HW, float eps,
                                           float momentum, float* running_mean, float* running_var, float* stats,
                                           int training, double* workspace, const float* weight, const float* bias,
                                           const float* residual, float* y, int act, float slope) {
    JAF_REQUIRE(x && stats && workspace && weight && bias && y && N >= 1 && C >= 1 && HW >= 1 && C <= 65535 && N <= 65535);
    JAF_REQUIRE(training || (running_mean && running_var));
    JAF_REQUIRE(!residual || act == JAF_ACT_NONE);
    if (training && (long)N * HW <= BN_SMALL_MAX) {
        if ((HW % 4 == 0) && al16(x, y, residual))
            hipLaunchKernelGGL(bn_fwd_small_kernel<4>, dim3(C), dim3(256), 0, (hipStream_t)s_, x, N, C, HW, eps, momentum, running_mean,
                               running_var, stats, weight, bias, residual, y, act, slope);
        else
            hipLaunchKernelGGL(bn_fwd_small_kernel<1>, dim3(C), dim3(256), 0, (hipStream_t)s_, x, N, C, HW, eps, momentum, running_mean,
                               running_var, stats, weight, bias, residual, y, act, slope);
        return jaf_launch_status();
    }
    const int rc = jaf_batchnorm_stats(s_, x, N, C, HW, eps, momentum, running_mean, running_var, stats, training, workspace);
    if (rc != JAF_OK) return rc;
    return jaf_batchnorm_act_fwd(s_, x, stats, weight, bias, residual, y, N, C, HW, act, slope);
}

__device__ __forceinline__ float bn_dz(float dy, float y, int act, float slope) {
    switch (act) {
        case JAF_ACT_LRELU: return dy * (y > 0.f ? 1.f : slope);
        case JAF_ACT_RELU: return dy * (y > 0.f ? 1.f : 0.f);
        case JAF_ACT_SIGMOID: return dy * y * (1.f - y);
        default: return dy;
    }
}

template <int V>
__global__ void bn_bwd_reduce_kernel(const float* dy, const float* x, const float* y, const float* stats, double* ws,
                                     int N, int C, int HW, int act, float slope) {
    const int c = blockIdx.x;
    const float mean = stats[c], r = stats[C + c];
    const int chunks = (HW + BN_CHUNK - 1) / BN_CHUNK;
    const int items = N * chunks;
    double sa = 0.0, sb = 0.0;
    for (int item = blockIdx.y; item < items; item += gridDim.y) {
        const int n = item / chunks;
        const int ch = item - n * chunks;
        const long base = ((long)n * C + c) * HW;
        const int lo = ch * BN_CHUNK;
        const int hi = lo + BN_CHUNK < HW ? lo + BN_CHUNK : HW;
        if (V == 4) {
            for (int i = lo + threadIdx.x * 4; i < hi; i += blockDim.x * 4) {
                const f32x4 dv = *(const f32x4*)(dy + base + i), xv = *(const f32x4*)(x + base + i);
                const f32x4 yv = *(const f32x4*)(y + base + i);
                float pa = 0.f, pb = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float dz = bn_dz(dv[k], yv[k], act, slope);
                    pa += dz;
                    pb += dz * ((xv[k] - mean) * r);
                }
                sa += (double)pa;
                sb += (double)pb;
            }
        } else {
            for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
                const float dz = bn_dz(dy[base + i], y[base + i], act, slope);
                sa += (double)dz;
                sb += (double)dz * (double)((x[base + i] - mean) * r);
            }
        }
    }
    block_sum2_atomic(sa, sb, &ws[2 * c], &ws[2 * c + 1]);
}

__global__ void bn_bwd_finalize_kernel(const double* ws, float* dweight, float* dbias, int C, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    dbias[c] = (accumulate ? dbias[c] : 0.f) + (float)ws[2 * c];
    dweight[c] = (accumulate ? dweight[c] : 0.f) + (float)ws[2 * c + 1];
}

// grid (pixel blocks, C, N)
template <int V>
__global__ void bn_bwd_apply_kernel(const float* dy, const float* x, const float* y, const float* stats,
                                    const float* w, const double* ws, float* dx, int N, int C, int HW, int act,
                                    float slope, int training) {
    const int c = blockIdx.y, n = blockIdx.z;
    const int pix = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (pix >= HW) return;
    const long e = ((long)n * C + c) * HW + pix;
    const float inv_cnt = 1.0f / ((float)N * (float)HW);
    const float mean = stats[c], r = stats[C + c], wr = w[c] * r;
    const float db = training ? (float)ws[2 * c] * inv_cnt : 0.f;
    const float dw = training ? (float)ws[2 * c + 1] * inv_cnt : 0.f;
    if (V == 4) {
        const f32x4 dv = *(const f32x4*)(dy + e), xv = *(const f32x4*)(x + e), yv = *(const f32x4*)(y + e);
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float dz = bn_dz(dv[k], yv[k], act, slope);
            o[k] = wr * (dz - db - (xv[k] - mean) * r * dw);
        }
        *(f32x4*)(dx + e) = o;
    } else {
        const float dz = bn_dz(dy[e], y[e], act, slope);
        dx[e] = wr * (dz - db - (x[e] - mean) * r * dw);
    }
}

// The backward counterpart of bn_fwd_small_kernel: one workgroup per channel reduces (sum dz, sum dz * xhat), writes the
// parameter gradients and applies -- instead of memset + reduce + finalize + apply.
template <int V>
__global__ __launch_bounds__(256) void bn_bwd_small_kernel(const float* dy, const float* x, const float* y, const float* stats,
                                                           const float* w, float* dx, float* dweight, float* dbias, int N, int C,
                                                           int HW, int act, float slope, int training, int accumulate) {
    const int c = blockIdx.x;
    const float mean = stats[c], r = stats[C + c];
    double sa = 0.0, sb = 0.0;
    if (V == 4) {
        const BnWalk wk = bn_walk(N, HW);
        const long c0 = (long)c * HW;
        for (int q0 = threadIdx.x; q0 < wk.tot4; q0 += 512) {       // (three operands: two float4 triples in flight)
            f32x4 dv[2], xv[2], yv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (q0 + 256 * u < wk.tot4) {
                    const long off = c0 + bn_off4(wk, q0 + 256 * u, C, HW);
                    dv[u] = *(const f32x4*)(dy + off); xv[u] = *(const f32x4*)(x + off); yv[u] = *(const f32x4*)(y + off);
                }
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (q0 + 256 * u < wk.tot4) {
                    float pa = 0.f, pb = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float dz = bn_dz(dv[u][k], yv[u][k], act, slope);
                        pa += dz;
                        pb += dz * ((xv[u][k] - mean) * r);
                    }
                    sa += (double)pa;
                    sb += (double)pb;
                }
        }
    } else {
        for (int n = 0; n < N; ++n) {
            const long base = ((long)n * C + c) * HW;
            for (int i = threadIdx.x; i < HW; i += 256) {
                const float dz = bn_dz(dy[base + i], y[base + i], act, slope);
                sa += (double)dz;
                sb += (double)dz * (double)((x[base + i] - mean) * r);
            }
        }
    }
    __shared__ double ra[4], rb[4];
    __shared__ float sh[2];
    sa = jaf_wave_sum(sa);
    sb = jaf_wave_sum(sb);
    if ((threadIdx.x & 63) == 0) { ra[threadIdx.x >> 6] = sa; rb[threadIdx.x >> 6] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double a = (ra[0] + ra[1]) + (ra[2] + ra[3]), bsum = (rb[0] + rb[1]) + (rb[2] + rb[3]);
        dbias[c] = (accumulate ? dbias[c] : 0.f) + (float)a;
        dweight[c] = (accumulate ? dweight[c] : 0.f) + (float)bsum;
        const float inv_cnt = 1.0f / ((float)N * (float)HW);
        sh[0] = training ? (float)a * inv_cnt : 0.f;
        sh[1] = training ? (float)bsum * inv_cnt : 0.f;
    }
    __syncthreads();
    const float db = sh[0], dw = sh[1], wr = w[c] * r;
    if (V == 4) {
        const BnWalk wk = bn_walk(N, HW);
        const long c0 = (long)c * HW;
        for (int q0 = threadIdx.x; q0 < wk.tot4; q0 += 512) {
            f32x4 dv[2], xv[2], yv[2];
            long off[2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (q0 + 256 * u < wk.tot4) {
                    off[u] = c0 + bn_off4(wk, q0 + 256 * u, C, HW);
                    dv[u] = *(const f32x4*)(dy + off[u]); xv[u] = *(const f32x4*)(x + off[u]); yv[u] = *(const f32x4*)(y + off[u]);
                }
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (q0 + 256 * u < wk.tot4) {
                    f32x4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float dz = bn_dz(dv[u][k], yv[u][k], act, slope);
                        o[k] = wr * (dz - db - (xv[u][k] - mean) * r * dw);
                    }
                    *(f32x4*)(dx + off[u]) = o;
                }
        }
    } else {
        for (int n = 0; n < N; ++n) {
            const long base = ((long)n * C + c) * HW;
            for (int i = threadIdx.x; i < HW; i += 256) {
                const float dz = bn_dz(dy[base + i], y[base + i], act, slope);
                dx[base + i] = wr * (dz - db - (x[base + i] - mean) * r * dw);
            }
        }
    }
}

// The backward counterpart for `parts` chunks in one launch (see bn_fwd_small_parts_kernel): bit-identical to bn_bwd_small_kernel
// called chunk after chunk with accumulate = 1 from the second chunk on.
template <int V>
__global__ __launch_bounds__(1024) void bn_bwd_small_parts_kernel(const float* dy, const float* x, const float* y, const float* stats,
                                                                  const float* w, float* dx, float* dweight, float* dbias, int n, int C,
                                                                  int HW, int act, float slope, int training, int accumulate, int parts) {
    const int c = blockIdx.x;
    const int part = threadIdx.x >> 8, tid = threadIdx.x & 255;
    const long poff = (long)part * n * C * HW;
    const float mean = stats[(long)part * 2 * C + c], r = stats[(long)part * 2 * C + C + c];
    double sa = 0.0, sb = 0.0;
    if (V == 4) {
        const BnWalk wk = bn_walk(n, HW);
        const long c0 = poff + (long)c * HW;
        for (int q0 = tid; q0 < wk.tot4; q0 += 512) {
            f32x4 dv[2], xv[2], yv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (q0 + 256 * u < wk.tot4) {
                    const long off = c0 + bn_off4(wk, q0 + 256 * u, C, HW);
                    dv[u] = *(const f32x4*)(dy + off); xv[u] = *(const f32x4*)(x + off); yv[u] = *(const f32x4*)(y + off);
                }
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (q0 + 256 * u < wk.tot4) {
                    float pa = 0.f, pb = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float dz = bn_dz(dv[u][k], yv[u][k], act, slope);
                        pa += dz;
                        pb += dz * ((xv[u][k] - mean) * r);
                    }
                    sa += (double)pa;
                    sb += (double)pb;
                }
        }
    } else {
        for (int i = 0; i < n; ++i) {
            const long base = poff + ((long)i * C + c) * HW;
            for (int e = tid; e < HW; e += 256) {
                const float dz = bn_dz(dy[base + e], y[base + e], act, slope);
                sa += (double)dz;
                sb += (double)dz * (double)((x[base + e] - mean) * r);
            }
        }
    }
    __shared__ double ra[4][4], rb[4][4];
    __shared__ float sh[4][2];
    sa = jaf_wave_sum(sa);
    sb = jaf_wave_sum(sb);
    if ((tid & 63) == 0) { ra[part][tid >> 6] = sa; rb[part][tid >> 6] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float db_run = accumulate ? dbias[c] : 0.f, dw_run = accumulate ? dweight[c] : 0.f;
        for (int k = 0; k < parts; ++k) {
            const double a = (ra[k][0] + ra[k][1]) + (ra[k][2] + ra[k][3]), bsum = (rb[k][0] + rb[k][1]) + (rb[k][2] + rb[k][3]);
            db_run = db_run + (float)a;
            dw_run = dw_run + (float)bsum;
            const float inv_cnt = 1.0f / ((float)n * (float)HW);
            sh[k][0] = training ? (float)a * inv_cnt : 0.f;
            sh[k][1] = training ? (float)bsum * inv_cnt : 0.f;
        }
        dbias[c] = db_run;
        dweight[c] = dw_run;
    }
    __syncthreads();
    const float db = sh[part][0], dw = sh[part][1], wr = w[c] * r;
    if (V == 4) {
        const BnWalk wk = bn_walk(n, HW);
        const long c0 = poff + (long)c * HW;
        for (int q0 = tid; q0 < wk.tot4; q0 += 512) {
            f32x4 dv[2], xv[2], yv[2];
            long off[2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (q0 + 256 * u < wk.tot4) {
                    off[u] = c0 + bn_off4(wk, q0 + 256 * u, C, HW);
                    dv[u] = *(const f32x4*)(dy + off[u]); xv[u] = *(const f32x4*)(x + off[u]); yv[u] = *(const f32x4*)(y + off[u]);
                }
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (q0 + 256 * u < wk.tot4) {
                    f32x4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float dz = bn_dz(dv[u][k], yv[u][k], act, slope);
                        o[k] = wr * (dz - db - (xv[u][k] - mean) * r * dw);
                    }
                    *(f32x4*)(dx + off[u]) = o;
                }
        }
    } else {
        for (int i = 0; i < n; ++i) {
            const long base = poff + ((long)i * C + c) * HW;
            for (int e = tid; e < HW; e += 256) {
                const float dz = bn_dz(dy[base + e], y[base + e], act, slope);
                dx[base + e] = wr * (dz - db - (x[base + e] - mean) * r * dw);
            }
        }
    }
}

extern "C" int jaf_batchnorm_act_bwd_split(jaf_stream_t s_, const float* dy, const float* x, const float* y, const float* stats,
                                           const float* weight, float* dx, float* dweight, float* dbias, int32_t N, int32_t C, int32_t HW,
                                           int act, float slope, int training, int accumulate, int32_t parts) {
    JAF_REQUIRE(dy && x && y && stats && weight && dx && dweight && dbias && N >= 1 && C >= 1 && HW >= 1 && C <= 65535);
    JAF_REQUIRE(parts >= 1 && parts <= 4 && N % parts == 0);
    const int n = N / parts;
    if ((long)n * HW > BN_SMALL_MAX) return JAF_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)s_;
    if ((HW % 4 == 0) && al16(dy, x, y, dx))
        hipLaunchKernelGGL(bn_bwd_small_parts_kernel<4>, dim3(C), dim3(256 * parts), 0, s, dy, x, y, stats, weight, dx, dweight, dbias, n, C, HW,
                           act, slope, training, accumulate, parts);
    else
        hipLaunchKernelGGL(bn_bwd_small_parts_kernel<1>, dim3(C), dim3(256 * parts), 0, s, dy, x, y, stats, weight, dx, dweight, dbias, n, C, HW,
                           act, slope, training, accumulate, parts);
    return jaf_launch_status();
}

extern "C" int jaf_batchnorm_act_bwd(jaf_stream_t s_, const float* dy, const float* x, const float* y,
                                     const float* stats, const float* weight, float* dx, float* dweight,
                                     float* dbias, int32_t N, int32_t C, int32_t HW, int act, float slope,
                                     int training, double* workspace, int accumulate) {
    JAF_REQUIRE(dy && x && y && stats && weight && dx && dweight && dbias && workspace && N >= 1 && C >= 1 && HW >= 1);
    JAF_REQUIRE(C <= 65535 && N <= 65535);
    hipStream_t s = (hipStream_t)s_;
    const bool v4 = (HW % 4 == 0) && al16(dy, x, y, dx);
    if ((long)N * HW <= BN_SMALL_MAX) {
        if (v4) hipLaunchKernelGGL(bn_bwd_small_kernel<4>, dim3(C), dim3(256), 0, s, dy, x, y, stats, weight, dx, dweight, dbias, N, C, HW,
                                   act, slope, training, accumulate);
        else hipLaunchKernelGGL(bn_bwd_small_kernel<1>, dim3(C), dim3(256), 0, s, dy, x, y, stats, weight, dx, dweight, dbias, N, C, HW,
                                act, slope, training, accumulate);
        return jaf_launch_status();
    }
    hipError_t e = hipMemsetAsync(workspace, 0, sizeof(double) * 2 * C, s);
    if (e != hipSuccess) return (int)e;
    const int ns = bn_nsplit(C, N, HW);
    if (v4) hipLaunchKernelGGL(bn_bwd_reduce_kernel<4>, dim3(C, ns), dim3(256), 0, s, dy, x, y, stats, workspace, N, C, HW, act, slope);
    else hipLaunchKernelGGL(bn_bwd_reduce_kernel<1>, dim3(C, ns), dim3(256), 0, s, dy, x, y, stats, workspace, N, C, HW, act, slope);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(jaf_cdiv(C, 64)), dim3(64), 0, s, workspace, dweight, dbias, C, accumulate);
    if (v4)
        hipLaunchKernelGGL(bn_bwd_apply_kernel<4>, dim3(jaf_cdiv(HW / 4, 256), C, N), dim3(256), 0, s, dy, x, y, stats, weight,
                           workspace, dx, N, C, HW, act, slope, training);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, dim3(jaf_cdiv(HW, 256), C, N), dim3(256), 0, s, dy, x, y, stats, weight,
                           workspace, dx, N, C, HW, act, slope, training);
    return jaf_launch_status();
}
