// Classifier heads of the discriminators (src/networks.py:408-410,447-449): Linear 4096->100,
// 2048->100, 100->1 on a handful of samples.  ~1 MFLOP: one wave per output element.
#include "jaf_common.h"

__global__ void linear_fwd_kernel(const float* x, const float* w, const float* b, float* y, int N, int I, int O,
                                  int act, float slope) {
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (wave >= N * O) return;
    const int n = wave / O, o = wave % O;
    const float* xp = x + (long)n * I;
    const float* wp = w + (long)o * I;
    float acc = 0.f;
    if ((I & 255) == 0 && ((((uintptr_t)x) | ((uintptr_t)w)) & 15) == 0) {
        // 16-byte loads, four independent partial sums (the 4096 -> 100 head of the image discriminator: 16 float4 pairs per lane in
        // flight instead of 64 dependent 4-byte loads; this launch sits in the discriminator phase's dependent chain: 21 -> 5 us)
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        const f32x4* x4 = (const f32x4*)xp;
        const f32x4* w4 = (const f32x4*)wp;
        f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
        for (int i = lane; i < (I >> 2); i += 64) {
            const f32x4 xv = x4[i], wv = w4[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) a4[k] = fmaf(xv[k], wv[k], a4[k]);
        }
        acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
    } else {
        for (int i = lane; i < I; i += 64) acc = fmaf(xp[i], wp[i], acc);
    }
    acc = jaf_wave_sum(acc);
    if (lane == 0) y[wave] = jaf_act(acc + (b ? b[o] : 0.f), act, slope);
}

extern "C" int jaf_linear_fwd(jaf_stream_t s, const float* x, const float* w, const float* b, float* y, int32_t N,
                              int32_t I, int32_t O, int act, float slope) {
    JAF_REQUIRE(x && w && y && N >= 1 && I >= 1 && O >= 1);
    const long waves = (long)N * O;
    hipLaunchKernelGGL(linear_fwd_kernel, dim3(jaf_cdiv(waves, 4)), dim3(256), 0, (hipStream_t)s, x, w, b, y, N, I, O, act, slope);
    return jaf_launch_status();
}

// dx[n,i] = sum_o dz[n,o] W[o,i];  dW[o,i] = sum_n dz[n,o] x[n,i];  db[o] = sum_n dz[n,o]
__global__ void linear_bwd_kernel(const float* dz, const float* x, const float* w, float* dx, float* dw, float* db,
                                  int N, int I, int O) {
    const long gs = (long)gridDim.x * blockDim.x;
    const long t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (dx) {
        for (long e = t0; e < (long)N * I; e += gs) {
            const int n = (int)(e / I), i = (int)(e % I);
            float acc = 0.f;
            for (int o = 0; o < O; ++o) acc = fmaf(dz[n * O + o], w[(long)o * I + i], acc);
            dx[e] = acc;
        }
    }
    for (long e = t0; e < (long)O * I; e += gs) {
        const int o = (int)(e / I), i = (int)(e % I);
        float acc = 0.f;
        for (int n = 0; n < N; ++n) acc = fmaf(dz[n * O + o], x[(long)n * I + i], acc);
        dw[e] = acc;
    }
    for (long e = t0; e < O; e += gs) {
        float acc = 0.f;
        for (int n = 0; n < N; ++n) acc += dz[n * O + e];
        db[e] = acc;
    }
}

extern "C" int jaf_linear_bwd(jaf_stream_t s, const float* dz, const float* x, const float* w, float* dx, float* dw,
                              float* db, int32_t N, int32_t I, int32_t O) {
    JAF_REQUIRE(dz && x && w && dw && db && N >= 1 && I >= 1 && O >= 1);
    long work = (long)O * I;
    if ((long)N * I > work) work = (long)N * I;
    hipLaunchKernelGGL(linear_bwd_kernel, dim3(jaf_ew_grid(work)), dim3(256), 0, (hipStream_t)s, dz, x, w, dx, dw, db, N, I, O);
    return jaf_launch_status();
}

// The same with the activation backward folded in (dz = dy * act'(y), recomputed where it is used: N x O is a few hundred values) and
// the parameter gradients ADDED to their buffers when `accumulate`: one launch instead of act_bwd + linear_bwd + two accumulation adds per
// classifier layer of the discriminators' dependent chain.  Same arithmetic: dz as jaf_act_bwd makes it, one add per gradient element.
__device__ __forceinline__ float lin_act_grad(float yv, int act, float slope) {
    switch (act) {
        case JAF_ACT_LRELU: return yv > 0.f ? 1.f : slope;
        case JAF_ACT_RELU: return yv > 0.f ? 1.f : 0.f;
        case JAF_ACT_SIGMOID: return yv * (1.f - yv);
        case JAF_ACT_TANH: return 1.f - yv * yv;
        default: return 1.f;
    }
}

__global__ void linear_bwd_fused_kernel(const float* dy, const float* y, const float* x, const float* w, float* dx, float* dw, float* db,
                                        int N, int I, int O, int act, float slope, int accumulate) {
#pragma clang fp contract(off)      // dz = dy * act'(y) is rounded as jaf_act_bwd stores it (HIP's __fmul_rn is a plain product: it would fuse into the sums)
    const long gs = (long)gridDim.x * blockDim.x;
    const long t0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (dx) {
        for (long e = t0; e < (long)N * I; e += gs) {
            const int n = (int)(e / I), i = (int)(e % I);
            float acc = 0.f;
            for (int o = 0; o < O; ++o) acc = fmaf(__fmul_rn(dy[n * O + o], lin_act_grad(y[n * O + o], act, slope)), w[(long)o * I + i], acc);
            dx[e] = acc;
        }
    }
    for (long e = t0; e < (long)O * I; e += gs) {
        const int o = (int)(e / I), i = (int)(e % I);
        float acc = 0.f;
        for (int n = 0; n < N; ++n) acc = fmaf(__fmul_rn(dy[n * O + o], lin_act_grad(y[n * O + o], act, slope)), x[(long)n * I + i], acc);
        dw[e] = accumulate ? dw[e] + acc : acc;
    }
    for (long e = t0; e < O; e += gs) {
        float acc = 0.f;
        for (int n = 0; n < N; ++n) acc += __fmul_rn(dy[n * O + e], lin_act_grad(y[n * O + e], act, slope));      // (dz rounded as jaf_act_bwd stores it: no contraction into the sum)
        db[e] = accumulate ? db[e] + acc : acc;
    }
}

extern "C" int jaf_linear_bwd_fused(jaf_stream_t s, const float* dy, const float* y, const float* x, const float* w, float* dx, float* dw,
                                    float* db, int32_t N, int32_t I, int32_t O, int act, float slope, int accumulate) {
    JAF_REQUIRE(dy && y && x && w && dw && db && N >= 1 && I >= 1 && O >= 1);
    long work = (long)O * I;
    if ((long)N * I > work) work = (long)N * I;
    hipLaunchKernelGGL(linear_bwd_fused_kernel, dim3(jaf_ew_grid(work)), dim3(256), 0, (hipStream_t)s, dy, y, x, w, dx, dw, db, N, I, O, act,
                       slope, accumulate);
    return jaf_launch_status();
}
