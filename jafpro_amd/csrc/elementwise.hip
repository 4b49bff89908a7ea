// HBM-bound elementwise kernels of the stage-4 step: blends, masks, activation backward,
// ConvLSTM gate backward, losses, Adam.  One pass over each operand, 16 B per lane where the
// layout allows it, grid capped at 2048 workgroups and strided (cdna_hip_programming.md G11/G13).
#include "jaf_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ activation backward
__device__ __forceinline__ float act_grad(float yv, int act, float slope) {
    switch (act) {
        case JAF_ACT_LRELU: return yv > 0.f ? 1.f : slope;
        case JAF_ACT_RELU: return yv > 0.f ? 1.f : 0.f;
        case JAF_ACT_SIGMOID: return yv * (1.f - yv);
        case JAF_ACT_TANH: return 1.f - yv * yv;
        default: return 1.f;
    }
}

__global__ void act_bwd_kernel(const float* dy, const float* y, float* dz, long n, int act, float slope) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dz[i] = dy[i] * act_grad(y[i], act, slope);
}

// 16 bytes per lane (cdna_hip_programming.md Guideline 13): the vector memory pipe works per
// instruction, a dword-per-lane stream reaches a fraction of the HBM rate.
__global__ void act_bwd_kernel4(const f32x4* dy, const f32x4* y, f32x4* dz, long n4, int act, float slope) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4 g = dy[i], yv = y[i];
        f32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = g[k] * act_grad(yv[k], act, slope);
        dz[i] = o;
    }
}

static inline bool jaf_al16(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr) {
    return ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)c) | ((uintptr_t)d)) & 15) == 0;
}

extern "C" int jaf_act_bwd(jaf_stream_t s, const float* dy, const float* y, float* dz, int64_t n, int act, float slope) {
    JAF_REQUIRE(dy && y && dz && n >= 0);
    if (n == 0) return JAF_OK;
    if ((n & 3) == 0 && jaf_al16(dy, y, dz))
        hipLaunchKernelGGL(act_bwd_kernel4, dim3(jaf_ew_grid(n >> 2)), dim3(256), 0, (hipStream_t)s, (const f32x4*)dy,
                           (const f32x4*)y, (f32x4*)dz, (long)(n >> 2), act, slope);
    else
        hipLaunchKernelGGL(act_bwd_kernel, dim3(jaf_ew_grid(n)), dim3(256), 0, (hipStream_t)s, dy, y, dz, (long)n, act, slope);
    return jaf_launch_status();
}

// ------------------------------------------------------------------ ConvLSTM gate backward
// gates: [N, G*4C, HW] (i,f,o,g per group), h/c tensors: [N, G*C, HW]  (src/convLSTM.py:48-54)
// grid: (pixel blocks, C, N*G); V = 4 pixels per lane when HW % 4 == 0.
template <int V>
__global__ void lstm_gates_bwd_kernel(int C, int HW, const float* dh, const float* dc_next, float* gates,
                                      const float* c_prev, const float* c_cur, float* dc_prev) {
    typedef float fv __attribute__((ext_vector_type(V == 1 ? 2 : V)));
    const int c = blockIdx.y;
    const long ng = blockIdx.z;
    const int pix = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    if (pix >= HW) return;
    const long e = (ng * C + c) * (long)HW + pix;
    float* gp = gates + (ng * 4 * C + c) * (long)HW + pix;
    const long cs = (long)C * HW;
    float gi[V], gf[V], go[V], gg[V], cc[V], dhv[V], dcn[V], cp[V];
    if (V == 4) {
        *(f32x4*)gi = *(const f32x4*)gp; *(f32x4*)gf = *(const f32x4*)(gp + cs);
        *(f32x4*)go = *(const f32x4*)(gp + 2 * cs); *(f32x4*)gg = *(const f32x4*)(gp + 3 * cs);
        *(f32x4*)cc = *(const f32x4*)(c_cur + e); *(f32x4*)dhv = *(const f32x4*)(dh + e);
        if (dc_next) *(f32x4*)dcn = *(const f32x4*)(dc_next + e);
        if (c_prev) *(f32x4*)cp = *(const f32x4*)(c_prev + e);
    } else {
        gi[0] = gp[0]; gf[0] = gp[cs]; go[0] = gp[2 * cs]; gg[0] = gp[3 * cs];
        cc[0] = c_cur[e]; dhv[0] = dh[e];
        if (dc_next) dcn[0] = dc_next[e];
        if (c_prev) cp[0] = c_prev[e];
    }
    float o0[V], o1[V], o2[V], o3[V], o4[V];
#pragma unroll
    for (int k = 0; k < V; ++k) {
        const float tc = jaf_tanh(cc[k]);
        float dc = dhv[k] * go[k] * (1.f - tc * tc);
        if (dc_next) dc += dcn[k];
        const float cpv = c_prev ? cp[k] : 0.f;
        o0[k] = dc * gg[k] * gi[k] * (1.f - gi[k]);
        o1[k] = dc * cpv * gf[k] * (1.f - gf[k]);
        o2[k] = dhv[k] * tc * go[k] * (1.f - go[k]);
        o3[k] = dc * gi[k] * (1.f - gg[k] * gg[k]);
        o4[k] = dc * gf[k];
    }
    if (V == 4) {
        *(f32x4*)gp = *(f32x4*)o0; *(f32x4*)(gp + cs) = *(f32x4*)o1;
        *(f32x4*)(gp + 2 * cs) = *(f32x4*)o2; *(f32x4*)(gp + 3 * cs) = *(f32x4*)o3;
        *(f32x4*)(dc_prev + e) = *(f32x4*)o4;
    } else {
        gp[0] = o0[0]; gp[cs] = o1[0]; gp[2 * cs] = o2[0]; gp[3 * cs] = o3[0]; dc_prev[e] = o4[0];
    }
}

extern "C" int jaf_convlstm_gates_bwd(jaf_stream_t s, int32_t N, int32_t G, int32_t C, int32_t HW,
                                      const float* dh, const float* dc_next, float* gates,
                                      const float* c_prev, const float* c_cur, float* dc_prev) {
    JAF_REQUIRE(dh && gates && c_cur && dc_prev && N >= 1 && G >= 1 && C >= 1 && HW >= 1);
    JAF_REQUIRE(C <= 65535 && (long)N * G <= 65535);
    const bool v4 = (HW % 4 == 0) && jaf_al16(dh, gates, c_cur, dc_prev) && jaf_al16(dc_next, c_prev);
    if (v4)
        hipLaunchKernelGGL(lstm_gates_bwd_kernel<4>, dim3(jaf_cdiv(HW / 4, 256), C, N * G), dim3(256), 0, (hipStream_t)s,
                           C, HW, dh, dc_next, gates, c_prev, c_cur, dc_prev);
    else
        hipLaunchKernelGGL(lstm_gates_bwd_kernel<1>, dim3(jaf_cdiv(HW, 256), C, N * G), dim3(256), 0, (hipStream_t)s,
                           C, HW, dh, dc_next, gates, c_prev, c_cur, dc_prev);
    return jaf_launch_status();
}

// ------------------------------------------------------------------ blends
// out = a*m + b*(1-m), m [N,1,HW] broadcast over C.
__global__ void blend_fwd_kernel(const float* a, const float* b, const float* m, float* out, int N, int C, int HW) {
    const long total = (long)N * C * HW;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int pix = (int)(e % HW);
        const long n = e / ((long)C * HW);
        const float mv = m[n * HW + pix];
        out[e] = a[e] * mv + b[e] * (1.f - mv);
    }
}

extern "C" int jaf_blend_fwd(jaf_stream_t s, const float* a, const float* b, const float* m, float* out,
                             int32_t N, int32_t C, int32_t HW) {
    JAF_REQUIRE(a && b && m && out && N >= 1 && C >= 1 && HW >= 1);
    hipLaunchKernelGGL(blend_fwd_kernel, dim3(jaf_ew_grid((long)N * C * HW)), dim3(256), 0, (hipStream_t)s, a, b, m, out, N, C, HW);
    return jaf_launch_status();
}

// one thread per (n, pixel): loops the C channels so dm needs no atomics
__global__ void blend_bwd_kernel(const float* dout, const float* a, const float* b, const float* m,
                                 float* da, float* db, float* dm, int N, int C, int HW) {
    const long total = (long)N * HW;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int pix = (int)(e % HW);
        const long n = e / HW;
        const float mv = m[e];
        float acc = 0.f;
        for (int c = 0; c < C; ++c) {
            const long i = (n * C + c) * HW + pix;
            const float g = dout[i];
            if (da) da[i] = g * mv;
            if (db) db[i] = g * (1.f - mv);
            acc += g * (a[i] - b[i]);
        }
        if (dm) dm[e] = acc;
    }
}

extern "C" int jaf_blend_bwd(jaf_stream_t s, const float* dout, const float* a, const float* b, const float* m,
                             float* da, float* db, float* dm, int32_t N, int32_t C, int32_t HW) {
    JAF_REQUIRE(dout && a && b && m && N >= 1 && C >= 1 && HW >= 1);
    hipLaunchKernelGGL(blend_bwd_kernel, dim3(jaf_ew_grid((long)N * HW)), dim3(256), 0, (hipStream_t)s, dout, a, b, m, da, db, dm, N, C, HW);
    return jaf_launch_status();
}

__global__ void mul_bcast_kernel(const float* x, const float* m, float* out, int N, int C, int MC, int HW) {
    const long total = (long)N * C * HW;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int pix = (int)(e % HW);
        const long nc = e / HW;
        const int c = (int)(nc % C);
        const long n = nc / C;
        const float mv = (MC == 1) ? m[n * HW + pix] : m[(n * MC + c) * HW + pix];
        out[e] = x[e] * mv;
    }
}

extern "C" int jaf_mul_bcast(jaf_stream_t s, const float* x, const float* m, float* out, int32_t N, int32_t C,
                             int32_t MC, int32_t HW) {
    JAF_REQUIRE(x && m && out && N >= 1 && C >= 1 && HW >= 1 && (MC == 1 || MC == C));
    hipLaunchKernelGGL(mul_bcast_kernel, dim3(jaf_ew_grid((long)N * C * HW)), dim3(256), 0, (hipStream_t)s, x, m, out, N, C, MC, HW);
    return jaf_launch_status();
}

// ------------------------------------------------------------------ texture-atlas helpers
// parts tensor: [B*T? , P*3, PSZ, PSZ]; atlas [B, T, 3, AH, AW]; part p sits at rows (p/6)*PSZ,
// cols (p%6)*PSZ (train/4...py:269-276).  Output image index = t*B + b (the reference
// concatenates the T references on the batch axis, src/networks.py:1317).
// grid (x blocks, row blocks, B*P*3 planes of reference frame t), block (tx, ty); V floats per lane along x.  The plane index is decoded
// once per workgroup with scalar arithmetic (the flat one-thread-per-element form spent six 64-bit divisions per
// element and ran at 1.8 TB/s).
template <int V>
__global__ void atlas_to_parts_kernel(const float* __restrict__ atlas, float* __restrict__ parts, int B, int T, int t, int AH, int AW, int PSZ) {
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= PSZ || y >= PSZ) return;
    const int pcols = AW / PSZ;
    const int P = (AH / PSZ) * pcols;
    int z = blockIdx.z;                       // (b*P + p)*3 + c
    const int c = z % 3; z /= 3;
    const int p = z % P;
    const int b = z / P;
    const int ay = (p / pcols) * PSZ + y;
    const int ax = (p % pcols) * PSZ + x;
    const float* src = atlas + ((((long)b * T + t) * 3 + c) * AH + ay) * AW + ax;
    float* dst = parts + ((((long)t * B * P * 3) + blockIdx.z) * PSZ + y) * PSZ + x;
    if (V == 4) *(f32x4*)dst = *(const f32x4*)src;
    else dst[0] = src[0];
}

extern "C" int jaf_atlas_to_parts(jaf_stream_t s, const float* atlas, float* parts, int32_t B, int32_t T,
                                  int32_t AH, int32_t AW, int32_t PSZ) {
    JAF_REQUIRE(atlas && parts && B >= 1 && T >= 1 && PSZ >= 1 && AH % PSZ == 0 && AW % PSZ == 0);
    const long planes = (long)B * (AH / PSZ) * (AW / PSZ) * 3;       // per reference frame
    if (planes > 65535) return JAF_EUNSUPPORTED;
    const bool v4 = (PSZ % 4 == 0) && (AW % 4 == 0) && ((((uintptr_t)atlas) | ((uintptr_t)parts)) & 15) == 0;
    const int wx = v4 ? PSZ / 4 : PSZ;
    int tx = 64;
    while (tx > 8 && (tx >> 1) >= wx) tx >>= 1;
    const int ty = 256 / tx;
    const dim3 grid(jaf_cdiv(wx, tx), jaf_cdiv(PSZ, ty), (unsigned)planes);
    for (int t = 0; t < T; ++t) {
        if (v4) hipLaunchKernelGGL(atlas_to_parts_kernel<4>, grid, dim3(tx, ty), 0, (hipStream_t)s, atlas, parts, B, T, t, AH, AW, PSZ);
        else hipLaunchKernelGGL(atlas_to_parts_kernel<1>, grid, dim3(tx, ty), 0, (hipStream_t)s, atlas, parts, B, T, t, AH, AW, PSZ);
    }
    return jaf_launch_status();
}

// The same slicing straight into the packed bf16 image the first part-encoder convolution reads (csrc/conv_dma.hip layout
// [image t*B+b][part][1 channel group][y*PSZ+x][8 channels], channels 3..7 zero; split: the residual plane behind the hi plane):
// the fp32 parts tensor (369 MB at B=8, T=4) is neither written nor read back by a packing pass.
// grid (x blocks of 4 pixels, rows, B*P) per reference frame.
__global__ void atlas_to_parts_packed_kernel(const float* __restrict__ atlas, unsigned char* __restrict__ img, int B, int T, int t, int AH,
                                             int AW, int PSZ, int split) {
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= PSZ || y >= PSZ) return;
    const int pcols = AW / PSZ;
    const int P = (AH / PSZ) * pcols;
    const int p = blockIdx.z % P;
    const int b = blockIdx.z / P;
    const int ay = (p / pcols) * PSZ + y;
    const int ax = (p % pcols) * PSZ + x;
    const float* src = atlas + ((((long)b * T + t) * 3) * AH + ay) * AW + ax;
    const long plane = (long)AH * AW;
    const f32x4 c0 = *(const f32x4*)src, c1 = *(const f32x4*)(src + plane), c2 = *(const f32x4*)(src + 2 * plane);
    const long HW = (long)PSZ * PSZ;
    unsigned char* dst = img + ((((long)t * B + b) * P + p) * (split ? 2 : 1)) * HW * 16 + ((long)y * PSZ + x) * 16;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const __bf16 h0 = (__bf16)c0[k], h1 = (__bf16)c1[k], h2 = (__bf16)c2[k];
        const unsigned int w0 = (unsigned int)__builtin_bit_cast(unsigned short, h0) | ((unsigned int)__builtin_bit_cast(unsigned short, h1) << 16);
        const unsigned int w1 = (unsigned int)__builtin_bit_cast(unsigned short, h2);
        *(jaf_u32x4*)(dst + k * 16) = (jaf_u32x4){w0, w1, 0u, 0u};
        if (split) {
            const __bf16 l0 = (__bf16)(c0[k] - (float)h0), l1 = (__bf16)(c1[k] - (float)h1), l2 = (__bf16)(c2[k] - (float)h2);
            const unsigned int v0 = (unsigned int)__builtin_bit_cast(unsigned short, l0) | ((unsigned int)__builtin_bit_cast(unsigned short, l1) << 16);
            const unsigned int v1 = (unsigned int)__builtin_bit_cast(unsigned short, l2);
            *(jaf_u32x4*)(dst + HW * 16 + k * 16) = (jaf_u32x4){v0, v1, 0u, 0u};
        }
    }
}

extern "C" int jaf_atlas_to_parts_packed(jaf_stream_t s, const float* atlas, void* image, int32_t B, int32_t T, int32_t AH, int32_t AW,
                                         int32_t PSZ, int32_t precision) {
    JAF_REQUIRE(atlas && image && B >= 1 && T >= 1 && PSZ >= 1 && AH % PSZ == 0 && AW % PSZ == 0);
    JAF_REQUIRE(precision == JAF_PREC_BF16 || precision == JAF_PREC_BF16X3);
    const long bp = (long)B * (AH / PSZ) * (AW / PSZ);
    if (bp > 65535) return JAF_EUNSUPPORTED;
    if (PSZ % 4 || AW % 4 || (((uintptr_t)atlas) & 15) || (((uintptr_t)image) & 15)) return JAF_EUNSUPPORTED;
    const int wx = PSZ / 4;
    int tx = 64;
    while (tx > 8 && (tx >> 1) >= wx) tx >>= 1;
    const int ty = 256 / tx;
    const dim3 grid(jaf_cdiv(wx, tx), jaf_cdiv(PSZ, ty), (unsigned)bp);
    for (int t = 0; t < T; ++t)
        hipLaunchKernelGGL(atlas_to_parts_packed_kernel, grid, dim3(tx, ty), 0, (hipStream_t)s, atlas, (unsigned char*)image, B, T, t, AH, AW,
                           PSZ, precision == JAF_PREC_BF16X3 ? 1 : 0);
    return jaf_launch_status();
}

// out[b, 3p+c, y, x] = tex[...] * (OR_t used[t] && masks[b,t,ay,ax] != 0)
// grid (x blocks, row blocks, B*P), block (tx, ty): no per-element index decoding
__global__ void part_mask_mul_kernel(const float* __restrict__ tex, const float* __restrict__ masks, const int* __restrict__ used,
                                     float* __restrict__ out, int B, int T, int AH, int AW, int P, int PSZ) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= PSZ || y >= PSZ) return;
    const int pcols = AW / PSZ;
    const int p = blockIdx.z % P;
    const int b = blockIdx.z / P;
    const int ay = (p / pcols) * PSZ + y;
    const int ax = (p % pcols) * PSZ + x;
    bool on = false;
    for (int t = 0; t < T; ++t)
        on = on || (used[t] && ((unsigned char)masks[(((long)b * T + t) * AH + ay) * AW + ax] != 0));
    const float mv = on ? 1.f : 0.f;
    for (int c = 0; c < 3; ++c) {
        const long i = ((((long)b * P + p) * 3 + c) * PSZ + y) * PSZ + x;
        out[i] = tex[i] * mv;
    }
}

extern "C" int jaf_part_mask_mul(jaf_stream_t s, const float* tex, const float* masks, const int32_t* used,
                                 float* out, int32_t B, int32_t T, int32_t AH, int32_t AW, int32_t P, int32_t PSZ) {
    JAF_REQUIRE(tex && masks && used && out && B >= 1 && T >= 1 && PSZ >= 1);
    JAF_REQUIRE(AH % PSZ == 0 && AW % PSZ == 0 && (AH / PSZ) * (AW / PSZ) == P);
    if ((long)B * P > 65535) return JAF_EUNSUPPORTED;
    int tx = 64;
    while (tx > 8 && (tx >> 1) >= PSZ) tx >>= 1;
    const int ty = 256 / tx;
    hipLaunchKernelGGL(part_mask_mul_kernel, dim3(jaf_cdiv(PSZ, tx), jaf_cdiv(PSZ, ty), (unsigned)(B * P)), dim3(tx, ty), 0,
                       (hipStream_t)s, tex, masks, used, out, B, T, AH, AW, P, PSZ);
    return jaf_launch_status();
}

// ------------------------------------------------------------------ losses
__global__ void vgg_preprocess_kernel(const float* x, float* y, int N, int HW) {
    const long total = (long)N * 3 * HW;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += stride) {
        const int c = (int)((e / HW) % 3);
        const float mean = c == 0 ? 103.939f : (c == 1 ? 116.779f : 123.68f);
        y[e] = 255.0f * (x[e] + 1.0f) / 2.0f - mean;
    }
}

extern "C" int jaf_vgg_preprocess(jaf_stream_t s, const float* x, float* y, int32_t N, int32_t HW) {
    JAF_REQUIRE(x && y && N >= 1 && HW >= 1);
    hipLaunchKernelGGL(vgg_preprocess_kernel, dim3(jaf_ew_grid((long)N * 3 * HW)), dim3(256), 0, (hipStream_t)s, x, y, N, HW);
    return jaf_launch_status();
}

__global__ void l1_loss_fwd_kernel(const float* a, const float* b, long n, float scale, float* loss) {
    double acc = 0.0;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += (double)fabsf(a[i] - b[i]);
    __shared__ double red[4];
    acc = jaf_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, (float)((red[0] + red[1] + red[2] + red[3]) * (double)scale));
}

extern "C" int jaf_l1_loss_fwd(jaf_stream_t s, const float* a, const float* b, int64_t n, float w, float* loss_accum) {
    JAF_REQUIRE(a && b && loss_accum && n >= 1);
    int grid = jaf_ew_grid(n, 4);
    if (grid > 512) grid = 512;
    hipLaunchKernelGGL(l1_loss_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)s, a, b, (long)n, w / (float)n, loss_accum);
    return jaf_launch_status();
}

__global__ void l1_loss_bwd_kernel(const float* a, const float* b, long n, float scale, const float* dloss,
                                   float* da, int accumulate) {
    const float g = scale * dloss[0];
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float dlt = a[i] - b[i];
        const float sg = dlt > 0.f ? 1.f : (dlt < 0.f ? -1.f : 0.f);
        da[i] = (accumulate ? da[i] : 0.f) + sg * g;
    }
}

extern "C" int jaf_l1_loss_bwd(jaf_stream_t s, const float* a, const float* b, int64_t n, float w,
                               const float* dloss, float* da, int accumulate) {
    JAF_REQUIRE(a && b && dloss && da && n >= 1);
    hipLaunchKernelGGL(l1_loss_bwd_kernel, dim3(jaf_ew_grid(n)), dim3(256), 0, (hipStream_t)s, a, b, (long)n, w / (float)n, dloss, da, accumulate);
    return jaf_launch_status();
}

__global__ void bce_fwd_kernel(const float* p, int n, float target, float* loss) {
    float acc = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float lp = fmaxf(__logf(p[i]), -100.f);
        const float lq = fmaxf(__logf(1.f - p[i]), -100.f);
        acc += -(target * lp + (1.f - target) * lq);
    }
    acc = jaf_wave_sum(acc);
    if (threadIdx.x == 0) loss[0] = acc / (float)n;
}

extern "C" int jaf_bce_fwd(jaf_stream_t s, const float* p, int32_t n, float target, float* loss) {
    JAF_REQUIRE(p && loss && n >= 1);
    hipLaunchKernelGGL(bce_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, p, n, target, loss);
    return jaf_launch_status();
}

__global__ void bce_bwd_kernel(const float* p, int n, float target, const float* dloss, float* dp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // d/dp [-(t log p + (1-t) log(1-p))], with torch's clamp: zero gradient where the log is clamped
    const float pv = p[i];
    float g = 0.f;
    if (__logf(pv) > -100.f) g += -target / pv;
    if (__logf(1.f - pv) > -100.f) g += (1.f - target) / (1.f - pv);
    dp[i] = g * dloss[0] / (float)n;
}

extern "C" int jaf_bce_bwd(jaf_stream_t s, const float* p, int32_t n, float target, const float* dloss, float* dp) {
    JAF_REQUIRE(p && dloss && dp && n >= 1);
    hipLaunchKernelGGL(bce_bwd_kernel, dim3(jaf_cdiv(n, 64)), dim3(64), 0, (hipStream_t)s, p, n, target, dloss, dp);
    return jaf_launch_status();
}

// Two BCE terms of one probability vector -- the first n1 entries against t1, the rest against t2 (the discriminators' real / generated
// halves of one batched pass, train/4...py:380-394) -- and their sum, in one launch; out = (term 1, term 2, sum).  One wave per term, the
// arithmetic of bce_fwd_kernel.
__global__ void bce_pair_fwd_kernel(const float* p, int n1, int n, float t1, float t2, float* o1, float* o2, float* osum) {
    __shared__ float part[2];
    const int which = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int lo = which ? n1 : 0, hi = which ? n : n1;
    const float target = which ? t2 : t1;
    float acc = 0.f;
    for (int i = lo + lane; i < hi; i += 64) {
        const float lp = fmaxf(__logf(p[i]), -100.f);
        const float lq = fmaxf(__logf(1.f - p[i]), -100.f);
        acc += -(target * lp + (1.f - target) * lq);
    }
    acc = jaf_wave_sum(acc);
    if (lane == 0) part[which] = acc / (float)(hi - lo);
    __syncthreads();
    if (threadIdx.x == 0) { o1[0] = part[0]; o2[0] = part[1]; osum[0] = part[0] + part[1]; }
}

extern "C" int jaf_bce_pair_fwd(jaf_stream_t s, const float* p, int32_t n1, int32_t n, float t1, float t2, float* loss1, float* loss2,
                                float* loss_sum) {
    JAF_REQUIRE(p && loss1 && loss2 && loss_sum && n1 >= 1 && n > n1);
    hipLaunchKernelGGL(bce_pair_fwd_kernel, dim3(1), dim3(128), 0, (hipStream_t)s, p, n1, n, t1, t2, loss1, loss2, loss_sum);
    return jaf_launch_status();
}

// d(g1 term1 + g2 term2 + gs (term1 + term2)) / dp; g1 / g2 / gs nullable (absent = 0), device scalars.
__global__ void bce_pair_bwd_kernel(const float* p, int n1, int n, float t1, float t2, const float* g1, const float* g2, const float* gs,
                                    float* dp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool first = i < n1;
    const float target = first ? t1 : t2;
    const float* gp = first ? g1 : g2;
    const float dl = (gp ? gp[0] : 0.f) + (gs ? gs[0] : 0.f);
    const float pv = p[i];
    float g = 0.f;
    if (__logf(pv) > -100.f) g += -target / pv;
    if (__logf(1.f - pv) > -100.f) g += (1.f - target) / (1.f - pv);
    dp[i] = g * dl / (float)(first ? n1 : n - n1);
}

extern "C" int jaf_bce_pair_bwd(jaf_stream_t s, const float* p, int32_t n1, int32_t n, float t1, float t2, const float* g1, const float* g2,
                                const float* gsum, float* dp) {
    JAF_REQUIRE(p && dp && n1 >= 1 && n > n1 && (g1 || g2 || gsum));
    hipLaunchKernelGGL(bce_pair_bwd_kernel, dim3(jaf_cdiv(n, 64)), dim3(64), 0, (hipStream_t)s, p, n1, n, t1, t2, g1, g2, gsum, dp);
    return jaf_launch_status();
}

// ------------------------------------------------------------------ optimiser
__global__ void adam_kernel(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2,
                            float eps, float bc1, float bc2_sqrt) {
    const long stride = (long)gridDim.x * blockDim.x;
    const long n4 = n >> 2;
    f32x4* p4 = (f32x4*)p; const f32x4* g4 = (const f32x4*)g; f32x4* m4 = (f32x4*)m; f32x4* v4 = (f32x4*)v;
    const float step_size = lr / bc1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 pv = p4[i], gv = g4[i], mv = m4[i], vv = v4[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mv[k] = mv[k] + (gv[k] - mv[k]) * (1.f - b1);       // torch: exp_avg.lerp_(grad, 1-beta1)
            vv[k] = vv[k] * b2 + (1.f - b2) * gv[k] * gv[k];
            const float denom = sqrtf(vv[k]) / bc2_sqrt + eps;
            pv[k] = pv[k] - step_size * (mv[k] / denom);
        }
        p4[i] = pv; m4[i] = mv; v4[i] = vv;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float mv = m[i] + (g[i] - m[i]) * (1.f - b1);
        float vv = v[i] * b2 + (1.f - b2) * g[i] * g[i];
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mv / denom);
        m[i] = mv; v[i] = vv;
    }
}

extern "C" int jaf_adam_step(jaf_stream_t s, float* p, const float* g, float* m, float* v, int64_t n,
                             float lr, float beta1, float beta2, float eps, int32_t step) {
    JAF_REQUIRE(p && g && m && v && n >= 1 && step >= 1);
    JAF_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    // bias corrections in double, as torch.optim.Adam takes them (Python floats), rounded once; adam_tick_kernel evaluates
    // the same two expressions on the device, so the eager and the graph-replayed step use the same numbers
    const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    hipLaunchKernelGGL(adam_kernel, dim3(jaf_ew_grid(n, 4)), dim3(256), 0, (hipStream_t)s, p, g, m, v, (long)n, lr,
                       beta1, beta2, eps, bc1, bc2_sqrt);
    return jaf_launch_status();
}

// The same update with the step count held ON THE DEVICE (state[0] as int32: steps taken so far; state[1], state[2]: the
// bias corrections of the step being taken): a captured hipGraph replays the launch with unchanged arguments, so the
// count cannot be a kernel argument there.  adam_tick advances the count and derives the corrections, adam_dev_kernel
// reads them.
__global__ void adam_tick_kernel(float* state, float b1, float b2) {
    int* cnt = (int*)state;
    const int step = cnt[0] + 1;
    cnt[0] = step;
    state[1] = (float)(1.0 - pow((double)b1, (double)step));           // the expressions of jaf_adam_step, in double
    state[2] = (float)sqrt(1.0 - pow((double)b2, (double)step));
}

__global__ void adam_dev_kernel(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                                const float* state) {
    const float bc1 = state[1], bc2_sqrt = state[2];
    const long stride = (long)gridDim.x * blockDim.x;
    const long n4 = n >> 2;
    f32x4* p4 = (f32x4*)p; const f32x4* g4 = (const f32x4*)g; f32x4* m4 = (f32x4*)m; f32x4* v4 = (f32x4*)v;
    const float step_size = lr / bc1;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f32x4 pv = p4[i], gv = g4[i], mv = m4[i], vv = v4[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            mv[k] = mv[k] + (gv[k] - mv[k]) * (1.f - b1);
            vv[k] = vv[k] * b2 + (1.f - b2) * gv[k] * gv[k];
            const float denom = sqrtf(vv[k]) / bc2_sqrt + eps;
            pv[k] = pv[k] - step_size * (mv[k] / denom);
        }
        p4[i] = pv; m4[i] = mv; v4[i] = vv;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float mv = m[i] + (g[i] - m[i]) * (1.f - b1);
        float vv = v[i] * b2 + (1.f - b2) * g[i] * g[i];
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mv / denom);
        m[i] = mv; v[i] = vv;
    }
}

extern "C" int jaf_adam_step_dev(jaf_stream_t s, float* p, const float* g, float* m, float* v, int64_t n, float lr,
                                 float beta1, float beta2, float eps, float* state) {
    JAF_REQUIRE(p && g && m && v && state && n >= 1);
    JAF_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)s, state, beta1, beta2);
    hipLaunchKernelGGL(adam_dev_kernel, dim3(jaf_ew_grid(n, 4)), dim3(256), 0, (hipStream_t)s, p, g, m, v, (long)n, lr, beta1,
                       beta2, eps, (const float*)state);
    return jaf_launch_status();
}

__global__ void sum_slots_kernel(const float* in, int slots, long n, float* out, int accumulate) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float t = 0.f;
        for (int s = 0; s < slots; ++s) t += in[(long)s * n + i];
        out[i] = accumulate ? out[i] + t : t;
    }
}

extern "C" int jaf_sum_slots(jaf_stream_t s, const float* in, int32_t slots, int64_t n, float* out, int accumulate) {
    JAF_REQUIRE(in && out && slots >= 1 && n >= 1);
    hipLaunchKernelGGL(sum_slots_kernel, dim3(jaf_ew_grid(n)), dim3(256), 0, (hipStream_t)s, in, slots, (long)n, out, accumulate);
    return jaf_launch_status();
}

__global__ void axpby_kernel(float a, const float* x, float b, float* y, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        y[i] = a * x[i] + (b == 0.f ? 0.f : b * y[i]);
}

extern "C" int jaf_axpby(jaf_stream_t s, float a, const float* x, float b, float* y, int64_t n) {
    JAF_REQUIRE(x && y && n >= 0);
    if (n == 0) return JAF_OK;
    hipLaunchKernelGGL(axpby_kernel, dim3(jaf_ew_grid(n)), dim3(256), 0, (hipStream_t)s, a, x, b, y, (long)n);
    return jaf_launch_status();
}

// ------------------------------------------------------------------ kernel names for the profiler rows
thread_local char jaf_kname_buf[160] = "";
int jaf_kname_on = 0;

extern "C" int jaf_kernel_names(int on) {
    const int prev = jaf_kname_on;
    jaf_kname_on = on ? 1 : 0;
    return prev;
}

extern "C" int jaf_last_kernel_name(char* buf, int32_t buflen) {
    JAF_REQUIRE(buf && buflen >= 1);
    snprintf(buf, (size_t)buflen, "%s", jaf_kname_buf);
    return JAF_OK;
}
