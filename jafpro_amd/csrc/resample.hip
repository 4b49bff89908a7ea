// Pools, bilinear resizes and reflection padding (HBM-bound; lanes walk the contiguous W axis).
#include "jaf_common.h"
#include <stdlib.h>

// ------------------------------------------------------------------ average pooling
// count_include_pad=True semantics (F.avg_pool2d default, src/crn_model.py:268-273): divisor k*k.
// grid (x blocks, output rows, planes): no lane divides.
__global__ void avgpool_fwd_kernel(const float* x, float* y, int H, int W, int OH, int OW, int k, int stride, int pad) {
    const int ox = blockIdx.x * blockDim.x + threadIdx.x;
    const int oy = blockIdx.y * blockDim.y + threadIdx.y;
    if (ox >= OW || oy >= OH) return;
    const long nc = blockIdx.z;
    const float inv = 1.0f / (float)(k * k);
    const float* p = x + nc * H * W;
    float acc = 0.f;
    for (int ky = 0; ky < k; ++ky) {
        const int iy = oy * stride - pad + ky;
        if (iy < 0 || iy >= H) continue;
        for (int kx = 0; kx < k; ++kx) {
            const int ix = ox * stride - pad + kx;
            if (ix < 0 || ix >= W) continue;
            acc += p[iy * W + ix];
        }
    }
    y[(nc * OH + oy) * OW + ox] = acc * inv;
}

// 2 x 2 / stride 2 (the VGG pools: the largest ones move 335 MB): 4 outputs per lane from two 32-byte row pieces, one 16-byte store;
// the same summation order as the general kernel
__global__ void avgpool_fwd_k2s2_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int OH, int OW) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int ox0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int oy = blockIdx.y * blockDim.y + threadIdx.y;
    if (ox0 >= OW || oy >= OH) return;
    const long nc = blockIdx.z;
    const float* p = x + (nc * H + 2 * oy) * (long)W + 2 * ox0;
    const f4 a0 = *(const f4*)p, a1 = *(const f4*)(p + 4);
    const f4 b0 = *(const f4*)(p + W), b1 = *(const f4*)(p + W + 4);
    f4 o;
    o[0] = (((a0[0] + a0[1]) + b0[0]) + b0[1]) * 0.25f;
    o[1] = (((a0[2] + a0[3]) + b0[2]) + b0[3]) * 0.25f;
    o[2] = (((a1[0] + a1[1]) + b1[0]) + b1[1]) * 0.25f;
    o[3] = (((a1[2] + a1[3]) + b1[2]) + b1[3]) * 0.25f;
    *(f4*)(y + (nc * OH + oy) * (long)OW + ox0) = o;
}

// (tx, ty) with tx a power of two >= 8 covering short rows and tx*ty = 256
static inline dim3 block2d(int w) {
    int tx = 256;
    while (tx > 8 && (tx >> 1) >= w) tx >>= 1;
    return dim3(tx, 256 / tx);
}

extern "C" int jaf_avgpool_fwd(jaf_stream_t s, const float* x, float* y, int32_t NC, int32_t H, int32_t W,
                               int32_t OH, int32_t OW, int32_t k, int32_t stride, int32_t pad) {
    JAF_REQUIRE(x && y && NC >= 1 && H >= 1 && W >= 1 && k >= 1 && stride >= 1 && pad >= 0);
    JAF_REQUIRE(OH == (H + 2 * pad - k) / stride + 1 && OW == (W + 2 * pad - k) / stride + 1);
    JAF_REQUIRE(OH <= 65535 && NC <= 65535);
    if (k == 2 && stride == 2 && pad == 0 && W % 8 == 0 && ((((uintptr_t)x) | ((uintptr_t)y)) & 15) == 0) {      // (W % 8 == 0: OW % 4 == 0, rows 16-byte aligned)
        const dim3 b = block2d(OW / 4);
        hipLaunchKernelGGL(avgpool_fwd_k2s2_kernel, dim3(jaf_cdiv(OW / 4, b.x), jaf_cdiv(OH, b.y), NC), b, 0, (hipStream_t)s, x, y, H, W, OH, OW);
        return jaf_launch_status();
    }
    const dim3 b = block2d(OW);
    hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(jaf_cdiv(OW, b.x), jaf_cdiv(OH, b.y), NC), b, 0, (hipStream_t)s, x, y, H, W, OH, OW, k, stride, pad);
    return jaf_launch_status();
}

__global__ void avgpool_bwd_kernel(const float* dy, float* dx, int H, int W, int OH, int OW, int k, int stride, int pad) {
    const int ix = blockIdx.x * blockDim.x + threadIdx.x;
    const int iy = blockIdx.y * blockDim.y + threadIdx.y;
    if (ix >= W || iy >= H) return;
    const long nc = blockIdx.z;
    const float inv = 1.0f / (float)(k * k);
    const float* p = dy + nc * OH * OW;
    // outputs oy with oy*stride - pad <= iy <= oy*stride - pad + k - 1
    int oy_lo = (iy + pad - k + 1 + stride - 1);
    oy_lo = oy_lo <= 0 ? 0 : oy_lo / stride;
    int oy_hi = (iy + pad) / stride;
    if (oy_hi > OH - 1) oy_hi = OH - 1;
    int ox_lo = (ix + pad - k + 1 + stride - 1);
    ox_lo = ox_lo <= 0 ? 0 : ox_lo / stride;
    int ox_hi = (ix + pad) / stride;
    if (ox_hi > OW - 1) ox_hi = OW - 1;
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy)
        for (int ox = ox_lo; ox <= ox_hi; ++ox) acc += p[oy * OW + ox];
    dx[(nc * H + iy) * W + ix] = acc * inv;
}

// stride-2 pools (VGG: 2 x 2 / 2, the discriminators: 3 x 3 / 2 pad 1) with 4 consecutive input pixels per lane: one 16-byte store, the
// window bounds by shifts (the general kernel spends two integer divisions and a 4-byte store per pixel: 1.7 TB/s)
template <int K>
__global__ void avgpool_bwd_s2_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W, int OH, int OW, int pad) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const int ix0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int iy = blockIdx.y * blockDim.y + threadIdx.y;
    if (ix0 >= W || iy >= H) return;
    const long nc = blockIdx.z;
    const float* p = dy + nc * OH * OW;
    const int ty = iy + pad - K + 2;                      // oy_lo = ceil((iy + pad - K + 1) / 2)
    const int oy_lo = ty <= 0 ? 0 : ty >> 1;
    int oy_hi = (iy + pad) >> 1;
    if (oy_hi > OH - 1) oy_hi = OH - 1;
    f4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ix = ix0 + j;
        const int tx = ix + pad - K + 2;
        const int ox_lo = tx <= 0 ? 0 : tx >> 1;
        int ox_hi = (ix + pad) >> 1;
        if (ox_hi > OW - 1) ox_hi = OW - 1;
        float acc = 0.f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
            for (int ox = ox_lo; ox <= ox_hi; ++ox) acc += p[oy * OW + ox];
        o[j] = acc * (1.0f / (float)(K * K));
    }
    *(f4*)(dx + (nc * H + iy) * W + ix0) = o;
}

extern "C" int jaf_avgpool_bwd(jaf_stream_t s, const float* dy, float* dx, int32_t NC, int32_t H, int32_t W,
                               int32_t OH, int32_t OW, int32_t k, int32_t stride, int32_t pad) {
    JAF_REQUIRE(dy && dx && NC >= 1 && H >= 1 && W >= 1 && k >= 1 && stride >= 1 && pad >= 0);
    JAF_REQUIRE(OH == (H + 2 * pad - k) / stride + 1 && OW == (W + 2 * pad - k) / stride + 1);
    JAF_REQUIRE(H <= 65535 && NC <= 65535);
    if (stride == 2 && (k == 2 || k == 3) && W % 4 == 0 && (((uintptr_t)dx) & 15) == 0) {
        const dim3 b = block2d(W / 4);
        const dim3 g(jaf_cdiv(W / 4, b.x), jaf_cdiv(H, b.y), NC);
        if (k == 2) hipLaunchKernelGGL(avgpool_bwd_s2_kernel<2>, g, b, 0, (hipStream_t)s, dy, dx, H, W, OH, OW, pad);
        else hipLaunchKernelGGL(avgpool_bwd_s2_kernel<3>, g, b, 0, (hipStream_t)s, dy, dx, H, W, OH, OW, pad);
        return jaf_launch_status();
    }
    const dim3 b = block2d(W);
    hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(jaf_cdiv(W, b.x), jaf_cdiv(H, b.y), NC), b, 0, (hipStream_t)s, dy, dx, H, W, OH, OW, k, stride, pad);
    return jaf_launch_status();
}

// ------------------------------------------------------------------ bilinear / nearest resize
// Source-index rules of ATen upsample_bilinear2d / upsample_nearest2d (torch 2.x):
//   align_corners: src = dst * (in-1)/(out-1);   else: src = max((dst+0.5)*in/out - 0.5, 0)
//   nearest: src = min(floor(dst * in/out), in-1)
typedef float f32x4r __attribute__((ext_vector_type(4)));

struct ResizeArgs {
    int N, C, H, W, y0, x0, ch, cw, OH, OW;
    float sy, sx;
    int align, nearest;
    int rw, rh;        // LDS kernels: pitch and rows of the staged region (floats)
    float inv_sy, inv_sx;
    int vec;           // LDS kernels: 16-byte staging allowed
    int rows;          // resize_bwd_lds_kernel: input rows per lane (tile height = blockDim.y * rows)
};

__device__ __forceinline__ void resize_src(int o, float scale, int in, int align, int& i0, int& i1, float& l) {
    float src = align ? scale * (float)o : fmaxf(scale * ((float)o + 0.5f) - 0.5f, 0.f);
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + ((i0 < in - 1) ? 1 : 0);
    l = src - (float)i0;
}

// grid (x blocks, output rows, planes); V consecutive output columns per lane (V = 4: one 16-byte store)
template <int V>
__global__ void resize_fwd_kernel(const float* x, float* y, ResizeArgs a) {
    const int ox = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    const int oy = blockIdx.y * blockDim.y + threadIdx.y;
    if (ox >= a.OW || oy >= a.OH) return;
    const long nc = blockIdx.z;
    const float* p = x + nc * a.H * a.W;
    float* q = y + (nc * a.OH + oy) * a.OW + ox;
    float o[V];
    if (a.nearest) {
        int iy = (int)floorf((float)oy * a.sy); if (iy > a.ch - 1) iy = a.ch - 1;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            int ix = (int)floorf((float)(ox + k) * a.sx); if (ix > a.cw - 1) ix = a.cw - 1;
            o[k] = p[(a.y0 + iy) * a.W + a.x0 + ix];
        }
    } else {
        int y0, y1; float ly;
        resize_src(oy, a.sy, a.ch, a.align, y0, y1, ly);
        const float hy = 1.f - ly;
        const float* r0 = p + (a.y0 + y0) * a.W + a.x0;
        const float* r1 = p + (a.y0 + y1) * a.W + a.x0;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            int x0, x1; float lx;
            resize_src(ox + k, a.sx, a.cw, a.align, x0, x1, lx);
            const float hx = 1.f - lx;
            o[k] = hy * (hx * r0[x0] + lx * r0[x1]) + ly * (hx * r1[x0] + lx * r1[x1]);
        }
    }
    if (V == 4) *(f32x4r*)q = *(f32x4r*)o;
    else q[0] = o[0];
}

// Bilinear forward with the source region of the workgroup's output tile staged in LDS: the direct kernel issues 16
// overlapping 4-byte global loads per lane (4 outputs x 4 taps) and reaches 1.9 TB/s on a store-bound operation.
// Block (tx, ty) = tx*4 x ty outputs of one plane; the source rows / columns of a tile are those of its first and
// last output (the source index is monotonic in the output index).
__global__ void resize_fwd_lds_kernel(const float* __restrict__ x, float* __restrict__ y, ResizeArgs a) {
    extern __shared__ float s_src[];
    const int ox0 = blockIdx.x * blockDim.x * 4, oy0 = blockIdx.y * blockDim.y;
    const long nc = blockIdx.z;
    int oxl = ox0 + (int)blockDim.x * 4 - 1, oyl = oy0 + (int)blockDim.y - 1;
    if (oxl > a.OW - 1) oxl = a.OW - 1;
    if (oyl > a.OH - 1) oyl = a.OH - 1;
    int rx0, rx1, ry0, ry1, t0, t1;
    float l;
    resize_src(ox0, a.sx, a.cw, a.align, rx0, t1, l);
    resize_src(oxl, a.sx, a.cw, a.align, t0, rx1, l);
    resize_src(oy0, a.sy, a.ch, a.align, ry0, t1, l);
    resize_src(oyl, a.sy, a.ch, a.align, t0, ry1, l);
    if (a.vec) rx0 &= ~3;                                    // 16-byte staging (W % 4 == 0, x0 % 4 == 0, pitch % 4 == 0)
    if (rx1 > rx0 + a.rw - 1) rx1 = rx0 + a.rw - 1;          // never taken: a.rw / a.rh bound the region (host)
    if (ry1 > ry0 + a.rh - 1) ry1 = ry0 + a.rh - 1;
    const float* p = x + nc * a.H * a.W + (long)(a.y0 + ry0) * a.W + a.x0 + rx0;
    if (a.vec) {
        const int w4 = (rx1 - rx0 + 4) >> 2;
        for (int r = threadIdx.y; r <= ry1 - ry0; r += blockDim.y)
            for (int c4 = threadIdx.x; c4 < w4; c4 += blockDim.x)
                *(f32x4r*)(s_src + r * a.rw + 4 * c4) = *(const f32x4r*)(p + (long)r * a.W + 4 * c4);
    } else {
        for (int r = threadIdx.y; r <= ry1 - ry0; r += blockDim.y)
            for (int c = threadIdx.x; c <= rx1 - rx0; c += blockDim.x) s_src[r * a.rw + c] = p[(long)r * a.W + c];
    }
    __syncthreads();
    const int ox = ox0 + threadIdx.x * 4;
    const int oy = oy0 + threadIdx.y;
    if (ox >= a.OW || oy >= a.OH) return;
    int y0, y1;
    float ly;
    resize_src(oy, a.sy, a.ch, a.align, y0, y1, ly);
    const float hy = 1.f - ly;
    const float* r0 = s_src + (y0 - ry0) * a.rw - rx0;
    const float* r1 = s_src + (y1 - ry0) * a.rw - rx0;
    f32x4r o;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int x0, x1;
        float lx;
        resize_src(ox + k, a.sx, a.cw, a.align, x0, x1, lx);
        const float hx = 1.f - lx;
        o[k] = hy * (hx * r0[x0] + lx * r0[x1]) + ly * (hx * r1[x0] + lx * r1[x1]);
    }
    *(f32x4r*)(y + (nc * a.OH + oy) * a.OW + ox) = o;
}

static void resize_scales(ResizeArgs& a) {
    if (a.nearest) {
        a.sy = (float)a.ch / (float)a.OH;
        a.sx = (float)a.cw / (float)a.OW;
    } else if (a.align) {
        a.sy = a.OH > 1 ? (float)(a.ch - 1) / (float)(a.OH - 1) : 0.f;
        a.sx = a.OW > 1 ? (float)(a.cw - 1) / (float)(a.OW - 1) : 0.f;
    } else {
        a.sy = (float)a.ch / (float)a.OH;
        a.sx = (float)a.cw / (float)a.OW;
    }
}

extern "C" int jaf_resize_fwd(jaf_stream_t s, const float* x, float* y, int32_t N, int32_t C, int32_t H, int32_t W,
                              int32_t y0, int32_t x0, int32_t ch, int32_t cw, int32_t OH, int32_t OW,
                              int align_corners, int nearest) {
    JAF_REQUIRE(x && y && N >= 1 && C >= 1 && OH >= 1 && OW >= 1 && ch >= 1 && cw >= 1);
    JAF_REQUIRE(y0 >= 0 && x0 >= 0 && y0 + ch <= H && x0 + cw <= W);
    ResizeArgs a = {N, C, H, W, y0, x0, ch, cw, OH, OW, 0.f, 0.f, align_corners, nearest};
    resize_scales(a);
    JAF_REQUIRE(OH <= 65535 && (long)N * C <= 65535);
    // 256-thread workgroups: (columns, rows) -- one-wave workgroups are dispatch-rate bound
    if (OW % 4 == 0 && (((uintptr_t)y) & 15) == 0) {
        const dim3 b = block2d(OW / 4);
        // source region of one tile of b.x*4 x b.y outputs: outputs*scale + 3 per axis, capped by the crop window
        long rw = (long)ceilf((float)(b.x * 4) * a.sx) + 3, rh = (long)ceilf((float)b.y * a.sy) + 3;
        if (rw > cw) rw = cw;
        if (rh > ch) rh = ch;
        // 16-byte staging needs whole 4-column groups inside the plane: W % 4 == 0 and the crop window starting on one
        const bool vec = (W % 4 == 0) && (x0 % 4 == 0) && ((((uintptr_t)x) & 15) == 0);
        if (vec) rw = (rw + 3 + 3) / 4 * 4;
        if (!nearest && a.sx <= 2.5f && a.sy <= 2.5f && rw * rh * 4 <= 48 * 1024) {
            a.rw = (int)rw;
            a.rh = (int)rh;
            a.vec = vec ? 1 : 0;
            hipLaunchKernelGGL(resize_fwd_lds_kernel, dim3(jaf_cdiv(OW / 4, b.x), jaf_cdiv(OH, b.y), N * C), b, (size_t)(rw * rh * 4),
                               (hipStream_t)s, x, y, a);
            return jaf_launch_status();
        }
        hipLaunchKernelGGL(resize_fwd_kernel<4>, dim3(jaf_cdiv(OW / 4, b.x), jaf_cdiv(OH, b.y), N * C), b, 0, (hipStream_t)s, x, y, a);
    } else {
        const dim3 b = block2d(OW);
        hipLaunchKernelGGL(resize_fwd_kernel<1>, dim3(jaf_cdiv(OW, b.x), jaf_cdiv(OH, b.y), N * C), b, 0, (hipStream_t)s, x, y, a);
    }
    return jaf_launch_status();
}

// Gather form of the adjoint (no atomics, deterministic): every input pixel visits the few
// output rows/columns whose two taps can touch it and re-derives their weights with the same
// source-index rule as the forward kernel.
__device__ __forceinline__ void resize_cand(int i, float scale, int out, int align, int& lo, int& hi) {
    if (scale <= 0.f) { lo = 0; hi = out - 1; return; }
    const float inv = 1.0f / scale;
    float a, b;
    if (align) { a = ((float)i - 1.f) * inv; b = ((float)i + 1.f) * inv; }
    else { a = ((float)i - 0.5f) * inv - 0.5f; b = ((float)i + 1.5f) * inv - 0.5f; }
    lo = (int)floorf(a - 0.01f);  // outputs with |src - i| < 1 lie in (a, b); the margin absorbs the rounding of 1/scale
    hi = (int)ceilf(b + 0.01f);
    if (lo < 0) lo = 0;
    if (hi > out - 1) hi = out - 1;
}

// Adjoint as a gather.  The weight of input i in output o of a bilinear resize is the tent function
// max(0, 1 - |src(o) - i|) of the (clamped) source coordinate -- plus, without align_corners, the mass
// that the right/bottom clamp i1 = min(i0 + 1, in - 1) folds onto the last index.  Candidates are the
// outputs with |src - i| < 1; x weights are resolved once per lane and reused for every row.
__device__ __forceinline__ float resize_w(int o, int i, float scale, int in, int align) {
    // align_corners: src = scale*o lies in [0, in-1], neither clamp of the forward rule is ever active, and the two
    // tap weights (1-l at i0, l at i0+1) are the tent function itself: 4 instructions instead of ~12 per tap
    // (the adjoint kernels evaluate ~14 of these per input pixel and were VALU-bound on them)
    if (align) return fmaxf(0.f, 1.f - fabsf(scale * (float)o - (float)i));
    const float src = fmaxf(scale * ((float)o + 0.5f) - 0.5f, 0.f);
    // forward rule: i0 = min((int)src, in-1), i1 = min(i0+1, in-1), l = src - i0 (resize_src)
    int i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    const int i1 = i0 + ((i0 < in - 1) ? 1 : 0);
    const float l = src - (float)i0;
    return (i0 == i ? 1.f - l : 0.f) + (i1 == i ? l : 0.f);
}

#define RB_MAXC 8
// grid (x blocks, input rows, planes)
template <typename DT>
__global__ void resize_bwd_kernel(const DT* dy, float* dx, ResizeArgs a) {
    const int gx = blockIdx.x * blockDim.x + threadIdx.x;
    const int gy = blockIdx.y * blockDim.y + threadIdx.y;
    if (gx >= a.W || gy >= a.H) return;
    const long nc = blockIdx.z;
    const int ix = gx - a.x0, iy = gy - a.y0;
    float acc = 0.f;
    if (ix >= 0 && iy >= 0 && ix < a.cw && iy < a.ch) {
        int ylo, yhi, xlo, xhi;
        resize_cand(iy, a.sy, a.OH, a.align, ylo, yhi);
        resize_cand(ix, a.sx, a.OW, a.align, xlo, xhi);
        const DT* p = dy + nc * a.OH * a.OW;
        if (xhi - xlo + 1 <= RB_MAXC) {
            float wx[RB_MAXC];
#pragma unroll
            for (int j = 0; j < RB_MAXC; ++j) wx[j] = (xlo + j <= xhi) ? resize_w(xlo + j, ix, a.sx, a.cw, a.align) : 0.f;
            for (int oy = ylo; oy <= yhi; ++oy) {
                const float wy = resize_w(oy, iy, a.sy, a.ch, a.align);      // row-uniform: scalar work
                if (wy == 0.f) continue;
                const DT* r = p + oy * a.OW + xlo;
                float row = 0.f;
#pragma unroll
                for (int j = 0; j < RB_MAXC; ++j) {
                    const int oxc = (xlo + j <= xhi) ? j : 0;              // clamped: the load is always in range,
                    row += wx[j] * (float)r[oxc];                          // padding taps carry weight 0
                }
                acc += wy * row;
            }
        } else {
            for (int oy = ylo; oy <= yhi; ++oy) {
                const float wy = resize_w(oy, iy, a.sy, a.ch, a.align);
                if (wy == 0.f) continue;
                float row = 0.f;
                for (int ox = xlo; ox <= xhi; ++ox) {
                    const float wxv = resize_w(ox, ix, a.sx, a.cw, a.align);
                    if (wxv != 0.f) row += wxv * (float)p[oy * a.OW + ox];
                }
                acc += wy * row;
            }
        }
    }
    dx[(nc * a.H + gy) * a.W + gx] = acc;
}

// Same adjoint with the candidate region of the workgroup's tile staged in LDS first: the direct kernel issues
// ~36 overlapping 4-byte global loads per input pixel (each dy element is fetched by 4-9 lanes) and runs at
// 1.4 TB/s; here every dy element is loaded once per tile, coalesced, and the taps read LDS (with the 4-instruction
// align_corners weights and the exact 5x5 support of <= 2.2x up-sampling: 161 -> 124 us per launch on average).
// Block (tx, ty) = one tile of tx x ty input pixels of one plane; the region is the union of the candidates of
// the tile's first and last pixel (candidate ranges are monotonic in the pixel index).
// DT: element type of dy (fp32, or bf16: the gradient of a bf16-stored handle, e.g. a decoder's lazily up-sampled input).
#ifndef JAF_RB_ROWS8_MIN_H
#define JAF_RB_ROWS8_MIN_H 50
#endif
template <typename DT>
__global__ void resize_bwd_lds_kernel(const DT* __restrict__ dy, float* __restrict__ dx, ResizeArgs a) {
    extern __shared__ float s_reg[];
    // a.rows input rows per lane: the tile is blockDim.x x (blockDim.y * a.rows) input pixels, so one staging round trip
    // (all of a lane's 16-byte loads in flight before the first LDS store) serves a.rows times the outputs
    const int TH = (int)blockDim.y * a.rows;
    const int gx0 = blockIdx.x * blockDim.x, gy0 = blockIdx.y * TH;
    const int gx = gx0 + threadIdx.x;
    const long nc = blockIdx.z;
    // tile -> crop-window coordinates, clipped (uniform)
    int ixf = gx0 - a.x0, ixl = gx0 + (int)blockDim.x - 1 - a.x0;
    int iyf = gy0 - a.y0, iyl = gy0 + TH - 1 - a.y0;
    if (ixf < 0) ixf = 0;
    if (iyf < 0) iyf = 0;
    if (ixl > a.cw - 1) ixl = a.cw - 1;
    if (iyl > a.ch - 1) iyl = a.ch - 1;
    const bool any = (ixf <= ixl) && (iyf <= iyl);
    int rx0 = 0, ry0 = 0, rx1 = 0, ry1 = 0;
    if (any) {
        int lo, hi;
        resize_cand(ixf, a.sx, a.OW, a.align, rx0, hi);
        resize_cand(ixl, a.sx, a.OW, a.align, lo, rx1);
        resize_cand(iyf, a.sy, a.OH, a.align, ry0, hi);
        resize_cand(iyl, a.sy, a.OH, a.align, lo, ry1);
        const DT* p = dy + nc * a.OH * a.OW;
        if (a.vec) {
            // 16-byte staging: the region starts on a multiple of 4 columns (OW % 4 == 0, pitch a.rw % 4 == 0);
            // three rows per pass so that three independent loads are in flight before the first LDS write
            rx0 &= ~3;
            if (rx1 > rx0 + a.rw - 1) rx1 = rx0 + a.rw - 1;  // never taken: a.rw / a.rh bound the region (host)
            if (ry1 > ry0 + a.rh - 1) ry1 = ry0 + a.rh - 1;
            const int w4 = (rx1 - rx0 + 4) >> 2, nr = ry1 - ry0 + 1;
            const int total4 = w4 * nr, tid = threadIdx.y * blockDim.x + threadIdx.x, nth = blockDim.x * blockDim.y;
            const float inv_w4 = 1.0f / (float)w4;
            for (int base = tid; base < total4; base += nth * 8) {
                f32x4r v[8];
                int so[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int idx = base + nth * u;
                    so[u] = -1;
                    if (idx < total4) {
                        int r = (int)(((float)idx + 0.5f) * inv_w4);
                        int c4 = idx - r * w4;
                        if (c4 < 0) { --r; c4 += w4; }
                        if (c4 >= w4) { ++r; c4 -= w4; }
                        float t4[4];
                        jaf_ldv<4, DT>(p + (long)(ry0 + r) * a.OW + rx0 + 4 * c4, t4);
                        v[u] = (f32x4r){t4[0], t4[1], t4[2], t4[3]};
                        so[u] = r * a.rw + 4 * c4;
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (so[u] >= 0) *(f32x4r*)(s_reg + so[u]) = v[u];
            }
        } else {
            if (rx1 > rx0 + a.rw - 1) rx1 = rx0 + a.rw - 1;
            if (ry1 > ry0 + a.rh - 1) ry1 = ry0 + a.rh - 1;
            for (int r = threadIdx.y; r <= ry1 - ry0; r += blockDim.y)
                for (int c = threadIdx.x; c <= rx1 - rx0; c += blockDim.x)
                    s_reg[r * a.rw + c] = (float)p[(ry0 + r) * a.OW + rx0 + c];
        }
    }
    __syncthreads();
    if (gx >= a.W) return;
    // the lane's column: its five x taps and their weights are the same for every row it visits (hoisted out of the row loop in
    // round 5: the kernel is vector-instruction bound, ~1200 per wave, and these were a seventh of them)
    const bool fast = a.align && a.sx >= 0.45f && a.sy >= 0.45f;
    float wx[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    int cx[5] = {0, 0, 0, 0, 0};
    {
        const int ix = gx - a.x0;
        if (fast && ix >= 0 && ix < a.cw) {
            int xl = (int)floorf((float)(ix - 1) * a.inv_sx) + 1;
            if (xl < rx0) xl = rx0;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int o = xl + j;
                wx[j] = o <= rx1 ? fmaxf(0.f, 1.f - fabsf(a.sx * (float)o - (float)ix)) : 0.f;
                cx[j] = (o <= rx1 ? o : rx1) - rx0;
            }
        }
    }
    if (fast) {
        // up-sampling by <= 2.2x with align_corners (every decoder of the path).  Row streaming: a lane owns a.rows CONTIGUOUS rows of one
        // input column; it walks the dy rows o that reach them once, takes the horizontal adjoint h(o) of its column (5 taps, weights
        // hoisted above) and adds (1 - f) h to row floor(sy o), f h to the row below -- two running accumulators, flushed as the source
        // row advances.  10 LDS reads + 14 fma per input pixel instead of the 25 + 25 of the per-pixel 5 x 5 gather (the kernel was
        // vector-instruction bound); o, the source row and the flushes are wave-uniform (a wave is one row of the block).
        const int r0 = gy0 + (int)threadIdx.y * a.rows;
        int r1 = r0 + a.rows;                                  // owned rows [r0, r1)
        if (r1 > a.H) r1 = a.H;
        float* out = dx + (nc * a.H) * a.W + gx;
        int cur = r0;                                          // next owned row to store
        for (; cur < r1 && cur < a.y0; ++cur) out[(long)cur * a.W] = 0.f;      // rows above the crop window
        int crow = cur - a.y0;                                 // crop row of `cur`
        const int cend = (r1 - a.y0 < a.ch ? r1 - a.y0 : a.ch);   // crop rows [crow, cend) are owned and inside the window
        float a0 = 0.f, a1 = 0.f;
        if (any && crow < cend) {
            int o = (int)floorf((float)(crow - 1) * a.inv_sy) - 1;
            if (o < ry0) o = ry0;
            for (; o <= ry1; ++o) {
                const float src = a.sy * (float)o;
                int i0 = (int)src;
                if (i0 > a.ch - 1) i0 = a.ch - 1;
                if (i0 < crow - 1) continue;
                if (i0 >= cend) break;
                while (crow < i0) {                            // rows above the source row are complete
                    out[(long)(crow + a.y0) * a.W] = a0;
                    a0 = a1; a1 = 0.f; ++crow;
                }
                const float l = (i0 < a.ch - 1) ? src - (float)i0 : 0.f;      // (the last row takes the whole weight: resize_src)
                const float* row = s_reg + (o - ry0) * a.rw;
                const float h = (wx[0] * row[cx[0]] + wx[1] * row[cx[1]]) + (wx[2] * row[cx[2]] + wx[3] * row[cx[3]]) + wx[4] * row[cx[4]];
                if (i0 == crow - 1) a0 += l * h;
                else { a0 += (1.f - l) * h; a1 += l * h; }
            }
        }
        for (; crow < cend; ++crow) { out[(long)(crow + a.y0) * a.W] = a0; a0 = a1; a1 = 0.f; }
        for (cur = (crow + a.y0 > cur ? crow + a.y0 : cur); cur < r1; ++cur) out[(long)cur * a.W] = 0.f;      // rows below the window
        return;
    }
    for (int rr = 0; rr < a.rows; ++rr) {
    const int gy = gy0 + threadIdx.y + (int)blockDim.y * rr;
    if (gy >= a.H) return;
    const int ix = gx - a.x0, iy = gy - a.y0;
    float acc = 0.f;
    if (ix >= 0 && iy >= 0 && ix < a.cw && iy < a.ch) {
        int ylo, yhi, xlo, xhi;
        resize_cand(iy, a.sy, a.OH, a.align, ylo, yhi);
        resize_cand(ix, a.sx, a.OW, a.align, xlo, xhi);
        if (xhi - xlo + 1 <= RB_MAXC) {
            float wx[RB_MAXC];
#pragma unroll
            for (int j = 0; j < RB_MAXC; ++j) wx[j] = (xlo + j <= xhi) ? resize_w(xlo + j, ix, a.sx, a.cw, a.align) : 0.f;
            for (int oy = ylo; oy <= yhi; ++oy) {
                const float wy = resize_w(oy, iy, a.sy, a.ch, a.align);
                if (wy == 0.f) continue;
                const float* r = s_reg + (oy - ry0) * a.rw + (xlo - rx0);
                float row = 0.f;
#pragma unroll
                for (int j = 0; j < RB_MAXC; ++j) {
                    const int oxc = (xlo + j <= xhi) ? j : 0;
                    row += wx[j] * r[oxc];
                }
                acc += wy * row;
            }
        } else {
            for (int oy = ylo; oy <= yhi; ++oy) {
                const float wy = resize_w(oy, iy, a.sy, a.ch, a.align);
                if (wy == 0.f) continue;
                float row = 0.f;
                for (int ox = xlo; ox <= xhi; ++ox) {
                    const float wxv = resize_w(ox, ix, a.sx, a.cw, a.align);
                    if (wxv != 0.f) row += wxv * s_reg[(oy - ry0) * a.rw + (ox - rx0)];
                }
                acc += wy * row;
            }
        }
    }
    dx[(nc * a.H + gy) * a.W + gx] = acc;
    }
}

extern "C" int jaf_resize_bwd(jaf_stream_t s, const float* dy, float* dx, int32_t N, int32_t C, int32_t H, int32_t W,
                              int32_t y0, int32_t x0, int32_t ch, int32_t cw, int32_t OH, int32_t OW,
                              int align_corners) {
    return jaf_resize_bwd_dt(s, dy, 0, dx, N, C, H, W, y0, x0, ch, cw, OH, OW, align_corners);
}

extern "C" int jaf_resize_bwd_dt(jaf_stream_t s, const void* dy, int dy_bf16, float* dx, int32_t N, int32_t C, int32_t H, int32_t W,
                                 int32_t y0, int32_t x0, int32_t ch, int32_t cw, int32_t OH, int32_t OW,
                                 int align_corners) {
    JAF_REQUIRE(dy && dx && N >= 1 && C >= 1 && OH >= 1 && OW >= 1 && ch >= 1 && cw >= 1);
    JAF_REQUIRE(y0 >= 0 && x0 >= 0 && y0 + ch <= H && x0 + cw <= W);
    ResizeArgs a = {N, C, H, W, y0, x0, ch, cw, OH, OW, 0.f, 0.f, align_corners, 0};
    resize_scales(a);
    JAF_REQUIRE(H <= 65535 && (long)N * C <= 65535);
    const dim3 b = block2d(W);      // (squarer 32x8 / 64x4 tiles stage fewer halo rows but measured 6-8 % slower)
    const bool vec = (OW % 4 == 0) && ((((uintptr_t)dy) & (dy_bf16 ? 7 : 15)) == 0);
    // candidate region of one (b.x, b.y * rows) tile: (pixels + 1) / scale + 4 per axis (resize_cand), capped by the image.
    // rows per lane: as many (<= 8, tile no taller than the plane) as keep the region within 40 KB and leave the launch >= 2048 workgroups
    long rw = 0, rh = 0;
    // 8 rows per lane on planes of at least 50 rows: -5..13 % against 4 on the 50 -> 100 .. 128 -> 256 adjoints (32 -> 64: +30 %)
    int rows = H >= JAF_RB_ROWS8_MIN_H ? 8 : 4;
    for (;; rows >>= 1) {
        rw = a.sx > 0.f ? (long)ceilf(((float)b.x + 1.f) / a.sx) + 4 : OW;
        rh = a.sy > 0.f ? (long)ceilf(((float)(b.y * rows) + 1.f) / a.sy) + 4 : OH;
        if (rw > OW) rw = OW;
        if (rh > OH) rh = OH;
        if (vec) rw = (rw + 3 + 3) / 4 * 4;        // start rounded down to a multiple of 4, pitch a multiple of 4
        const long blocks = (long)jaf_cdiv(W, b.x) * jaf_cdiv(H, b.y * rows) * N * C;
        if (rows == 1 || (rw * rh * 4 <= 40 * 1024 && blocks >= 2048)) break;
    }
    if (rw * rh * 4 <= 48 * 1024) {
        a.rw = (int)rw;
        a.rh = (int)rh;
        a.vec = vec ? 1 : 0;
        a.rows = rows;
        a.inv_sx = a.sx > 0.f ? 1.0f / a.sx : 0.f;
        a.inv_sy = a.sy > 0.f ? 1.0f / a.sy : 0.f;
        const dim3 grid(jaf_cdiv(W, b.x), jaf_cdiv(H, b.y * rows), N * C);
        if (dy_bf16) hipLaunchKernelGGL(resize_bwd_lds_kernel<__bf16>, grid, b, (size_t)(rw * rh * 4), (hipStream_t)s, (const __bf16*)dy, dx, a);
        else hipLaunchKernelGGL(resize_bwd_lds_kernel<float>, grid, b, (size_t)(rw * rh * 4), (hipStream_t)s, (const float*)dy, dx, a);
    } else {
        const dim3 grid(jaf_cdiv(W, b.x), jaf_cdiv(H, b.y), N * C);
        if (dy_bf16) hipLaunchKernelGGL(resize_bwd_kernel<__bf16>, grid, b, 0, (hipStream_t)s, (const __bf16*)dy, dx, a);
        else hipLaunchKernelGGL(resize_bwd_kernel<float>, grid, b, 0, (hipStream_t)s, (const float*)dy, dx, a);
    }
    return jaf_launch_status();
}

// ------------------------------------------------------------------ reflection padding
__device__ __forceinline__ int reflect_idx(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

// grid (x blocks, row blocks, planes), block (tx, ty): no per-element index decoding
__global__ void reflect_pad_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int p) {
    const int PH = H + 2 * p, PW = W + 2 * p;
    const int px = blockIdx.x * blockDim.x + threadIdx.x;
    const int py = blockIdx.y * blockDim.y + threadIdx.y;
    if (px >= PW || py >= PH) return;
    const long nc = blockIdx.z;
    y[(nc * PH + py) * PW + px] = x[(nc * H + reflect_idx(py - p, H)) * W + reflect_idx(px - p, W)];
}

extern "C" int jaf_reflect_pad_fwd(jaf_stream_t s, const float* x, float* y, int32_t NC, int32_t H, int32_t W, int32_t p) {
    JAF_REQUIRE(x && y && NC >= 1 && p >= 0 && p < H && p < W);
    JAF_REQUIRE(NC <= 65535 && H + 2 * p <= 65535);
    const dim3 b = block2d(W + 2 * p);
    hipLaunchKernelGGL(reflect_pad_fwd_kernel, dim3(jaf_cdiv(W + 2 * p, b.x), jaf_cdiv(H + 2 * p, b.y), NC), b, 0, (hipStream_t)s, x, y, H, W, p);
    return jaf_launch_status();
}

// gather form of the adjoint: each input pixel sums the padded positions that mirror onto it.
__global__ void reflect_pad_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int H, int W, int p) {
    const int PH = H + 2 * p, PW = W + 2 * p;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= W || y >= H) return;
    const long nc = blockIdx.z;
    int ys[3], xs[3], ny = 0, nx = 0;
    ys[ny++] = y + p;
    if (y >= 1 && y <= p) ys[ny++] = p - y;
    if (y <= H - 2 && y >= H - 1 - p) ys[ny++] = p + 2 * (H - 1) - y;
    xs[nx++] = x + p;
    if (x >= 1 && x <= p) xs[nx++] = p - x;
    if (x <= W - 2 && x >= W - 1 - p) xs[nx++] = p + 2 * (W - 1) - x;
    const float* q = dy + nc * PH * PW;
    float acc = 0.f;
    for (int i = 0; i < ny; ++i)
        for (int j = 0; j < nx; ++j) acc += q[ys[i] * PW + xs[j]];
    dx[(nc * H + y) * W + x] = acc;
}

extern "C" int jaf_reflect_pad_bwd(jaf_stream_t s, const float* dy, float* dx, int32_t NC, int32_t H, int32_t W, int32_t p) {
    JAF_REQUIRE(dy && dx && NC >= 1 && p >= 0 && p < H && p < W);
    JAF_REQUIRE(NC <= 65535 && H <= 65535);
    const dim3 b = block2d(W);
    hipLaunchKernelGGL(reflect_pad_bwd_kernel, dim3(jaf_cdiv(W, b.x), jaf_cdiv(H, b.y), NC), b, 0, (hipStream_t)s, dy, dx, H, W, p);
    return jaf_launch_status();
}
