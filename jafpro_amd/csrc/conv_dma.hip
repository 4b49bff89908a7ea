// Packed-input convolution: the bf16 matrix-core kernel of conv_bf16.hip with its staging done by
// the DMA path (buffer_load ... lds) instead of vector loads + conversion + ds_write.
//
// The input is first re-laid out ONCE per tensor by jaf_conv2d_pack_input into
//     packed[n][g][group8][y][x][8 channels]   (bf16, channel counts padded to 8 with zeros),
// i.e. exactly the 16-byte (position, 8 channels) items of the LDS patch image.  The forward
// convolution and the weight-gradient of a layer share one packed input, the data-gradient and the
// weight-gradient share one packed dz, so every tensor is converted to bf16 once instead of once
// per consumer and per Cout block.
//
// In the kernel a patch plane is filled by 16-byte-per-lane buffer loads that land directly in LDS:
// lane l of round r supplies slot 64*r + l, its source offset (or an out-of-range offset for
// padding, which the buffer unit turns into zeros -- measured, profiles/experiments/dma_test.hip) is chunk
// invariant, and the plane (wave-uniform) is a scalar buffer resource.  Per 32-channel chunk a wave
// issues <= 8 patch DMAs and <= 3 weight DMAs and no VALU staging work at all; the previous
// kernel spent ~800 of its ~1100 instructions per chunk there.
#include "conv_internal.h"
#include <stdlib.h>

#include "conv_dma_kernel.h"

// ---------------------------------------------------------------------------------------------
// input packing: fp32 NCHW (up to three concatenated sources, grouped) -> bf16 [n][g][group8][y][x][8]
// grid (x blocks, H, N*G*ngroups8); V = 4 columns per lane when W % 4 == 0.
// ---------------------------------------------------------------------------------------------
struct PackInArgs {
    const float* src[3];
    unsigned char* out;
    jaf_conv_desc d;
    int ngroups8;
    int split;             // d.precision == JAF_PREC_BF16X3: every group of 8 channels gets TWO planes, hi = bf16(v) and right behind
                           // it lo = bf16(v - hi): [n][g][group8][hi, lo][y][x][8] (the operand images of conv_dma_split.hip)
};

// One pixel's 8 channels as a packed item; `lo`: the residual item v - bf16(v) of the split-bf16 images.
__device__ __forceinline__ u32x4 cd_item8(float v0, float v1, float v2, float v3, float v4, float v5, float v6, float v7, bool lo) {
    if (lo) {
        v0 -= (float)(__bf16)v0; v1 -= (float)(__bf16)v1; v2 -= (float)(__bf16)v2; v3 -= (float)(__bf16)v3;
        v4 -= (float)(__bf16)v4; v5 -= (float)(__bf16)v5; v6 -= (float)(__bf16)v6; v7 -= (float)(__bf16)v7;
    }
    u32x4 w;
    w[0] = cd_pack2(v0, v1);
    w[1] = cd_pack2(v2, v3);
    w[2] = cd_pack2(v4, v5);
    w[3] = cd_pack2(v6, v7);
    return w;
}

// Byte offset of item (pixel `pix`) of plane `cg` of (image, group) `ng` in an image of `ng8` channel groups; split images
// hold the hi plane of a group at 2 cg and its lo plane at 2 cg + 1.
__device__ __forceinline__ long cd_item_off(long ng, int ng8, int cg, long HW, long pix, int split) {
    return split ? (((ng * ng8 + cg) * 2) * HW + pix) * 16 : ((ng * ng8 + cg) * HW + pix) * 16;
}

template <int V>
__global__ void conv_pack_input_kernel(const PackInArgs a) {
    const jaf_conv_desc& d = a.d;
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= d.W || y >= d.H) return;
    int z = blockIdx.z;
    const int cg = z % a.ngroups8;
    z /= a.ngroups8;
    const int g = z % d.G;
    const int n = z / d.G;
    const int c0 = d.src_c[0];
    const int c01 = c0 + (d.nsrc > 1 ? d.src_c[1] : 0);
    const long HW = (long)d.H * d.W;
    // base of this (image, group) in each source, with constant indices: a descriptor array indexed by a run-time source
    // number is re-read from the kernel arguments (scalar load + wait) at every use
    const float* sb[3];
#pragma unroll
    for (int s = 0; s < 3; ++s)
        sb[s] = (s < d.nsrc) ? a.src[s] + ((long)n * d.src_ctot[s] + d.src_coff[s] + (long)g * d.src_gstride[s]) * HW : nullptr;
    const unsigned pix = (unsigned)(y * d.W + x);
    float v[8][V];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cg * 8 + j;
#pragma unroll
        for (int i = 0; i < V; ++i) v[j][i] = 0.f;
        if (c < d.Cin) {
            const int s = (c < c0) ? 0 : ((c < c01) ? 1 : 2);
            const int cl = (s == 0) ? c : ((s == 1) ? c - c0 : c - c01);
            const float* p = ((s == 0) ? sb[0] : ((s == 1) ? sb[1] : sb[2])) + cl * HW + pix;
            if (V == 4) {
                const f32x4 t = *(const f32x4*)p;
                v[j][0] = t[0]; v[j][1] = t[1]; v[j][2] = t[2]; v[j][3] = t[3];
            } else {
                v[j][0] = p[0];
            }
        }
    }
    unsigned char* o = a.out + cd_item_off((long)n * d.G + g, a.ngroups8, cg, HW, (long)y * d.W + x, a.split);
#pragma unroll
    for (int i = 0; i < V; ++i) {
        *(u32x4*)(o + i * 16) = cd_item8(v[0][i], v[1][i], v[2][i], v[3][i], v[4][i], v[5][i], v[6][i], v[7][i], false);
        if (a.split) *(u32x4*)(o + HW * 16 + i * 16) = cd_item8(v[0][i], v[1][i], v[2][i], v[3][i], v[4][i], v[5][i], v[6][i], v[7][i], true);
    }
}

static bool pack_desc_ok(const jaf_conv_desc* d) {
    if (!d) return false;
    if (d->N < 1 || d->G < 1 || d->Cin < 1 || d->H < 1 || d->W < 1) return false;
    if (d->nsrc < 1 || d->nsrc > 3) return false;
    int c = 0;
    for (int i = 0; i < d->nsrc; ++i) {
        if (d->src_c[i] < 1 || d->src_ctot[i] < 1 || d->src_coff[i] < 0 || d->src_gstride[i] < 0) return false;
        if (d->src_coff[i] + (d->G - 1) * d->src_gstride[i] + d->src_c[i] > d->src_ctot[i]) return false;
        c += d->src_c[i];
    }
    return c == d->Cin;
}

extern "C" int64_t jaf_conv2d_packed_input_bytes(const jaf_conv_desc* d) {
    if (!pack_desc_ok(d)) return -1;
    return (int64_t)d->N * d->G * jaf_cdiv(d->Cin, 8) * d->H * d->W * 16 * (d->precision == JAF_PREC_BF16X3 ? 2 : 1);
}

extern "C" int jaf_conv2d_pack_input(jaf_stream_t s, const jaf_conv_desc* d, const float* src0, const float* src1,
                                     const float* src2, void* packed) {
    JAF_REQUIRE(pack_desc_ok(d) && src0 && packed);
    JAF_REQUIRE(d->nsrc < 2 || src1);
    JAF_REQUIRE(d->nsrc < 3 || src2);
    PackInArgs a;
    a.src[0] = src0; a.src[1] = src1; a.src[2] = src2;
    a.out = (unsigned char*)packed;
    a.d = *d;
    a.ngroups8 = jaf_cdiv(d->Cin, 8);
    a.split = d->precision == JAF_PREC_BF16X3 ? 1 : 0;
    const long nz = (long)d->N * d->G * a.ngroups8;
    if (nz > 65535 || d->H > 65535) return JAF_EUNSUPPORTED;
    const bool al = ((((uintptr_t)src0) | ((uintptr_t)src1) | ((uintptr_t)src2)) & 15) == 0;
    // packing is elementwise over the plane: when only H*W is a multiple of 4 (50x50 levels) the plane is walked
    // as one row so that the 16-byte path still applies
    if (al && d->W % 4 != 0 && ((long)d->H * d->W) % 4 == 0 && (long)d->H * d->W < (1L << 30)) { a.d.W = d->H * d->W; a.d.H = 1; }
    const int H = a.d.H, W = a.d.W;
    const bool v4 = (W % 4 == 0) && al;
    // 256-thread workgroups: tx lanes along x (a power of two covering the row when it is short), ty rows
    const int wx = v4 ? W / 4 : W;
    int tx = H == 1 ? 256 : 64;
    while (tx > 1 && (tx >> 1) >= wx) tx >>= 1;
    if (tx < 8) tx = 8;
    const int ty = 256 / tx;
    const dim3 grid(jaf_cdiv(wx, tx), jaf_cdiv(H, ty), (unsigned)nz);
    if (v4) hipLaunchKernelGGL(conv_pack_input_kernel<4>, grid, dim3(tx, ty), 0, (hipStream_t)s, a);
    else hipLaunchKernelGGL(conv_pack_input_kernel<1>, grid, dim3(tx, ty), 0, (hipStream_t)s, a);
    return jaf_launch_status();
}

// Packing with bilinear up-sampling fused in: a source flagged lazy is given at its LOW resolution [N][ctot][sh][sw] and
// is sampled at the layer's H x W on the fly (ATen upsample_bilinear2d index rules, as resample.hip) -- the up-sampled
// fp32 tensor of the decoders (`cat[up(x), skip]`, src/networks.py:896-909; the CRN's `cat[label, pool, up(net)]`,
// src/crn_model.py:276-299) is neither written nor read back.  One lane = one pixel x 8 channels.
struct PackLazyArgs {
    PackInArgs b;
    int lazy[3], sh[3], sw[3], align[3];
    float sy[3], sx[3];
    int cap;               // LDS-staged form: floats per channel of the staged source tile
    int cg0, cgn;          // this launch covers the channel groups [cg0, cg0 + cgn) (the others: conv_pack_up_kernel)
};

__device__ __forceinline__ void cd_resize_src(int o, float scale, int in, int align, int& i0, int& i1, float& l) {
    float src = align ? scale * (float)o : fmaxf(scale * ((float)o + 0.5f) - 0.5f, 0.f);
    i0 = (int)src;
    if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + ((i0 < in - 1) ? 1 : 0);
    l = src - (float)i0;
}

__global__ __launch_bounds__(256) void conv_pack_input_lazy_kernel(const PackLazyArgs a) {
    const jaf_conv_desc& d = a.b.d;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int y = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= d.W || y >= d.H) return;
    int z = blockIdx.z;
    const int cg = a.cg0 + z % a.cgn;
    z /= a.cgn;
    const int g = z % d.G;
    const int n = z / d.G;
    const int c0 = d.src_c[0];
    const int c01 = c0 + (d.nsrc > 1 ? d.src_c[1] : 0);
    const long HW = (long)d.H * d.W;
    int o00[3], o01[3], o10[3], o11[3];
    float ly[3], lx[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        o00[s] = o01[s] = o10[s] = o11[s] = 0;
        ly[s] = lx[s] = 0.f;
        if (s < d.nsrc && a.lazy[s]) {
            int y0, y1, x0, x1;
            cd_resize_src(y, a.sy[s], a.sh[s], a.align[s], y0, y1, ly[s]);
            cd_resize_src(x, a.sx[s], a.sw[s], a.align[s], x0, x1, lx[s]);
            o00[s] = y0 * a.sw[s] + x0; o01[s] = y0 * a.sw[s] + x1;
            o10[s] = y1 * a.sw[s] + x0; o11[s] = y1 * a.sw[s] + x1;
        }
    }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cg * 8 + j;
        v[j] = 0.f;
        if (c < d.Cin) {
            const int s = (c < c0) ? 0 : ((c < c01) ? 1 : 2);
            const int cl = (s == 0) ? c : ((s == 1) ? c - c0 : c - c01);
            const float* sp = (s == 0) ? a.b.src[0] : ((s == 1) ? a.b.src[1] : a.b.src[2]);
            const long ch = (long)n * d.src_ctot[s] + d.src_coff[s] + g * d.src_gstride[s] + cl;
            if (a.lazy[s]) {
                const float* p = sp + ch * (long)(a.sh[s] * a.sw[s]);
                const float hy = 1.f - ly[s], hx = 1.f - lx[s];
                v[j] = hy * (hx * p[o00[s]] + lx[s] * p[o01[s]]) + ly[s] * (hx * p[o10[s]] + lx[s] * p[o11[s]]);
            } else {
                v[j] = sp[ch * HW + (long)y * d.W + x];
            }
        }
    }
    unsigned char* o = a.b.out + cd_item_off((long)n * d.G + g, a.b.ngroups8, cg, HW, (long)y * d.W + x, a.b.split);
    *(u32x4*)o = cd_item8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], false);
    if (a.b.split) *(u32x4*)(o + HW * 16) = cd_item8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], true);
}

// The same packing with the low-resolution source tiles staged through LDS: a 64 x 16 output tile of an up-sampled source
// needs about (64 sx + 2) x (16 sy + 2) source pixels per channel, which the workgroup loads once, coalesced, instead of
// four gathers per output value.  V = 4 (W % 4 == 0): a lane owns 4 consecutive pixels of one row -- the sources that are
// not resized arrive by 16-byte loads and the lane writes 64 contiguous bytes; V = 1: lane = column, 4 rows per lane.
// Same expression tree as the gather kernel: results are identical.
#define PL_TW 64
#define PL_TH 16
#define PL_CAP 1280        // floats per channel of the staged source tile (host checks the bound)
template <int V>
__global__ __launch_bounds__(256) void conv_pack_input_lazy_lds_kernel(const PackLazyArgs a) {
    extern __shared__ float s_dyn[];       // [8][a.cap]
    constexpr int LX = PL_TW / V;          // lanes along x
    constexpr int RP = 256 / LX;           // rows in flight
    const jaf_conv_desc& d = a.b.d;
    const int tid = threadIdx.x;
    const int tx = tid % LX, ty = tid / LX;
    const int x = blockIdx.x * PL_TW + tx * V;
    const int yb = blockIdx.y * PL_TH;
    int z = blockIdx.z;
    const int cg = a.cg0 + z % a.cgn;
    z /= a.cgn;
    const int g = z % d.G;
    const int n = z / d.G;
    const int c0 = d.src_c[0];
    const int c01 = c0 + (d.nsrc > 1 ? d.src_c[1] : 0);
    const long HW = (long)d.H * d.W;
    // per-source scalars with constant indices (a descriptor array indexed by a run-time source number is re-read from
    // the kernel arguments, with a wait, at every use): plane size, lazy flag, base of this (image, group)
    int lz[3];
    long pl[3];
    const float* sb[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        lz[s] = (s < d.nsrc) ? a.lazy[s] : 0;
        pl[s] = lz[s] ? (long)a.sh[s] * a.sw[s] : HW;
        sb[s] = (s < d.nsrc) ? a.b.src[s] + ((long)n * d.src_ctot[s] + d.src_coff[s] + (long)g * d.src_gstride[s]) * pl[s] : nullptr;
    }
    const int xl = min(blockIdx.x * PL_TW + PL_TW - 1, d.W - 1), yl = min(yb + PL_TH - 1, d.H - 1);     // last column / row of the tile
    int ys0[3], xs0[3], tw[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        ys0[s] = xs0[s] = 0; tw[s] = 1;
        if (lz[s]) {
            int i0, i1; float l;
            cd_resize_src(yb, a.sy[s], a.sh[s], a.align[s], ys0[s], i1, l);
            cd_resize_src(blockIdx.x * PL_TW, a.sx[s], a.sw[s], a.align[s], xs0[s], i1, l);
            cd_resize_src(xl, a.sx[s], a.sw[s], a.align[s], i0, i1, l);
            tw[s] = i1 - xs0[s] + 1;
        }
    }
    // stage: channel j of this group of 8, rows ys0 .. y1(last row), columns xs0 .. x1(last column).  All of a lane's loads
    // (two per channel cover tiles of up to 512 source pixels) are issued before the first LDS store.
    int cnt[3], soff[3][2];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        cnt[s] = 0; soff[s][0] = soff[s][1] = 0;
        if (lz[s]) {
            int i0, i1; float l;
            cd_resize_src(yl, a.sy[s], a.sh[s], a.align[s], i0, i1, l);
            cnt[s] = (i1 - ys0[s] + 1) * tw[s];
            const float inv_tw = 1.0f / (float)tw[s];          // (e < 512: the reciprocal form is exact; six integer divisions per lane before)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int e = tid + 256 * u;
                const int r = (int)(((float)e + 0.5f) * inv_tw), q = e - r * tw[s];
                soff[s][u] = (ys0[s] + r) * a.sw[s] + xs0[s] + q;
            }
        }
    }
    float st[8][2];
    const float* sbase[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cg * 8 + j;
        sbase[j] = nullptr;
        st[j][0] = st[j][1] = 0.f;
        if (c >= d.Cin) continue;
        const int s = (c < c0) ? 0 : ((c < c01) ? 1 : 2);
        if (!((s == 0) ? lz[0] : ((s == 1) ? lz[1] : lz[2]))) continue;
        const int cl = (s == 0) ? c : ((s == 1) ? c - c0 : c - c01);
        sbase[j] = ((s == 0) ? sb[0] : ((s == 1) ? sb[1] : sb[2])) + cl * ((s == 0) ? pl[0] : ((s == 1) ? pl[1] : pl[2]));
        const int cn = (s == 0) ? cnt[0] : ((s == 1) ? cnt[1] : cnt[2]);
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (tid + 256 * u < cn) st[j][u] = sbase[j][(s == 0) ? soff[0][u] : ((s == 1) ? soff[1][u] : soff[2][u])];
    }
    // the sources that are not resized do not depend on the staged tiles: their loads go out with the staging loads
    constexpr int ITER = PL_TH / RP;
    float nv[ITER][8][V];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int y = yb + ty + RP * it;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cg * 8 + j;
#pragma unroll
            for (int i = 0; i < V; ++i) nv[it][j][i] = 0.f;
            if (c < d.Cin && x < d.W && y < d.H) {
                const int s = (c < c0) ? 0 : ((c < c01) ? 1 : 2);
                if (!((s == 0) ? lz[0] : ((s == 1) ? lz[1] : lz[2]))) {
                    const int cl = (s == 0) ? c : ((s == 1) ? c - c0 : c - c01);
                    const float* q = ((s == 0) ? sb[0] : ((s == 1) ? sb[1] : sb[2])) + cl * HW + (unsigned)(y * d.W + x);
                    if (V == 4) {
                        const f32x4 tq = *(const f32x4*)q;
                        nv[it][j][0] = tq[0]; nv[it][j][V > 1 ? 1 : 0] = tq[1]; nv[it][j][V > 2 ? 2 : 0] = tq[2]; nv[it][j][V > 3 ? 3 : 0] = tq[3];
                    } else {
                        nv[it][j][0] = q[0];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (!sbase[j]) continue;
        const int c = cg * 8 + j;
        const int s = (c < c0) ? 0 : ((c < c01) ? 1 : 2);
        const int cn = (s == 0) ? cnt[0] : ((s == 1) ? cnt[1] : cnt[2]);
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (tid + 256 * u < cn) s_dyn[j * a.cap + tid + 256 * u] = st[j][u];
        if (cn > 512) {                                           // tiles beyond 512 source pixels (ratios near 1)
            const int tws = (s == 0) ? tw[0] : ((s == 1) ? tw[1] : tw[2]);
            const int y0s = (s == 0) ? ys0[0] : ((s == 1) ? ys0[1] : ys0[2]);
            const int x0s = (s == 0) ? xs0[0] : ((s == 1) ? xs0[1] : xs0[2]);
            const int sws = (s == 0) ? a.sw[0] : ((s == 1) ? a.sw[1] : a.sw[2]);
            for (int e = tid + 512; e < cn; e += 256) {
                const int r = e / tws, q = e - r * tws;
                s_dyn[j * a.cap + e] = sbase[j][(y0s + r) * sws + x0s + q];
            }
        }
    }
    __syncthreads();
    if (x >= d.W) return;
    int ox0[3][V], ox1[3][V];
    float lx[3][V];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int i = 0; i < V; ++i) {
            ox0[s][i] = ox1[s][i] = 0; lx[s][i] = 0.f;
            if (lz[s]) {
                int x0, x1;
                cd_resize_src(x + i, a.sx[s], a.sw[s], a.align[s], x0, x1, lx[s][i]);
                ox0[s][i] = x0 - xs0[s]; ox1[s][i] = x1 - xs0[s];
            }
        }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const int y = yb + ty + RP * it;
        if (y >= d.H) break;
        int r0[3], r1[3];
        float ly[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            r0[s] = r1[s] = 0; ly[s] = 0.f;
            if (lz[s]) {
                int y0, y1;
                cd_resize_src(y, a.sy[s], a.sh[s], a.align[s], y0, y1, ly[s]);
                r0[s] = (y0 - ys0[s]) * tw[s]; r1[s] = (y1 - ys0[s]) * tw[s];
            }
        }
        float v[8][V];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cg * 8 + j;
#pragma unroll
            for (int i = 0; i < V; ++i) v[j][i] = 0.f;
            if (c < d.Cin) {
                const int s = (c < c0) ? 0 : ((c < c01) ? 1 : 2);
                if ((s == 0) ? lz[0] : ((s == 1) ? lz[1] : lz[2])) {
                    const float* t = s_dyn + j * a.cap;
                    const float lys = (s == 0) ? ly[0] : ((s == 1) ? ly[1] : ly[2]);
                    const int r0s = (s == 0) ? r0[0] : ((s == 1) ? r0[1] : r0[2]);
                    const int r1s = (s == 0) ? r1[0] : ((s == 1) ? r1[1] : r1[2]);
                    const float hy = 1.f - lys;
#pragma unroll
                    for (int i = 0; i < V; ++i) {
                        const float lxs = (s == 0) ? lx[0][i] : ((s == 1) ? lx[1][i] : lx[2][i]);
                        const int o0 = (s == 0) ? ox0[0][i] : ((s == 1) ? ox0[1][i] : ox0[2][i]);
                        const int o1 = (s == 0) ? ox1[0][i] : ((s == 1) ? ox1[1][i] : ox1[2][i]);
                        const float hx = 1.f - lxs;
                        v[j][i] = hy * (hx * t[r0s + o0] + lxs * t[r0s + o1]) + lys * (hx * t[r1s + o0] + lxs * t[r1s + o1]);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < V; ++i) v[j][i] = nv[it][j][i];
                }
            }
        }
        unsigned char* o = a.b.out + cd_item_off((long)n * d.G + g, a.b.ngroups8, cg, HW, (long)y * d.W + x, a.b.split);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            *(u32x4*)(o + i * 16) = cd_item8(v[0][i], v[1][i], v[2][i], v[3][i], v[4][i], v[5][i], v[6][i], v[7][i], false);
            if (a.b.split) *(u32x4*)(o + HW * 16 + i * 16) = cd_item8(v[0][i], v[1][i], v[2][i], v[3][i], v[4][i], v[5][i], v[6][i], v[7][i], true);
        }
    }
}

// Channel groups that lie inside ONE up-sampled source (the bulk of a decoder's input: 256-512 channels of `up(net)` next to a few
// label / skip channels) on a kernel of their own.  conv_pack_input_lazy_lds_kernel serves any mix of three sources, pays for it
// with per-channel source selection, and reads its four taps per output value as four 4-byte LDS loads (instruction mix in
// profiles/round3_k_pmc_instmix.txt: 912 VALU + 877 SALU + 95 LDS instructions per wave for 32 outputs per lane, 1 TB/s on the
// 256-channel 128 -> 256 layer).  Here the staged source tile is CHANNEL-INNERMOST, [position][8 channels] fp32, so a tap of all
// eight channels is two 16-byte LDS reads and the lane's four pixels take 32 reads for 32 outputs; the 16-byte chunks are
// XOR-swizzled so that lanes two source columns apart (2x up-sampling) fall into different banks.  Same expression tree
// hy (hx a + lx b) + ly (hx c + lx d) as the other two kernels.  A lane = 4 consecutive pixels of one row; the 1024-pixel tile is as
// wide as the image allows (LX lanes along x: 256 x 4, 128 x 8 or 64 x 16 outputs), so that a workgroup's stores are few long runs
// of the plane (a whole 16 KB at W = 256) instead of sixteen 1 KB pieces.
struct PackUpArgs {
    const float* src;          // the source's tensor [N][ctot][sh][sw]
    unsigned char* out;
    int ctot, coff, gstride;   // its channel geometry (jaf_conv_desc.src_*)
    int cbase;                 // concatenated channel of the source's channel 0
    int cend;                  // channels >= cend of a group are padding (zeros)
    int cg0, cgn;              // channel groups [cg0, cg0 + cgn) of the ngroups8 planes per (image, group)
    int ngroups8, G, H, W, sh, sw, align, split;
    float sy, sx;
};

__device__ __forceinline__ int pu_chunk(int pos, int h) {       // 16-byte chunk of (position, channel half) in the staged tile
    const int c = pos * 2 + h;
    return c ^ ((c >> 4) & 3);
}

template <int LX>
__global__ __launch_bounds__(256) void conv_pack_up_kernel(const PackUpArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_up[];      // [position][8] fp32, chunk-swizzled
    constexpr int TW = LX * 4, TH = 256 / LX;
    const int tid = threadIdx.x;
    const int tx = tid % LX, ty = tid / LX;
    const int xb = blockIdx.x * TW, yb = blockIdx.y * TH;
    const int x = xb + tx * 4, y = yb + ty;
    int z = blockIdx.z;
    const int cg = a.cg0 + z % a.cgn;
    z /= a.cgn;
    const int g = z % a.G;
    const int n = z / a.G;
    const long HW = (long)a.H * a.W;
    const long pl = (long)a.sh * a.sw;
    const int xl = min(xb + TW - 1, a.W - 1), yl = min(yb + TH - 1, a.H - 1);
    int ys0, xs0, tw, th;
    {
        int i0, i1; float l;
        cd_resize_src(yb, a.sy, a.sh, a.align, ys0, i1, l);
        cd_resize_src(xb, a.sx, a.sw, a.align, xs0, i1, l);
        cd_resize_src(xl, a.sx, a.sw, a.align, i0, i1, l);
        tw = i1 - xs0 + 1;
        cd_resize_src(yl, a.sy, a.sh, a.align, i0, i1, l);
        th = i1 - ys0 + 1;
    }
    // stage: position e = (row, column) of the tile's source region, 8 channels per lane and position; two positions' loads
    // (16) are in flight before the first LDS store
    const int npos = tw * th;
    const float inv_tw = 1.0f / (float)tw;
    const int cl0 = cg * 8 - a.cbase;                   // source-local channel of this group's channel 0
    const float* sbase = a.src + ((long)n * a.ctot + a.coff + (long)g * a.gstride + cl0) * pl;
    const int nch = min(8, a.cend - cg * 8);            // live channels of this group
    for (int e0 = tid; e0 < npos; e0 += 512) {
        float v[2][8];
        int pe[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = e0 + 256 * u;
            pe[u] = e < npos ? e : -1;
            int r = (int)(((float)e + 0.5f) * inv_tw);
            int q = e - r * tw;
            if (q < 0) { --r; q += tw; }
            if (q >= tw) { ++r; q -= tw; }
            const float* p = sbase + (long)(ys0 + r) * a.sw + xs0 + q;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[u][j] = (pe[u] >= 0 && j < nch) ? p[j * pl] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (pe[u] >= 0) {
                *(f32x4*)(s_up + pu_chunk(pe[u], 0) * 4) = (f32x4){v[u][0], v[u][1], v[u][2], v[u][3]};
                *(f32x4*)(s_up + pu_chunk(pe[u], 1) * 4) = (f32x4){v[u][4], v[u][5], v[u][6], v[u][7]};
            }
    }
    __syncthreads();
    if (x >= a.W || y >= a.H) return;
    int r0, r1;
    float ly;
    {
        int y0, y1;
        cd_resize_src(y, a.sy, a.sh, a.align, y0, y1, ly);
        r0 = (y0 - ys0) * tw - xs0;
        r1 = (y1 - ys0) * tw - xs0;
    }
    const float hy = 1.f - ly;
    unsigned char* o = a.out + cd_item_off((long)n * a.G + g, a.ngroups8, cg, HW, (long)y * a.W + x, a.split);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int x0, x1;
        float lx;
        cd_resize_src(x + i, a.sx, a.sw, a.align, x0, x1, lx);
        const float hx = 1.f - lx;
        // (the second half of a position is the neighbouring chunk: pu_chunk(p, 1) == pu_chunk(p, 0) ^ 1)
        const int ia = pu_chunk(r0 + x0, 0), ib = pu_chunk(r0 + x1, 0), ic = pu_chunk(r1 + x0, 0), id = pu_chunk(r1 + x1, 0);
        float v[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 ta = *(const f32x4*)(s_up + (ia ^ h) * 4), tb = *(const f32x4*)(s_up + (ib ^ h) * 4);
            const f32x4 tc = *(const f32x4*)(s_up + (ic ^ h) * 4), td = *(const f32x4*)(s_up + (id ^ h) * 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[4 * h + k] = hy * (hx * ta[k] + lx * tb[k]) + ly * (hx * tc[k] + lx * td[k]);
        }
        *(u32x4*)(o + i * 16) = cd_item8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], false);
        if (a.split) *(u32x4*)(o + HW * 16 + i * 16) = cd_item8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], true);
    }
}

extern "C" int jaf_conv2d_pack_input_resized(jaf_stream_t s, const jaf_conv_desc* d, const float* src0, const float* src1,
                                             const float* src2, const int32_t* src_h, const int32_t* src_w,
                                             const int32_t* align_corners, void* packed) {
    JAF_REQUIRE(pack_desc_ok(d) && src0 && packed && src_h && src_w && align_corners);
    JAF_REQUIRE(d->nsrc < 2 || src1);
    JAF_REQUIRE(d->nsrc < 3 || src2);
    PackLazyArgs a;
    a.b.src[0] = src0; a.b.src[1] = src1; a.b.src[2] = src2;
    a.b.out = (unsigned char*)packed;
    a.b.d = *d;
    a.b.ngroups8 = jaf_cdiv(d->Cin, 8);
    a.b.split = d->precision == JAF_PREC_BF16X3 ? 1 : 0;
    for (int i = 0; i < 3; ++i) {
        a.lazy[i] = (i < d->nsrc && src_h[i] > 0) ? 1 : 0;
        a.sh[i] = a.lazy[i] ? src_h[i] : 1;
        a.sw[i] = a.lazy[i] ? src_w[i] : 1;
        a.align[i] = a.lazy[i] ? (align_corners[i] ? 1 : 0) : 0;
        JAF_REQUIRE(!a.lazy[i] || (src_w[i] > 0 && (long)src_h[i] * src_w[i] < (1L << 30)));
        if (a.align[i]) {
            a.sy[i] = d->H > 1 ? (float)(a.sh[i] - 1) / (float)(d->H - 1) : 0.f;
            a.sx[i] = d->W > 1 ? (float)(a.sw[i] - 1) / (float)(d->W - 1) : 0.f;
        } else {
            a.sy[i] = (float)a.sh[i] / (float)d->H;
            a.sx[i] = (float)a.sw[i] / (float)d->W;
        }
    }
    const long nz = (long)d->N * d->G * a.b.ngroups8;
    if (nz > 65535 || d->H > 65535) return JAF_EUNSUPPORTED;
    // the LDS-staged form when the image is at least a tile wide and every lazy source tile fits (up-sampling: it always does)
    bool staged = d->W >= 48;
    long cap = 32;
    for (int i = 0; i < 3; ++i)
        if (a.lazy[i]) {
            const long th = (long)ceilf(a.sy[i] * PL_TH) + 3, tw = (long)ceilf(a.sx[i] * PL_TW) + 3;
            if (th * tw > cap) cap = th * tw;
        }
    if (cap > PL_CAP) staged = false;
    const bool al = ((((uintptr_t)src0) | ((uintptr_t)src1) | ((uintptr_t)src2)) & 15) == 0;
    // Channel groups inside one resized source: conv_pack_up_kernel; the others (label / skip channels, groups that straddle two
    // sources): the general kernels over the remaining group ranges.  JAF_PACK_UP=0 sends everything through the general kernels.
    const int up_env = 1;
    const bool up_ok = up_env && staged && d->W % 4 == 0;
    const int cb[4] = {0, d->src_c[0], d->src_c[0] + (d->nsrc > 1 ? d->src_c[1] : 0), d->Cin};
    const float* srcs[3] = {src0, src1, src2};
    auto cls = [&](int cg) {      // the resized source that holds all live channels of group cg, or -1
        if (!up_ok) return -1;
        const int lo = cg * 8, hi = (cg * 8 + 8 < d->Cin ? cg * 8 + 8 : d->Cin) - 1;
        for (int i = 0; i < d->nsrc; ++i) {
            const int e = (i + 1 < d->nsrc) ? cb[i + 1] : d->Cin;
            if (a.lazy[i] && lo >= cb[i] && hi < e) return i;
        }
        return -1;
    };
    for (int cg0 = 0; cg0 < a.b.ngroups8;) {
        const int c = cls(cg0);
        int cg1 = cg0 + 1;
        while (cg1 < a.b.ngroups8 && cls(cg1) == c) ++cg1;
        const int cgn = cg1 - cg0;
        const unsigned gz = (unsigned)((long)d->N * d->G * cgn);
        if (c >= 0) {
            PackUpArgs u;
            u.src = srcs[c];
            u.out = (unsigned char*)packed;
            u.ctot = d->src_ctot[c]; u.coff = d->src_coff[c]; u.gstride = d->src_gstride[c];
            u.cbase = cb[c];
            u.cend = d->Cin;
            u.cg0 = cg0; u.cgn = cgn;
            u.ngroups8 = a.b.ngroups8; u.G = d->G; u.H = d->H; u.W = d->W;
            u.sh = a.sh[c]; u.sw = a.sw[c]; u.align = a.align[c]; u.split = a.b.split;
            u.sy = a.sy[c]; u.sx = a.sx[c];
            int lx = d->W >= 192 ? 64 : (d->W >= 96 ? 32 : 16);
            long th = 0, tw = 0;
            for (;; lx >>= 1) {          // (a wide tile of a source that is not up-sampled may not fit the staging buffer)
                th = (long)ceilf(u.sy * (256 / lx)) + 3;
                tw = (long)ceilf(u.sx * (lx * 4)) + 3;
                if (th * tw <= PL_CAP || lx == 16) break;
            }
            const size_t lds = (size_t)((th * tw + 7) / 8 * 8) * 32;      // [position][8] fp32; <= 40 KB (PL_CAP)
            const dim3 grid(jaf_cdiv(d->W, lx * 4), jaf_cdiv(d->H, 256 / lx), gz);
            JAF_NOTE_KERNEL("conv_pack_up_kernel<%d>", lx);
            if (lx == 64) hipLaunchKernelGGL(conv_pack_up_kernel<64>, grid, dim3(256), lds, (hipStream_t)s, u);
            else if (lx == 32) hipLaunchKernelGGL(conv_pack_up_kernel<32>, grid, dim3(256), lds, (hipStream_t)s, u);
            else hipLaunchKernelGGL(conv_pack_up_kernel<16>, grid, dim3(256), lds, (hipStream_t)s, u);
        } else {
            a.cg0 = cg0;
            a.cgn = cgn;
            if (staged) {
                a.cap = (int)((cap + 31) / 32 * 32);
                const size_t lds = (size_t)8 * a.cap * sizeof(float);       // <= 40 KB
                const dim3 grid(jaf_cdiv(d->W, PL_TW), jaf_cdiv(d->H, PL_TH), gz);
                JAF_NOTE_KERNEL("conv_pack_input_lazy_lds_kernel<%d>", (al && d->W % 4 == 0) ? 4 : 1);
                if (al && d->W % 4 == 0) hipLaunchKernelGGL(conv_pack_input_lazy_lds_kernel<4>, grid, dim3(256), lds, (hipStream_t)s, a);
                else hipLaunchKernelGGL(conv_pack_input_lazy_lds_kernel<1>, grid, dim3(256), lds, (hipStream_t)s, a);
            } else {
                a.cap = 0;
                int tx = 64;
                while (tx > 8 && (tx >> 1) >= d->W) tx >>= 1;
                const int ty = 256 / tx;
                JAF_NOTE_KERNEL("conv_pack_input_lazy_kernel");
                hipLaunchKernelGGL(conv_pack_input_lazy_kernel, dim3(jaf_cdiv(d->W, tx), jaf_cdiv(d->H, ty), gz), dim3(tx, ty), 0, (hipStream_t)s, a);
            }
        }
        const int rc = jaf_launch_status();
        if (rc != JAF_OK) return rc;
        cg0 = cg1;
    }
    return JAF_OK;
}

// ---------------------------------------------------------------------------------------------
// dz packing for the backward pass: dz = dy * act'(y) (activation backward), its packed bf16 image for
// the data / weight gradient kernels, the bias gradient (per-channel sum of dz) and -- only when a
// non-packed kernel still needs it -- the fp32 dz, all in ONE pass over dy (and y).
// Replaces act_bwd + channel_sum + pack_input (three passes) for every packed layer.
// grid (x blocks, row blocks, N*G*ngroups8), block (tx, ty) = 256 threads of one (image, group, group8).
// ---------------------------------------------------------------------------------------------
struct PackDzArgs {
    const float* dy;
    const float* y;        // activation output (nullable when act == NONE)
    const unsigned char* yp;   // ... or the same values as the packed bf16 image the forward epilogue wrote (jaf_packed_io)
    int yp_ng8, yp_cg0;        // its planes per (image, group) and the first plane of this tensor's channels
    unsigned char* out;    // packed dz
    float* dz;             // fp32 dz (nullable)
    float* dbias;          // [G*C] += sum over n, pixels (nullable)
    int N, G, C, H, W, ngroups8, act;
    float slope;
    int split;             // split-bf16 image (hi and lo planes per channel group): see PackInArgs
    int dy_bf16;           // dy holds bf16 (the gradient of a bf16-stored tensor: jaf_conv2d_pack_dz_dt)
    int yp_split;          // the sign image `yp` is a split-bf16 image (hi planes read) whatever `split` says of the output
};

__device__ __forceinline__ float dz_of(float g, float yv, int act, float slope) {
    switch (act) {
        case JAF_ACT_LRELU: return g * (yv > 0.f ? 1.f : slope);
        case JAF_ACT_RELU: return g * (yv > 0.f ? 1.f : 0.f);
        case JAF_ACT_SIGMOID: return g * yv * (1.f - yv);
        case JAF_ACT_TANH: return g * (1.f - yv * yv);
        default: return g;
    }
}

template <int V>
__global__ __launch_bounds__(256) void conv_pack_dz_kernel(const PackDzArgs a) {
    const int x = (blockIdx.x * blockDim.x + threadIdx.x) * V;
    const int yy = blockIdx.y * blockDim.y + threadIdx.y;
    int z = blockIdx.z;
    const int cg = z % a.ngroups8;
    z /= a.ngroups8;
    const int g = z % a.G;
    const int n = z / a.G;
    const bool live = (x < a.W) && (yy < a.H);
    const long HW = (long)a.H * a.W;
    float v[8][V];
    float part[8];
    float ypk[8][V];       // y from the packed image: item (pixel i) holds this lane's 8 channels
    if (a.yp && live) {
        const unsigned char* ip = a.yp + cd_item_off(((long)n * a.G + g) * a.yp_ng8 + a.yp_cg0, 1, cg, HW, (long)yy * a.W + x, a.yp_split);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            const u32x4 w = *(const u32x4*)(ip + i * 16);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                ypk[2 * u][i] = __builtin_bit_cast(float, w[u] << 16);
                ypk[2 * u + 1][i] = __builtin_bit_cast(float, w[u] & 0xffff0000u);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cg * 8 + j;
        part[j] = 0.f;
#pragma unroll
        for (int i = 0; i < V; ++i) v[j][i] = 0.f;
        if (live && c < a.C) {
            const long e = ((long)n * a.G * a.C + (long)g * a.C + c) * HW + (long)yy * a.W + x;
            if (V == 4) {
                f32x4 gq;
                if (a.dy_bf16) {
                    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                    gq = __builtin_convertvector(*(const bf16x4*)((const __bf16*)a.dy + e), f32x4);
                } else {
                    gq = *(const f32x4*)(a.dy + e);
                }
                f32x4 yq = {0.f, 0.f, 0.f, 0.f};
                if (a.yp) { yq[0] = ypk[j][0]; yq[1] = ypk[j][V > 1 ? 1 : 0]; yq[2] = ypk[j][V > 2 ? 2 : 0]; yq[3] = ypk[j][V > 3 ? 3 : 0]; }
                else if (a.act != JAF_ACT_NONE) yq = *(const f32x4*)(a.y + e);
                f32x4 o;
                o[0] = dz_of(gq[0], yq[0], a.act, a.slope);
                o[1] = dz_of(gq[1], yq[1], a.act, a.slope);
                o[2] = dz_of(gq[2], yq[2], a.act, a.slope);
                o[3] = dz_of(gq[3], yq[3], a.act, a.slope);
                v[j][0] = o[0]; v[j][1] = o[1]; v[j][2] = o[2]; v[j][3] = o[3];
                if (a.dz) *(f32x4*)(a.dz + e) = o;
                part[j] = (o[0] + o[1]) + (o[2] + o[3]);
            } else {
                const float o = dz_of(a.dy_bf16 ? (float)((const __bf16*)a.dy)[e] : a.dy[e], a.yp ? ypk[j][0] : (a.act != JAF_ACT_NONE ? a.y[e] : 0.f),
                                      a.act, a.slope);
                v[j][0] = o;
                if (a.dz) a.dz[e] = o;
                part[j] = o;
            }
        }
    }
    if (live) {
        unsigned char* o = a.out + cd_item_off((long)n * a.G + g, a.ngroups8, cg, HW, (long)yy * a.W + x, a.split);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            *(u32x4*)(o + i * 16) = cd_item8(v[0][i], v[1][i], v[2][i], v[3][i], v[4][i], v[5][i], v[6][i], v[7][i], false);
            if (a.split) *(u32x4*)(o + HW * 16 + i * 16) = cd_item8(v[0][i], v[1][i], v[2][i], v[3][i], v[4][i], v[5][i], v[6][i], v[7][i], true);
        }
    }
    if (a.dbias) {
        __shared__ float red[4][8];
        const int t = threadIdx.y * blockDim.x + threadIdx.x;
#pragma unroll
        for (int j = 0; j < 8; ++j) part[j] = jaf_wave_sum(part[j]);
        if ((t & 63) == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) red[t >> 6][j] = part[j];
        }
        __syncthreads();
        if (t < 8) {
            const int c = cg * 8 + t;
            if (c < a.C) atomicAdd(&a.dbias[g * a.C + c], (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]));
        }
    }
}

extern "C" int jaf_conv2d_pack_dz(jaf_stream_t s, const float* dy, const float* y, int32_t N, int32_t G, int32_t C,
                                  int32_t H, int32_t W, int act, float slope, void* packed, float* dz, float* dbias) {
    return jaf_conv2d_pack_dz_ex(s, dy, y, nullptr, 0, 0, N, G, C, H, W, act, slope, packed, dz, dbias);
}

extern "C" int jaf_conv2d_pack_dz_ex(jaf_stream_t s, const float* dy, const float* y, const void* y_packed, int32_t y_ng8_tot,
                                     int32_t y_coff, int32_t N, int32_t G, int32_t C, int32_t H, int32_t W, int act, float slope,
                                     void* packed, float* dz, float* dbias) {
    return jaf_conv2d_pack_dz_prec(s, dy, y, y_packed, y_ng8_tot, y_coff, N, G, C, H, W, act, slope, packed, dz, dbias, JAF_PREC_BF16);
}

extern "C" int jaf_conv2d_pack_dz_prec(jaf_stream_t s, const float* dy, const float* y, const void* y_packed, int32_t y_ng8_tot,
                                       int32_t y_coff, int32_t N, int32_t G, int32_t C, int32_t H, int32_t W, int act, float slope,
                                       void* packed, float* dz, float* dbias, int precision) {
    return jaf_conv2d_pack_dz_dt(s, dy, 0, y, y_packed, y_ng8_tot, y_coff, N, G, C, H, W, act, slope, packed, dz, dbias, precision);
}

extern "C" int jaf_conv2d_pack_dz_dt(jaf_stream_t s, const void* dy_, int dy_bf16, const float* y, const void* y_packed, int32_t y_ng8_tot,
                                     int32_t y_coff, int32_t N, int32_t G, int32_t C, int32_t H, int32_t W, int act, float slope,
                                     void* packed, float* dz, float* dbias, int precision) {
    return jaf_conv2d_pack_dz_dt2(s, dy_, dy_bf16, y, y_packed, y_ng8_tot, y_coff, precision == JAF_PREC_BF16X3 ? 1 : 0, N, G, C, H, W, act,
                                  slope, packed, dz, dbias, precision);
}

extern "C" int jaf_conv2d_pack_dz_dt2(jaf_stream_t s, const void* dy_, int dy_bf16, const float* y, const void* y_packed, int32_t y_ng8_tot,
                                      int32_t y_coff, int y_split, int32_t N, int32_t G, int32_t C, int32_t H, int32_t W, int act, float slope,
                                      void* packed, float* dz, float* dbias, int precision) {
    const float* dy = (const float*)dy_;
    JAF_REQUIRE(dy && packed && N >= 1 && G >= 1 && C >= 1 && H >= 1 && W >= 1);
    JAF_REQUIRE(precision == JAF_PREC_BF16 || precision == JAF_PREC_BF16X3);
    JAF_REQUIRE(!dy_bf16 || precision == JAF_PREC_BF16);
    JAF_REQUIRE(act == JAF_ACT_NONE || y || y_packed);
    // y from the packed image: only its sign is used (ReLU / LeakyReLU), which bf16 rounding keeps
    JAF_REQUIRE(!y_packed || ((act == JAF_ACT_LRELU || act == JAF_ACT_RELU) && y_coff >= 0 && (y_coff & 7) == 0 &&
                              y_coff / 8 + jaf_cdiv(C, 8) <= y_ng8_tot));
    PackDzArgs a;
    a.dy = dy; a.y = y; a.out = (unsigned char*)packed; a.dz = dz; a.dbias = dbias;
    a.yp = (const unsigned char*)y_packed; a.yp_ng8 = y_ng8_tot; a.yp_cg0 = y_coff / 8;
    a.split = precision == JAF_PREC_BF16X3 ? 1 : 0;       // (then y_packed is a split image too: the sign is read from its hi planes)
    a.dy_bf16 = dy_bf16 ? 1 : 0;
    a.yp_split = y_split ? 1 : 0;
    JAF_REQUIRE(!(precision == JAF_PREC_BF16X3 && y_packed && !y_split));      // (a split output next to a plain sign image: no such caller)
    const bool al = ((((uintptr_t)dy) | ((uintptr_t)y) | ((uintptr_t)dz)) & 15) == 0;
    if (al && W % 4 != 0 && ((long)H * W) % 4 == 0 && (long)H * W < (1L << 30)) { W = H * W; H = 1; }   // as in jaf_conv2d_pack_input
    a.N = N; a.G = G; a.C = C; a.H = H; a.W = W; a.ngroups8 = jaf_cdiv(C, 8); a.act = act; a.slope = slope;
    const long nz = (long)N * G * a.ngroups8;
    if (nz > 65535 || H > 65535) return JAF_EUNSUPPORTED;
    const bool v4 = (W % 4 == 0) && al;
    const int wx = v4 ? W / 4 : W;
    int tx = H == 1 ? 256 : 64;
    while (tx > 1 && (tx >> 1) >= wx) tx >>= 1;
    if (tx < 8) tx = 8;
    const int ty = 256 / tx;
    const dim3 grid(jaf_cdiv(wx, tx), jaf_cdiv(H, ty), (unsigned)nz);
    if (v4) hipLaunchKernelGGL(conv_pack_dz_kernel<4>, grid, dim3(tx, ty), 0, (hipStream_t)s, a);
    else hipLaunchKernelGGL(conv_pack_dz_kernel<1>, grid, dim3(tx, ty), 0, (hipStream_t)s, a);
    return jaf_launch_status();
}

// ---------------------------------------------------------------------------------------------
// ConvLSTM gate backward straight into the packed image (src/convLSTM.py:48-54 adjoint): reads the
// saved gates i,f,o,g, c_{t-1}, c_t, dh, dc_{t+1}; writes dc_{t-1} (fp32), the PRE-activation gate
// gradients as a packed bf16 image for the data / weight gradient kernels, and adds their
// per-channel sums to the bias gradient.  The fp32 gate-gradient tensor (4C channels) is never
// written, re-read for packing, or re-read for the bias sum.
//
// The packed gate gradients are CHANNEL-MAJOR: packed channel 4 c + gate, so that one 16-byte item = 2 hidden channels x
// (i, f, o, g).  A lane then owns 2 hidden channels x V pixels and everything it touches is whole: 16-byte loads of the five
// fp32 planes and of the (gate-innermost) saved gates at V = 4, one 16-byte item store per pixel.  (Rounds 1-3 kept the
// gradients gate-major, gate * C + c: a whole item then needs 8 -- or, at C = 12, all 12 -- hidden channels in one lane, which
// left room for only 2 pixels, i.e. 8-byte accesses on every fp32 plane: 5.0 TB/s.)  The consumers follow the order:
// jaf_conv2d_pack(JAF_PACK_DGRAD_LSTM) permutes the reduction channels of the data gradient's weight image,
// jaf_conv2d_wgrad_packed_lstm the rows of dW.
// grid (pixel blocks, C/2, N*G), block 256.
// ---------------------------------------------------------------------------------------------
// DT: element type of dh (fp32: the gradient autograd hands in for the last step; bf16: the d h_{t-1} the data-gradient launch of
// step t+1 wrote, jaf_packed_io.out2_bf16); ST: element type of the cell-state tensors c_prev, c_cur, dc_next, dc_prev (bf16
// storage of BASELINE configs[2]: 26 instead of 36 bytes per hidden-channel pixel).
template <int V, typename T>
__device__ __forceinline__ void lg_load(const T* __restrict__ p, float (&o)[V]) {
    if constexpr (V > 1) {
        typedef T tv __attribute__((ext_vector_type(V)));
        const tv t = *(const tv*)p;
#pragma unroll
        for (int k = 0; k < V; ++k) o[k] = (float)t[k];
    } else {
        o[0] = (float)p[0];
    }
}
template <int V, typename T>
__device__ __forceinline__ void lg_store(T* __restrict__ p, const float (&o)[V]) {
    if constexpr (V > 1) {
        typedef T tv __attribute__((ext_vector_type(V)));
        typedef float fv __attribute__((ext_vector_type(V)));
        fv f;
#pragma unroll
        for (int k = 0; k < V; ++k) f[k] = o[k];
        *(tv*)p = __builtin_convertvector(f, tv);
    } else {
        p[0] = (T)o[0];
    }
}

template <int V, typename GT, typename DT = float, typename ST = float>
__global__ __launch_bounds__(256) void lstm_gates_bwd_cmajor_kernel(int G, int C, int HW, const DT* __restrict__ dh,
                                                                    const ST* __restrict__ dc_next, const GT* __restrict__ gates,
                                                                    const ST* __restrict__ c_prev, const ST* __restrict__ c_cur,
                                                                    ST* __restrict__ dc_prev, unsigned char* __restrict__ packed,
                                                                    float* __restrict__ dbias, int iters, int split) {
    const int kp = blockIdx.y;                     // hidden-channel pair 2 kp, 2 kp + 1 = item kp of every pixel
    const long ng = blockIdx.z;
    const int g = (int)(ng % G);
    const int ng8 = C >> 1;
    const long cs = (long)C * HW;
    __shared__ float red[8];
    if (threadIdx.x < 8) red[threadIdx.x] = 0.f;
    __syncthreads();
    float sums[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) sums[k] = 0.f;
    for (int it = 0; it < iters; ++it) {
        const int pix0 = (blockIdx.x * iters + it) * 256 * V;          // workgroup-uniform
        if (pix0 >= HW) break;
        const int pix = pix0 + threadIdx.x * V;
        if (pix >= HW) continue;
        float o[4][2][V];                                               // [gate][channel of the pair][pixel]
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = 2 * kp + h;
            const long e = (ng * C + c) * (long)HW + pix;
            float gi[V], gf[V], go[V], gg[V], cc[V], dhv[V], dcn[V], cp[V];
            if constexpr (sizeof(GT) == 2) {                            // saved gates gate-innermost: [c][pixel][i, f, o, g]
                typedef GT g4v __attribute__((ext_vector_type(4 * V)));
                const g4v t = *(const g4v*)(gates + ng * 4 * cs + ((long)c * HW + pix) * 4);
#pragma unroll
                for (int k = 0; k < V; ++k) { gi[k] = (float)t[4 * k]; gf[k] = (float)t[4 * k + 1]; go[k] = (float)t[4 * k + 2]; gg[k] = (float)t[4 * k + 3]; }
            } else {                                                    // fp32 gates: planes [gate][c][pixel]
                const GT* gp = gates + (ng * 4 * C + c) * (long)HW + pix;
#pragma unroll
                for (int k = 0; k < V; ++k) { gi[k] = (float)gp[k]; gf[k] = (float)gp[cs + k]; go[k] = (float)gp[2 * cs + k]; gg[k] = (float)gp[3 * cs + k]; }
            }
            lg_load<V, ST>(c_cur + e, cc);
            lg_load<V, DT>(dh + e, dhv);
            if (dc_next) lg_load<V, ST>(dc_next + e, dcn);
            if (c_prev) lg_load<V, ST>(c_prev + e, cp);
            float dcp[V];
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float tc = jaf_tanh(cc[k]);
                float dc = dhv[k] * go[k] * (1.f - tc * tc);
                if (dc_next) dc += dcn[k];
                const float cpv = c_prev ? cp[k] : 0.f;
                o[0][h][k] = dc * gg[k] * gi[k] * (1.f - gi[k]);
                o[1][h][k] = dc * cpv * gf[k] * (1.f - gf[k]);
                o[2][h][k] = dhv[k] * tc * go[k] * (1.f - go[k]);
                o[3][h][k] = dc * gi[k] * (1.f - gg[k] * gg[k]);
                dcp[k] = dc * gf[k];
            }
            lg_store<V, ST>(dc_prev + e, dcp);
        }
        unsigned char* op = packed + cd_item_off(ng, ng8, kp, HW, pix, split);        // (split-bf16: hi plane, lo plane behind it)
#pragma unroll
        for (int k = 0; k < V; ++k) {
            *(u32x4*)(op + k * 16) = cd_item8(o[0][0][k], o[1][0][k], o[2][0][k], o[3][0][k], o[0][1][k], o[1][1][k], o[2][1][k], o[3][1][k], false);
            if (split)
                *(u32x4*)(op + (long)HW * 16 + k * 16) = cd_item8(o[0][0][k], o[1][0][k], o[2][0][k], o[3][0][k], o[0][1][k], o[1][1][k], o[2][1][k], o[3][1][k], true);
#pragma unroll
            for (int a = 0; a < 4; ++a) { sums[2 * a] += o[a][0][k]; sums[2 * a + 1] += o[a][1][k]; }
        }
    }
    // bias gradient: 8 (gate, channel) sums per workgroup
#pragma unroll
    for (int k = 0; k < 8; ++k) sums[k] = jaf_wave_sum(sums[k]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) atomicAdd(&red[k], sums[k]);
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        const int gate = threadIdx.x >> 1, h = threadIdx.x & 1;
        atomicAdd(&dbias[g * 4 * C + gate * C + 2 * kp + h], red[threadIdx.x]);
    }
}

extern "C" int jaf_convlstm_gates_bwd_packed(jaf_stream_t s, int32_t N, int32_t G, int32_t C, int32_t HW, const float* dh,
                                             const float* dc_next, const void* gates, int gates_bf16, const float* c_prev,
                                             const float* c_cur, float* dc_prev, void* packed, float* dbias) {
    return jaf_convlstm_gates_bwd_packed_prec(s, N, G, C, HW, dh, dc_next, gates, gates_bf16, c_prev, c_cur, dc_prev, packed, dbias,
                                              JAF_PREC_BF16);
}

extern "C" int jaf_convlstm_gates_bwd_packed_prec(jaf_stream_t s, int32_t N, int32_t G, int32_t C, int32_t HW, const float* dh,
                                                  const float* dc_next, const void* gates, int gates_bf16, const float* c_prev,
                                                  const float* c_cur, float* dc_prev, void* packed, float* dbias, int precision) {
    return jaf_convlstm_gates_bwd_packed_dt(s, N, G, C, HW, dh, 0, dc_next, gates, gates_bf16, c_prev, c_cur, dc_prev, 0, packed, dbias,
                                            precision);
}

extern "C" int jaf_convlstm_gates_bwd_packed_dt(jaf_stream_t s, int32_t N, int32_t G, int32_t C, int32_t HW, const void* dh,
                                                int dh_bf16, const void* dc_next, const void* gates, int gates_bf16,
                                                const void* c_prev, const void* c_cur, void* dc_prev, int state_bf16, void* packed,
                                                float* dbias, int precision) {
    JAF_REQUIRE(dh && gates && c_cur && dc_prev && packed && dbias && N >= 1 && G >= 1 && C >= 4 && HW >= 1);
    JAF_REQUIRE(precision == JAF_PREC_BF16 || precision == JAF_PREC_BF16X3);
    JAF_REQUIRE(!(dh_bf16 || state_bf16) || (precision == JAF_PREC_BF16 && gates_bf16));
    const int split = precision == JAF_PREC_BF16X3 ? 1 : 0;
    if (C % 4) return JAF_EUNSUPPORTED;
    JAF_REQUIRE(C / 2 <= 65535 && (long)N * G <= 65535);
    const uintptr_t al = ((uintptr_t)dh) | ((uintptr_t)gates) | ((uintptr_t)c_cur) | ((uintptr_t)dc_prev) |
                         ((uintptr_t)dc_next) | ((uintptr_t)c_prev);
    // (bf16 tensors: V elements are 2 V bytes, so the fp32 alignment conditions more than cover them)
    const bool v4 = (HW % 4 == 0) && (al & 15) == 0;
    const bool v2 = (HW % 2 == 0) && (al & 7) == 0;
    const int V = v4 ? 4 : (v2 ? 2 : 1);
    const int per_block = 256 * V;
    // one pixel block per workgroup: measured 0.63 / 0.31 / 0.155 ms at the 200 / 100 / 50 levels against 0.68 / 0.34-0.45 / 0.17
    // with 4-8 blocks per workgroup (fewer bias atomics, but fewer and unevenly loaded workgroups)
    const int iters = 1;
    const dim3 grid(jaf_cdiv(HW, per_block * iters), C / 2, N * G);
#define JAF_LGC(V_, T_, D_, S_)                                                                                  \
    hipLaunchKernelGGL((lstm_gates_bwd_cmajor_kernel<V_, T_, D_, S_>), grid, dim3(256), 0, (hipStream_t)s, G, C, HW, (const D_*)dh, \
                       (const S_*)dc_next, (const T_*)gates, (const S_*)c_prev, (const S_*)c_cur, (S_*)dc_prev, (unsigned char*)packed, \
                       dbias, iters, split)
#define JAF_LGV(T_, D_, S_) do { if (v4) JAF_LGC(4, T_, D_, S_); else if (v2) JAF_LGC(2, T_, D_, S_); else JAF_LGC(1, T_, D_, S_); } while (0)
    if (state_bf16) { if (dh_bf16) JAF_LGV(__bf16, __bf16, __bf16); else JAF_LGV(__bf16, float, __bf16); }
    else if (dh_bf16) return JAF_EUNSUPPORTED;          // (a bf16 dh only arises beside a bf16 state)
    else if (gates_bf16) JAF_LGV(__bf16, float, float);
    else JAF_LGV(float, float, float);
#undef JAF_LGV
#undef JAF_LGC
    return jaf_launch_status();
}

// ---------------------------------------------------------------------------------------------
// the convolution kernel
// ---------------------------------------------------------------------------------------------
// Minimum resident workgroups per CU the register allocator must leave room for (256 threads = one wave per SIMD each, so
// k workgroups = k waves per SIMD = at most 512 / k registers per lane).  Without it the allocator spreads: <4,4> plain took
// 148 registers (3 waves per SIMD) where 110 do.
#ifndef CD_FORCE_SPLIT
#define CD_FORCE_SPLIT 0      // probe builds: the CD_FORCE_* hooks apply to the split-bf16 plans instead of the bf16 ones
#endif
#ifndef CD_WS_CAND
#define CD_WS_CAND 3      // k-steps per weight sub-load the planner may choose (jaf_conv_plan.pf)
#endif
#ifndef CD_MIN_WG
#define CD_MIN_WG(MT, NT, LSTM, DZ, PLAIN) (((NT) == 4 && (MT) >= 3 && !((LSTM) && (MT) == 4)) ? 4 : 1)
#endif
template <int MT, int NT, bool LSTM, bool DZ = false, bool PLAIN = false>
__global__ __launch_bounds__(256, CD_MIN_WG(MT, NT, LSTM, DZ, PLAIN)) void conv_dma_kernel(const ConvDArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const jaf_conv_desc& d = a.d;
    const jaf_conv_plan& P = a.p;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15;
    const int q = lane >> 4;
    constexpr int MR = 16 * MT;
    const int NG = P.NG;
    const int npos = P.npos, plane = P.plane, PW = P.PW, PWp = P.PWp;
    // Pixel interleave: lane li of tile nt owns pixel NT*li + nt of the wave's 16*NT pixels, so a lane's NT
    // accumulators of one output channel are NT consecutive pixels (one vector store).  The patch rows are
    // de-interleaved into NT column classes (slot = row*PWp + (x % NT)*PWq + x / NT) so that the 16 lanes
    // of a tile still read 16 consecutive 16-byte slots; with DMA staging that is only a different
    // slot -> source mapping.
    const int lg = a.ilv ? (NT == 4 ? 2 : (NT == 2 ? 1 : 0)) : 0;
    const int cmask = (1 << lg) - 1;
    const int PWq = PWp >> lg;

    unsigned char* s_patch = smem;
    unsigned char* s_w = smem + a.off_w;
    int* s_tab = (int*)(smem + a.off_tab);

    // ---- block -> (row block, pixel tile, image, group), XCD-contiguous ----
    int L;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, j = bid >> 3, qn = nblk >> 3, rn = nblk & 7;
        L = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + j;
    }
    // (divisions by launch constants: multiply-high + shift, jaf_fdiv.h)
    const int Lm = (int)jaf_fdiv_q((unsigned)L, a.dv_mblocks);
    const int mb = L - Lm * P.mblocks;
    const int ngi = (int)jaf_fdiv_q((unsigned)Lm, a.dv_ntiles);
    const int tile = Lm - ngi * a.ntiles;
    const int n = (int)jaf_fdiv_q((unsigned)ngi, a.dv_G);
    const int g = ngi - n * d.G;
    const int tb = (int)jaf_fdiv_q((unsigned)tile, a.dv_tiles_x);
    const int tx = tile - tb * P.tiles_x;
    const int x0 = tx * P.TWIN;
    const int pbase = tb * (64 * NT);
    const int oy0 = (int)jaf_fdiv_q((unsigned)pbase, a.dv_twin);
    const int iy0 = oy0 * d.stride - d.pad_t;
    const int ix0 = x0 * d.stride - d.pad_l;
    const int OHW = d.OH * d.OW;
    const int HW = d.H * d.W;

    // ---- one-time table: (slot, tile nt) -> patch byte offset; [0]: full chunk, [1]: last chunk ----
    {
        const int taps = d.KH * d.KW;
        const float inv_kw = a.inv_kw;
        for (int e = tid; e < 2 * 16 * P.nsteps; e += 256) {
            const int nt = e & 3;
            int s = e >> 2;
            const int which = s >= 4 * P.nsteps;
            s -= which * 4 * P.nsteps;
            const int ngc = which ? P.ng_last : NG;
            int v = 0;
            if (s < taps * ngc) {
                const int tap = (int)(((float)s + 0.5f) * (which ? a.inv_ng_last : a.inv_ng)), grp = s - __mul24(tap, ngc);
                const int ky = (int)(((float)tap + 0.5f) * inv_kw), kx = tap - __mul24(ky, d.KW);
                const int xk = (a.ilv ? d.stride * nt : 0) + kx;
                v = __mul24(grp, plane) + (__mul24(ky, PWp) + __mul24(xk & cmask, PWq) + (xk >> lg)) * 16;
            }
            s_tab[e] = v;
        }
    }

    // ---- per-lane output pixels ----
    int boff[NT];
    int opix[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int p = a.ilv ? (pbase + wave * 16 * NT + li * NT + nt) : (pbase + (wave * NT + nt) * 16 + li);
        const int oy = (int)(((float)p + 0.5f) * a.inv_twin);
        const int oxr = p - __mul24(oy, P.TWIN);
        const int ox = x0 + oxr;
        const bool valid = (oy < d.OH) && (ox < d.OW);
        boff[nt] = valid ? ((__mul24(oy - oy0, d.stride * PWp) + __mul24(oxr >> lg, d.stride)) * 16) : 0;
        opix[nt] = valid ? (__mul24(oy, d.OW) + ox) : -1;
    }

    // ---- DMA source offsets: wave w fills rounds w and w+4 (64 slots each) of every group plane ----
    const int nrounds = (npos + 63) >> 6;
    int dvoff[CD_RPW];
    {
        const int dil = d.dil_in;
        const int Hd = (d.H - 1) * dil + 1;
        const int Wd = (d.W - 1) * dil + 1;
#pragma unroll
        for (int j = 0; j < CD_RPW; ++j) {
            const int slot = lane + 64 * (wave + 4 * j);
            const int r = (int)(((float)slot + 0.5f) * a.inv_pwp);
            const int rem = slot - __mul24(r, PWp);
            const int cls = (int)(((float)rem + 0.5f) * a.inv_pwq);
            const int x = ((rem - __mul24(cls, PWq)) << lg) + cls;               // patch column held by this slot
            const int iyd = iy0 + r, ixd = ix0 + x;
            bool ok = (slot < npos) && (x < PW) && (iyd >= 0) && (ixd >= 0) && (iyd < Hd) && (ixd < Wd);
            int iy = iyd, ix = ixd;
            if (dil == 2) {
                ok = ok && !((iyd | ixd) & 1);
                iy = iyd >> 1;
                ix = ixd >> 1;
            }
            dvoff[j] = ok ? ((__mul24(iy, d.W) + ix) * 16) : CD_OOB;
        }
    }
    // plane (group8 = 0) of this (image, group); consecutive group8 planes are HW*16 bytes apart
    const unsigned char* xbase = a.xp + (((long)n * d.G + g) * a.in_ng8) * (long)HW * 16;
    const int plane_bytes = HW * 16;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ConvLSTM: the previous cell state of this lane's 4 consecutive pixels (vector epilogue, NT == 4) is requested NOW, so
    // that its HBM latency runs under the patch DMA and the matrix-core loop instead of inside the epilogue (with 3
    // workgroups per CU nothing else hides it: the 200 x 200 level sat at 3.1 TB/s)
    f32x4 cpre[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) cpre[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (LSTM && NT == 4 && a.c_prev && a.vec && opix[0] >= 0) {
        const int C = d.Cout >> 2;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int ch = ((mb * MR + mt * 16) >> 2) + q;
            if (ch < C) {
                const long co_ = (((long)n * d.G + g) * C + ch) * OHW + opix[0];
                if (a.state_bf16) {
                    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                    cpre[mt] = __builtin_convertvector(*(const bf16x4*)((const __bf16*)a.c_prev + co_), f32x4);
                } else {
                    cpre[mt] = *(const f32x4*)(a.c_prev + co_);
                }
            }
        }
    }

    const long wchunk_bytes = (long)P.nsteps * MT * 1024;
    const unsigned char* wbase = a.wpk + ((long)(g * P.mblocks + mb) * P.nchunks) * wchunk_bytes;

    // plan.pf > 0: the chunk's weights pass through LDS `pf` k-steps at a time (the patch stays): a 4-group chunk (9 exact k-steps
    // of a 3 x 3 layer, no padded half step) then needs the LDS of a 2-group one and keeps 4 workgroups per CU
    const int wsub = P.pf > 0 ? P.pf : P.nsteps;
    for (int chunk = 0; chunk < P.nchunks; ++chunk) {
        const bool last = (chunk == P.nchunks - 1);
        const int ngc = last ? P.ng_last : NG;
        const int nst = last ? P.nsteps_last : P.nsteps;
        const int* tab = s_tab + (last ? 16 * P.nsteps : 0) + q * 4;
        for (int s0 = 0; s0 < nst; s0 += wsub) {
            const int s1 = s0 + wsub < nst ? s0 + wsub : nst;
            __syncthreads();   // previous (sub-)chunk consumed (first pass: slot table visible)

            // ---- weights and patch: DMA straight into LDS.  Overlap with the matrix cores comes from the
            // 2-6 workgroups resident per CU (LDS/VGPR footprint is small); an intra-workgroup second
            // buffer measured slower than the extra resident workgroup it costs. ----
            {
                const unsigned char* wsrc = wbase + (long)chunk * wchunk_bytes + (long)s0 * MT * 1024;
                for (int e = wave; e < (s1 - s0) * MT; e += 4)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + e * 1024 + lane * 16),
                                                     (__attribute__((address_space(3))) void*)(s_w + e * 1024), 16, 0, 0);
                if (s0 == 0) {
                    const unsigned char* cbase = xbase + (long)(chunk * NG) * plane_bytes;
                    for (int grp = 0; grp < ngc; ++grp) {
                        const __amdgpu_buffer_rsrc_t rs =
                            __builtin_amdgcn_make_buffer_rsrc((void*)(cbase + (long)grp * plane_bytes), 0, plane_bytes, 0x00020000);
#pragma unroll
                        for (int j = 0; j < CD_RPW; ++j) {
                            const int round = wave + 4 * j;
                            if (round < nrounds)
                                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(s_patch + grp * plane + round * 1024),
                                                                         16, dvoff[j], 0, 0, 0);
                        }
                    }
                }
            }
            __builtin_amdgcn_s_waitcnt(0);      // vmcnt(0): the DMAs of this wave have landed
            __syncthreads();

            // ---- MFMA over the (sub-)chunk's steps ----
            u32x4 tnext = *(const u32x4*)(tab + 16 * s0);     // slot-table entry fetched one step ahead
            for (int st = s0; st < s1; ++st) {
                const u32x4 t4 = tnext;
                tnext = *(const u32x4*)(tab + 16 * (st + 1 < s1 ? st + 1 : st));
                const int off[4] = {(int)t4.x, (int)t4.y, (int)t4.z, (int)t4.w};
                bf16x8 bh[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bh[nt] = *(const bf16x8*)(s_patch + off[nt] + boff[nt]);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const bf16x8 ah = *(const bf16x8*)(s_w + ((st - s0) * MT + mt) * 1024 + lane * 16);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[nt], acc[mt][nt], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue ----
    cd_epilogue<MT, NT, LSTM, DZ, PLAIN>(a, acc, opix, n, g, mb, q, OHW, smem, cpre);
}

// ---------------------------------------------------------------------------------------------
// planning (same tiling vocabulary as conv_bf16.hip; the patch must fit 512 DMA slots)
// ---------------------------------------------------------------------------------------------
static inline int rup_d(int v, int m) { return (v + m - 1) / m * m; }

static bool cd_desc_ok(const jaf_conv_desc* d) {
    if (!pack_desc_ok(d)) return false;
    if (d->Cout < 1 || d->OH < 1 || d->OW < 1) return false;
    if (d->KH < 1 || d->KW < 1 || d->KH > 7 || d->KW > 7) return false;
    if (d->stride < 1 || d->stride > 2) return false;
    if (d->dil_in != 1 && d->dil_in != 2) return false;
    if (d->dil_in == 2 && d->stride != 1) return false;
    if (d->w_cin_off < 0 || d->w_cin_tot < 1) return false;
    if (d->out_coff < 0 || d->out_coff + d->G * d->Cout > d->out_ctot) return false;
    if (d->pad_t < 0 || d->pad_l < 0) return false;
    // (the kernels' per-lane index arithmetic uses 24-bit multiplies)
    if (d->H >= (1 << 22) || d->W >= (1 << 22) || d->OH >= (1 << 22) || d->OW >= (1 << 22)) return false;
    // JAF_PREC_BF16X3: the same kernels' split-bf16 form (conv_dma_split.hip) over hi / lo operand images
    if (d->precision != JAF_PREC_BF16 && d->precision != JAF_PREC_BF16X3) return false;
    return true;
}

extern "C" int jaf_conv2d_plan_packed(const jaf_conv_desc* d, int lstm, jaf_conv_plan* plan) {
    return jaf_conv2d_plan_packed_ex(d, lstm, 0, plan);
}

extern "C" int jaf_conv2d_plan_packed_ex(const jaf_conv_desc* d, int lstm, int flags, jaf_conv_plan* plan) {
    JAF_REQUIRE(cd_desc_ok(d) && plan);
    const bool no_ilv = (flags & JAF_PLAN_NO_INTERLEAVE) != 0;
    if (lstm) JAF_REQUIRE((d->Cout & 3) == 0 && d->KH == 3 && d->KW == 3 && d->stride == 1);
    const int M = d->Cout;
    const int taps = d->KH * d->KW;
    int mt_lo = 1, mt_hi = 4;
    if (lstm) {
        mt_lo = mt_hi = (M % 48 == 0) ? 3 : ((M % 64 == 0) ? 4 : ((M % 32 == 0) ? 2 : 1));
        JAF_REQUIRE(M % (16 * mt_lo) == 0);
    }
    const int groups = jaf_cdiv(d->Cin, 8);
    const long OHW = (long)d->OH * d->OW;
    // split-bf16 (conv_dma_split.hip): hi and lo planes of the patch and of the weight image in LDS (twice the bytes), three
    // matrix-core instructions per operand pair, two LDS reads per operand
    const bool split = d->precision == JAF_PREC_BF16X3;
    const int sb = split ? 2 : 1;

    double bestCost = 1e300;
    int bTW = 0, bNT = 0, bNG = 0, bMT = 0, bWS = 0;
    const int ngcap = 4;          // channel groups of 8 per chunk (slot table and DMA rounds are sized for <= 4)
    const int cand_tw[4] = {16, 32, 64, d->OW};
    // experiment hooks (scratch/mb_part.py): restrict the search to one NT / one tile width
    const int force_nt = 0, force_tw = 0;
    for (int cMT = mt_hi; cMT >= mt_lo; --cMT)
    for (int ci = 0; ci < 4; ++ci) {
        const int TW = cand_tw[ci];
        if (ci < 3 && TW >= d->OW) continue;
        if (force_tw && d->G >= 8 && ci < 3 && TW != force_tw) continue;
        for (int NT = 4; NT >= 1; NT >>= 1) {
            if (force_nt && d->G >= 8 && NT != force_nt) continue;           // (the 24-part networks only)
            const int MT = cMT;
            const int Pn = 64 * NT;
            int rows_span, tiles_x, tiles_p;
            if (ci < 3) {
                if (Pn % TW) continue;
                rows_span = Pn / TW;
                tiles_x = jaf_cdiv(d->OW, TW);
                tiles_p = jaf_cdiv(d->OH, rows_span);
            } else {
                rows_span = (Pn % TW == 0) ? Pn / TW : (Pn + TW - 2) / TW + 1;
                if (rows_span > d->OH) rows_span = d->OH;
                tiles_x = 1;
                tiles_p = jaf_cdiv(OHW, Pn);
            }
            const int PH = (rows_span - 1) * d->stride + d->KH;
            const int PW = (TW - 1) * d->stride + d->KW;
            const int ilv = (ci < 3 && NT > 1 && !no_ilv) ? 1 : 0;
            const int PWp = ilv ? rup_d(PW, NT) : PW;
            const int npos = PH * PWp;
            if (npos > 64 * 4 * CD_RPW) continue;
            const int plane = rup_d(npos * 16, 1024);
            for (int NG = (groups < ngcap ? groups : ngcap); NG >= 1; --NG) {
                const int nchunks = jaf_cdiv(groups, NG);
                const int ng_last = groups - (nchunks - 1) * NG;
                const int nsteps = jaf_cdiv(taps * NG, 4);
                const int nsteps_last = jaf_cdiv(taps * ng_last, 4);
#ifdef CD_FORCE_NG
                if (split == (CD_FORCE_SPLIT != 0) && d->G == 1 && M >= 64 && groups >= 4 && NG != CD_FORCE_NG) continue;
#endif
                // WS: k-steps of weights resident at a time (0: the whole chunk).  Sub-loads keep the LDS of a chunk with many channel
                // groups (no padded half k-step: 3 x 3 taps x 4 groups = 9 exact steps) at that of a small one.
                for (int wi = 0; wi < 2; ++wi) {
                const int WS = wi == 0 ? 0 : CD_WS_CAND;
                if (WS && WS >= nsteps) continue;
#ifdef CD_FORCE_WSUB
                if (split == (CD_FORCE_SPLIT != 0) && d->G == 1 && M >= 64 && groups >= 4 && WS != ((CD_FORCE_WSUB) < nsteps ? (CD_FORCE_WSUB) : 0)) continue;
#endif
                const int wl = WS ? WS : nsteps;
                const long lds = (long)sb * NG * plane + (long)sb * wl * MT * 1024 + 2L * 16 * nsteps * 4 + 64;
                if (lds > 150 * 1024) continue;
                const double total_steps = (double)(nchunks - 1) * nsteps + nsteps_last;
                const double mfma = (double)MT * NT * 16.0 * (split ? 3.0 : 1.0);
                const double ldsrd = 4.0 * (MT + NT) * 4.0 * sb;
                const double t_step = (mfma > ldsrd ? mfma : ldsrd) + 40.0;
                const double stage = 600.0 + sb * ((double)npos * NG * 16.0 + (double)nsteps * MT * 1024.0) / 48.0;
                const int blocks_cu = (int)(160 * 1024 / lds);
                const int bl = blocks_cu > 6 ? 6 : blocks_cu;
                // two resident workgroups do not overlap each other's phases (DMA, matrix cores and epilogue of
                // the 256 -> 256 layer add up): where a launch has at least 4 workgroups per CU to run, 4 resident ones on half-size
                // chunks beat 2 on full-size chunks by 2-11 % in spite of 10 % of padded k-steps (measured, 64 x 64 .. 256 x 256
                // layers); with fewer workgroups than that the larger chunks win by 15-19 % (32 x 32 layers)
                // (split-bf16: three matrix-core instructions per operand pair, so the phases a resident neighbour hides weigh less --
                // profiles/experiments/round5_weight_subloads.txt)
                const double pen2 = split ? 1.15 : 1.25;
                const double nblocks0 = (double)tiles_x * tiles_p * jaf_cdiv(M, 16 * MT) * d->N * d->G;
                const bool big = nblocks0 >= 1024.0;
                const double occ_pen = bl >= 4 ? 1.0 : (bl == 3 ? (big ? 0.5 * (1.0 + pen2) : 1.0) : (bl == 2 ? (big ? pen2 : 1.1) : 1.4));
                const double fixed = 2500.0 + 60.0 * MT * NT;
                // whole-launch cost: workgroups / 256 CUs, but never less than one workgroup's own
                // latency (deep 4x4 .. 16x16 layers launch fewer workgroups than there are CUs, and
                // then smaller tiles that spread over more CUs win)
                const double nblocks = (double)tiles_x * tiles_p * jaf_cdiv(M, 16 * MT) * d->N * d->G;
                const double per_cu = nblocks / 256.0;
                const int conc = per_cu >= bl ? bl : (per_cu <= 1.0 ? 1 : (int)per_cu);
                const double ovl = 1.0 / (double)(conc < 1 ? 1 : conc);   // resident workgroups hide each other's staging
                const double subs = WS ? (double)(nchunks - 1) * (jaf_cdiv(nsteps, WS) - 1) + (jaf_cdiv(nsteps_last, WS) - 1) : 0.0;
                // (calibrated on profiles/experiments/round5_weight_subloads.txt: a sub-load costs its own latency where nothing else is
                // resident, a quarter of it at four workgroups per CU)
                const double wstage = 500.0 + (double)wl * MT * 1024.0 / 48.0;
                const double per_block = total_steps * t_step * occ_pen + nchunks * (stage * ovl + 250.0) + subs * (wstage * ovl + 50.0) + fixed;
                const double cost = (per_cu > 1.0 ? per_cu : 1.0) * per_block;
                if (cost < bestCost) { bestCost = cost; bTW = TW; bNT = NT; bNG = NG; bMT = MT; bWS = WS; }
                }
            }
        }
    }
    JAF_REQUIRE(bTW > 0);
    const int MT = bMT;
    const int Pn = 64 * bNT;
    const bool linear = (bTW == d->OW);
    int rows_span;
    if (!linear) {
        rows_span = Pn / bTW;
        plan->tiles_x = jaf_cdiv(d->OW, bTW);
        plan->tiles_p = jaf_cdiv(d->OH, rows_span);
    } else {
        rows_span = (Pn % bTW == 0) ? Pn / bTW : (Pn + bTW - 2) / bTW + 1;
        if (rows_span > d->OH) rows_span = d->OH;
        plan->tiles_x = 1;
        plan->tiles_p = jaf_cdiv(OHW, Pn);
    }
    plan->precision = d->precision;
    plan->MT = MT;
    plan->NT = bNT;
    plan->NG = bNG;
    plan->CK = 8 * bNG;
    plan->TWIN = bTW;
    plan->PH = (rows_span - 1) * d->stride + d->KH;
    plan->PW = (bTW - 1) * d->stride + d->KW;
    plan->ilv = (!linear && bNT > 1 && !no_ilv) ? 1 : 0;
    plan->pf = bWS;
    plan->PWp = plan->ilv ? rup_d(plan->PW, bNT) : plan->PW;
    plan->npos = plan->PH * plan->PWp;
    plan->plane = rup_d(plan->npos * 16, 1024);
    plan->PS = plan->plane;
    plan->MRp = 16 * MT;
    plan->nchunks = jaf_cdiv(groups, bNG);
    plan->ng_last = groups - (plan->nchunks - 1) * bNG;
    plan->nsteps = jaf_cdiv(taps * bNG, 4);
    plan->nsteps_last = jaf_cdiv(taps * plan->ng_last, 4);
    plan->mblocks = jaf_cdiv(M, 16 * MT);
    plan->lds_bytes = (int)((long)sb * bNG * plan->plane + (long)sb * (bWS ? bWS : plan->nsteps) * MT * 1024 + 2L * 16 * plan->nsteps * 4 + 64);
    // (split: the hi image and the residual image of every chunk, [chunk][image][k-step]: the layout jaf_conv2d_pack makes)
    plan->packed_floats = ((int64_t)d->G * plan->mblocks * plan->nchunks * sb * plan->nsteps * MT * 1024) / 4;
    return JAF_OK;
}

static bool cd_plan_ok(const jaf_conv_desc* d, const jaf_conv_plan* p) {
    if (!p || p->precision != d->precision) return false;
    const int sb = d->precision == JAF_PREC_BF16X3 ? 2 : 1;
    if (p->MT < 1 || p->MT > 4) return false;
    if (p->NT != 1 && p->NT != 2 && p->NT != 4) return false;
    if (p->NG < 1 || p->NG > 4) return false;
    const int groups = jaf_cdiv(d->Cin, 8);
    if (p->nchunks != jaf_cdiv(groups, p->NG)) return false;
    if (p->ng_last != groups - (p->nchunks - 1) * p->NG) return false;
    const int taps = d->KH * d->KW;
    if (p->nsteps != jaf_cdiv(taps * p->NG, 4) || p->nsteps_last != jaf_cdiv(taps * p->ng_last, 4)) return false;
    if (p->mblocks != jaf_cdiv(d->Cout, 16 * p->MT)) return false;
    if (p->ilv != 0 && p->ilv != 1) return false;
    if (p->ilv && (p->NT < 2 || (p->TWIN == d->OW && p->tiles_x == 1) || p->PWp % p->NT)) return false;
    if (p->PWp < p->PW || (!p->ilv && p->PWp != p->PW)) return false;
    if (p->npos != p->PH * p->PWp || p->npos > 64 * 4 * CD_RPW) return false;
    if (p->plane < ((p->npos + 63) / 64) * 1024 || (p->plane & 1023)) return false;
    if (p->TWIN < 1 || p->tiles_x < 1 || p->tiles_p < 1) return false;
    const int Pn = 64 * p->NT;
    int rows_span;
    if (p->TWIN == d->OW && p->tiles_x == 1) {
        rows_span = (Pn % p->TWIN == 0) ? Pn / p->TWIN : (Pn + p->TWIN - 2) / p->TWIN + 1;
        if (rows_span > d->OH) rows_span = d->OH;
    } else {
        if (Pn % p->TWIN) return false;
        rows_span = Pn / p->TWIN;
    }
    if (p->PH < (rows_span - 1) * d->stride + d->KH) return false;
    if (p->PW < (p->TWIN - 1) * d->stride + d->KW) return false;
    if (p->pf < 0 || p->pf > p->nsteps) return false;
    if (p->lds_bytes < sb * p->NG * p->plane + sb * (p->pf ? p->pf : p->nsteps) * p->MT * 1024 + 2 * 16 * p->nsteps * 4) return false;
    if (p->lds_bytes > 160 * 1024) return false;
    if ((long)d->H * d->W * 16 >= CD_OOB) return false;
    return true;
}

template <int MT, int NT, bool LSTM>
static int cd_launch_one(const ConvDArgs& a_in, hipStream_t s) {
    const ConvDArgs& a = a_in;
    const int lds = a.p.lds_bytes;
    const long nblk = (long)a.ntiles * a.p.mblocks * a.d.N * a.d.G;
    if (nblk < 1 || nblk > 0x7fffffffL) return JAF_EINVAL;
    if constexpr (!LSTM) {
        if (a.dz_mask) {          // the fused activation backward has its own instantiation (see cd_epilogue)
            auto kz = conv_dma_kernel<MT, NT, false, true>;
            static int optin_z[JAF_MAX_DEVICES];
            if (lds > 48 * 1024) {
                const int e = jaf_lds_optin((const void*)kz, optin_z);
                if (e) return e;
            }
            JAF_NOTE_KERNEL("conv_dma_kernel<%d, %d, false, true, false>", MT, NT);
            hipLaunchKernelGGL(kz, dim3((unsigned)nblk), dim3(256), (size_t)lds, s, a);
            return jaf_launch_status();
        }
    }
    if constexpr (!LSTM) {
        if (!a.dst && !a.stats && !a.acc_out && !a.out2 && !a.skip_f32) {
            auto kp = conv_dma_kernel<MT, NT, false, false, true>;
            static int optin_p[JAF_MAX_DEVICES];
            if (lds > 48 * 1024) {
                const int e = jaf_lds_optin((const void*)kp, optin_p);
                if (e) return e;
            }
            JAF_NOTE_KERNEL("conv_dma_kernel<%d, %d, false, false, true>", MT, NT);
            hipLaunchKernelGGL(kp, dim3((unsigned)nblk), dim3(256), (size_t)lds, s, a);
            return jaf_launch_status();
        }
    }
    auto k = conv_dma_kernel<MT, NT, LSTM, false>;
    static int optin[JAF_MAX_DEVICES];
    if (lds > 48 * 1024) {
        const int e = jaf_lds_optin((const void*)k, optin);
        if (e) return e;
    }
    JAF_NOTE_KERNEL("conv_dma_kernel<%d, %d, %s, false, false>", MT, NT, LSTM ? "true" : "false");
    hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(256), (size_t)lds, s, a);
    return jaf_launch_status();
}

template <int MT, bool LSTM>
static int cd_launch_nt(const ConvDArgs& a, hipStream_t s) {
    switch (a.p.NT) {
        case 1: return cd_launch_one<MT, 1, LSTM>(a, s);
        case 2: return cd_launch_one<MT, 2, LSTM>(a, s);
        case 4: return cd_launch_one<MT, 4, LSTM>(a, s);
    }
    return JAF_EINVAL;
}

template <bool LSTM>
static int cd_launch_mt(const ConvDArgs& a, hipStream_t s) {
    switch (a.p.MT) {
        case 1: return cd_launch_nt<1, LSTM>(a, s);
        case 2: return cd_launch_nt<2, LSTM>(a, s);
        case 3: return cd_launch_nt<3, LSTM>(a, s);
        case 4: return cd_launch_nt<4, LSTM>(a, s);
    }
    return JAF_EINVAL;
}

static void cd_fill(ConvDArgs& a, const jaf_conv_desc* d, const jaf_conv_plan* plan) {
    a.d = *d;
    a.p = *plan;
    const int sb = d->precision == JAF_PREC_BF16X3 ? 2 : 1;        // split: [hi planes][lo planes][hi weights][lo weights][table]
    a.off_w = sb * plan->NG * plan->plane;
    a.off_tab = a.off_w + sb * (plan->pf ? plan->pf : plan->nsteps) * plan->MT * 1024;
    a.ntiles = plan->tiles_x * plan->tiles_p;
    a.ngroups8 = jaf_cdiv(d->Cin, 8);
    a.inv_pwp = 1.0f / (float)plan->PWp;
    a.inv_pwq = 1.0f / (float)(plan->ilv ? plan->PWp / plan->NT : plan->PWp);
    a.inv_twin = 1.0f / (float)plan->TWIN;
    a.dv_mblocks = jaf_fdiv_make((uint32_t)plan->mblocks);
    a.dv_ntiles = jaf_fdiv_make((uint32_t)a.ntiles);
    a.dv_G = jaf_fdiv_make((uint32_t)d->G);
    a.dv_tiles_x = jaf_fdiv_make((uint32_t)plan->tiles_x);
    a.dv_twin = jaf_fdiv_make((uint32_t)plan->TWIN);
    a.inv_kw = 1.0f / (float)d->KW;
    a.inv_ng = 1.0f / (float)plan->NG;
    a.inv_ng_last = 1.0f / (float)plan->ng_last;
    a.ilv = plan->ilv;
    a.vec = (plan->ilv && d->OW % plan->NT == 0) ? 1 : 0;
    a.c_prev = nullptr;
    a.c_out = nullptr;
    a.h_out = nullptr;
    a.gates_out = nullptr;
    a.gates_bf16 = 0;
    a.stats = nullptr;
    a.stat_slots = 1;
    a.in_ng8 = a.ngroups8;
    a.dst = nullptr;
    a.dst_ng8 = a.dst_coff = a.dst_img_off = a.dst_pad_tail = 0;
    a.skip_f32 = 0;
    a.acc_out = 0;
    a.out2 = nullptr;
    a.split = 0;
    a.dz_mask = nullptr;
    a.dz_mask_ng8 = a.dz_mask_coff = 0;
    a.dz_slope = 0.f;
    a.dz_dbias = nullptr;
    a.out_bf16 = a.out2_bf16 = a.state_bf16 = 0;
    a.dz_mask_split = 0;
}

static bool cd_io_ok(const jaf_conv_desc* d, const jaf_packed_io* io, bool lstm) {
    if (!io) return true;
    if (io->in_ng8_tot != 0 && io->in_ng8_tot < jaf_cdiv(d->Cin, 8)) return false;
    const int cout = lstm ? (d->Cout >> 2) : d->Cout;        // channels this launch writes per group (LSTM: the hidden state)
    if (io->dst) {
        if (io->dst_ng8_tot < 1 || io->dst_coff < 0 || (io->dst_coff & 3) || io->dst_img_off < 0) return false;
        const int wr = (!lstm && io->dz_mask && io->out2) ? io->split_rows : cout;          // channels actually written to dst
        const int end = io->dst_coff + (io->dst_pad_tail ? ((wr + 7) & ~7) : wr);
        if (end > io->dst_ng8_tot * 8) return false;
        if (lstm && io->dst_pad_tail) return false;
    } else if (io->skip_f32 && !lstm) {
        return false;                                         // a launch that writes nothing
    }
    if (io->dz_mask) {
        // fused activation backward: a plain data-gradient launch whose only output is the packed dz image
        if (lstm || !io->dst || io->dst_coff != 0 || !io->dst_pad_tail) return false;
        const int dzc = io->out2 ? io->split_rows : d->Cout;
        if (io->dz_mask_ng8 < 1 || io->dz_mask_coff < 0 || (io->dz_mask_coff & 7) ||
            io->dz_mask_coff / 8 + jaf_cdiv(dzc, 8) > io->dz_mask_ng8) return false;
        if (jaf_cdiv(dzc, 8) > io->dst_ng8_tot) return false;
        if (io->skip_f32 && io->accumulate_f32) return false;      // the first consumer's gradient is read from `out`
    }
    if (io->accumulate_f32 && (lstm || (io->skip_f32 && !io->dz_mask))) return false;
    if (io->out2 && (lstm || io->skip_f32 || (io->dst && !io->dz_mask) || io->split_rows < 1 || io->split_rows >= d->Cout)) return false;
    // bf16 storage of the NCHW outputs / the cell state: bf16 arithmetic only (the parity-grade modes keep fp32 tensors)
    if ((io->out_bf16 || io->out2_bf16 || io->state_bf16) && d->precision != JAF_PREC_BF16) return false;
    if (io->dz_mask_split && (d->precision != JAF_PREC_BF16 || !io->dz_mask)) return false;
    if (io->state_bf16 && !lstm) return false;
    if ((io->out_bf16 || io->out2_bf16) && lstm) return false;
    return true;
}

static void cd_apply_io(ConvDArgs& a, const jaf_packed_io* io) {
    if (!io) return;
    if (io->in_ng8_tot) a.in_ng8 = io->in_ng8_tot;
    a.dst = (unsigned char*)io->dst;
    a.dst_ng8 = io->dst_ng8_tot;
    a.dst_coff = io->dst_coff;
    a.dst_img_off = io->dst_img_off;
    a.dst_pad_tail = io->dst_pad_tail ? 1 : 0;
    a.skip_f32 = io->skip_f32 ? 1 : 0;
    a.acc_out = io->accumulate_f32 ? 1 : 0;
    a.dz_mask = (const unsigned char*)io->dz_mask;
    a.dz_mask_ng8 = io->dz_mask_ng8;
    a.dz_mask_coff = io->dz_mask_coff;
    a.dz_slope = io->dz_slope;
    a.dz_dbias = io->dz_dbias;
    a.out2 = io->out2;
    a.split = io->out2 ? io->split_rows : 0;
    a.out_bf16 = io->out_bf16 ? 1 : 0;
    a.out2_bf16 = (io->out2 && io->out2_bf16) ? 1 : 0;
    a.state_bf16 = io->state_bf16 ? 1 : 0;
    a.dz_mask_split = (io->dz_mask && io->dz_mask_split) ? 1 : 0;
}

extern "C" int jaf_conv2d_fwd_packed_io(jaf_stream_t s, const jaf_conv_desc* d, const jaf_conv_plan* plan,
                                        const void* packed_in, const void* packed_w, const float* bias, float* out,
                                        double* stats, int32_t stat_slots, const jaf_packed_io* io) {
    JAF_REQUIRE(cd_desc_ok(d) && cd_plan_ok(d, plan) && packed_in && packed_w && cd_io_ok(d, io, false));
    JAF_REQUIRE(out || (io && io->skip_f32));
    JAF_REQUIRE(!stats || (d->act == JAF_ACT_NONE && d->G == 1 && stat_slots >= 1 && stat_slots <= 64 && !(io && (io->skip_f32 || io->out2))));
    ConvDArgs a;
    cd_fill(a, d, plan);
    cd_apply_io(a, io);
    a.xp = (const unsigned char*)packed_in;
    a.wpk = (const unsigned char*)packed_w;
    a.bias = bias;
    a.out = out;
    a.stats = stats;
    a.stat_slots = stats ? stat_slots : 1;
    if (d->precision == JAF_PREC_BF16X3) {
        // split-bf16: same epilogue; a packed destination / sign image is a split image too (hi + lo planes)
        return cd_split_launch(a, (hipStream_t)s, false);
    }
    return cd_launch_mt<false>(a, (hipStream_t)s);
}

extern "C" int jaf_conv2d_fwd_packed_stats(jaf_stream_t s, const jaf_conv_desc* d, const jaf_conv_plan* plan,
                                           const void* packed_in, const void* packed_w, const float* bias, float* out,
                                           double* stats, int32_t stat_slots) {
    JAF_REQUIRE(cd_desc_ok(d) && cd_plan_ok(d, plan) && packed_in && packed_w && out);
    JAF_REQUIRE(!stats || (d->act == JAF_ACT_NONE && d->G == 1 && stat_slots >= 1 && stat_slots <= 64));
    ConvDArgs a;
    cd_fill(a, d, plan);
    a.xp = (const unsigned char*)packed_in;
    a.wpk = (const unsigned char*)packed_w;
    a.bias = bias;
    a.out = out;
    a.stats = stats;
    a.stat_slots = stats ? stat_slots : 1;
    if (d->precision == JAF_PREC_BF16X3) return cd_split_launch(a, (hipStream_t)s, false);
    return cd_launch_mt<false>(a, (hipStream_t)s);
}

extern "C" int jaf_conv2d_fwd_packed(jaf_stream_t s, const jaf_conv_desc* d, const jaf_conv_plan* plan,
                                     const void* packed_in, const void* packed_w, const float* bias, float* out) {
    return jaf_conv2d_fwd_packed_stats(s, d, plan, packed_in, packed_w, bias, out, nullptr, 1);
}

extern "C" int jaf_convlstm_cell_fwd_packed(jaf_stream_t s, const jaf_conv_desc* d, const jaf_conv_plan* plan,
                                            const void* packed_in, const void* packed_w, const float* bias,
                                            const float* c_prev, float* h_out, float* c_out, void* gates_out,
                                            int gates_bf16) {
    return jaf_convlstm_cell_fwd_packed_io(s, d, plan, packed_in, packed_w, bias, c_prev, h_out, c_out, gates_out, gates_bf16,
                                           nullptr);
}

extern "C" int jaf_convlstm_cell_fwd_packed_io(jaf_stream_t s, const jaf_conv_desc* d, const jaf_conv_plan* plan,
                                               const void* packed_in, const void* packed_w, const float* bias,
                                               const float* c_prev, float* h_out, float* c_out, void* gates_out,
                                               int gates_bf16, const jaf_packed_io* io) {
    JAF_REQUIRE(cd_desc_ok(d) && cd_plan_ok(d, plan) && packed_in && packed_w && bias && c_out && cd_io_ok(d, io, true));
    JAF_REQUIRE(h_out || (io && io->skip_f32 && io->dst));
    JAF_REQUIRE(d->KH == 3 && d->KW == 3 && d->stride == 1 && (d->Cout & 3) == 0);
    JAF_REQUIRE(d->Cout % (16 * plan->MT) == 0 && d->H == d->OH && d->W == d->OW);
    ConvDArgs a;
    cd_fill(a, d, plan);
    cd_apply_io(a, io);
    a.xp = (const unsigned char*)packed_in;
    a.wpk = (const unsigned char*)packed_w;
    a.bias = bias;
    a.out = nullptr;
    a.c_prev = c_prev;
    a.c_out = c_out;
    a.h_out = h_out;
    a.gates_out = (float*)gates_out;
    a.gates_bf16 = gates_bf16 ? 1 : 0;
    if (d->precision == JAF_PREC_BF16X3) {
        return cd_split_launch(a, (hipStream_t)s, true);
    }
    return cd_launch_mt<true>(a, (hipStream_t)s);
}
