// Wavefront-level gathers of the stage-4 step (HBM / L2 bound):
//  * texture warp  -- train/4.convLSTM_flowpro_interval.py:43-76 (== test/conv_pro_test.py:41-74,
//    src/networks.py:36-68): the reference loops 24 parts x (2 where + grid_sample + where) per
//    SAMPLE; here one pass reads each IUV pixel once (3 bytes), picks the part, does one
//    bilinear gather of 3 channels and writes the pixel.  Lanes walk x, so the uint8 IUV read,
//    and the output writes are contiguous per wave; taps are L2-resident (24 x 3 x 200 x 200).
//  * its adjoint   -- bilinear scatter-add into the 24 part textures (grid carries no gradient).
//  * grid_sample   -- src/cal_flow.py:38 (padding_mode='border') and the generic zeros mode.
#include "jaf_common.h"

__device__ __forceinline__ float gs_unnormalize(float g, int size, int align) {
    // ATen grid_sampler_unnormalize
    return align ? ((g + 1.f) / 2.f) * (float)(size - 1) : ((g + 1.f) * (float)size - 1.f) / 2.f;
}

struct WarpTap {
    int x0, y0;
    float wnw, wne, wsw, wse;
};

__device__ __forceinline__ WarpTap make_tap(float ix, float iy) {
    WarpTap t;
    const float fx = floorf(ix), fy = floorf(iy);
    t.x0 = (int)fx;
    t.y0 = (int)fy;
    const float ax = ix - fx, ay = iy - fy;     // distance to the west / north tap
    t.wnw = (1.f - ax) * (1.f - ay);
    t.wne = ax * (1.f - ay);
    t.wsw = (1.f - ax) * ay;
    t.wse = ax * ay;
    return t;
}

__global__ void texture_warp_fwd_kernel(const float* tex, const uint8_t* iuv, float* out, int B, int S, int TH,
                                        int TW, int align) {
    const long total = (long)B * S * S;
    const long gs = (long)gridDim.x * blockDim.x;
    const long SS = (long)S * S;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gs) {
        const long b = e / SS;
        const long pix = e - b * SS;
        const uint8_t* q = iuv + e * 3;
        const int I = q[0];
        float r0 = 0.f, r1 = 0.f, r2 = 0.f;
        if (I >= 1 && I <= 24) {
            const float U = (float)q[1], V = (float)q[2];
            const float gx = ((255.f - V) / 255.f - 0.5f) * 2.f;
            const float gy = (U / 255.f - 0.5f) * 2.f;
            const WarpTap t = make_tap(gs_unnormalize(gx, TW, align), gs_unnormalize(gy, TH, align));
            const float* tp = tex + ((b * 24 + (I - 1)) * 3) * (long)TH * TW;
            const long cs = (long)TH * TW;
            const bool xw = t.x0 >= 0 && t.x0 < TW, xe = t.x0 + 1 >= 0 && t.x0 + 1 < TW;
            const bool yn = t.y0 >= 0 && t.y0 < TH, ys = t.y0 + 1 >= 0 && t.y0 + 1 < TH;
            const long o = (long)t.y0 * TW + t.x0;
            if (yn && xw) { const float w = t.wnw; r0 += tp[o] * w; r1 += tp[cs + o] * w; r2 += tp[2 * cs + o] * w; }
            if (yn && xe) { const float w = t.wne; r0 += tp[o + 1] * w; r1 += tp[cs + o + 1] * w; r2 += tp[2 * cs + o + 1] * w; }
            if (ys && xw) { const float w = t.wsw; r0 += tp[o + TW] * w; r1 += tp[cs + o + TW] * w; r2 += tp[2 * cs + o + TW] * w; }
            if (ys && xe) { const float w = t.wse; r0 += tp[o + TW + 1] * w; r1 += tp[cs + o + TW + 1] * w; r2 += tp[2 * cs + o + TW + 1] * w; }
        }
        float* op = out + b * 3 * SS + pix;
        op[0] = r0;
        op[SS] = r1;
        op[2 * SS] = r2;
    }
}

extern "C" int jaf_texture_warp_fwd(jaf_stream_t s, const float* tex, const uint8_t* iuv, float* out, int32_t B,
                                    int32_t S, int32_t TH, int32_t TW, int align_corners) {
    JAF_REQUIRE(tex && iuv && out && B >= 1 && S >= 1 && TH >= 1 && TW >= 1);
    hipLaunchKernelGGL(texture_warp_fwd_kernel, dim3(jaf_ew_grid((long)B * S * S)), dim3(256), 0, (hipStream_t)s, tex, iuv, out, B, S, TH, TW, align_corners);
    return jaf_launch_status();
}

__global__ void texture_warp_bwd_kernel(const float* dout, const uint8_t* iuv, float* dtex, int B, int S, int TH,
                                        int TW, int align) {
    const long total = (long)B * S * S;
    const long gs = (long)gridDim.x * blockDim.x;
    const long SS = (long)S * S;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gs) {
        const long b = e / SS;
        const long pix = e - b * SS;
        const uint8_t* q = iuv + e * 3;
        const int I = q[0];
        if (I < 1 || I > 24) continue;
        const float U = (float)q[1], V = (float)q[2];
        const float gx = ((255.f - V) / 255.f - 0.5f) * 2.f;
        const float gy = (U / 255.f - 0.5f) * 2.f;
        const WarpTap t = make_tap(gs_unnormalize(gx, TW, align), gs_unnormalize(gy, TH, align));
        float* tp = dtex + ((b * 24 + (I - 1)) * 3) * (long)TH * TW;
        const long cs = (long)TH * TW;
        const float* gp = dout + b * 3 * SS + pix;
        const float g0 = gp[0], g1 = gp[SS], g2 = gp[2 * SS];
        const bool xw = t.x0 >= 0 && t.x0 < TW, xe = t.x0 + 1 >= 0 && t.x0 + 1 < TW;
        const bool yn = t.y0 >= 0 && t.y0 < TH, ys = t.y0 + 1 >= 0 && t.y0 + 1 < TH;
        const long o = (long)t.y0 * TW + t.x0;
        if (yn && xw) { atomicAdd(&tp[o], g0 * t.wnw); atomicAdd(&tp[cs + o], g1 * t.wnw); atomicAdd(&tp[2 * cs + o], g2 * t.wnw); }
        if (yn && xe) { atomicAdd(&tp[o + 1], g0 * t.wne); atomicAdd(&tp[cs + o + 1], g1 * t.wne); atomicAdd(&tp[2 * cs + o + 1], g2 * t.wne); }
        if (ys && xw) { atomicAdd(&tp[o + TW], g0 * t.wsw); atomicAdd(&tp[cs + o + TW], g1 * t.wsw); atomicAdd(&tp[2 * cs + o + TW], g2 * t.wsw); }
        if (ys && xe) { atomicAdd(&tp[o + TW + 1], g0 * t.wse); atomicAdd(&tp[cs + o + TW + 1], g1 * t.wse); atomicAdd(&tp[2 * cs + o + TW + 1], g2 * t.wse); }
    }
}

extern "C" int jaf_texture_warp_bwd(jaf_stream_t s, const float* dout, const uint8_t* iuv, float* dtex, int32_t B,
                                    int32_t S, int32_t TH, int32_t TW, int align_corners) {
    JAF_REQUIRE(dout && iuv && dtex && B >= 1 && S >= 1 && TH >= 1 && TW >= 1);
    hipLaunchKernelGGL(texture_warp_bwd_kernel, dim3(jaf_ew_grid((long)B * S * S)), dim3(256), 0, (hipStream_t)s, dout, iuv, dtex, B, S, TH, TW, align_corners);
    return jaf_launch_status();
}

// one thread per output pixel, loops the channels (the grid is read once)
__global__ void grid_sample_fwd_kernel(const float* src, const float* grid, float* out, int B, int C, int H, int W,
                                       int OH, int OW, int border, int align) {
    const long total = (long)B * OH * OW;
    const long gs = (long)gridDim.x * blockDim.x;
    const long OS = (long)OH * OW;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gs) {
        const long b = e / OS;
        const long pix = e - b * OS;
        float ix = gs_unnormalize(grid[e * 2], W, align);
        float iy = gs_unnormalize(grid[e * 2 + 1], H, align);
        if (border) {
            ix = fminf((float)(W - 1), fmaxf(ix, 0.f));
            iy = fminf((float)(H - 1), fmaxf(iy, 0.f));
        }
        const WarpTap t = make_tap(ix, iy);
        const bool xw = t.x0 >= 0 && t.x0 < W, xe = t.x0 + 1 >= 0 && t.x0 + 1 < W;
        const bool yn = t.y0 >= 0 && t.y0 < H, ys = t.y0 + 1 >= 0 && t.y0 + 1 < H;
        const long o = (long)t.y0 * W + t.x0;
        for (int c = 0; c < C; ++c) {
            const float* sp = src + (b * C + c) * (long)H * W;
            float r = 0.f;
            if (yn && xw) r += sp[o] * t.wnw;
            if (yn && xe) r += sp[o + 1] * t.wne;
            if (ys && xw) r += sp[o + W] * t.wsw;
            if (ys && xe) r += sp[o + W + 1] * t.wse;
            out[(b * C + c) * OS + pix] = r;
        }
    }
}

extern "C" int jaf_grid_sample_fwd(jaf_stream_t s, const float* src, const float* grid, float* out, int32_t B,
                                   int32_t C, int32_t H, int32_t W, int32_t OH, int32_t OW, int padding_border,
                                   int align_corners) {
    JAF_REQUIRE(src && grid && out && B >= 1 && C >= 1 && H >= 1 && W >= 1 && OH >= 1 && OW >= 1);
    hipLaunchKernelGGL(grid_sample_fwd_kernel, dim3(jaf_ew_grid((long)B * OH * OW)), dim3(256), 0, (hipStream_t)s, src, grid, out, B, C, H, W, OH, OW, padding_border, align_corners);
    return jaf_launch_status();
}

// Adjoint of grid_sample_fwd_kernel (ATen grid_sampler_2d_backward, bilinear): dsrc (+=, nullable) by scatter-add,
// dgrid (nullable) [B,OH,OW,2].  Border mode: the clamp passes a gradient of 1 strictly inside (0, size-1) and 0
// where it clipped (clip_coordinates_set_grad).
__global__ void grid_sample_bwd_kernel(const float* dout, const float* src, const float* grid, float* dsrc, float* dgrid, int B,
                                       int C, int H, int W, int OH, int OW, int border, int align) {
    const long total = (long)B * OH * OW;
    const long gs = (long)gridDim.x * blockDim.x;
    const long OS = (long)OH * OW;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gs) {
        const long b = e / OS;
        const long pix = e - b * OS;
        float ix = gs_unnormalize(grid[e * 2], W, align);
        float iy = gs_unnormalize(grid[e * 2 + 1], H, align);
        float gix_mult = align ? (float)(W - 1) / 2.f : (float)W / 2.f;
        float giy_mult = align ? (float)(H - 1) / 2.f : (float)H / 2.f;
        if (border) {
            if (ix <= 0.f) { ix = 0.f; gix_mult = 0.f; } else if (ix >= (float)(W - 1)) { ix = (float)(W - 1); gix_mult = 0.f; }
            if (iy <= 0.f) { iy = 0.f; giy_mult = 0.f; } else if (iy >= (float)(H - 1)) { iy = (float)(H - 1); giy_mult = 0.f; }
        }
        const WarpTap t = make_tap(ix, iy);
        const float ax = ix - (float)t.x0, ay = iy - (float)t.y0;
        const bool xw = t.x0 >= 0 && t.x0 < W, xe = t.x0 + 1 >= 0 && t.x0 + 1 < W;
        const bool yn = t.y0 >= 0 && t.y0 < H, ys = t.y0 + 1 >= 0 && t.y0 + 1 < H;
        const long o = (long)t.y0 * W + t.x0;
        float gix = 0.f, giy = 0.f;
        for (int c = 0; c < C; ++c) {
            const float g = dout[(b * C + c) * OS + pix];
            const long pb = (b * C + c) * (long)H * W;
            if (dsrc && g != 0.f) {
                if (yn && xw) atomicAdd(&dsrc[pb + o], g * t.wnw);
                if (yn && xe) atomicAdd(&dsrc[pb + o + 1], g * t.wne);
                if (ys && xw) atomicAdd(&dsrc[pb + o + W], g * t.wsw);
                if (ys && xe) atomicAdd(&dsrc[pb + o + W + 1], g * t.wse);
            }
            if (dgrid) {
                const float* sp = src + pb;
                const float nw = (yn && xw) ? sp[o] : 0.f, ne = (yn && xe) ? sp[o + 1] : 0.f;
                const float sw = (ys && xw) ? sp[o + W] : 0.f, se = (ys && xe) ? sp[o + W + 1] : 0.f;
                gix += g * ((ne - nw) * (1.f - ay) + (se - sw) * ay);
                giy += g * ((sw - nw) * (1.f - ax) + (se - ne) * ax);
            }
        }
        if (dgrid) {
            dgrid[e * 2] = gix_mult * gix;
            dgrid[e * 2 + 1] = giy_mult * giy;
        }
    }
}

extern "C" int jaf_grid_sample_bwd(jaf_stream_t s, const float* dout, const float* src, const float* grid, float* dsrc,
                                   float* dgrid, int32_t B, int32_t C, int32_t H, int32_t W, int32_t OH, int32_t OW,
                                   int padding_border, int align_corners) {
    JAF_REQUIRE(dout && src && grid && (dsrc || dgrid) && B >= 1 && C >= 1 && H >= 1 && W >= 1 && OH >= 1 && OW >= 1);
    hipLaunchKernelGGL(grid_sample_bwd_kernel, dim3(jaf_ew_grid((long)B * OH * OW)), dim3(256), 0, (hipStream_t)s, dout, src, grid,
                       dsrc, dgrid, B, C, H, W, OH, OW, padding_border, align_corners);
    return jaf_launch_status();
}

// float_estimate.forward in one pass (src/cal_flow.py:28-39 + src/flow_net.py:91): barycentric flow T of a pixel from the
// target face-index / weight maps and the SOURCE faces (cal_bc_transform, src/nmr.py:617-659), the border-mode bilinear
// sample of the source image at T, and (optionally) the multiplication with the target's SMPL mask that the propagater
// applies first.  Same arithmetic, in the same order, as jaf_bc_transform -> jaf_grid_sample_fwd -> jaf_mul_bcast; the flow
// field T [B,S,S,2] (8 B per pixel written and read back) never exists.
__global__ void flow_warp_fwd_kernel(const float* __restrict__ src, const float* __restrict__ src_faces, const int* __restrict__ fim,
                                     const float* __restrict__ wim, const float* __restrict__ mask, float* __restrict__ out, int B,
                                     int C, int H, int W, int NF, int S, int mask_c, int align) {
    const long total = (long)B * S * S;
    const long gs = (long)gridDim.x * blockDim.x;
    const long OS = (long)S * S;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gs) {
        const long b = e / OS;
        const long pix = e - b * OS;
        const int f = fim[e];
        float tx = -2.f, ty = -2.f;
        if (f >= 0 && f < NF) {
            // products then sums, no fused multiply-add: the expression tree of bc_transform_kernel (raster.hip is built
            // with -ffp-contract=off) and of the reference's `(pts * w).sum(1)`, so the sample position is the same to the bit
#pragma clang fp contract(off)
            const float* v = src_faces + (b * NF + f) * 9;
            const float w0 = wim[e * 3], w1 = wim[e * 3 + 1], w2 = wim[e * 3 + 2];
            tx = (v[0] * w0 + v[3] * w1) + v[6] * w2;
            ty = ((-v[1]) * w0 + (-v[4]) * w1) + (-v[7]) * w2;
        }
        float ix = gs_unnormalize(tx, W, align);
        float iy = gs_unnormalize(ty, H, align);
        ix = fminf((float)(W - 1), fmaxf(ix, 0.f));
        iy = fminf((float)(H - 1), fmaxf(iy, 0.f));
        const WarpTap t = make_tap(ix, iy);
        const bool xw = t.x0 >= 0 && t.x0 < W, xe = t.x0 + 1 >= 0 && t.x0 + 1 < W;
        const bool yn = t.y0 >= 0 && t.y0 < H, ys = t.y0 + 1 >= 0 && t.y0 + 1 < H;
        const long o = (long)t.y0 * W + t.x0;
        for (int c = 0; c < C; ++c) {
            const float* sp = src + (b * C + c) * (long)H * W;
            float r = 0.f;
            if (yn && xw) r += sp[o] * t.wnw;
            if (yn && xe) r += sp[o + 1] * t.wne;
            if (ys && xw) r += sp[o + W] * t.wsw;
            if (ys && xe) r += sp[o + W + 1] * t.wse;
            if (mask) r = r * mask[(b * mask_c + (mask_c == 1 ? 0 : c)) * OS + pix];
            out[(b * C + c) * OS + pix] = r;
        }
    }
}

extern "C" int jaf_flow_warp_fwd(jaf_stream_t s, const float* src, const float* src_faces, const int32_t* fim, const float* wim,
                                 const float* mask, float* out, int32_t B, int32_t C, int32_t H, int32_t W, int32_t NF, int32_t S,
                                 int32_t mask_c, int align_corners) {
    JAF_REQUIRE(src && src_faces && fim && wim && out && B >= 1 && C >= 1 && H >= 1 && W >= 1 && NF >= 1 && S >= 1);
    JAF_REQUIRE(!mask || mask_c == 1 || mask_c == C);
    hipLaunchKernelGGL(flow_warp_fwd_kernel, dim3(jaf_ew_grid((long)B * S * S)), dim3(256), 0, (hipStream_t)s, src, src_faces, fim, wim,
                       mask, out, B, C, H, W, NF, S, mask_c, align_corners);
    return jaf_launch_status();
}
