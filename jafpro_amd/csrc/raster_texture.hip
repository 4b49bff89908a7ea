// Texture branch of the neural_renderer rasteriser and its lighting:
//   forward_texture_sampling_cuda_kernel  rasterize_cuda_kernel.cu:171-243  (+ forward_background, rasterize.py:194-202)
//   backward_textures_cuda_kernel         rasterize_cuda_kernel.cu:506-541
//   lighting                              neural_renderer/lighting.py:6-58  (called from src/nmr.py:219-229)
//
// MI355X notes.  The sampler is HBM-bound gather work: lanes run along x, a lane owns one pixel, the per-pixel maps
// are read once and the 3 x 8 texel taps come from a face's ts^3 x 3 block (324 B at ts = 3), which sits in L1/L2
// after the first pixel of the face.  The reference stores 8 sampling indices + 8 weights per pixel for its backward
// pass (64 B/pixel written, 64 B/pixel read back); here both maps are OPTIONAL: the backward kernel can rebuild them
// from the weight / depth maps the rasteriser keeps anyway (a dozen VALU ops), which removes 128 B/pixel of traffic --
// the FFI-compatible form with the two maps is still there for callers that hold them.  The background fill of
// forward_background is folded into the sampler's store.  Lighting is one wave per face: the light colour is
// wave-uniform, lanes stride over the face's texels, and the adjoint reduces its three sums with DPP/shuffles.
// Built with -ffp-contract=off: the sampler's fp32 expression tree is the reference's (bit-exact vs the C oracle).
#include "jaf_common.h"

namespace {

struct Taps {
    int idx[8];
    float w[8];
};

// the 8 trilinear taps of a foreground pixel (rasterize_cuda_kernel.cu:206-236)
__device__ __forceinline__ Taps texel_taps(const float* face, const float* weight, float depth, int ts, float eps) {
    float pos[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float t = weight[k] * (ts - 1) * (depth / face[3 * k + 2]);
        t = (float)fmax((double)t, 0.);
        t = (float)fmin((double)t, (double)(ts - 1 - eps));
        pos[k] = t;
    }
    Taps r;
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
        float w = 1;
        int at[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int base = (int)pos[k];
            if (((corner >> k) & 1) == 0) { w *= 1 - (pos[k] - base); at[k] = base; }
            else { w *= pos[k] - base; at[k] = base + 1; }
        }
        r.idx[corner] = at[0] * ts * ts + at[1] * ts + at[2];
        r.w[corner] = w;
    }
    return r;
}

}  // namespace

__global__ __launch_bounds__(256) void raster_texture_fwd_kernel(const float* __restrict__ faces,
                                                                 const float* __restrict__ textures,
                                                                 const int* __restrict__ fim, const float* __restrict__ wim,
                                                                 const float* __restrict__ depth, float* __restrict__ rgb,
                                                                 int* __restrict__ tap_index, float* __restrict__ tap_weight,
                                                                 const float* __restrict__ background, int bg_per_image,
                                                                 int NF, int S, int ts, float eps) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), image = blockIdx.z;
    if (x >= S || y >= S) return;
    const long o = ((long)image * S + y) * S + x;
    const int fn = fim[o];
    const float* bg = background + (bg_per_image ? image * 3 : 0);
    float px[3] = {0.f, 0.f, 0.f};
    Taps t;
#pragma unroll
    for (int c = 0; c < 8; ++c) { t.idx[c] = 0; t.w[c] = 0.f; }
    if (fn >= 0) {
        const long f = (long)image * NF + fn;
        const float* tex = textures + f * ts * ts * ts * 3;
        t = texel_taps(faces + f * 9, wim + o * 3, depth[o], ts, eps);
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int k = 0; k < 3; ++k) px[k] += t.w[c] * tex[t.idx[c] * 3 + k];
    }
    const float mask = fn >= 0 ? 1.f : 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k) rgb[o * 3 + k] = px[k] * mask + (1 - mask) * bg[k];
    if (tap_index) {
        int4* d = (int4*)(tap_index + o * 8);
        d[0] = make_int4(t.idx[0], t.idx[1], t.idx[2], t.idx[3]);
        d[1] = make_int4(t.idx[4], t.idx[5], t.idx[6], t.idx[7]);
    }
    if (tap_weight) {
        float4* d = (float4*)(tap_weight + o * 8);
        d[0] = make_float4(t.w[0], t.w[1], t.w[2], t.w[3]);
        d[1] = make_float4(t.w[4], t.w[5], t.w[6], t.w[7]);
    }
}

extern "C" int jaf_rasterize_texture_fwd(jaf_stream_t s, const float* faces, const float* textures,
                                         const int32_t* face_index_map, const float* weight_map, const float* depth_map,
                                         float* rgb_map, int32_t* sampling_index_map, float* sampling_weight_map,
                                         const float* background, int bg_per_image, int32_t B, int32_t NF, int32_t S,
                                         int32_t ts, float eps) {
    JAF_REQUIRE(faces && textures && face_index_map && weight_map && depth_map && rgb_map && background);
    // ts >= 2 and eps > 0 keep the upper tap (floor(pos) + 1) inside the texture block: pos <= ts - 1 - eps
    JAF_REQUIRE(B >= 1 && B <= 65535 && NF >= 1 && S >= 1 && ts >= 2 && ts <= 64 && eps > 0.f);
    JAF_REQUIRE((long)B * NF * ts * ts * ts <= 0x7fffffffL / 3 && (long)B * S * S <= 0x7fffffffL / 8);
    hipLaunchKernelGGL(raster_texture_fwd_kernel, dim3(jaf_cdiv(S, 64), jaf_cdiv(S, 4), B), dim3(256), 0, (hipStream_t)s, faces,
                       textures, face_index_map, weight_map, depth_map, rgb_map, sampling_index_map, sampling_weight_map,
                       background, bg_per_image ? 1 : 0, NF, S, ts, eps);
    return jaf_launch_status();
}

// grad_textures[image, face, tap] += w * grad_rgb[pixel]   (fp32 atomics resolved in L2; the texel block of a face is 81-192 floats)
template <bool REBUILD>
__global__ __launch_bounds__(256) void raster_texture_bwd_kernel(const float* __restrict__ faces, const int* __restrict__ fim,
                                                                 const float* __restrict__ wim, const float* __restrict__ depth,
                                                                 const int* __restrict__ tap_index,
                                                                 const float* __restrict__ tap_weight,
                                                                 const float* __restrict__ g_rgb, float* __restrict__ g_textures,
                                                                 int NF, int S, int ts, float eps) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), image = blockIdx.z;
    if (x >= S || y >= S) return;
    const long o = ((long)image * S + y) * S + x;
    const int fn = fim[o];
    if (fn < 0) return;
    const float g[3] = {g_rgb[o * 3], g_rgb[o * 3 + 1], g_rgb[o * 3 + 2]};
    if (g[0] == 0.f && g[1] == 0.f && g[2] == 0.f) return;           // only exact zeros would be added
    const long f = (long)image * NF + fn;
    Taps t;
    if (REBUILD) {
        t = texel_taps(faces + f * 9, wim + o * 3, depth[o], ts, eps);
    } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) { t.idx[c] = tap_index[o * 8 + c]; t.w[c] = tap_weight[o * 8 + c]; }
    }
    float* gt = g_textures + f * ts * ts * ts * 3;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        if (t.w[c] == 0.f) continue;
#pragma unroll
        for (int k = 0; k < 3; ++k) atomicAdd(&gt[t.idx[c] * 3 + k], t.w[c] * g[k]);
    }
}

extern "C" int jaf_rasterize_texture_bwd(jaf_stream_t s, const int32_t* face_index_map, const float* sampling_weight_map,
                                         const int32_t* sampling_index_map, const float* grad_rgb_map, float* grad_textures,
                                         int32_t B, int32_t NF, int32_t S, int32_t ts) {
    JAF_REQUIRE(face_index_map && sampling_weight_map && sampling_index_map && grad_rgb_map && grad_textures);
    JAF_REQUIRE(B >= 1 && B <= 65535 && NF >= 1 && S >= 1 && ts >= 1 && ts <= 64);
    hipLaunchKernelGGL((raster_texture_bwd_kernel<false>), dim3(jaf_cdiv(S, 64), jaf_cdiv(S, 4), B), dim3(256), 0, (hipStream_t)s,
                       (const float*)nullptr, face_index_map, (const float*)nullptr, (const float*)nullptr, sampling_index_map,
                       sampling_weight_map, grad_rgb_map, grad_textures, NF, S, ts, 0.f);
    return jaf_launch_status();
}

extern "C" int jaf_rasterize_texture_bwd_rebuild(jaf_stream_t s, const float* faces, const int32_t* face_index_map,
                                                 const float* weight_map, const float* depth_map, const float* grad_rgb_map,
                                                 float* grad_textures, int32_t B, int32_t NF, int32_t S, int32_t ts, float eps) {
    JAF_REQUIRE(faces && face_index_map && weight_map && depth_map && grad_rgb_map && grad_textures);
    JAF_REQUIRE(B >= 1 && B <= 65535 && NF >= 1 && S >= 1 && ts >= 2 && ts <= 64 && eps > 0.f);
    hipLaunchKernelGGL((raster_texture_bwd_kernel<true>), dim3(jaf_cdiv(S, 64), jaf_cdiv(S, 4), B), dim3(256), 0, (hipStream_t)s,
                       faces, face_index_map, weight_map, depth_map, (const int*)nullptr, (const float*)nullptr, grad_rgb_map,
                       grad_textures, NF, S, ts, eps);
    return jaf_launch_status();
}

// ---------------------------------------------------------------------------------------------
// lighting (neural_renderer/lighting.py:6-58): textures[b,f,...,c] *= ia*ca[c] + id*cd[c]*relu(n_f . dir)
// ---------------------------------------------------------------------------------------------
struct LightArgs {
    float ia, id;              // intensities
    float ca[3], cd[3], dir[3];
};

// unit normal of a face (v10 x v12, F.normalize eps 1e-5) and its cosine against the light direction
__device__ __forceinline__ void face_normal(const float* f, float n[3], float& len) {
    const float a[3] = {f[0] - f[3], f[1] - f[4], f[2] - f[5]};          // v0 - v1
    const float b[3] = {f[6] - f[3], f[7] - f[4], f[8] - f[5]};          // v2 - v1
    n[0] = a[1] * b[2] - a[2] * b[1];
    n[1] = a[2] * b[0] - a[0] * b[2];
    n[2] = a[0] * b[1] - a[1] * b[0];
    len = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
}

__global__ __launch_bounds__(256) void lighting_fwd_kernel(const float* __restrict__ faces, const float* __restrict__ tex_in,
                                                           float* __restrict__ tex_out, float* __restrict__ light_out,
                                                           LightArgs L, long total, int texels3) {
    const int lane = threadIdx.x & 63;
    const long fid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (fid >= total) return;
    float light[3] = {0.f, 0.f, 0.f};
    if (L.ia != 0.f)
        for (int c = 0; c < 3; ++c) light[c] += L.ia * L.ca[c];
    if (L.id != 0.f) {
        float n[3], len;
        face_normal(faces + fid * 9, n, len);
        const float d = fmaxf(len, 1e-5f);
        const float cs = fmaxf((n[0] / d) * L.dir[0] + (n[1] / d) * L.dir[1] + (n[2] / d) * L.dir[2], 0.f);
        for (int c = 0; c < 3; ++c) light[c] += L.id * (L.cd[c] * cs);
    }
    if (light_out && lane < 3) light_out[fid * 3 + lane] = lane == 0 ? light[0] : (lane == 1 ? light[1] : light[2]);
    const float* src = tex_in + fid * texels3;
    float* dst = tex_out + fid * texels3;
    for (int i = lane; i < texels3; i += 64) {
        const int c = i % 3;
        dst[i] = src[i] * (c == 0 ? light[0] : (c == 1 ? light[1] : light[2]));
    }
}

extern "C" int jaf_lighting_fwd(jaf_stream_t s, const float* faces, const float* textures_in, float* textures_out,
                                float* light_out, float intensity_ambient, float intensity_directional,
                                const float* color_ambient, const float* color_directional, const float* direction, int32_t B,
                                int32_t NF, int32_t ts) {
    JAF_REQUIRE(faces && textures_in && textures_out && color_ambient && color_directional && direction);
    JAF_REQUIRE(B >= 1 && NF >= 1 && ts >= 1 && ts <= 64);
    LightArgs L;
    L.ia = intensity_ambient; L.id = intensity_directional;
    for (int c = 0; c < 3; ++c) { L.ca[c] = color_ambient[c]; L.cd[c] = color_directional[c]; L.dir[c] = direction[c]; }
    const long total = (long)B * NF;
    hipLaunchKernelGGL(lighting_fwd_kernel, dim3((unsigned)jaf_cdiv(total, 4)), dim3(256), 0, (hipStream_t)s, faces, textures_in,
                       textures_out, light_out, L, total, ts * ts * ts * 3);
    return jaf_launch_status();
}

// adjoint: g_tex_in = g_tex_out * light; g_faces (directional light only) through relu, the dot product, F.normalize and the cross product
__global__ __launch_bounds__(256) void lighting_bwd_kernel(const float* __restrict__ faces, const float* __restrict__ tex_in,
                                                           const float* __restrict__ g_out, float* __restrict__ g_in,
                                                           float* __restrict__ g_faces, LightArgs L, long total, int texels3) {
    const int lane = threadIdx.x & 63;
    const long fid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (fid >= total) return;
    float light[3] = {0.f, 0.f, 0.f};
    if (L.ia != 0.f)
        for (int c = 0; c < 3; ++c) light[c] += L.ia * L.ca[c];
    float n[3] = {0.f, 0.f, 0.f}, len = 0.f, d = 1.f, cs_raw = 0.f;
    if (L.id != 0.f) {
        face_normal(faces + fid * 9, n, len);
        d = fmaxf(len, 1e-5f);
        cs_raw = (n[0] / d) * L.dir[0] + (n[1] / d) * L.dir[1] + (n[2] / d) * L.dir[2];
        const float cs = fmaxf(cs_raw, 0.f);
        for (int c = 0; c < 3; ++c) light[c] += L.id * (L.cd[c] * cs);
    }
    const float* src = tex_in + fid * texels3;
    const float* go = g_out + fid * texels3;
    float part[3] = {0.f, 0.f, 0.f};                 // d loss / d light[c] = sum over the face's texels of g_out * tex_in
    for (int i = lane; i < texels3; i += 64) {
        const int c = i % 3;
        const float g = go[i];
        if (g_in) g_in[fid * texels3 + i] = g * (c == 0 ? light[0] : (c == 1 ? light[1] : light[2]));
        if (g_faces) {
            const float v = g * src[i];
            if (c == 0) part[0] += v; else if (c == 1) part[1] += v; else part[2] += v;
        }
    }
    if (!g_faces) return;
    float gf[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (L.id != 0.f) {
        for (int c = 0; c < 3; ++c)
            for (int off = 32; off >= 1; off >>= 1) part[c] += __shfl_xor(part[c], off);
        if (cs_raw > 0.f) {
            const float g_cs = L.id * (L.cd[0] * part[0] + L.cd[1] * part[1] + L.cd[2] * part[2]);
            float g_unit[3] = {g_cs * L.dir[0], g_cs * L.dir[1], g_cs * L.dir[2]};     // w.r.t. the unit normal
            float g_n[3];
            if (len > 1e-5f) {       // u = n / |n|:  dn = (du - u (u . du)) / |n|
                const float u[3] = {n[0] / d, n[1] / d, n[2] / d};
                const float ud = u[0] * g_unit[0] + u[1] * g_unit[1] + u[2] * g_unit[2];
                for (int k = 0; k < 3; ++k) g_n[k] = (g_unit[k] - u[k] * ud) / d;
            } else {                 // u = n / eps
                for (int k = 0; k < 3; ++k) g_n[k] = g_unit[k] / d;
            }
            const float* f = faces + fid * 9;
            const float a[3] = {f[0] - f[3], f[1] - f[4], f[2] - f[5]};
            const float b[3] = {f[6] - f[3], f[7] - f[4], f[8] - f[5]};
            // n = a x b:  da = b x g_n,  db = g_n x a
            const float da[3] = {b[1] * g_n[2] - b[2] * g_n[1], b[2] * g_n[0] - b[0] * g_n[2], b[0] * g_n[1] - b[1] * g_n[0]};
            const float db[3] = {g_n[1] * a[2] - g_n[2] * a[1], g_n[2] * a[0] - g_n[0] * a[2], g_n[0] * a[1] - g_n[1] * a[0]};
            for (int k = 0; k < 3; ++k) { gf[k] = da[k]; gf[6 + k] = db[k]; gf[3 + k] = -da[k] - db[k]; }
        }
    }
    if (lane < 9) {
        float mine = gf[0];
#pragma unroll
        for (int k = 1; k < 9; ++k) mine = (lane == k) ? gf[k] : mine;
        g_faces[fid * 9 + lane] = mine;
    }
}

extern "C" int jaf_lighting_bwd(jaf_stream_t s, const float* faces, const float* textures_in, const float* grad_out,
                                float* grad_textures_in, float* grad_faces, float intensity_ambient,
                                float intensity_directional, const float* color_ambient, const float* color_directional,
                                const float* direction, int32_t B, int32_t NF, int32_t ts) {
    JAF_REQUIRE(faces && textures_in && grad_out && (grad_textures_in || grad_faces));
    JAF_REQUIRE(color_ambient && color_directional && direction && B >= 1 && NF >= 1 && ts >= 1 && ts <= 64);
    LightArgs L;
    L.ia = intensity_ambient; L.id = intensity_directional;
    for (int c = 0; c < 3; ++c) { L.ca[c] = color_ambient[c]; L.cd[c] = color_directional[c]; L.dir[c] = direction[c]; }
    const long total = (long)B * NF;
    hipLaunchKernelGGL(lighting_bwd_kernel, dim3((unsigned)jaf_cdiv(total, 4)), dim3(256), 0, (hipStream_t)s, faces, textures_in,
                       grad_out, grad_textures_in, grad_faces, L, total, ts * ts * ts * 3);
    return jaf_launch_status();
}

// ---------------------------------------------------------------------------------------------
// SMPLRenderer.dynamic_sampler (src/nmr.py:388-395 = batch_orth_proj_idrot :445-458 -> points_to_faces :397-417 ->
// points_to_sampler :460-477) and the layout half of extract_tex (:366-386).
// One wave per face; lanes over the face's T*T texels.  sampler[b,f,j,:] = clamp(p2 + (p0-p2)*a_j + (p1-p2)*b_j, -1, 1)
// with p_k = sc * (v_k.xy + t) the orthographic projection of the face's vertices and (a_j, b_j) = coords[:, j].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void face_sampler_fwd_kernel(const float* __restrict__ verts, const float* __restrict__ cam,
                                                               const int* __restrict__ fidx, const float* __restrict__ coords,
                                                               float* __restrict__ sampler, long total, int NV, int NF, int TT) {
    const int lane = threadIdx.x & 63;
    const long fid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (fid >= total) return;
    const int b = (int)(fid / NF), f = (int)(fid % NF);
    const float sc = cam[b * 3], tx = cam[b * 3 + 1], ty = cam[b * 3 + 2];
    float p[3][2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float* v = verts + ((long)b * NV + fidx[f * 3 + k]) * 3;
        p[k][0] = sc * (v[0] + tx);
        p[k][1] = sc * (v[1] + ty);
    }
    for (int j = lane; j < TT; j += 64) {
        const float a = coords[j], bb = coords[TT + j];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const float s = ((p[0][d] - p[2][d]) * a + (p[1][d] - p[2][d]) * bb) + p[2][d];
            sampler[(fid * TT + j) * 2 + d] = fminf(fmaxf(s, -1.f), 1.f);
        }
    }
}

extern "C" int jaf_face_sampler_fwd(jaf_stream_t s, const float* verts, const float* cam, const int32_t* faces_idx,
                                    const float* coords, float* sampler, int32_t B, int32_t NV, int32_t NF, int32_t TT) {
    JAF_REQUIRE(verts && cam && faces_idx && coords && sampler && B >= 1 && NV >= 1 && NF >= 1 && TT >= 1);
    const long total = (long)B * NF;
    hipLaunchKernelGGL(face_sampler_fwd_kernel, dim3((unsigned)jaf_cdiv(total, 4)), dim3(256), 0, (hipStream_t)s, verts, cam,
                       faces_idx, coords, sampler, total, NV, NF, TT);
    return jaf_launch_status();
}

// adjoint: dverts[B,NV,3] += (xy only), dcam[B,3] += (nullable); the clamp passes the gradient where -1 <= s <= 1
__global__ __launch_bounds__(256) void face_sampler_bwd_kernel(const float* __restrict__ verts, const float* __restrict__ cam,
                                                               const int* __restrict__ fidx, const float* __restrict__ coords,
                                                               const float* __restrict__ g_sampler, float* __restrict__ dverts,
                                                               float* __restrict__ dcam, long total, int NV, int NF, int TT) {
    const int lane = threadIdx.x & 63;
    const long fid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (fid >= total) return;
    const int b = (int)(fid / NF), f = (int)(fid % NF);
    const float sc = cam[b * 3], tx = cam[b * 3 + 1], ty = cam[b * 3 + 2];
    int vi[3];
    float q[3][2], p[3][2];                        // q = v.xy + t, p = sc * q
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        vi[k] = fidx[f * 3 + k];
        const float* v = verts + ((long)b * NV + vi[k]) * 3;
        q[k][0] = v[0] + tx; q[k][1] = v[1] + ty;
        p[k][0] = sc * q[k][0]; p[k][1] = sc * q[k][1];
    }
    float dp[3][2] = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    for (int j = lane; j < TT; j += 64) {
        const float a = coords[j], bb = coords[TT + j];
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            const float s = ((p[0][d] - p[2][d]) * a + (p[1][d] - p[2][d]) * bb) + p[2][d];
            const float g = (s >= -1.f && s <= 1.f) ? g_sampler[(fid * TT + j) * 2 + d] : 0.f;
            dp[0][d] += g * a;
            dp[1][d] += g * bb;
            dp[2][d] += g * ((1.f - a) - bb);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int d = 0; d < 2; ++d)
            for (int off = 32; off >= 1; off >>= 1) dp[k][d] += __shfl_xor(dp[k][d], off);
    if (lane == 0) {
        float dsc = 0.f, dtx = 0.f, dty = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float* dv = dverts + ((long)b * NV + vi[k]) * 3;
            if (dp[k][0] != 0.f) atomicAdd(&dv[0], sc * dp[k][0]);
            if (dp[k][1] != 0.f) atomicAdd(&dv[1], sc * dp[k][1]);
            dsc += dp[k][0] * q[k][0] + dp[k][1] * q[k][1];
            dtx += sc * dp[k][0];
            dty += sc * dp[k][1];
        }
        if (dcam) {
            if (dsc != 0.f) atomicAdd(&dcam[b * 3], dsc);
            if (dtx != 0.f) atomicAdd(&dcam[b * 3 + 1], dtx);
            if (dty != 0.f) atomicAdd(&dcam[b * 3 + 2], dty);
        }
    }
}

extern "C" int jaf_face_sampler_bwd(jaf_stream_t s, const float* verts, const float* cam, const int32_t* faces_idx,
                                    const float* coords, const float* grad_sampler, float* dverts, float* dcam, int32_t B,
                                    int32_t NV, int32_t NF, int32_t TT) {
    JAF_REQUIRE(verts && cam && faces_idx && coords && grad_sampler && dverts && B >= 1 && NV >= 1 && NF >= 1 && TT >= 1);
    const long total = (long)B * NF;
    hipLaunchKernelGGL(face_sampler_bwd_kernel, dim3((unsigned)jaf_cdiv(total, 4)), dim3(256), 0, (hipStream_t)s, verts, cam,
                       faces_idx, coords, grad_sampler, dverts, dcam, total, NV, NF, TT);
    return jaf_launch_status();
}

// extract_tex's view / permute / unsqueeze / repeat (src/nmr.py:379-384): sampled[B,3,NF,T*T] -> tex[B,NF,T,T,T,3], the T x T
// samples repeated along the third texture axis; adjoint = sum over that axis.
__global__ void tex_expand_fwd_kernel(const float* __restrict__ sampled, float* __restrict__ tex, long total, int NF, int T) {
    const long gs = (long)gridDim.x * blockDim.x;
    const int TT = T * T;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gs) {        // e over [B,NF,T,T,T,3]
        const int c = (int)(e % 3);
        const long r = e / 3 / T;                      // drops the channel and the repeated axis
        const int ij = (int)(r % TT);
        const long bf = r / TT;
        const long b = bf / NF, f = bf % NF;
        tex[e] = sampled[((b * 3 + c) * NF + f) * TT + ij];
    }
}

__global__ void tex_expand_bwd_kernel(const float* __restrict__ g_tex, float* __restrict__ g_sampled, long total, int NF, int T) {
    const long gs = (long)gridDim.x * blockDim.x;
    const int TT = T * T;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gs) {        // e over [B,3,NF,T*T]
        const int ij = (int)(e % TT);
        const long r = e / TT;
        const long f = r % NF;
        const long bc = r / NF;
        const long b = bc / 3;
        const int c = (int)(bc % 3);
        const float* src = g_tex + (((b * NF + f) * TT + ij) * T) * 3 + c;
        float acc = 0.f;
        for (int k = 0; k < T; ++k) acc += src[k * 3];
        g_sampled[e] = acc;
    }
}

extern "C" int jaf_tex_expand_fwd(jaf_stream_t s, const float* sampled, float* tex, int32_t B, int32_t NF, int32_t T) {
    JAF_REQUIRE(sampled && tex && B >= 1 && NF >= 1 && T >= 1);
    const long total = (long)B * NF * T * T * T * 3;
    hipLaunchKernelGGL(tex_expand_fwd_kernel, dim3(jaf_ew_grid(total)), dim3(256), 0, (hipStream_t)s, sampled, tex, total, NF, T);
    return jaf_launch_status();
}

extern "C" int jaf_tex_expand_bwd(jaf_stream_t s, const float* grad_tex, float* grad_sampled, int32_t B, int32_t NF, int32_t T) {
    JAF_REQUIRE(grad_tex && grad_sampled && B >= 1 && NF >= 1 && T >= 1);
    const long total = (long)B * 3 * NF * T * T;
    hipLaunchKernelGGL(tex_expand_bwd_kernel, dim3(jaf_ew_grid(total)), dim3(256), 0, (hipStream_t)s, grad_tex, grad_sampled, total, NF, T);
    return jaf_launch_status();
}

// neural_renderer.vertices_to_faces (vertices_to_faces.py:4-22): faces[B,NF,3,3] = verts[b, faces_idx[f,k], :]; adjoint = scatter-add.
__global__ void gather_face_vertices_kernel(const float* __restrict__ verts, const int* __restrict__ fidx, float* __restrict__ faces,
                                            long total, int NV, int NF) {
    const long gs = (long)gridDim.x * blockDim.x;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gs) {        // e over [B,NF,3] corners
        const long b = e / (3L * NF);
        const float* v = verts + (b * NV + fidx[e % (3L * NF)]) * 3;
        faces[e * 3] = v[0]; faces[e * 3 + 1] = v[1]; faces[e * 3 + 2] = v[2];
    }
}

__global__ void scatter_face_vertices_kernel(const float* __restrict__ dfaces, const int* __restrict__ fidx,
                                             float* __restrict__ dverts, long total, int NV, int NF) {
    const long gs = (long)gridDim.x * blockDim.x;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gs) {
        const long b = e / (3L * NF);
        float* dv = dverts + (b * NV + fidx[e % (3L * NF)]) * 3;
        for (int c = 0; c < 3; ++c) {
            const float g = dfaces[e * 3 + c];
            if (g != 0.f) atomicAdd(&dv[c], g);
        }
    }
}

extern "C" int jaf_vertices_to_faces(jaf_stream_t s, const float* verts, const int32_t* faces_idx, float* faces, int32_t B,
                                     int32_t NV, int32_t NF) {
    JAF_REQUIRE(verts && faces_idx && faces && B >= 1 && NV >= 1 && NF >= 1);
    const long total = (long)B * NF * 3;
    hipLaunchKernelGGL(gather_face_vertices_kernel, dim3(jaf_ew_grid(total)), dim3(256), 0, (hipStream_t)s, verts, faces_idx, faces, total, NV, NF);
    return jaf_launch_status();
}

extern "C" int jaf_vertices_to_faces_bwd(jaf_stream_t s, const float* dfaces, const int32_t* faces_idx, float* dverts, int32_t B,
                                         int32_t NV, int32_t NF) {
    JAF_REQUIRE(dfaces && faces_idx && dverts && B >= 1 && NV >= 1 && NF >= 1);
    const long total = (long)B * NF * 3;
    hipLaunchKernelGGL(scatter_face_vertices_kernel, dim3(jaf_ew_grid(total)), dim3(256), 0, (hipStream_t)s, dfaces, faces_idx, dverts, total, NV, NF);
    return jaf_launch_status();
}
