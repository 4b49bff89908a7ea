// Gradients of the rasteriser w.r.t. the projected faces: the silhouette / colour gradient of
// neural_renderer's backward_pixel_map and the depth gradient of backward_depth_map
// (third_party/neural_renderer/neural_renderer/cuda/rasterize_cuda_kernel.cu:245-491, :537-593; called from
// RasterizeFunction.backward, rasterize.py:101-160).
//
// MI355X design (not the reference's one-thread-per-face / one-thread-per-pixel decomposition):
//
//  * raster_edge_sweep_bwd_kernel -- ONE 64-LANE WAVE PER FACE.  The approximate gradient is a sum over pixel runs
//    that start at an edge crossing: for every edge, scan axis and scan line `u` there is one run that leaves the
//    face ("outside": from the crossing to the image border, up to S pixels) and one that enters it ("inside": up
//    to the opposite edge).  The wave keeps the few-iteration (edge, axis, u) loop uniform and spreads each RUN
//    over its lanes: the map reads of a run are one coalesced request per 64 pixels for the x-runs and one strided
//    gather for the y-runs, and the two divisions per pixel run lane-parallel.  fp32 addition is not associative,
//    so the run's terms are then folded into the wave-uniform accumulator IN PIXEL ORDER (ballot of the lanes with a
//    positive term, v_readlane per set bit): the result is bit-identical to a serial walk -- which is what the
//    reference's single thread produces and what oracle/raster_oracle.c restates -- while everything expensive ran
//    64 wide.  No atomics: a wave owns its face's nine outputs.
//
//  * raster_depth_bwd_tile_kernel -- ONE WORKGROUP PER 16x16 PIXEL TILE.  Every foreground pixel contributes nine
//    values to ITS face; neighbouring pixels mostly share a face, so the tile first merges its contributions in an
//    LDS hash table keyed by face index (ds_add_f32) and then issues one global atomic per (face, component) it
//    touched instead of nine per pixel (~20x fewer L2 atomics on the SMPL mesh at 256x256).
//
// Built with -ffp-contract=off like raster.hip: the per-term fp32 / fp64 expression trees are the reference's.
#include "jaf_common.h"

namespace {

// folds `term` of the lanes in `mask` into the wave-uniform `acc`, lowest lane first: acc = (..(acc - t0) - t1 ..)
__device__ __forceinline__ float fold_in_lane_order(float acc, float term, unsigned long long mask) {
    while (mask) {
        const int l = __builtin_ctzll(mask);
        acc = acc - __int_as_float(__builtin_amdgcn_readlane(__float_as_int(term), l));
        mask &= mask - 1ull;
    }
    return acc;
}

struct RunMaps {            // the maps a run reads (nullable pairs: colour and / or coverage)
    const float* rgb;       // [B,S,S,3]
    const float* alpha;     // [B,S,S]
    const float* g_rgb;
    const float* g_alpha;
    const int* fim;         // [B,S,S]
};

// image-plane position of vertex coordinate c of a face in pixel units (rasterize_cuda_kernel.cu:286)
__device__ __forceinline__ float to_pixel(float c, int S) { return (float)(0.5 * (double)(c * S + S - 1)); }

// signed pixel distance of the run pixel `v` from the crossing, scaled along the edge towards one of its end points
// (:395-412): `span` = Bu - Au, `reach` = distance of that end point from the scan line
__device__ __forceinline__ float pull(float span, float reach, int v, float v_edge, int S, float eps) {
    float dist = (float)((double)(span / reach * (v - v_edge)) * 2. / S);
    return (0 < dist) ? dist + eps : dist - eps;
}

}  // namespace

template <bool RGB, bool ALPHA>
__global__ __launch_bounds__(256) void raster_edge_sweep_bwd_kernel(const float* __restrict__ faces, RunMaps maps,
                                                                    float* __restrict__ grad_faces, long total, int NF,
                                                                    int S, float eps) {
    const int lane = threadIdx.x & 63;
    const long fid = (long)blockIdx.x * 4 + (threadIdx.x >> 6);          // wave-uniform: (image, face)
    if (fid >= total) return;
    const int image = (int)(fid / NF), fn = (int)(fid % NF);
    const float* face = faces + fid * 9;
    float* out = grad_faces + fid * 9;
    // back faces receive no gradient (:278-279); their entries stay as the caller zeroed them
    if ((face[7] - face[1]) * (face[3] - face[0]) < (face[4] - face[1]) * (face[6] - face[0])) return;

    float px[3], py[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { px[k] = to_pixel(face[3 * k], S); py[k] = to_pixel(face[3 * k + 1], S); }
    float acc[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const long plane = (long)image * S * S;

#pragma unroll
    for (int e = 0; e < 3; ++e) {
        const int ia = e, ib = (e + 1) % 3, ic = (e + 2) % 3;            // edge a->b, c = the vertex opposite to it
#pragma unroll
        for (int axis = 0; axis < 2; ++axis) {
            // u: the scan-line coordinate, v: the coordinate the runs extend along (axis 0: u = x, v = y)
            const float Au = axis ? py[ia] : px[ia], Av = axis ? px[ia] : py[ia];
            const float Bu = axis ? py[ib] : px[ib], Bv = axis ? px[ib] : py[ib];
            const float Cu = axis ? py[ic] : px[ic], Cv = axis ? px[ic] : py[ic];
            const int step = (axis == 0) ? ((Au < Bu) ? -1 : 1) : ((Au < Bu) ? 1 : -1);   // from inside to outside along v
            const long v_pitch = axis ? 1 : S, u_pitch = axis ? S : 1;
            const int u_first = (int)fmax((double)ceilf(fminf(Au, Bu)), 0.);
            const int u_last = (int)fmin((double)fmaxf(Au, Bu), S - 1.);
            const int comp = 1 - axis;                                     // which coordinate of a / b the runs push
            float ga = acc[ia * 3 + comp], gb = acc[ib * 3 + comp];
            for (int u = u_first; u <= u_last; ++u) {
                const float v_edge = (Bv - Av) / (Bu - Au) * (u - Au) + Av;
                const int v_in = (0 < step) ? (int)floorf(v_edge) : (int)ceilf(v_edge);
                const int v_out = v_in + step;
                if (v_in < 0 || S <= v_in || v_out < 0 || S <= v_out) continue;
                const long line = plane + (long)u * u_pitch;
                const long at_in = line + (long)v_in * v_pitch, at_out = line + (long)v_out * v_pitch;
                float a_in = 0.f, a_out = 0.f, c_in[3] = {0.f, 0.f, 0.f}, c_out[3] = {0.f, 0.f, 0.f};
                if (ALPHA) { a_in = maps.alpha[at_in]; a_out = maps.alpha[at_out]; }
                if (RGB) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) { c_in[k] = maps.rgb[at_in * 3 + k]; c_out[k] = maps.rgb[at_out * 3 + k]; }
                }
                const bool to_a = Bu != u, to_b = Au != u;                // the end point on the scan line gets no pull
                const float span = Bu - Au, reach_a = Bu - u, reach_b = u - Au;

                // run 1, outside: from the first pixel beyond the edge to the image border (:352-414)
                if (maps.fim[at_in] == fn) {
                    const int border = (0 < step) ? S - 1 : 0;
                    const int lo = max(min(v_out, border), 0), hi = min(max(v_out, border), S - 1);
                    for (int v0 = lo; v0 <= hi; v0 += 64) {
                        const int v = v0 + lane;
                        float diff = 0.f;
                        if (v <= hi) {
                            const long mp = line + (long)v * v_pitch;
                            if (ALPHA) diff += (maps.alpha[mp] - a_in) * maps.g_alpha[mp];
                            if (RGB) {
#pragma unroll
                                for (int k = 0; k < 3; ++k) diff += (maps.rgb[mp * 3 + k] - c_in[k]) * maps.g_rgb[mp * 3 + k];
                            }
                        }
                        const bool hit = v <= hi && diff > 0;
                        const unsigned long long m = __ballot(hit);
                        if (!m) continue;
                        if (to_a) ga = fold_in_lane_order(ga, hit ? diff / pull(span, reach_a, v, v_edge, S, eps) : 0.f, m);
                        if (to_b) gb = fold_in_lane_order(gb, hit ? diff / pull(span, reach_b, v, v_edge, S, eps) : 0.f, m);
                    }
                }
                // run 2, inside: this face's pixels from the edge back to where the scan line leaves the triangle (:416-482)
                {
                    float v_far;
                    if ((u - Au) * (u - Cu) < 0) v_far = (Cv - Av) / (Cu - Au) * (u - Au) + Av;
                    else v_far = (Bv - Cv) / (Bu - Cu) * (u - Cu) + Cv;
                    const int stop = (0 < step) ? (int)ceilf(v_far) : (int)floorf(v_far);
                    const int lo = max(min(v_in, stop), 0), hi = min(max(v_in, stop), S - 1);
                    for (int v0 = lo; v0 <= hi; v0 += 64) {
                        const int v = v0 + lane;
                        float diff = 0.f;
                        bool mine = false;
                        if (v <= hi) {
                            const long mp = line + (long)v * v_pitch;
                            mine = maps.fim[mp] == fn;
                            if (mine) {
                                if (ALPHA) diff += (maps.alpha[mp] - a_out) * maps.g_alpha[mp];
                                if (RGB) {
#pragma unroll
                                    for (int k = 0; k < 3; ++k) diff += (maps.rgb[mp * 3 + k] - c_out[k]) * maps.g_rgb[mp * 3 + k];
                                }
                            }
                        }
                        const bool hit = mine && diff > 0;
                        const unsigned long long m = __ballot(hit);
                        if (!m) continue;
                        if (to_a) ga = fold_in_lane_order(ga, hit ? diff / pull(span, reach_a, v, v_edge, S, eps) : 0.f, m);
                        if (to_b) gb = fold_in_lane_order(gb, hit ? diff / pull(span, reach_b, v, v_edge, S, eps) : 0.f, m);
                    }
                }
            }
            acc[ia * 3 + comp] = ga;
            acc[ib * 3 + comp] = gb;
        }
    }
    if (lane < 9) {
        float mine = acc[0];
#pragma unroll
        for (int k = 1; k < 9; ++k) mine = (lane == k) ? acc[k] : mine;
        out[lane] = mine;
    }
}

extern "C" int jaf_rasterize_bwd_pixel_map(jaf_stream_t s, const float* faces, const int32_t* face_index_map,
                                           const float* rgb_map, const float* alpha_map, const float* grad_rgb_map,
                                           const float* grad_alpha_map, float* grad_faces, int32_t B, int32_t NF, int32_t S,
                                           float eps) {
    JAF_REQUIRE(faces && face_index_map && grad_faces && B >= 1 && NF >= 1 && S >= 1);
    const bool rgb = rgb_map && grad_rgb_map, alpha = alpha_map && grad_alpha_map;
    JAF_REQUIRE(rgb || alpha);
    JAF_REQUIRE((long)B * NF <= 0x7fffffffL / 9 && (long)B * S * S <= 0x7fffffffL / 3);
    const long total = (long)B * NF;
    RunMaps maps{rgb ? rgb_map : nullptr, alpha ? alpha_map : nullptr, rgb ? grad_rgb_map : nullptr,
                 alpha ? grad_alpha_map : nullptr, face_index_map};
    const dim3 grid((unsigned)jaf_cdiv(total, 4)), block(256);
    hipStream_t st = (hipStream_t)s;
    if (rgb && alpha) hipLaunchKernelGGL((raster_edge_sweep_bwd_kernel<true, true>), grid, block, 0, st, faces, maps, grad_faces, total, NF, S, eps);
    else if (rgb) hipLaunchKernelGGL((raster_edge_sweep_bwd_kernel<true, false>), grid, block, 0, st, faces, maps, grad_faces, total, NF, S, eps);
    else hipLaunchKernelGGL((raster_edge_sweep_bwd_kernel<false, true>), grid, block, 0, st, faces, maps, grad_faces, total, NF, S, eps);
    return jaf_launch_status();
}

// ---------------------------------------------------------------------------------------------
// depth gradient (rasterize_cuda_kernel.cu:537-593): d depth / d face vertices, ADDED to grad_faces
// ---------------------------------------------------------------------------------------------
#define DT 16          // tile edge
#define DSLOTS 512     // hash slots per tile (a 16x16 tile sees at most 256 distinct faces)

__global__ __launch_bounds__(256) void raster_depth_bwd_tile_kernel(const float* __restrict__ faces,
                                                                    const float* __restrict__ depth_map,
                                                                    const int* __restrict__ fim,
                                                                    const float* __restrict__ finv_map,
                                                                    const float* __restrict__ wim,
                                                                    const float* __restrict__ g_depth,
                                                                    float* __restrict__ grad_faces, int NF, int S) {
    __shared__ int s_face[DSLOTS];
    __shared__ float s_sum[DSLOTS * 9];
    const int tid = threadIdx.x;
    for (int i = tid; i < DSLOTS; i += 256) s_face[i] = -1;
    for (int i = tid; i < DSLOTS * 9; i += 256) s_sum[i] = 0.f;
    __syncthreads();

    const int tiles_x = (S + DT - 1) / DT;
    const int x = (blockIdx.x % tiles_x) * DT + (tid % DT), y = (blockIdx.x / tiles_x) * DT + (tid / DT);
    const int image = blockIdx.y;
    if (x < S && y < S) {
        const long o = ((long)image * S + y) * S + x;
        const int fn = fim[o];
        const float gd = g_depth[o];
        if (fn >= 0 && gd != 0.f) {                       // a zero upstream gradient only adds exact zeros
            const float* face = faces + ((long)image * NF + fn) * 9;
            const float* inv = finv_map + o * 9;
            const float* w = wim + o * 3;
            const float d = depth_map[o];
            const float d2 = d * d;
            float c[9];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float zk = face[3 * k + 2];
                c[3 * k + 2] = gd * w[k] * d2 / (zk * zk);
            }
            float t[2] = {0.f, 0.f};                       // column sums of -inv / z  (only x and y are used)
#pragma unroll
            for (int l = 0; l < 2; ++l)
#pragma unroll
                for (int r = 0; r < 3; ++r) t[l] += -inv[3 * r + l] / face[3 * r + 2];
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int l = 0; l < 2; ++l) c[3 * k + l] = -gd * t[l] * w[k] * d2 * S / 2;
            unsigned slot = ((unsigned)fn * 2654435761u) >> 23;           // 9 bits
            for (;;) {
                const int prev = atomicCAS(&s_face[slot], -1, fn);
                if (prev == -1 || prev == fn) break;
                slot = (slot + 1) & (DSLOTS - 1);
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) atomicAdd(&s_sum[slot * 9 + k], c[k]);
        }
    }
    __syncthreads();
    for (int i = tid; i < DSLOTS * 9; i += 256) {
        const int fn = s_face[i / 9];
        const float v = s_sum[i];
        if (fn >= 0 && v != 0.f) atomicAdd(&grad_faces[((long)image * NF + fn) * 9 + (i % 9)], v);
    }
}

extern "C" int jaf_rasterize_bwd_depth_map(jaf_stream_t s, const float* faces, const float* depth_map,
                                           const int32_t* face_index_map, const float* face_inv_map, const float* weight_map,
                                           const float* grad_depth_map, float* grad_faces, int32_t B, int32_t NF, int32_t S) {
    JAF_REQUIRE(faces && depth_map && face_index_map && face_inv_map && weight_map && grad_depth_map && grad_faces);
    JAF_REQUIRE(B >= 1 && NF >= 1 && S >= 1 && B <= 65535);
    const int tiles = jaf_cdiv(S, DT);
    hipLaunchKernelGGL(raster_depth_bwd_tile_kernel, dim3(tiles * tiles, B), dim3(256), 0, (hipStream_t)s, faces, depth_map,
                       face_index_map, face_inv_map, weight_map, grad_depth_map, grad_faces, NF, S);
    return jaf_launch_status();
}
