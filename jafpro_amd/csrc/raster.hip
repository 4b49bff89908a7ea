// Renderer path of float_estimate (src/cal_flow.py:21-39): SMPL projection, the neural_renderer
// face-index / weight-map rasteriser (rasterize_cuda_kernel.cu:24-169) and cal_bc_transform
// (src/nmr.py:617-659).
//
// The reference rasteriser tests every pixel against all 13 776 faces.  Here a workgroup owns a
// 16x16 pixel tile, sweeps the face list once with a bounding-box overlap test (256 faces per
// sweep step, order-preserving wave compaction), stages the survivors (vertices + inverse) in
// LDS and lets its 256 pixels walk that short list with the reference's exact per-pixel
// arithmetic.  Faces are visited in ascending index order, so "first strictly smaller z wins"
// (rasterize_cuda_kernel.cu:142) is preserved.  This file is compiled with -ffp-contract=off so
// the fp32 expression trees match the C restatement in oracle/ bit for bit.
#include "jaf_common.h"

#define RT 16        // tile edge
#define LCAP 768     // LDS face list capacity

struct FaceRec {     // 13 x 4 bytes per face in the workspace
    float adj[9];
    int bb[4];       // xmin, xmax, ymin, ymax (pixel units, kernel orientation); xmin>xmax = culled
};

// verts[B,NV,3], cam[B,3] -> faces[B,NF,3,3]   (src/nmr.py:19-28,269-276 + look_at with
// eye=(0,0,eye_z), at=0, up=+y, whose rotation is exactly the identity)
__global__ void project_faces_kernel(const float* verts, const float* cam, const int* fidx, float* faces, int B,
                                     int NV, int NF, float eye_z) {
    const long total = (long)B * NF * 3;
    const long gs = (long)gridDim.x * blockDim.x;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gs) {
        const int k = (int)(e % 3);
        const int f = (int)((e / 3) % NF);
        const int b = (int)(e / (3L * NF));
        const int vi = fidx[f * 3 + k];
        const float* v = verts + ((long)b * NV + vi) * 3;
        const float sc = cam[b * 3], tx = cam[b * 3 + 1], ty = cam[b * 3 + 2];
        float* o = faces + e * 3;
        o[0] = sc * (v[0] + tx);
        o[1] = -(sc * (v[1] + ty));
        o[2] = v[2] - eye_z;
    }
}

extern "C" int jaf_project_faces(jaf_stream_t s, const float* verts, const float* cam, const int32_t* faces_idx,
                                 float* faces, int32_t B, int32_t NV, int32_t NF, float eye_z) {
    JAF_REQUIRE(verts && cam && faces_idx && faces && B >= 1 && NV >= 1 && NF >= 1);
    hipLaunchKernelGGL(project_faces_kernel, dim3(jaf_ew_grid((long)B * NF * 3)), dim3(256), 0, (hipStream_t)s, verts, cam, faces_idx, faces, B, NV, NF, eye_z);
    return jaf_launch_status();
}

// rasterize_cuda_kernel.cu:24-67 plus a conservative pixel bounding box
__global__ void raster_setup_kernel(const float* faces, FaceRec* rec, int total, int is) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float* tri = faces + (long)i * 9;
    FaceRec r;
    for (int k = 0; k < 9; ++k) r.adj[k] = 0.f;
    r.bb[0] = 1; r.bb[1] = 0; r.bb[2] = 1; r.bb[3] = 0;
    const bool back = (tri[7] - tri[1]) * (tri[3] - tri[0]) < (tri[4] - tri[1]) * (tri[6] - tri[0]);
    if (!back) {
        float pt[3][2];
        for (int v = 0; v < 3; v++)
            for (int ax = 0; ax < 2; ax++) pt[v][ax] = (float)(0.5 * (double)(tri[3 * v + ax] * is + is - 1));
        float adj[9] = {
            pt[1][1] - pt[2][1], pt[2][0] - pt[1][0], pt[1][0] * pt[2][1] - pt[2][0] * pt[1][1],
            pt[2][1] - pt[0][1], pt[0][0] - pt[2][0], pt[2][0] * pt[0][1] - pt[0][0] * pt[2][1],
            pt[0][1] - pt[1][1], pt[1][0] - pt[0][0], pt[0][0] * pt[1][1] - pt[1][0] * pt[0][1]};
        const float area2 = (pt[2][0] * (pt[0][1] - pt[1][1]) + pt[0][0] * (pt[1][1] - pt[2][1]) + pt[1][0] * (pt[2][1] - pt[0][1]));
        for (int k = 0; k < 9; ++k) r.adj[k] = adj[k] / area2;
        const float xmn = fminf(pt[0][0], fminf(pt[1][0], pt[2][0])), xmx = fmaxf(pt[0][0], fmaxf(pt[1][0], pt[2][0]));
        const float ymn = fminf(pt[0][1], fminf(pt[1][1], pt[2][1])), ymx = fmaxf(pt[0][1], fmaxf(pt[1][1], pt[2][1]));
        // clamp before the int conversion; one pixel of slack on every side
        r.bb[0] = (int)floorf(fmaxf(xmn, -4.f)) - 1;
        r.bb[1] = (int)ceilf(fminf(xmx, (float)is + 4.f)) + 1;
        r.bb[2] = (int)floorf(fmaxf(ymn, -4.f)) - 1;
        r.bb[3] = (int)ceilf(fminf(ymx, (float)is + 4.f)) + 1;
        if (!(xmn == xmn) || !(xmx == xmx) || !(ymn == ymn) || !(ymx == ymx)) {   // NaN: keep everywhere
            r.bb[0] = -1; r.bb[1] = is + 1; r.bb[2] = -1; r.bb[3] = is + 1;
        }
    }
    rec[i] = r;
}

__global__ __launch_bounds__(256) void raster_tile_kernel(const float* faces, const FaceRec* rec, int* fim,
                                                          float* wim, float* depth, float* finv_map, float* alpha,
                                                          int NF, int is, float near_, float far_, int flip) {
    __shared__ int s_list[LCAP];
    __shared__ float s_face[LCAP * 9];
    __shared__ float s_inv[LCAP * 9];
    __shared__ int s_wcount[4];
    __shared__ int s_count;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int tiles_x = (is + RT - 1) / RT;
    const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
    const int bn = blockIdx.y;
    const int xi = tx * RT + (tid % RT);
    const int yi = ty * RT + (tid / RT);
    const bool inside_img = xi < is && yi < is;
    const float yp = (float)((2. * yi + 1 - is) / is);
    const float xp = (float)((2. * xi + 1 - is) / is);
    const int tx0 = tx * RT, tx1 = tx * RT + RT - 1, ty0 = ty * RT, ty1 = ty * RT + RT - 1;

    const float* fbase = faces + (long)bn * NF * 9;
    const FaceRec* rbase = rec + (long)bn * NF;

    float z_best = far_;
    int f_best = -1;
    int slot_best = -1;          // LDS slot of the winning tri in the batch it came from (return_depth only)
    float bc_best[3] = {0.f, 0.f, 0.f};
    float bary_best[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    if (tid == 0) s_count = 0;
    __syncthreads();

    for (int base = 0; base < NF; base += 256) {
        const int f = base + tid;
        bool ov = false;
        if (f < NF) {
            const int* bb = rbase[f].bb;
            ov = bb[0] <= tx1 && bb[1] >= tx0 && bb[2] <= ty1 && bb[3] >= ty0 && bb[0] <= bb[1];
        }
        const unsigned long long m = __ballot(ov);
        const int wprefix = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_wcount[wave] = __popcll(m);
        __syncthreads();
        int off = s_count;
        for (int bc = 0; bc < wave; ++bc) off += s_wcount[bc];
        if (ov) s_list[off + wprefix] = f;
        __syncthreads();
        if (tid == 0) s_count += s_wcount[0] + s_wcount[1] + s_wcount[2] + s_wcount[3];
        __syncthreads();
        const int cnt = s_count;
        const bool last = base + 256 >= NF;
        if (cnt > LCAP - 256 || (last && cnt > 0)) {
            for (int e = tid; e < cnt * 9; e += 256) {
                const int i = e / 9, k = e - i * 9;
                const int ff = s_list[i];
                s_face[e] = fbase[(long)ff * 9 + k];
                s_inv[e] = rbase[ff].adj[k];
            }
            __syncthreads();
            if (inside_img) {
                for (int i = 0; i < cnt; ++i) {
                    const float* tri = s_face + i * 9;
                    const float* bary = s_inv + i * 9;
                    if (((yp - tri[1]) * (tri[3] - tri[0]) < (xp - tri[0]) * (tri[4] - tri[1])) ||
                        ((yp - tri[4]) * (tri[6] - tri[3]) < (xp - tri[3]) * (tri[7] - tri[4])) ||
                        ((yp - tri[7]) * (tri[0] - tri[6]) < (xp - tri[6]) * (tri[1] - tri[7])))
                        continue;
                    float bc[3];
                    bc[0] = bary[0] * xi + bary[1] * yi + bary[2];
                    bc[1] = bary[3] * xi + bary[4] * yi + bary[5];
                    bc[2] = bary[6] * xi + bary[7] * yi + bary[8];
                    float bc_sum = 0;
                    for (int k = 0; k < 3; k++) {
                        bc[k] = fminf(fmaxf(bc[k], 0.f), 1.f);
                        bc_sum += bc[k];
                    }
                    for (int k = 0; k < 3; k++) bc[k] /= bc_sum;
                    const float z_here = (float)(1. / (double)(bc[0] / tri[2] + bc[1] / tri[5] + bc[2] / tri[8]));
                    if (z_here <= near_ || far_ <= z_here) continue;
                    if (z_here < z_best) {
                        z_best = z_here;
                        f_best = s_list[i];
                        slot_best = i;
                        bc_best[0] = bc[0]; bc_best[1] = bc[1]; bc_best[2] = bc[2];
                    }
                }
                if (finv_map && slot_best >= 0) {          // the winner of this batch: keep its inverse before the LDS is reused
                    for (int k = 0; k < 9; ++k) bary_best[k] = s_inv[slot_best * 9 + k];
                    slot_best = -1;
                }
            }
            __syncthreads();
            if (tid == 0) s_count = 0;
            __syncthreads();
        }
    }

    if (inside_img) {
        // vertical flip of rasterize.py:334-338 folded into the store (flip == 0: the maps RasterizeFunction saves)
        const long o = ((long)bn * is + (flip ? (is - 1 - yi) : yi)) * is + xi;
        fim[o] = f_best;
        wim[o * 3 + 0] = bc_best[0];
        wim[o * 3 + 1] = bc_best[1];
        wim[o * 3 + 2] = bc_best[2];
        if (depth) depth[o] = z_best;                                   // far where nothing was hit (rasterize.py:52)
        if (alpha) alpha[o] = f_best >= 0 ? 1.f : 0.f;             // forward_alpha_map (rasterize.py:188-192)
        if (finv_map)
            for (int k = 0; k < 9; ++k) finv_map[o * 9 + k] = bary_best[k];
    }
}

extern "C" int64_t jaf_rasterize_workspace(int32_t B, int32_t NF, int32_t S) {
    (void)S;
    return (int64_t)B * NF * (int64_t)sizeof(FaceRec);
}

extern "C" int jaf_rasterize_fim_wim(jaf_stream_t s_, const float* faces, int32_t* fim, float* wim, void* workspace,
                                     int32_t B, int32_t NF, int32_t S, float near_, float far_) {
    JAF_REQUIRE(faces && fim && wim && workspace && B >= 1 && NF >= 1 && S >= 1 && B <= 65535);
    JAF_REQUIRE((long)B * NF <= 0x7fffffffL / 9 && (long)B * S * S <= 0x7fffffffL / 3);      // 32-bit element indices below
    hipStream_t s = (hipStream_t)s_;
    FaceRec* rec = (FaceRec*)workspace;
    hipLaunchKernelGGL(raster_setup_kernel, dim3(jaf_cdiv((long)B * NF, 256)), dim3(256), 0, s, faces, rec, B * NF, S);
    const int tiles = jaf_cdiv(S, RT);
    hipLaunchKernelGGL(raster_tile_kernel, dim3(tiles * tiles, B), dim3(256), 0, s, faces, rec, fim, wim, (float*)nullptr,
                       (float*)nullptr, (float*)nullptr, NF, S, near_, far_, 1);
    return jaf_launch_status();
}

extern "C" int jaf_rasterize_maps(jaf_stream_t s_, const float* faces, int32_t* fim, float* wim, float* depth,
                                  float* face_inv_map, float* alpha, void* workspace, int32_t B, int32_t NF, int32_t S,
                                  float near_, float far_, int flip) {
    JAF_REQUIRE(faces && fim && wim && workspace && B >= 1 && NF >= 1 && S >= 1 && B <= 65535);
    JAF_REQUIRE((long)B * NF <= 0x7fffffffL / 9 && (long)B * S * S <= 0x7fffffffL / 9);
    hipStream_t s = (hipStream_t)s_;
    FaceRec* rec = (FaceRec*)workspace;
    hipLaunchKernelGGL(raster_setup_kernel, dim3(jaf_cdiv((long)B * NF, 256)), dim3(256), 0, s, faces, rec, B * NF, S);
    const int tiles = jaf_cdiv(S, RT);
    hipLaunchKernelGGL(raster_tile_kernel, dim3(tiles * tiles, B), dim3(256), 0, s, faces, rec, fim, wim, depth, face_inv_map,
                       alpha, NF, S, near_, far_, flip ? 1 : 0);
    return jaf_launch_status();
}

// The rasteriser's backward kernels (silhouette / colour and depth gradients) live in raster_bwd.hip.

// Adjoint of project_faces_kernel: dfaces[B,NF,3,3] -> dverts[B,NV,3] (+=, a vertex belongs to ~6 faces) and
// dcam[B,3] (+=): x = sc*(vx+tx), y = -sc*(vy+ty), z = vz - eye_z.
__global__ void project_faces_bwd_kernel(const float* dfaces, const float* verts, const float* cam, const int* fidx,
                                         float* dverts, float* dcam, int B, int NV, int NF) {
    const long total = (long)B * NF * 3;
    const long gs = (long)gridDim.x * blockDim.x;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gs) {
        const int k = (int)(e % 3);
        const int f = (int)((e / 3) % NF);
        const int b = (int)(e / (3L * NF));
        const int vi = fidx[f * 3 + k];
        const float* g = dfaces + e * 3;
        const float gx = g[0], gy = g[1], gz = g[2];
        if (gx == 0.f && gy == 0.f && gz == 0.f) continue;
        const float sc = cam[b * 3], tx = cam[b * 3 + 1], ty = cam[b * 3 + 2];
        float* dv = dverts + ((long)b * NV + vi) * 3;
        atomicAdd(&dv[0], sc * gx);
        atomicAdd(&dv[1], -sc * gy);
        atomicAdd(&dv[2], gz);
        if (dcam) {
            const float* v = verts + ((long)b * NV + vi) * 3;
            atomicAdd(&dcam[b * 3], gx * (v[0] + tx) - gy * (v[1] + ty));
            atomicAdd(&dcam[b * 3 + 1], sc * gx);
            atomicAdd(&dcam[b * 3 + 2], -sc * gy);
        }
    }
}

extern "C" int jaf_project_faces_bwd(jaf_stream_t s, const float* dfaces, const float* verts, const float* cam,
                                     const int32_t* faces_idx, float* dverts, float* dcam, int32_t B, int32_t NV, int32_t NF) {
    JAF_REQUIRE(dfaces && verts && cam && faces_idx && dverts && B >= 1 && NV >= 1 && NF >= 1);
    hipLaunchKernelGGL(project_faces_bwd_kernel, dim3(jaf_ew_grid((long)B * NF * 3)), dim3(256), 0, (hipStream_t)s, dfaces, verts,
                       cam, faces_idx, dverts, dcam, B, NV, NF);
    return jaf_launch_status();
}

// src/nmr.py:617-659 with src/cal_flow.py:30-31 folded in (x,y of the SOURCE faces, y negated)
__global__ void bc_transform_kernel(const float* src_faces, const int* fim, const float* wim, float* T, int B, int NF,
                                    int S) {
    const long total = (long)B * S * S;
    const long gs = (long)gridDim.x * blockDim.x;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gs) {
        const long b = e / ((long)S * S);
        const int f = fim[e];
        float tx = -2.f, ty = -2.f;
        if (f >= 0 && f < NF) {
            const float* v = src_faces + (b * NF + f) * 9;
            const float w0 = wim[e * 3], w1 = wim[e * 3 + 1], w2 = wim[e * 3 + 2];
            tx = (v[0] * w0 + v[3] * w1) + v[6] * w2;
            ty = ((-v[1]) * w0 + (-v[4]) * w1) + (-v[7]) * w2;
        }
        T[e * 2] = tx;
        T[e * 2 + 1] = ty;
    }
}

// Adjoint of bc_transform_kernel w.r.t. the SOURCE faces (the weight and face-index maps carry no gradient in the
// reference either: RasterizeFunction.backward ignores grad_weight_map, rasterize.py:101-104): dsrc_faces[B,NF,3,3] +=.
__global__ void bc_transform_bwd_kernel(const float* dT, const int* fim, const float* wim, float* dsrc_faces, int B,
                                        int NF, int S) {
    const long total = (long)B * S * S;
    const long gs = (long)gridDim.x * blockDim.x;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gs) {
        const int f = fim[e];
        if (f < 0 || f >= NF) continue;
        const float gx = dT[e * 2], gy = dT[e * 2 + 1];
        if (gx == 0.f && gy == 0.f) continue;
        const long b = e / ((long)S * S);
        float* v = dsrc_faces + (b * NF + f) * 9;
        for (int k = 0; k < 3; ++k) {
            const float w = wim[e * 3 + k];
            atomicAdd(&v[3 * k], gx * w);
            atomicAdd(&v[3 * k + 1], -(gy * w));
        }
    }
}

extern "C" int jaf_bc_transform_bwd(jaf_stream_t s, const float* dT, const int32_t* fim, const float* wim, float* dsrc_faces,
                                    int32_t B, int32_t NF, int32_t S) {
    JAF_REQUIRE(dT && fim && wim && dsrc_faces && B >= 1 && NF >= 1 && S >= 1);
    hipLaunchKernelGGL(bc_transform_bwd_kernel, dim3(jaf_ew_grid((long)B * S * S)), dim3(256), 0, (hipStream_t)s, dT, fim, wim,
                       dsrc_faces, B, NF, S);
    return jaf_launch_status();
}

extern "C" int jaf_bc_transform(jaf_stream_t s, const float* src_faces, const int32_t* fim, const float* wim, float* T,
                                int32_t B, int32_t NF, int32_t S) {
    JAF_REQUIRE(src_faces && fim && wim && T && B >= 1 && NF >= 1 && S >= 1);
    hipLaunchKernelGGL(bc_transform_kernel, dim3(jaf_ew_grid((long)B * S * S)), dim3(256), 0, (hipStream_t)s, src_faces, fim, wim, T, B, NF, S);
    return jaf_launch_status();
}
