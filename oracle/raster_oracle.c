/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (never linked or called by the product path).
 *
 * Plain-C restatement of the neural_renderer face-index / weight-map rasteriser the reference
 * reaches through src/nmr.py:277:
 *   third_party/neural_renderer/neural_renderer/cuda/rasterize_cuda_kernel.cu:24-67   (face inverse)
 *   third_party/neural_renderer/neural_renderer/cuda/rasterize_cuda_kernel.cu:69-169  (pixel loop, incl. the
 *                                                   depth map and the per-pixel face inverse of return_depth)
 *   third_party/neural_renderer/neural_renderer/cuda/rasterize_cuda_kernel.cu:245-491 (backward_pixel_map)
 *   third_party/neural_renderer/neural_renderer/cuda/rasterize_cuda_kernel.cu:537-593 (backward_depth_map)
 *   third_party/neural_renderer/neural_renderer/rasterize.py:50-52 (fill values), :334-338 (flip)
 * The reference kernel is CUDA-only and cannot be built here (no nvcc), so parity is pinned on
 * the reference's own golden image instead: oracle/pin_rasterizer.py renders
 * tests/data/teapot.obj with this file and compares the silhouette with
 * tests/data/teapot_blender.png exactly as tests/test_rasterize_silhouettes.py:16-35 does, the depth map with
 * tests/data/test_depth.png as tests/test_rasterize_depth.py:37-54 does, and the backward kernels are checked against
 * the known-answer gradients of tests/test_rasterize_silhouettes.py:37-99 (tests/test_oracle_golden.py).
 *
 * Build with -ffp-contract=off: fp32 expression trees as written (the product kernel is built
 * the same way), double where the CUDA source has double literals.
 */
#include <stdint.h>
#include <stdlib.h>

/* faces [B,NF,3,3] -> fim int32 [B,S,S] (-1), wim float [B,S,S,3] (0); optional depth [B,S,S] (far) and
 * face_inv_map [B,S,S,9] (0) as RasterizeFunction.forward fills them (rasterize.py:50-70, kernel :136-168).
 * flip != 0: outputs vertically flipped as rasterize_rgbad does (:334-338); flip == 0: the maps as the autograd
 * Function saves them for its backward pass. */
int raster_oracle_maps(const float* faces, int32_t* fim, float* wim, float* depth, float* face_inv_map, int B, int NF,
                       int is, float near_, float far_, int flip) {
    float* faces_inv = (float*)calloc((size_t)B * NF * 9, sizeof(float));
    if (!faces_inv) return -1;
    /* kernel_1 (:24-67) */
    for (long i = 0; i < (long)B * NF; ++i) {
        const float* face = faces + i * 9;
        float* face_inv_g = faces_inv + i * 9;
        if ((face[7] - face[1]) * (face[3] - face[0]) < (face[4] - face[1]) * (face[6] - face[0])) continue;
        float p[3][2];
        for (int num = 0; num < 3; num++)
            for (int dim = 0; dim < 2; dim++) p[num][dim] = (float)(0.5 * (double)(face[3 * num + dim] * is + is - 1));
        float face_inv[9] = {
            p[1][1] - p[2][1], p[2][0] - p[1][0], p[1][0] * p[2][1] - p[2][0] * p[1][1],
            p[2][1] - p[0][1], p[0][0] - p[2][0], p[2][0] * p[0][1] - p[0][0] * p[2][1],
            p[0][1] - p[1][1], p[1][0] - p[0][0], p[0][0] * p[1][1] - p[1][0] * p[0][1]};
        float den = (p[2][0] * (p[0][1] - p[1][1]) + p[0][0] * (p[1][1] - p[2][1]) + p[1][0] * (p[2][1] - p[0][1]));
        for (int k = 0; k < 9; k++) face_inv_g[k] = face_inv[k] / den;
    }
    /* kernel_2 (:69-169) */
    for (long i = 0; i < (long)B * is * is; ++i) {
        const int bn = (int)(i / ((long)is * is));
        const int pn = (int)(i % ((long)is * is));
        const int yi = pn / is, xi = pn % is;
        const float yp = (float)((2. * yi + 1 - is) / is);
        const float xp = (float)((2. * xi + 1 - is) / is);
        float depth_min = far_;
        int face_index_min = -1;
        float weight_min[3] = {0.f, 0.f, 0.f};
        float face_inv_min[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int fn = 0; fn < NF; fn++) {
            const float* face = faces + ((long)bn * NF + fn) * 9;
            const float* face_inv = faces_inv + ((long)bn * NF + fn) * 9;
            if ((face[7] - face[1]) * (face[3] - face[0]) < (face[4] - face[1]) * (face[6] - face[0])) continue;
            if (((yp - face[1]) * (face[3] - face[0]) < (xp - face[0]) * (face[4] - face[1])) ||
                ((yp - face[4]) * (face[6] - face[3]) < (xp - face[3]) * (face[7] - face[4])) ||
                ((yp - face[7]) * (face[0] - face[6]) < (xp - face[6]) * (face[1] - face[7])))
                continue;
            float w[3];
            w[0] = face_inv[0] * xi + face_inv[1] * yi + face_inv[2];
            w[1] = face_inv[3] * xi + face_inv[4] * yi + face_inv[5];
            w[2] = face_inv[6] * xi + face_inv[7] * yi + face_inv[8];
            float w_sum = 0;
            for (int k = 0; k < 3; k++) {
                float t = w[k] > 0.f ? w[k] : 0.f;     /* min(max(w,0.),1.) */
                w[k] = t < 1.f ? t : 1.f;
                w_sum += w[k];
            }
            for (int k = 0; k < 3; k++) w[k] /= w_sum;
            const float zp = (float)(1. / (double)(w[0] / face[2] + w[1] / face[5] + w[2] / face[8]));
            if (zp <= near_ || far_ <= zp) continue;
            if (zp < depth_min) {
                depth_min = zp;
                face_index_min = fn;
                for (int k = 0; k < 3; k++) weight_min[k] = w[k];
                for (int k = 0; k < 9; k++) face_inv_min[k] = face_inv[k];
            }
        }
        const long o = ((long)bn * is + (flip ? (is - 1 - yi) : yi)) * is + xi;     /* torch.flip(dims=(1,)) */
        fim[o] = face_index_min;
        for (int k = 0; k < 3; k++) wim[o * 3 + k] = weight_min[k];
        if (depth) depth[o] = depth_min;                               /* pre-filled with far (rasterize.py:52) */
        if (face_inv_map)
            for (int k = 0; k < 9; k++) face_inv_map[o * 9 + k] = face_inv_min[k];
    }
    free(faces_inv);
    return 0;
}

int raster_oracle_fim_wim(const float* faces, int32_t* fim, float* wim, int B, int NF, int is,
                          float near_, float far_) {
    return raster_oracle_maps(faces, fim, wim, 0, 0, B, NF, is, near_, far_, 1);
}

static float fminf_(float a, float b) { return a < b ? a : b; }
static float fmaxf_(float a, float b) { return a > b ? a : b; }
static double dmin_(double a, double b) { return a < b ? a : b; }
static double dmax_(double a, double b) { return a > b ? a : b; }
static int imin_(int a, int b) { return a < b ? a : b; }
static int imax_(int a, int b) { return a > b ? a : b; }
#include <math.h>

/* backward_pixel_map_cuda_kernel (rasterize_cuda_kernel.cu:245-491): one pass per face over its three edges and the
 * two scan axes.  Maps are the UNFLIPPED ones the Function saved; rgb_map/grad_rgb_map [B,S,S,3] and
 * alpha_map/grad_alpha_map [B,S,S] are nullable (return_rgb / return_alpha); grad_faces [B,NF,3,3] is overwritten
 * for front faces (the kernel returns early for back faces and leaves their pre-zeroed entries alone). */
int raster_oracle_bwd_pixel_map(const float* faces, const int32_t* face_index_map, const float* rgb_map,
                                const float* alpha_map, const float* grad_rgb_map, const float* grad_alpha_map,
                                float* grad_faces, int B, int NF, int is, float eps) {
    const int return_rgb = rgb_map && grad_rgb_map, return_alpha = alpha_map && grad_alpha_map;
    for (long i = 0; i < (long)B * NF; ++i) {
        const int bn = (int)(i / NF);
        const int fn = (int)(i % NF);
        const float* face = faces + i * 9;
        float grad_face[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if ((face[7] - face[1]) * (face[3] - face[0]) < (face[4] - face[1]) * (face[6] - face[0])) continue;
        for (int edge_num = 0; edge_num < 3; edge_num++) {
            int pi[3];
            float pp[3][2];
            for (int num = 0; num < 3; num++) pi[num] = (edge_num + num) % 3;
            for (int num = 0; num < 3; num++)
                for (int dim = 0; dim < 2; dim++) pp[num][dim] = (float)(0.5 * (double)(face[3 * pi[num] + dim] * is + is - 1));
            for (int axis = 0; axis < 2; axis++) {
                float p[3][2];
                for (int num = 0; num < 3; num++)
                    for (int dim = 0; dim < 2; dim++) p[num][dim] = pp[num][(dim + axis) % 2];
                int direction;
                if (axis == 0) direction = (p[0][0] < p[1][0]) ? -1 : 1;
                else direction = (p[0][0] < p[1][0]) ? 1 : -1;
                const int d0_from = (int)dmax_((double)ceilf(fminf_(p[0][0], p[1][0])), 0.);
                const int d0_to = (int)dmin_((double)fmaxf_(p[0][0], p[1][0]), is - 1.);
                for (int d0 = d0_from; d0 <= d0_to; d0++) {
                    int d1_in, d1_out;
                    const float d1_cross = (p[1][1] - p[0][1]) / (p[1][0] - p[0][0]) * (d0 - p[0][0]) + p[0][1];
                    if (0 < direction) d1_in = (int)floorf(d1_cross);
                    else d1_in = (int)ceilf(d1_cross);
                    d1_out = d1_in + direction;
                    if (d1_in < 0 || is <= d1_in) continue;
                    if (d1_out < 0 || is <= d1_out) continue;
                    float alpha_in = 0.f, alpha_out = 0.f;
                    const float *rgb_in = 0, *rgb_out = 0;
                    int map_index_in, map_index_out;
                    if (axis == 0) {
                        map_index_in = bn * is * is + d1_in * is + d0;
                        map_index_out = bn * is * is + d1_out * is + d0;
                    } else {
                        map_index_in = bn * is * is + d0 * is + d1_in;
                        map_index_out = bn * is * is + d0 * is + d1_out;
                    }
                    if (return_alpha) { alpha_in = alpha_map[map_index_in]; alpha_out = alpha_map[map_index_out]; }
                    if (return_rgb) { rgb_in = &rgb_map[(long)map_index_in * 3]; rgb_out = &rgb_map[(long)map_index_out * 3]; }
                    const int map_offset = (axis == 0) ? is : 1;
                    /* out */
                    if (face_index_map[map_index_in] == fn) {
                        const int d1_limit = (0 < direction) ? is - 1 : 0;
                        const int d1_from = imax_(imin_(d1_out, d1_limit), 0);
                        const int d1_to = imin_(imax_(d1_out, d1_limit), is - 1);
                        const int map_index_from = (axis == 0) ? bn * is * is + d1_from * is + d0 : bn * is * is + d0 * is + d1_from;
                        long mp = map_index_from;
                        for (int d1 = d1_from; d1 <= d1_to; d1++, mp += map_offset) {
                            float diff_grad = 0;
                            if (return_alpha) diff_grad += (alpha_map[mp] - alpha_in) * grad_alpha_map[mp];
                            if (return_rgb)
                                for (int k = 0; k < 3; k++) diff_grad += (rgb_map[mp * 3 + k] - rgb_in[k]) * grad_rgb_map[mp * 3 + k];
                            if (diff_grad <= 0) continue;
                            if (p[1][0] != d0) {
                                float dist = (float)((double)((p[1][0] - p[0][0]) / (p[1][0] - d0) * (d1 - d1_cross)) * 2. / is);
                                dist = (0 < dist) ? dist + eps : dist - eps;
                                grad_face[pi[0] * 3 + (1 - axis)] -= diff_grad / dist;
                            }
                            if (p[0][0] != d0) {
                                float dist = (float)((double)((p[1][0] - p[0][0]) / (d0 - p[0][0]) * (d1 - d1_cross)) * 2. / is);
                                dist = (0 < dist) ? dist + eps : dist - eps;
                                grad_face[pi[1] * 3 + (1 - axis)] -= diff_grad / dist;
                            }
                        }
                    }
                    /* in */
                    {
                        float d0_cross2;
                        if ((d0 - p[0][0]) * (d0 - p[2][0]) < 0)
                            d0_cross2 = (p[2][1] - p[0][1]) / (p[2][0] - p[0][0]) * (d0 - p[0][0]) + p[0][1];
                        else
                            d0_cross2 = (p[1][1] - p[2][1]) / (p[1][0] - p[2][0]) * (d0 - p[2][0]) + p[2][1];
                        const int d1_limit = (0 < direction) ? (int)ceilf(d0_cross2) : (int)floorf(d0_cross2);
                        const int d1_from = imax_(imin_(d1_in, d1_limit), 0);
                        const int d1_to = imin_(imax_(d1_in, d1_limit), is - 1);
                        const int map_index_from = (axis == 0) ? bn * is * is + d1_from * is + d0 : bn * is * is + d0 * is + d1_from;
                        long mp = map_index_from;
                        for (int d1 = d1_from; d1 <= d1_to; d1++, mp += map_offset) {
                            if (face_index_map[mp] != fn) continue;
                            float diff_grad = 0;
                            if (return_alpha) diff_grad += (alpha_map[mp] - alpha_out) * grad_alpha_map[mp];
                            if (return_rgb)
                                for (int k = 0; k < 3; k++) diff_grad += (rgb_map[mp * 3 + k] - rgb_out[k]) * grad_rgb_map[mp * 3 + k];
                            if (diff_grad <= 0) continue;
                            if (p[1][0] != d0) {
                                float dist = (float)((double)((p[1][0] - p[0][0]) / (p[1][0] - d0) * (d1 - d1_cross)) * 2. / is);
                                dist = (0 < dist) ? dist + eps : dist - eps;
                                grad_face[pi[0] * 3 + (1 - axis)] -= diff_grad / dist;
                            }
                            if (p[0][0] != d0) {
                                float dist = (float)((double)((p[1][0] - p[0][0]) / (d0 - p[0][0]) * (d1 - d1_cross)) * 2. / is);
                                dist = (0 < dist) ? dist + eps : dist - eps;
                                grad_face[pi[1] * 3 + (1 - axis)] -= diff_grad / dist;
                            }
                        }
                    }
                }
            }
        }
        for (int k = 0; k < 9; k++) grad_faces[i * 9 + k] = grad_face[k];
    }
    return 0;
}

/* backward_depth_map_cuda_kernel (rasterize_cuda_kernel.cu:537-593): ADDS to grad_faces (the reference uses
 * atomicAdd on the buffer backward_pixel_map wrote); pixel order here is ascending. */
int raster_oracle_bwd_depth_map(const float* faces, const float* depth_map, const int32_t* face_index_map,
                                const float* face_inv_map, const float* weight_map, const float* grad_depth_map,
                                float* grad_faces, int B, int NF, int is) {
    for (long i = 0; i < (long)B * is * is; ++i) {
        const int fn = face_index_map[i];
        if (0 <= fn) {
            const int bn = (int)(i / ((long)is * is));
            const float* face = faces + ((long)bn * NF + fn) * 9;
            const float depth = depth_map[i];
            const float depth2 = depth * depth;
            const float* face_inv = face_inv_map + i * 9;
            const float* weight = weight_map + i * 3;
            const float grad_depth = grad_depth_map[i];
            float* grad_face = grad_faces + ((long)bn * NF + fn) * 9;
            for (int k = 0; k < 3; k++) {
                const float z_k = face[3 * k + 2];
                grad_face[3 * k + 2] += grad_depth * weight[k] * depth2 / (z_k * z_k);
            }
            float tmp[3] = {0.f, 0.f, 0.f};
            for (int k = 0; k < 3; k++)
                for (int l = 0; l < 3; l++) tmp[k] += -face_inv[3 * l + k] / face[3 * l + 2];
            for (int k = 0; k < 3; k++)
                for (int l = 0; l < 2; l++) grad_face[3 * k + l] += -grad_depth * tmp[l] * weight[k] * depth2 * is / 2;
        }
    }
    return 0;
}

/* forward_texture_sampling_cuda_kernel (rasterize_cuda_kernel.cu:171-243) followed by forward_background
 * (rasterize.py:194-202).  textures [B,NF,ts,ts,ts,3]; maps UNFLIPPED; rgb_map [B,S,S,3]; sampling_index_map int32
 * [B,S,S,8] and sampling_weight_map [B,S,S,8] keep their fill value 0 where no face is hit (rasterize.py:56-58).
 * background [3] (bg_per_image == 0) or [B,3]. */
int raster_oracle_texture_fwd(const float* faces, const float* textures, const int32_t* face_index_map,
                              const float* weight_map, const float* depth_map, float* rgb_map, int32_t* sampling_index_map,
                              float* sampling_weight_map, const float* background, int bg_per_image, int B, int NF, int is,
                              int ts, float eps) {
    for (long i = 0; i < (long)B * is * is; ++i) {
        const int face_index = face_index_map[i];
        const int bn = (int)(i / ((long)is * is));
        float* pixel = rgb_map + i * 3;
        int32_t* sampling_indices = sampling_index_map + i * 8;
        float* sampling_weights = sampling_weight_map + i * 8;
        for (int k = 0; k < 3; k++) pixel[k] = 0.f;
        for (int pn = 0; pn < 8; pn++) { sampling_indices[pn] = 0; sampling_weights[pn] = 0.f; }
        if (face_index >= 0) {
            const float* face = faces + ((long)bn * NF + face_index) * 9;
            const float* texture = textures + ((long)bn * NF + face_index) * ts * ts * ts * 3;
            const float* weight = weight_map + i * 3;
            const float depth = depth_map[i];
            float texture_index_float[3];
            for (int k = 0; k < 3; k++) {
                float tif = weight[k] * (ts - 1) * (depth / (face[3 * k + 2]));
                tif = (float)dmax_((double)tif, 0.);                    /* max(tif, 0.) promotes to double in the CUDA source */
                tif = (float)dmin_((double)tif, (double)(ts - 1 - eps));   /* min(tif, ts - 1 - eps): int - float = float, then double */
                texture_index_float[k] = tif;
            }
            float new_pixel[3] = {0, 0, 0};
            for (int pn = 0; pn < 8; pn++) {
                float w = 1;
                int texture_index_int[3];
                for (int k = 0; k < 3; k++) {
                    if ((pn >> k) % 2 == 0) {
                        w *= 1 - (texture_index_float[k] - (int)texture_index_float[k]);
                        texture_index_int[k] = (int)texture_index_float[k];
                    } else {
                        w *= texture_index_float[k] - (int)texture_index_float[k];
                        texture_index_int[k] = (int)texture_index_float[k] + 1;
                    }
                }
                const int isc = texture_index_int[0] * ts * ts + texture_index_int[1] * ts + texture_index_int[2];
                for (int k = 0; k < 3; k++) new_pixel[k] += w * texture[isc * 3 + k];
                sampling_indices[pn] = isc;
                sampling_weights[pn] = w;
            }
            for (int k = 0; k < 3; k++) pixel[k] = new_pixel[k];
        }
        /* rgb_map * mask + (1 - mask) * background_color */
        const float mask = face_index >= 0 ? 1.f : 0.f;
        const float* bg = background + (bg_per_image ? bn * 3 : 0);
        for (int k = 0; k < 3; k++) pixel[k] = pixel[k] * mask + (1 - mask) * bg[k];
    }
    return 0;
}

/* backward_textures_cuda_kernel (rasterize_cuda_kernel.cu:506-541): ADDS into grad_textures (pixel order ascending). */
int raster_oracle_texture_bwd(const int32_t* face_index_map, const float* sampling_weight_map,
                              const int32_t* sampling_index_map, const float* grad_rgb_map, float* grad_textures, int B,
                              int NF, int is, int ts) {
    for (long i = 0; i < (long)B * is * is; ++i) {
        const int face_index = face_index_map[i];
        if (0 <= face_index) {
            const int bn = (int)(i / ((long)is * is));
            float* grad_texture = grad_textures + ((long)bn * NF + face_index) * ts * ts * ts * 3;
            for (int pn = 0; pn < 8; pn++) {
                const float w = sampling_weight_map[i * 8 + pn];
                const int isc = sampling_index_map[i * 8 + pn];
                for (int k = 0; k < 3; k++) grad_texture[isc * 3 + k] += w * grad_rgb_map[i * 3 + k];
            }
        }
    }
    return 0;
}
