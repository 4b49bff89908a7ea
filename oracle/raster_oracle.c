/*
 * ORACLE -- TEST INFRASTRUCTURE ONLY (never linked or called by the product path).
 *
 * Plain-C restatement of the neural_renderer face-index / weight-map rasteriser the reference
 * reaches through src/nmr.py:277:
 *   third_party/neural_renderer/neural_renderer/cuda/rasterize_cuda_kernel.cu:24-67   (face inverse)
 *   third_party/neural_renderer/neural_renderer/cuda/rasterize_cuda_kernel.cu:69-169  (pixel loop)
 *   third_party/neural_renderer/neural_renderer/rasterize.py:50-52 (fill values), :334-338 (flip)
 * The reference kernel is CUDA-only and cannot be built here (no nvcc), so parity is pinned on
 * the reference's own golden image instead: oracle/pin_rasterizer.py renders
 * tests/data/teapot.obj with this file and compares the silhouette with
 * tests/data/teapot_blender.png exactly as tests/test_rasterize_silhouettes.py:16-35 does.
 *
 * Build with -ffp-contract=off: fp32 expression trees as written (the product kernel is built
 * the same way), double where the CUDA source has double literals.
 */
#include <stdint.h>
#include <stdlib.h>

/* faces [B,NF,3,3] -> fim int32 [B,S,S] (-1), wim float [B,S,S,3] (0), both already flipped along H */
int raster_oracle_fim_wim(const float* faces, int32_t* fim, float* wim, int B, int NF, int is,
                          float near_, float far_) {
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
            }
        }
        const long o = ((long)bn * is + (is - 1 - yi)) * is + xi;     /* torch.flip(dims=(1,)) */
        fim[o] = face_index_min;
        for (int k = 0; k < 3; k++) wim[o * 3 + k] = weight_min[k];
    }
    free(faces_inv);
    return 0;
}
