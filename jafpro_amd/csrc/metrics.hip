// Evaluation metrics on the device (test/video_evaluation.py:165-212): grayscale conversion, windowed
// SSIM statistics (skimage compare_ssim / the MS-SSIM scales) and per-frame error sums.  Frames are 256x256, so these
// are latency-sized kernels; what matters is that a whole video is scored without a host round trip per frame.
#include "jaf_common.h"

// cv2.cvtColor(COLOR_BGR2GRAY) on uint8: fixed-point Y = (B*1868 + G*9617 + R*4899 + 8192) >> 14 (OpenCV's 14-bit
// coefficients of 0.114 / 0.587 / 0.299).  in [N][HW][3] (B,G,R) -> out [N][HW].
__global__ void bgr_to_gray_u8_kernel(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, long total) {
    const long gs = (long)gridDim.x * blockDim.x;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gs) {
        const uint8_t* p = in + e * 3;
        out[e] = (uint8_t)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + 8192) >> 14);
    }
}

extern "C" int jaf_bgr_to_gray_u8(jaf_stream_t s, const uint8_t* in, uint8_t* out, int64_t npix) {
    JAF_REQUIRE(in && out && npix >= 1);
    hipLaunchKernelGGL(bgr_to_gray_u8_kernel, dim3(jaf_ew_grid(npix)), dim3(256), 0, (hipStream_t)s, in, out, (long)npix);
    return jaf_launch_status();
}

// Windowed SSIM of fp32 image pairs x, y [N][H][W] over the VALID window positions (the border a 'reflect'/'constant'
// filter would fill is exactly what skimage crops away, structural_similarity: crop(S, (win-1)//2)):
//   ux = sum w x, uxx = sum w x^2, ... ; vx = cn*(uxx - ux^2), vxy = cn*(uxy - ux*uy)
//   S = (2 ux uy + C1)(2 vxy + C2) / ((ux^2 + uy^2 + C1)(vx + vy + C2)),  CS = (2 vxy + C2) / (vx + vy + C2)
// w: win*win weights (uniform 1/49 for skimage's default, Gaussian 11/1.5 for MS-SSIM), cn = NP/(NP-1) or 1.
// sums[n][0] += sum S, sums[n][1] += sum CS (fp64).  One lane per window position, arithmetic in fp64.
__global__ __launch_bounds__(256) void ssim_window_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                         const double* __restrict__ w, double* __restrict__ sums, int H, int W,
                                                         int win, double cn, double C1, double C2) {
    const int n = blockIdx.y;
    const int OW = W - win + 1, OH = H - win + 1;
    const long total = (long)OH * OW;
    const float* xp = x + (long)n * H * W;
    const float* yp = y + (long)n * H * W;
    double sS = 0.0, sC = 0.0;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        const int oy = (int)(e / OW), ox = (int)(e - (long)oy * OW);
        double ux = 0, uy = 0, uxx = 0, uyy = 0, uxy = 0;
        for (int ky = 0; ky < win; ++ky)
            for (int kx = 0; kx < win; ++kx) {
                const double wk = w[ky * win + kx];
                const double a = (double)xp[(oy + ky) * W + ox + kx], b = (double)yp[(oy + ky) * W + ox + kx];
                ux += wk * a; uy += wk * b; uxx += wk * a * a; uyy += wk * b * b; uxy += wk * a * b;
            }
        const double vx = cn * (uxx - ux * ux), vy = cn * (uyy - uy * uy), vxy = cn * (uxy - ux * uy);
        const double A1 = 2 * ux * uy + C1, A2 = 2 * vxy + C2, B1 = ux * ux + uy * uy + C1, B2 = vx + vy + C2;
        sS += (A1 * A2) / (B1 * B2);
        sC += A2 / B2;
    }
    sS = jaf_wave_sum(sS);
    sC = jaf_wave_sum(sC);
    __shared__ double red[4][2];
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = sS; red[threadIdx.x >> 6][1] = sC; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&sums[n * 2], (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]));
        atomicAdd(&sums[n * 2 + 1], (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]));
    }
}

extern "C" int jaf_ssim_window_sums(jaf_stream_t s, const float* x, const float* y, const double* weights, double* sums,
                                    int32_t N, int32_t H, int32_t W, int32_t win, double cov_norm, double C1, double C2) {
    JAF_REQUIRE(x && y && weights && sums && N >= 1 && N <= 65535 && win >= 1 && H >= win && W >= win);
    const long total = (long)(H - win + 1) * (W - win + 1);
    int blocks = jaf_cdiv(total, 256);
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(ssim_window_kernel, dim3(blocks, N), dim3(256), 0, (hipStream_t)s, x, y, weights, sums, H, W, win, cov_norm, C1, C2);
    return jaf_launch_status();
}

// Per-frame sums of (a-b)^2 and |a-b| of uint8 frames [N][P] -> sums[n][2] (fp64, exact integers): PSNR and L1.
__global__ __launch_bounds__(256) void frame_error_sums_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b,
                                                              double* __restrict__ sums, long P) {
    const int n = blockIdx.y;
    const uint8_t* ap = a + (long)n * P;
    const uint8_t* bp = b + (long)n * P;
    long long s2 = 0, s1 = 0;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < P; e += (long)gridDim.x * blockDim.x) {
        const int d = (int)ap[e] - (int)bp[e];
        s2 += d * d;
        s1 += d < 0 ? -d : d;
    }
    double d2 = jaf_wave_sum((double)s2), d1 = jaf_wave_sum((double)s1);
    __shared__ double red[4][2];
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][0] = d2; red[threadIdx.x >> 6][1] = d1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&sums[n * 2], (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]));
        atomicAdd(&sums[n * 2 + 1], (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]));
    }
}

extern "C" int jaf_frame_error_sums_u8(jaf_stream_t s, const uint8_t* a, const uint8_t* b, double* sums, int32_t N, int64_t P) {
    JAF_REQUIRE(a && b && sums && N >= 1 && N <= 65535 && P >= 1);
    int blocks = jaf_cdiv(P, 1024);
    if (blocks > 64) blocks = 64;
    hipLaunchKernelGGL(frame_error_sums_kernel, dim3(blocks, N), dim3(256), 0, (hipStream_t)s, a, b, sums, (long)P);
    return jaf_launch_status();
}
