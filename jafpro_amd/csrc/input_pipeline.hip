// Device-side input pipeline: what Fusion_dataset_smpl_interval.__getitem__ (src/data.py:640-773) and
// the permutes of train/4.convLSTM_flowpro_interval.py:216-237 do per sample on the host in float64 NumPy -- uint8 HWC
// -> normalised fp32 CHW, mask /255, the TransferTexture-style silhouette of an IUV map (src/utils.py:369-394) -- as
// HBM-bound kernels over the raw uint8 frames.  The clip is uploaded once as uint8 (a quarter of the bytes of the
// reference's float tensors, which it also round-trips through float64) and never touched by the host again.
//
// Arithmetic: the reference normalises in float64 and casts to float32 (`(x / 255.0 - 0.5) * 2`, then `.float()`); a
// uint8 has 256 values, so the kernels evaluate the same float64 expression (bit-identical results, tests/test_gpu_data.py).
#include "jaf_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float ip_norm(unsigned int v, int mode) {
    const double x = (double)v;
    return mode == 0 ? (float)((x / 255.0 - 0.5) * 2.0) : (float)(x / 255.0);
}

// in u8 [N][HW][C] -> out f32 [N][C][HW]; a lane converts 4 consecutive pixels of all C (<= 4) channels:
// 4*C contiguous bytes in, C float4 out.  grid (pixel blocks, N).
template <int C>
__global__ __launch_bounds__(256) void u8_hwc_to_f32_chw_kernel(const uint8_t* __restrict__ in, float* __restrict__ out, int HW,
                                                                 int mode) {
    const long n = blockIdx.y;
    const int p4 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (p4 >= HW) return;
    const uint8_t* ip = in + (n * HW + p4) * C;
    float* op = out + n * C * (long)HW + p4;
    if (p4 + 4 <= HW && (((uintptr_t)ip) & 3) == 0 && (HW & 3) == 0) {
        unsigned int w[C];
#pragma unroll
        for (int i = 0; i < C; ++i) w[i] = ((const unsigned int*)ip)[i];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            f32x4 o;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int byte = k * C + c;
                o[k] = ip_norm((w[byte >> 2] >> ((byte & 3) * 8)) & 0xffu, mode);
            }
            *(f32x4*)(op + (long)c * HW) = o;
        }
    } else {
        for (int k = 0; k < 4 && p4 + k < HW; ++k)
            for (int c = 0; c < C; ++c) op[(long)c * HW + k] = ip_norm(ip[k * C + c], mode);
    }
}

extern "C" int jaf_u8_hwc_to_f32_chw(jaf_stream_t s, const uint8_t* in, float* out, int32_t N, int32_t HW, int32_t C, int mode) {
    JAF_REQUIRE(in && out && N >= 1 && N <= 65535 && HW >= 1 && (C == 1 || C == 3) && (mode == 0 || mode == 1));
    const dim3 grid(jaf_cdiv(jaf_cdiv(HW, 4), 256), N);
    if (C == 3) hipLaunchKernelGGL(u8_hwc_to_f32_chw_kernel<3>, grid, dim3(256), 0, (hipStream_t)s, in, out, HW, mode);
    else hipLaunchKernelGGL(u8_hwc_to_f32_chw_kernel<1>, grid, dim3(256), 0, (hipStream_t)s, in, out, HW, mode);
    return jaf_launch_status();
}

// TransferTexture(np.ones((800,1200,3)), IUV) (src/data.py:690-695): 1 where the part index is 1..24, on 3 equal channels.
__global__ void iuv_part_mask_kernel(const uint8_t* __restrict__ iuv, float* __restrict__ out, int N, int SS) {
    const long total = (long)N * SS;
    const long gs = (long)gridDim.x * blockDim.x;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gs) {
        const long n = e / SS;
        const long p = e - n * SS;
        const int I = iuv[e * 3];
        const float v = (I >= 1 && I <= 24) ? 1.f : 0.f;
        float* o = out + n * 3 * (long)SS + p;
        o[0] = v; o[SS] = v; o[2 * (long)SS] = v;
    }
}

extern "C" int jaf_iuv_part_mask(jaf_stream_t s, const uint8_t* iuv, float* out, int32_t N, int32_t S) {
    JAF_REQUIRE(iuv && out && N >= 1 && S >= 1);
    hipLaunchKernelGGL(iuv_part_mask_kernel, dim3(jaf_ew_grid((long)N * S * S)), dim3(256), 0, (hipStream_t)s, iuv, out, N, S * S);
    return jaf_launch_status();
}

// TransferTexture (src/utils.py:369-394): out[y][x] = tex[part cell][rint(U/255*199)][199 - rint(V/255*199)] for part
// 1..24, else 0; with `im`, pixels whose transferred value is 0 take im's (per channel, :390-392).  uint8 in and out;
// tex [NT][AH][AW][3] with NT == N or 1 (one atlas for the whole batch), cell edge psz = AH / 4.
__global__ void transfer_texture_u8_kernel(const uint8_t* __restrict__ tex, const uint8_t* __restrict__ iuv,
                                           const uint8_t* __restrict__ im, uint8_t* __restrict__ out, int N, int SS, int AH,
                                           int AW, int psz, int tex_batched) {
    const long total = (long)N * SS;
    const long gs = (long)gridDim.x * blockDim.x;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gs) {
        const long n = e / SS;
        const uint8_t* q = iuv + e * 3;
        const int I = q[0];
        unsigned char r[3] = {0, 0, 0};
        if (I >= 1 && I <= 24) {
            const int U = (int)(unsigned char)rint((double)q[1] / 255. * (double)(psz - 1));
            const int V = (int)(unsigned char)rint((double)q[2] / 255. * (double)(psz - 1));
            const int i_cor = (I - 1) / 6, j_cor = I - i_cor * 6 - 1;
            const uint8_t* t = tex + (((tex_batched ? n : 0) * AH + i_cor * psz + U) * (long)AW + j_cor * psz + (psz - 1 - V)) * 3;
            r[0] = t[0]; r[1] = t[1]; r[2] = t[2];
        }
        if (im) {
            const uint8_t* b = im + e * 3;
            for (int c = 0; c < 3; ++c)
                if (r[c] == 0) r[c] = b[c];
        }
        uint8_t* o = out + e * 3;
        o[0] = r[0]; o[1] = r[1]; o[2] = r[2];
    }
}

extern "C" int jaf_transfer_texture_u8(jaf_stream_t s, const uint8_t* tex, const uint8_t* iuv, const uint8_t* im, uint8_t* out,
                                       int32_t N, int32_t S, int32_t AH, int32_t AW, int tex_batched) {
    JAF_REQUIRE(tex && iuv && out && N >= 1 && S >= 1 && AH >= 4 && AW >= 6 && AH % 4 == 0 && AW / 6 == AH / 4);
    hipLaunchKernelGGL(transfer_texture_u8_kernel, dim3(jaf_ew_grid((long)N * S * S)), dim3(256), 0, (hipStream_t)s, tex, iuv, im,
                       out, N, S * S, AH, AW, AH / 4, tex_batched ? 1 : 0);
    return jaf_launch_status();
}
