// Two micro-benchmarks that measure this box's ceilings (include/jafpro_hip.h, "Measured ceilings").
#include "jaf_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void ubench_mfma_bf16_kernel(float* sink, int iters) {
    bf16x8 a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)(0.001f * (float)((threadIdx.x + i) & 7));
        b[i] = (__bf16)(0.002f * (float)((threadIdx.x + 3 * i) & 7));
    }
    f32x4 acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    if (s == -1.0f) sink[0] = s;          // never true: keeps the accumulators live
}

extern "C" int jaf_ubench_mfma_bf16(jaf_stream_t s, int32_t blocks, int32_t iters, float* sink) {
    JAF_REQUIRE(blocks >= 1 && iters >= 1 && sink);
    hipLaunchKernelGGL(ubench_mfma_bf16_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)s, sink, iters);
    return jaf_launch_status();
}

__global__ __launch_bounds__(256) void ubench_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long n16) {
    // four independent 16-byte loads in flight per lane before the first store
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const u32x4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

extern "C" int jaf_ubench_copy(jaf_stream_t s, const void* src, void* dst, int64_t n16) {
    JAF_REQUIRE(src && dst && n16 >= 1);
    JAF_REQUIRE(((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0);
    hipLaunchKernelGGL(ubench_copy_kernel, dim3(256 * 16), dim3(256), 0, (hipStream_t)s, (const u32x4*)src, (u32x4*)dst, (long)n16);
    return jaf_launch_status();
}

// Variants of the streaming copy (measurement only): nontemporal accesses and different amounts of work in flight.
template <int U, bool NT>
__global__ __launch_bounds__(256) void ubench_copy_var_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long n16) {
    const long stride = (long)gridDim.x * blockDim.x;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        u32x4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) v[k] = NT ? __builtin_nontemporal_load(src + i + k * stride) : src[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            if (NT) __builtin_nontemporal_store(v[k], dst + i + k * stride);
            else dst[i + k * stride] = v[k];
        }
    }
    for (; i < n16; i += stride) dst[i] = src[i];
}

extern "C" int jaf_ubench_copy_variant(jaf_stream_t s, const void* src, void* dst, int64_t n16, int32_t variant, int32_t blocks) {
    JAF_REQUIRE(src && dst && n16 >= 1 && blocks >= 1);
    JAF_REQUIRE(((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0);
    const u32x4* a = (const u32x4*)src;
    u32x4* b = (u32x4*)dst;
    switch (variant) {
        case 0: hipLaunchKernelGGL((ubench_copy_var_kernel<4, false>), dim3(blocks), dim3(256), 0, (hipStream_t)s, a, b, (long)n16); break;
        case 1: hipLaunchKernelGGL((ubench_copy_var_kernel<4, true>), dim3(blocks), dim3(256), 0, (hipStream_t)s, a, b, (long)n16); break;
        case 2: hipLaunchKernelGGL((ubench_copy_var_kernel<8, true>), dim3(blocks), dim3(256), 0, (hipStream_t)s, a, b, (long)n16); break;
        case 3: hipLaunchKernelGGL((ubench_copy_var_kernel<2, true>), dim3(blocks), dim3(256), 0, (hipStream_t)s, a, b, (long)n16); break;
        case 4: hipLaunchKernelGGL((ubench_copy_var_kernel<1, true>), dim3(blocks), dim3(256), 0, (hipStream_t)s, a, b, (long)n16); break;
        case 5: hipLaunchKernelGGL((ubench_copy_var_kernel<1, false>), dim3(blocks), dim3(256), 0, (hipStream_t)s, a, b, (long)n16); break;
        default: return JAF_EINVAL;
    }
    return jaf_launch_status();
}
