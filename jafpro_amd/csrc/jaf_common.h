// Shared device/host helpers for the jafpro_amd HIP library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/jafpro_hip.h"

#define JAF_WAVE 64

// Every entry point returns 0 on success, a negative JAF_E* on bad arguments, or the
// (positive) hipError_t of a failed launch.  Nothing throws across the ABI
// (reference: rasterize_cuda.cpp:66-68 raises from AT_CHECK, launch errors are only
// printf'd at rasterize_cuda_kernel.cu:624-626 -- here they are returned).
static inline int jaf_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? JAF_OK : (int)e;
}

#define JAF_REQUIRE(cond) do { if (!(cond)) return JAF_EINVAL; } while (0)

// Name of the kernel instantiation a launch path picked, as rocprofv3 prints it (jaf_last_kernel_name): written by the launch
// code itself -- the one place that knows -- while jaf_kernel_names(1) is on (bench.py's roofline step, profiling scripts).
extern thread_local char jaf_kname_buf[160];
extern int jaf_kname_on;
#define JAF_NOTE_KERNEL(...) do { if (jaf_kname_on) snprintf(jaf_kname_buf, sizeof(jaf_kname_buf), __VA_ARGS__); } while (0)

// Opt a kernel into > 48 KB of dynamic LDS.  The attribute belongs to the (kernel, DEVICE) pair, so the
// "already done" flag is kept per device of the calling thread (`cache`: one zero-initialised int[JAF_MAX_DEVICES]
// per kernel instantiation); racing callers store the same value.  Returns 0 or the hipError_t.
#define JAF_MAX_DEVICES 64
static inline int jaf_lds_optin(const void* kernel, int* cache) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= JAF_MAX_DEVICES) return JAF_EINVAL;
    if (!cache[dev]) {
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        cache[dev] = 1;
    }
    return 0;
}

static inline int jaf_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// grid for HBM-bound elementwise kernels: cap at 2048 blocks of 256 and grid-stride
// the rest (cdna_hip_programming.md Guideline 11).
static inline int jaf_ew_grid(long n, int per_thread = 1) {
    long blocks = (n + 256L * per_thread - 1) / (256L * per_thread);
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    return (int)blocks;
}

// v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 instructions): the ConvLSTM epilogue
// evaluates 5 of these per hidden pixel and was VALU-issue bound on the divisions (PMC: 1455 VALU
// instructions per wave).  |error| of sigmoid / tanh stays ~2e-7.
__device__ __forceinline__ float jaf_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float jaf_sigmoid(float x) { return jaf_rcp(1.0f + __expf(-x)); }
__device__ __forceinline__ float jaf_tanh(float x) {
    // tanh via exp; exact enough for fp32 parity (|err| ~2e-7), saturates cleanly.
    float e = __expf(-2.0f * fabsf(x));
    float t = (1.0f - e) * jaf_rcp(1.0f + e);
    return copysignf(t, x);
}

__device__ __forceinline__ float jaf_act(float v, int act, float slope) {
    switch (act) {
        case JAF_ACT_LRELU: return v > 0.f ? v : v * slope;
        case JAF_ACT_RELU: return v > 0.f ? v : 0.f;
        case JAF_ACT_SIGMOID: return jaf_sigmoid(v);
        case JAF_ACT_TANH: return jaf_tanh(v);
        default: return v;
    }
}

// 16-byte-per-lane buffer load straight into LDS (lane l lands at lds_addr + 16 l), issued from inline assembly.
// The builtin (__builtin_amdgcn_raw_ptr_buffer_load_lds) is tracked by the compiler as an LDS WRITE: with one dynamic LDS
// array it cannot prove that a later ds_read touches another buffer, and it puts `s_waitcnt vmcnt(0)` in front of the first
// LDS read that follows the DMA -- which serialises exactly the overlap a second tile buffer is for (seen in the ISA of the
// double-buffered weight-gradient kernel: the wait sat between the DMA of tile i+1 and the first ds_read of tile i, and the
// kernel ran no faster than the single-buffered one).  From assembly the compiler inserts nothing; the caller owns the
// ordering: `s_waitcnt vmcnt(0)` + a workgroup barrier before anybody reads the buffer, a barrier after the last read
// before it is filled again.  `rsrc`: buffer resource words (base, base_hi | stride, bytes, 0x00020000), wave-uniform.
typedef unsigned int jaf_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ jaf_u32x4 jaf_make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    jaf_u32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}
__device__ __forceinline__ void jaf_dma16_async(jaf_u32x4 rsrc, unsigned lds_addr, int voffset) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                 :: "s"(lds_addr), "v"(voffset), "s"(rsrc) : "memory");
}

__device__ __forceinline__ double jaf_wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float jaf_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// V consecutive elements of an fp32 or bf16 tensor as floats (one 4V- or 2V-byte access) and back (RNE): the I/O of the HBM-bound
// kernels that serve both storage formats of the bf16 mode (fp32 tensors, and the bf16 ones of BASELINE configs[2]'s storage).
template <int V, typename T>
__device__ __forceinline__ void jaf_ldv(const T* __restrict__ p, float (&o)[V]) {
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
__device__ __forceinline__ void jaf_stv(T* __restrict__ p, const float (&o)[V]) {
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
