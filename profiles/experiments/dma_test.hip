#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const unsigned* src, unsigned* out, int nrec) {
    __shared__ __attribute__((aligned(16))) unsigned sm[64 * 4];
    for (int i = threadIdx.x; i < 256; i += 64) sm[i] = 0xdeadbeefu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nrec, 0x00020000);
    int voff = threadIdx.x * 16;
    if (threadIdx.x == 7) voff = 0x40000000;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)sm, 16, voff, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = sm[i];
}
int main() {
    unsigned *src, *out; unsigned h[256], ho[256];
    for (int i = 0; i < 256; ++i) h[i] = 1000 + i;
    hipMalloc(&src, 1024); hipMalloc(&out, 1024);
    hipMemcpy(src, h, 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, out, 160);   // 160 bytes valid: lanes 0..9
    hipMemcpy(ho, out, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 16; ++l) printf("lane %2d: %08x %08x %08x %08x\n", l, ho[4*l], ho[4*l+1], ho[4*l+2], ho[4*l+3]);
    return 0;
}
