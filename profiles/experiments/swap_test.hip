#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    const unsigned lane = threadIdx.x;
    // register r holds value 100*r + row (row = lane >> 4)
    unsigned t0 = 0 * 100 + (lane >> 4), t1 = 100 + (lane >> 4), t2 = 200 + (lane >> 4), t3 = 300 + (lane >> 4);
    auto s02 = __builtin_amdgcn_permlane32_swap(t0, t2, false, false);
    auto s13 = __builtin_amdgcn_permlane32_swap(t1, t3, false, false);
    auto s01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
    auto s23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
    out[lane * 4 + 0] = s01[0]; out[lane * 4 + 1] = s01[1]; out[lane * 4 + 2] = s23[0]; out[lane * 4 + 3] = s23[1];
    // raw semantics
    auto a = __builtin_amdgcn_permlane32_swap(t0, t1, false, false);
    out[256 + lane * 2] = a[0]; out[256 + lane * 2 + 1] = a[1];
    auto b = __builtin_amdgcn_permlane16_swap(t0, t1, false, false);
    out[384 + lane * 2] = b[0]; out[384 + lane * 2 + 1] = b[1];
}
int main() {
    unsigned* d; hipMalloc(&d, 4096); unsigned h[512];
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); hipMemcpy(h, d, 2048, hipMemcpyDeviceToHost);
    for (int row = 0; row < 4; ++row) printf("row %d (lane %d): regs after transpose = %u %u %u %u\n", row, row * 16, h[row * 64], h[row * 64 + 1], h[row * 64 + 2], h[row * 64 + 3]);
    for (int row = 0; row < 4; ++row) printf("permlane32_swap(t0,t1) row %d: a0=%u a1=%u | permlane16_swap: b0=%u b1=%u\n", row, h[256 + row * 32], h[256 + row * 32 + 1], h[384 + row * 32], h[384 + row * 32 + 1]);
    return 0;
}
