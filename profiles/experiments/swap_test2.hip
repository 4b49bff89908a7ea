#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* out) {
    const unsigned lane = threadIdx.x;
    unsigned t0 = 0 * 100 + (lane >> 4), t1 = 100 + (lane >> 4), t2 = 200 + (lane >> 4), t3 = 300 + (lane >> 4);
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\ts_nop 1\n\t"
                 "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1"
                 : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
    out[lane * 4 + 0] = t0; out[lane * 4 + 1] = t1; out[lane * 4 + 2] = t2; out[lane * 4 + 3] = t3;
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 4096); unsigned h[256];
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); (void)hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    for (int row = 0; row < 4; ++row) printf("row %d: %u %u %u %u\n", row, h[row * 64], h[row * 64 + 1], h[row * 64 + 2], h[row * 64 + 3]);
    return 0;
}
