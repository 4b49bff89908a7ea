// Division of a wave-uniform index by a launch constant without the ~30-instruction integer-division sequence (v_rcp_iflag,
// readfirstlane, two correction steps: profiles/experiments/round5_conv_stamps.txt shows a convolution workgroup spending 8-28 %
// of its life before its first barrier, most of it in five such divisions of the block index).
// Granlund-Montgomery: for 0 <= n < 2^31, d >= 2, l = ceil(log2 d), m = ceil(2^(31 + l) / d) (fits 32 bits):
//   floor(n / d) = (m * n) >> (31 + l) = mulhi32(n, m) >> (l - 1).        d = 1: identity (flag in the shift word).
// Plain C on the host side so that tests/test_host_logic.py can compile it with gcc and check it exhaustively.
#pragma once
#include <stdint.h>

typedef struct jaf_fdiv { uint32_t m, sh; } jaf_fdiv;

static inline jaf_fdiv jaf_fdiv_make(uint32_t d) {
    jaf_fdiv f;
    if (d <= 1) { f.m = 0; f.sh = 0x80000000u; return f; }
    uint32_t l = 0;
    while (((uint64_t)1 << l) < d) ++l;                  /* l = ceil(log2 d), 1 <= l <= 32 */
    const uint64_t p = (uint64_t)1 << (31 + l);          /* <= 2^63 */
    f.m = (uint32_t)((p + d - 1) / d);
    f.sh = l - 1;
    return f;
}

static inline uint32_t jaf_fdiv_host(uint32_t n, jaf_fdiv f) {
    return (f.sh & 0x80000000u) ? n : (uint32_t)(((uint64_t)n * f.m) >> 32) >> f.sh;
}

#ifdef __HIPCC__
__device__ __forceinline__ unsigned jaf_fdiv_q(unsigned n, jaf_fdiv f) {
    return (f.sh & 0x80000000u) ? n : (__umulhi(n, f.m) >> f.sh);
}
#endif
