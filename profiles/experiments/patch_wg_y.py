"""Timing-only flags for conv_wgrad_dma_kernel (JAF_WG_X): 1 DMA only for a workgroup's first tile, 2 no LDS reads / MFMA,
4 no epilogue, 16 B operand read for ky == 0 only (3 of 9 transposed reads), 32 A operand read for mt == 0 only."""
import sys
s = sys.stdin.read()
def rep(a, b):
    global s
    assert s.count(a) == 1, (a, s.count(a))
    s = s.replace(a, b)
rep("    float inv_pwp;\n", "    float inv_pwp;\n    int xf;\n")
rep("    a.inv_pwp = 1.0f / (float)a.PWp;", "    a.inv_pwp = 1.0f / (float)a.PWp;\n    { static const int xf = getenv(\"JAF_WG_X\") ? atoi(getenv(\"JAF_WG_X\")) : 0; a.xf = xf; }")
rep("            __syncthreads();   // previous tile consumed\n            issue(item, 0);\n            __builtin_amdgcn_s_waitcnt(0);\n            __syncthreads();",
    "            __syncthreads();\n            if (!(a.xf & 1) || item == split) issue(item, 0);\n            __builtin_amdgcn_s_waitcnt(0);\n            __syncthreads();")
rep("        for (int ks = wk; ks < 4; ks += WK) {", "        for (int ks = wk; ks < ((a.xf & 2) ? 0 : 4); ks += WK) {")
rep("    const int cit = ci0 + wc * 16;\n", "    const int cit = ci0 + wc * 16;\n    if ((a.xf & 4) && acc[0][0][0] != 123456.789f) return;\n")
rep("""            for (int mt = 0; mt < MTW; ++mt) {
                const unsigned char* ap = c_dz + mt * 4096 + ks * 1024;
                const s16x4 a0""", """            for (int mt = 0; mt < MTW; ++mt) {
                if ((a.xf & 32) && mt > 0) { af[mt] = af[0]; continue; }
                const unsigned char* ap = c_dz + mt * 4096 + ks * 1024;
                const s16x4 a0""")
rep("""            for (int ky = 0; ky < KS; ++ky) {
                const int rowoff = ((2 * ks * s + ky) * PWp) * 32;
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const int a0 = (bbase[0][kx] + rowoff) ^ bswz[0][kx];
                    const int a1 = (bbase[1][kx] + rowoff) ^ bswz[1][kx];
                    const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(c_x + a0));
                    const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(c_x + a1));
                    const bf16x8 bf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));""",
"""            for (int ky = 0; ky < KS; ++ky) {
                const int rowoff = ((2 * ks * s + ky) * PWp) * 32;
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const int a0 = (bbase[0][kx] + rowoff) ^ bswz[0][kx];
                    const int a1 = (bbase[1][kx] + rowoff) ^ bswz[1][kx];
                    bf16x8 bf;
                    if ((a.xf & 16) && ky > 0) bf = bsv[kx];
                    else {
                    const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(c_x + a0));
                    const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(c_x + a1));
                    bf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
                    bsv[kx] = bf; }""")
rep("            bf16x8 af[MTW];\n", "            bf16x8 af[MTW];\n            bf16x8 bsv[KS];\n")
sys.stdout.write(s)
