"""Timing-only flags for conv_wgrad_dma_kernel (JAF_WG_X): 1 DMA only for a workgroup's first tile, 2 no LDS reads / MFMA,
4 no epilogue, 8 no barriers inside the loop (with 1: pure compute issue rate)."""
import sys
s = sys.stdin.read()
def rep(a, b):
    global s
    assert s.count(a) == 1, (a, s.count(a))
    s = s.replace(a, b)
rep("    float inv_pwp;\n};", "    float inv_pwp;\n    int xf;\n};")
rep("    a.inv_pwp = 1.0f / (float)a.PWp;", "    a.inv_pwp = 1.0f / (float)a.PWp;\n    { static const int xf = getenv(\"JAF_WG_X\") ? atoi(getenv(\"JAF_WG_X\")) : 0; a.xf = xf; }")
rep("            __syncthreads();   // previous tile consumed\n            issue(item, 0);\n            __builtin_amdgcn_s_waitcnt(0);\n            __syncthreads();",
    "            if (!(a.xf & 8)) __syncthreads();\n            if (!(a.xf & 1) || item == split) issue(item, 0);\n            __builtin_amdgcn_s_waitcnt(0);\n            if (!(a.xf & 8)) __syncthreads();")
rep("        for (int ks = wk; ks < 4; ks += WK) {", "        for (int ks = wk; ks < ((a.xf & 2) ? 0 : 4); ks += WK) {")
rep("    const int cit = ci0 + wc * 16;\n", "    const int cit = ci0 + wc * 16;\n    if ((a.xf & 4) && acc[0][0][0] != 123456.789f) return;\n")
sys.stdout.write(s)
