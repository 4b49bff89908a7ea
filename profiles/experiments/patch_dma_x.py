"""Timing-only experiment flags for conv_dma_kernel (JAF_CD_X bitmask): 1 skip weight DMA, 2 skip patch DMA, 4 skip epilogue,
8 skip MFMA loop.  Only applied to launches with G >= 8 (the part networks)."""
import sys
s = sys.stdin.read()
def rep(a, b, cnt=1):
    global s
    assert s.count(a) >= 1, a
    s = s.replace(a, b, cnt)
rep("    float* dz_dbias;\n};", "    float* dz_dbias;\n    int xf;\n};")
rep("    a.dz_dbias = nullptr;\n}", "    a.dz_dbias = nullptr;\n    static const int xf = getenv(\"JAF_CD_X\") ? atoi(getenv(\"JAF_CD_X\")) : 0;\n    static const int xg = getenv(\"JAF_CD_XG\") ? atoi(getenv(\"JAF_CD_XG\")) : 8; a.xf = d->G >= xg ? xf : 0;\n}")
rep("            for (int e = wave; e < nst * MT; e += 4)\n", "            if (!(a.xf & 1)) for (int e = wave; e < nst * MT; e += 4)\n")
rep("            for (int grp = 0; grp < ngc; ++grp) {\n                const __amdgpu_buffer_rsrc_t rs", "            if (!(a.xf & 2)) for (int grp = 0; grp < ngc; ++grp) {\n                const __amdgpu_buffer_rsrc_t rs")
rep("        for (int st = 0; st < nst; ++st) {\n            const u32x4 t4 = tnext;", "        for (int st = 0; st < ((a.xf & 8) ? 1 : nst); ++st) {\n            const u32x4 t4 = tnext;")
rep("    cd_epilogue<MT, NT, LSTM, DZ, PLAIN>(a, acc, opix, n, g, mb, q, OHW, smem, cpre);", "    if (!(a.xf & 4) || acc[0][0][0] == 123456.789f) cd_epilogue<MT, NT, LSTM, DZ, PLAIN>(a, acc, opix, n, g, mb, q, OHW, smem, cpre);")

# epilogue decomposition: 16 skip LSTM packed-h stores, 32 skip gate stores, 64 skip c/h fp32 stores,
# 128 DZ: no partial-sum loads, 256 DZ: no mask loads, 512 DZ: no packed dst stores, 1024: no fp32 stores after the dz part
rep("                if (a.dst) {     // h_t straight into the consumer's packed image", "                if (a.dst && !(a.xf & 16)) {     // h_t straight into the consumer's packed image")
rep("                if (a.gates_out) {\n                    if (a.gates_bf16) {\n                        // bf16 gates are kept gate-innermost", "                if (a.gates_out && !(a.xf & 32)) {\n                    if (a.gates_bf16) {\n                        // bf16 gates are kept gate-innermost")
rep("                *(fvec*)(a.c_out + hc + opix[0]) = vc;\n                if (!a.skip_f32) *(fvec*)(a.h_out + hc + opix[0]) = vh;", "                if (!(a.xf & 64)) *(fvec*)(a.c_out + hc + opix[0]) = vc;\n                if (!a.skip_f32 && !(a.xf & 64)) *(fvec*)(a.h_out + hc + opix[0]) = vh;")
rep("                if (DZ && a.acc_out) {\n", "                if (DZ && a.acc_out && !(a.xf & 128)) {\n")
rep("                                if (a.acc_out) t += part[j][nt];", "                                if (a.acc_out && !(a.xf & 128)) t += part[j][nt];")
rep("                        const unsigned int x01 = xp2[0], x23 = xp2[1];", "                        const unsigned int x01 = (a.xf & 256) ? 0x3f803f80u : xp2[0], x23 = (a.xf & 256) ? 0x3f803f80u : xp2[1];")
rep("                    if (co0 + 4 <= cpad || a.dst_pad_tail) {\n                        *(u32x2*)cd_dst_ptr", "                    if (a.xf & 512) { if (w[0] == 0x12345678u) *(u32x2*)cd_dst_ptr(a.dst, ngd, a.dst_ng8, a.dst_coff + co0, OHW, opix[nt]) = w; } else if (co0 + 4 <= cpad || a.dst_pad_tail) {\n                        *(u32x2*)cd_dst_ptr")
rep("        if (a.skip_f32 || (DZ && !a.out2)) return;", "        if (a.skip_f32 || (DZ && !a.out2) || (a.xf & 1024)) return;")

# 2048: stagger the co-resident workgroups: those in an odd wave slot of their SIMD sleep JAF_CD_SLEEP x 64 cycles before their first DMA
rep("    const int mb = L % P.mblocks;\n", "    if ((a.xf & 2048) && blockIdx.x < (unsigned)a.xsl_blocks) { const unsigned hw = __builtin_amdgcn_s_getreg(6148); if (hw & 1) { for (int zz = 0; zz < a.xsl; ++zz) __builtin_amdgcn_s_sleep(16); } }\n    const int mb = L % P.mblocks;\n")
rep("    int xf;\n};", "    int xf, xsl, xsl_blocks;\n};")
rep("a.xf = d->G >= xg ? xf : 0;", "a.xf = d->G >= xg ? xf : 0; { static const int sl = getenv(\"JAF_CD_SLEEP\") ? atoi(getenv(\"JAF_CD_SLEEP\")) : 4; static const int sb = getenv(\"JAF_CD_SLEEP_BLOCKS\") ? atoi(getenv(\"JAF_CD_SLEEP_BLOCKS\")) : 1024; a.xsl = sl; a.xsl_blocks = sb; }")

# 4096: DZ: no bias-gradient reduction
rep("                if (DZ && a.dz_dbias) {       // the 16 lanes of a q-group hold the same 4 channels: fold them", "                if (DZ && a.dz_dbias && !(a.xf & 4096)) {       // the 16 lanes of a q-group hold the same 4 channels: fold them")
rep("            if (DZ && a.dz_dbias) {\n                // workgroup sum of the four waves in LDS", "            if (DZ && a.dz_dbias && !(a.xf & 4096)) {\n                // workgroup sum of the four waves in LDS")
sys.stdout.write(s)
