// Split-bf16 ("bf16x3") form of the packed-input convolution of conv_dma.hip: fp32-grade products on the bf16 matrix
// cores, staged by the same LDS-DMA path.
//
// Every fp32 operand is carried as two bf16 images, hi = bf16(v) and lo = bf16(v - hi) (jaf_conv2d_pack_input with
// d.precision == JAF_PREC_BF16X3 writes the two planes of a channel group next to each other, jaf_conv2d_pack the hi and the
// residual image of every weight chunk), and a product is taken as  ah*bh + al*bh + ah*bl  (the al*bl term is below 2^-16 of
// the product): three v_mfma_f32_16x16x32_bf16 per fragment pair, fp32 accumulation, relative error ~2^-17 per product --
// the arithmetic of conv_bf16_kernel<.., 3, ..> (conv_bf16.hip), whose fp32-input staging (scalar loads, conversion, ds_write:
// ~1100 instructions per 144 matrix-core instructions) this kernel replaces by DMA.
//
// Per 16 x 16 x 32 step a wave reads 2 (MT + NT) fragments for 3 MT NT matrix-core instructions: a better read-to-math ratio
// than the bf16 kernel (which reads MT + NT per MT NT).  LDS: [hi planes][lo planes][hi weights][lo weights][slot table].
// Tiling, slot table, pixel interleave, block -> tile mapping and the epilogue are those of conv_dma_kernel (conv_dma_kernel.h).
#include "conv_dma_kernel.h"

template <int MT, int NT, bool LSTM, bool DZ>
__global__ __launch_bounds__(256) void conv_dma_split_kernel(const ConvDArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const jaf_conv_desc& d = a.d;
    const jaf_conv_plan& P = a.p;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15;
    const int q = lane >> 4;
    constexpr int MR = 16 * MT;
    const int NG = P.NG;
    const int npos = P.npos, plane = P.plane, PW = P.PW, PWp = P.PWp;
    const int lg = a.ilv ? (NT == 4 ? 2 : (NT == 2 ? 1 : 0)) : 0;
    const int cmask = (1 << lg) - 1;
    const int PWq = PWp >> lg;

    const int lo_patch = NG * plane;                    // lo planes behind the hi planes
    const int lo_w = (P.pf > 0 ? P.pf : P.nsteps) * MT * 1024;              // lo weight image behind the hi one
    unsigned char* s_patch = smem;
    unsigned char* s_w = smem + a.off_w;
    int* s_tab = (int*)(smem + a.off_tab);

    // ---- block -> (row block, pixel tile, image, group), XCD-contiguous ----
    int L;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, j = bid >> 3, qn = nblk >> 3, rn = nblk & 7;
        L = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + j;
    }
    // (divisions by launch constants: multiply-high + shift, jaf_fdiv.h)
    const int Lm = (int)jaf_fdiv_q((unsigned)L, a.dv_mblocks);
    const int mb = L - Lm * P.mblocks;
    const int ngi = (int)jaf_fdiv_q((unsigned)Lm, a.dv_ntiles);
    const int tile = Lm - ngi * a.ntiles;
    const int n = (int)jaf_fdiv_q((unsigned)ngi, a.dv_G);
    const int g = ngi - n * d.G;
    const int tb = (int)jaf_fdiv_q((unsigned)tile, a.dv_tiles_x);
    const int tx = tile - tb * P.tiles_x;
    const int x0 = tx * P.TWIN;
    const int pbase = tb * (64 * NT);
    const int oy0 = (int)jaf_fdiv_q((unsigned)pbase, a.dv_twin);
    const int iy0 = oy0 * d.stride - d.pad_t;
    const int ix0 = x0 * d.stride - d.pad_l;
    const int OHW = d.OH * d.OW;
    const int HW = d.H * d.W;

    // ---- one-time table: (slot, tile nt) -> patch byte offset; [0]: full chunk, [1]: last chunk ----
    {
        const int taps = d.KH * d.KW;
        const float inv_kw = a.inv_kw;
        for (int e = tid; e < 2 * 16 * P.nsteps; e += 256) {
            const int nt = e & 3;
            int s = e >> 2;
            const int which = s >= 4 * P.nsteps;
            s -= which * 4 * P.nsteps;
            const int ngc = which ? P.ng_last : NG;
            int v = 0;
            if (s < taps * ngc) {
                const int tap = (int)(((float)s + 0.5f) * (which ? a.inv_ng_last : a.inv_ng)), grp = s - __mul24(tap, ngc);
                const int ky = (int)(((float)tap + 0.5f) * inv_kw), kx = tap - __mul24(ky, d.KW);
                const int xk = (a.ilv ? d.stride * nt : 0) + kx;
                v = __mul24(grp, plane) + (__mul24(ky, PWp) + __mul24(xk & cmask, PWq) + (xk >> lg)) * 16;
            }
            s_tab[e] = v;
        }
    }

    // ---- per-lane output pixels ----
    int boff[NT];
    int opix[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int p = a.ilv ? (pbase + wave * 16 * NT + li * NT + nt) : (pbase + (wave * NT + nt) * 16 + li);
        const int oy = (int)(((float)p + 0.5f) * a.inv_twin);
        const int oxr = p - __mul24(oy, P.TWIN);
        const int ox = x0 + oxr;
        const bool valid = (oy < d.OH) && (ox < d.OW);
        boff[nt] = valid ? ((__mul24(oy - oy0, d.stride * PWp) + __mul24(oxr >> lg, d.stride)) * 16) : 0;
        opix[nt] = valid ? (__mul24(oy, d.OW) + ox) : -1;
    }

    // ---- DMA source offsets: wave w fills rounds w and w+4 (64 slots each) of every group plane ----
    const int nrounds = (npos + 63) >> 6;
    int dvoff[CD_RPW];
    {
        const int dil = d.dil_in;
        const int Hd = (d.H - 1) * dil + 1;
        const int Wd = (d.W - 1) * dil + 1;
#pragma unroll
        for (int j = 0; j < CD_RPW; ++j) {
            const int slot = lane + 64 * (wave + 4 * j);
            const int r = (int)(((float)slot + 0.5f) * a.inv_pwp);
            const int rem = slot - __mul24(r, PWp);
            const int cls = (int)(((float)rem + 0.5f) * a.inv_pwq);
            const int x = ((rem - __mul24(cls, PWq)) << lg) + cls;
            const int iyd = iy0 + r, ixd = ix0 + x;
            bool ok = (slot < npos) && (x < PW) && (iyd >= 0) && (ixd >= 0) && (iyd < Hd) && (ixd < Wd);
            int iy = iyd, ix = ixd;
            if (dil == 2) {
                ok = ok && !((iyd | ixd) & 1);
                iy = iyd >> 1;
                ix = ixd >> 1;
            }
            dvoff[j] = ok ? ((__mul24(iy, d.W) + ix) * 16) : CD_OOB;
        }
    }
    // split image: channel group cg of this (image, group) has its hi plane at 2 cg and its lo plane at 2 cg + 1
    const int plane_bytes = HW * 16;
    const unsigned char* xbase = a.xp + (((long)n * d.G + g) * a.in_ng8) * 2L * plane_bytes;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 cpre[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) cpre[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (LSTM && NT == 4 && a.c_prev && a.vec && opix[0] >= 0) {
        const int C = d.Cout >> 2;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int ch = ((mb * MR + mt * 16) >> 2) + q;
            if (ch < C) cpre[mt] = *(const f32x4*)(a.c_prev + (((long)n * d.G + g) * C + ch) * OHW + opix[0]);
        }
    }

    // weight image: [g][row block][chunk][hi, residual][k-step][MT][1 KB]
    const long wimg_bytes = (long)P.nsteps * MT * 1024;
    const unsigned char* wbase = a.wpk + ((long)(g * P.mblocks + mb) * P.nchunks) * 2 * wimg_bytes;

    // plan.pf > 0: the chunk's weights pass through LDS `pf` k-steps at a time (conv_dma.hip)
    const int wsub = P.pf > 0 ? P.pf : P.nsteps;
    for (int chunk = 0; chunk < P.nchunks; ++chunk) {
        const bool last = (chunk == P.nchunks - 1);
        const int ngc = last ? P.ng_last : NG;
        const int nst = last ? P.nsteps_last : P.nsteps;
        const int* tab = s_tab + (last ? 16 * P.nsteps : 0) + q * 4;
        for (int s0 = 0; s0 < nst; s0 += wsub) {
            const int s1 = s0 + wsub < nst ? s0 + wsub : nst;
            __syncthreads();   // previous (sub-)chunk consumed (first pass: slot table visible)
            {
                const unsigned char* wsrc = wbase + (long)chunk * 2 * wimg_bytes + (long)s0 * MT * 1024;
                for (int e = wave; e < (s1 - s0) * MT; e += 4) {
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + e * 1024 + lane * 16),
                                                     (__attribute__((address_space(3))) void*)(s_w + e * 1024), 16, 0, 0);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wsrc + wimg_bytes + e * 1024 + lane * 16),
                                                     (__attribute__((address_space(3))) void*)(s_w + lo_w + e * 1024), 16, 0, 0);
                }
                if (s0 == 0) {
                    const unsigned char* cbase = xbase + (long)(chunk * NG) * 2 * plane_bytes;
                    for (int grp = 0; grp < ngc; ++grp) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const __amdgpu_buffer_rsrc_t rs =
                                __builtin_amdgcn_make_buffer_rsrc((void*)(cbase + (long)(2 * grp + h) * plane_bytes), 0, plane_bytes, 0x00020000);
#pragma unroll
                            for (int j = 0; j < CD_RPW; ++j) {
                                const int round = wave + 4 * j;
                                if (round < nrounds)
                                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                                        rs, (__attribute__((address_space(3))) void*)(s_patch + h * lo_patch + grp * plane + round * 1024), 16, dvoff[j], 0, 0, 0);
                            }
                        }
                    }
                }
            }
            __builtin_amdgcn_s_waitcnt(0);      // vmcnt(0): the DMAs of this wave have landed
            __syncthreads();

            u32x4 tnext = *(const u32x4*)(tab + 16 * s0);
            for (int st = s0; st < s1; ++st) {
                const u32x4 t4 = tnext;
                tnext = *(const u32x4*)(tab + 16 * (st + 1 < s1 ? st + 1 : st));
                const int off[4] = {(int)t4.x, (int)t4.y, (int)t4.z, (int)t4.w};
                bf16x8 bh[NT], bl[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    bh[nt] = *(const bf16x8*)(s_patch + off[nt] + boff[nt]);
                    bl[nt] = *(const bf16x8*)(s_patch + lo_patch + off[nt] + boff[nt]);
                }
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const bf16x8 ah = *(const bf16x8*)(s_w + ((st - s0) * MT + mt) * 1024 + lane * 16);
                    const bf16x8 al = *(const bf16x8*)(s_w + lo_w + ((st - s0) * MT + mt) * 1024 + lane * 16);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        // the two cross terms first: they are 2^-8 of the main one and the sum stays fp32 either way
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[nt], acc[mt][nt], 0, 0, 0);
                    }
                }
            }
        }
    }

    cd_epilogue<MT, NT, LSTM, DZ, false, true>(a, acc, opix, n, g, mb, q, OHW, smem, cpre);
}

template <int MT, int NT, bool LSTM>
static int cds_launch_one(const ConvDArgs& a, hipStream_t s) {
    const int lds = a.p.lds_bytes;
    const long nblk = (long)a.ntiles * a.p.mblocks * a.d.N * a.d.G;
    if (nblk < 1 || nblk > 0x7fffffffL) return JAF_EINVAL;
    if constexpr (!LSTM) {
        if (a.dz_mask) {          // the fused activation backward has its own instantiation (see cd_epilogue)
            auto kz = conv_dma_split_kernel<MT, NT, false, true>;
            static int optin_z[JAF_MAX_DEVICES];
            if (lds > 48 * 1024) {
                const int e = jaf_lds_optin((const void*)kz, optin_z);
                if (e) return e;
            }
            JAF_NOTE_KERNEL("conv_dma_split_kernel<%d, %d, false, true>", MT, NT);
            hipLaunchKernelGGL(kz, dim3((unsigned)nblk), dim3(256), (size_t)lds, s, a);
            return jaf_launch_status();
        }
    }
    auto k = conv_dma_split_kernel<MT, NT, LSTM, false>;
    static int optin[JAF_MAX_DEVICES];
    if (lds > 48 * 1024) {
        const int e = jaf_lds_optin((const void*)k, optin);
        if (e) return e;
    }
    JAF_NOTE_KERNEL("conv_dma_split_kernel<%d, %d, %s, false>", MT, NT, LSTM ? "true" : "false");
    hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(256), (size_t)lds, s, a);
    return jaf_launch_status();
}

template <int MT, bool LSTM>
static int cds_launch_nt(const ConvDArgs& a, hipStream_t s) {
    switch (a.p.NT) {
        case 1: return cds_launch_one<MT, 1, LSTM>(a, s);
        case 2: return cds_launch_one<MT, 2, LSTM>(a, s);
        case 4: return cds_launch_one<MT, 4, LSTM>(a, s);
    }
    return JAF_EINVAL;
}

template <bool LSTM>
static int cds_launch_mt(const ConvDArgs& a, hipStream_t s) {
    switch (a.p.MT) {
        case 1: return cds_launch_nt<1, LSTM>(a, s);
        case 2: return cds_launch_nt<2, LSTM>(a, s);
        case 3: return cds_launch_nt<3, LSTM>(a, s);
        case 4: return cds_launch_nt<4, LSTM>(a, s);
    }
    return JAF_EINVAL;
}

int cd_split_launch(const ConvDArgs& a, hipStream_t s, bool lstm) {
    return lstm ? cds_launch_mt<true>(a, s) : cds_launch_mt<false>(a, s);
}
