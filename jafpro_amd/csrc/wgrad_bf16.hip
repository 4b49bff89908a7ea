// Weight gradient of 3x3 convolutions on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16).
//
//   dW[co][ci][tap] = sum over images n and output pixels p of  dz[n][co][p] * X[n][ci][p*s + tap]
//
// GEMM view: rows = output channels (A operand = dz), columns = (tap, input channel) with the 16
// input channels of a column tile on the MFMA lane (B operand = the forward input patch),
// reduction K = output pixels, 32 per MFMA.
//
// * a workgroup owns [16*MTW output channels] x [16*WC input channels] x 9 taps of one group's dW
//   and walks a strided share of the 16x8-pixel tiles of all images; its 4 waves are WC waves
//   along the input-channel tiles times WK = 4/WC waves along the k-steps of a tile;
// * dz tile in LDS: [co][128 pixels] bf16, row pitch 288 B: an A fragment (8 consecutive pixels of
//   one output channel) is one conflict-free ds_read_b128;
// * input patch in LDS: [ci tile][position][16 channels] bf16 (32 B per position, bit 7 of the byte
//   address XORed with bit 3 of the patch column; rows padded to whole 256-byte lines).  The B fragment needs 8 consecutive PIXELS of
//   one channel, i.e. a transposed read of this channel-innermost image: two
//   ds_read_b64_tr_b16 (cdna_hip_programming.md T10), whose four row addresses per 16-lane group
//   are free -- so every tap is just a shifted row address and no im2col copy exists anywhere;
// * accumulators (MTW x 9 tiles) leave through an LDS transpose so that the fp32 atomics that
//   combine the pixel splits run along the memory order of dW (co rows of [ci][tap]);
// * SPLIT (JAF_PREC_BF16X3, the parity-grade mode): both operands are staged as a bf16 head and a bf16 residual
//   (v = hi + lo up to 2^-17 relative) in two LDS images each, and every product is three MFMAs
//   dz_hi*x_hi + dz_lo*x_hi + dz_hi*x_lo -- fp32-grade weight gradients at a third of the bf16 matrix-core rate instead
//   of the 33 TFLOP/s of the fp32 MFMA kernel (wgrad.hip) that mode used before.
#include "conv_internal.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) float* gfptr;
typedef const __attribute__((address_space(1))) f32x4* gf4ptr;
typedef __attribute__((address_space(3))) s16x4* lds_s16x4;

#define WB_TW 16        // pixel tile: 16 wide x 8 high = 128 pixels = 4 MFMA k-steps
#define WB_TH 8
#define WB_DZP 288      // dz row pitch in bytes (2*128 + 32: conflict-free b128 fragment reads)
#define WB_ITX 3        // input-patch items (position, 8 channels) per thread per staging round
#define WB_EP 145       // floats per output-channel row of the epilogue transpose (16 ci x 9 taps + 1)

struct WgBArgs {
    const float* src[3];
    const float* dz;
    float* dw;
    jaf_conv_desc d;
    int WC, WK;
    int coblocks, ciblocks, nsplit;
    int tiles_x, tiles_y;
    int PH, PW, PWp, npos;   // PWp: row pitch in positions, multiple of 8 (row = whole 256-byte lines)
    int xplane;
    int off_dz, off_cptr;
    int lo_x, lo_dz;         // SPLIT: byte distance from the head image to the residual image (patch / dz)
    int nitems_x;
    int vec4;
    float inv_pw, inv_npos;
};

__device__ __forceinline__ unsigned int wb_pack2(float a, float b) {
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(unsigned int, r);
}

// residual of the bf16 head: v - float(bf16(v)), itself rounded to bf16 by wb_pack2
__device__ __forceinline__ unsigned int wb_pack2_lo(float a, float b) {
    f32x2 v = {a, b};
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    const f32x2 hf = __builtin_convertvector(h, f32x2);
    return wb_pack2(a - hf[0], b - hf[1]);
}

template <int MTW, bool SPLIT>
__global__ __launch_bounds__(256, SPLIT ? 1 : 2) void conv_wgrad_bf16_kernel(const WgBArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const jaf_conv_desc& d = a.d;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 15;
    const int q = lane >> 4;
    const int WC = a.WC, WK = a.WK;
    const int wc = wave % WC;
    const int wk = wave / WC;
    const int s = d.stride;
    const int PW = a.PW, PWp = a.PWp;

    unsigned char* s_x = smem;
    unsigned char* s_dz = smem + a.off_dz;
    unsigned long long* s_cptr = (unsigned long long*)(smem + a.off_cptr);     // [16*WC] plane pointers (n = 0)
    int* s_cstr = (int*)(smem + a.off_cptr + 16 * WC * 8);                     // [16*WC] image stride (floats)

    // ---- block -> (ci block fastest, co block, pixel split, group), XCD-contiguous ----
    int L;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, j = bid >> 3, qn = nblk >> 3, rn = nblk & 7;
        L = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + j;
    }
    const int cib = L % a.ciblocks;
    L /= a.ciblocks;
    const int cob = L % a.coblocks;
    L /= a.coblocks;
    const int split = L % a.nsplit;
    const int g = L / a.nsplit;
    const int ci0 = cib * 16 * WC;
    const int co0 = cob * 16 * MTW;
    const int HW = d.H * d.W;
    const int OHW = d.OH * d.OW;

    {
        const int c0 = d.src_c[0];
        const int c01 = c0 + (d.nsrc > 1 ? d.src_c[1] : 0);
        for (int c = tid; c < 16 * WC; c += 256) {
            const int cg = ci0 + c;
            const float* ptr = a.src[0];
            int istr = 0;
            if (cg < d.Cin) {
                const int sidx = (cg < c0) ? 0 : ((cg < c01) ? 1 : 2);
                const int cl = (sidx == 0) ? cg : ((sidx == 1) ? cg - c0 : cg - c01);
                const float* sp = (sidx == 0) ? a.src[0] : ((sidx == 1) ? a.src[1] : a.src[2]);
                ptr = sp + ((long)d.src_coff[sidx] + g * d.src_gstride[sidx] + cl) * (long)HW;
                istr = d.src_ctot[sidx] * HW;
            }
            s_cptr[c] = (unsigned long long)ptr;
            s_cstr[c] = istr;
        }
    }

    // ---- per-lane constants of the transposed B reads ----
    // lane = 16q + 4q' + p supplies the address of row (pixel) 8q + 4h + q', channels 4p..4p+3
    const int qp = (lane >> 2) & 3;
    const int pp = lane & 3;
    int bbase[2][3], bswz[2][3];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c0h = (8 * (q & 1) + 4 * h + qp) * s;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            bbase[h][kx] = ((q >> 1) * s * PWp + c0h + kx) * 32 + pp * 8 + wc * a.xplane;
            bswz[h][kx] = ((c0h + kx) & 8) << 4;
        }
    }

    f32x4 acc[MTW][9];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int tiles = a.tiles_x * a.tiles_y;
    const int items = d.N * tiles;
    const long dz_g = ((long)d.out_coff + (long)g * d.Cout) * OHW;

    for (int item = split; item < items; item += a.nsplit) {
        const int n = item / tiles;
        const int tile = item - n * tiles;
        const int ty = tile / a.tiles_x;
        const int tx = tile - ty * a.tiles_x;
        const int oy0 = ty * WB_TH, ox0 = tx * WB_TW;
        const int iy0 = oy0 * s - d.pad_t;
        const int ix0 = ox0 * s - d.pad_l;

        __syncthreads();   // previous tile consumed (first pass: pointer table visible)

        // ---- dz tile: item = (output channel, 8 consecutive pixels) ----
        {
            f32x4 v0[MTW], v1[MTW];
            int ldso[MTW];
#pragma unroll
            for (int it = 0; it < MTW; ++it) {
                const int e = tid + 256 * it;
                const int col = e >> 4;
                const int kg = e & 15;
                const int oy = oy0 + (kg >> 1);
                const int ox = ox0 + 8 * (kg & 1);
                const bool rowok = (co0 + col < d.Cout) && (oy < d.OH);
                ldso[it] = col * WB_DZP + kg * 16;
                const long base = ((long)n * d.out_ctot) * OHW + dz_g + (long)(co0 + col) * OHW + (long)oy * d.OW + ox;
                if (a.vec4) {
                    const bool ok0 = rowok && (ox + 4 <= d.OW);
                    const bool ok1 = rowok && (ox + 8 <= d.OW);
                    const gf4ptr p0 = (gf4ptr)((gfptr)a.dz + (ok0 ? base : 0));
                    const gf4ptr p1 = (gf4ptr)((gfptr)a.dz + (ok1 ? base + 4 : 0));
                    // (mask as whole vectors: __builtin_bit_cast of a vector ELEMENT lvalue reads element 0)
                    const u32x4 t0 = __builtin_bit_cast(u32x4, *p0) & (ok0 ? 0xffffffffu : 0u);
                    const u32x4 t1 = __builtin_bit_cast(u32x4, *p1) & (ok1 ? 0xffffffffu : 0u);
                    v0[it] = __builtin_bit_cast(f32x4, t0);
                    v1[it] = __builtin_bit_cast(f32x4, t1);
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const bool ok = rowok && (ox + i < d.OW);
                        const gfptr p = (gfptr)a.dz + (ok ? base + i : 0);
                        const unsigned int bits = __builtin_bit_cast(unsigned int, *p) & (ok ? 0xffffffffu : 0u);
                        if (i < 4) v0[it][i] = __builtin_bit_cast(float, bits);
                        else v1[it][i - 4] = __builtin_bit_cast(float, bits);
                    }
                }
            }
#pragma unroll
            for (int it = 0; it < MTW; ++it) {
                u32x4 w;
                w[0] = wb_pack2(v0[it][0], v0[it][1]);
                w[1] = wb_pack2(v0[it][2], v0[it][3]);
                w[2] = wb_pack2(v1[it][0], v1[it][1]);
                w[3] = wb_pack2(v1[it][2], v1[it][3]);
                *(u32x4*)(s_dz + ldso[it]) = w;
                if (SPLIT) {
                    u32x4 l;
                    l[0] = wb_pack2_lo(v0[it][0], v0[it][1]);
                    l[1] = wb_pack2_lo(v0[it][2], v0[it][3]);
                    l[2] = wb_pack2_lo(v1[it][0], v1[it][1]);
                    l[3] = wb_pack2_lo(v1[it][2], v1[it][3]);
                    *(u32x4*)(s_dz + a.lo_dz + ldso[it]) = l;
                }
            }
        }

        // ---- input patch: item = (position, 8 channels) ----
        for (int e0 = 0; e0 < a.nitems_x; e0 += 256 * WB_ITX) {
            float v[WB_ITX][8];
            int ldso[WB_ITX];
#pragma unroll
            for (int it = 0; it < WB_ITX; ++it) {
                const int e = e0 + tid + 256 * it;
                const int grp = (int)(((float)e + 0.5f) * a.inv_npos);
                const int pos = e - grp * a.npos;
                const int r = (int)(((float)pos + 0.5f) * a.inv_pw);
                const int c = pos - r * PW;
                const int iy = iy0 + r, ix = ix0 + c;
                const bool inr = e < a.nitems_x;
                const bool ok = inr && (iy >= 0) && (ix >= 0) && (iy < d.H) && (ix < d.W);
                ldso[it] = inr ? ((grp >> 1) * a.xplane + (((r * PWp + c) * 32) ^ ((c & 8) << 4)) + (grp & 1) * 16) : -1;
                const int go = ok ? (iy * d.W + ix) : 0;
                const int cb = inr ? grp * 8 : 0;
                const int nvalid = ok ? (d.Cin - ci0 - cb) : 0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const gfptr p = (gfptr)s_cptr[cb + j] + (long)n * s_cstr[cb + j];
                    const unsigned int bits = __builtin_bit_cast(unsigned int, p[go]);
                    const unsigned int m = (unsigned int)((j - nvalid) >> 31);
                    v[it][j] = __builtin_bit_cast(float, bits & m);
                }
            }
#pragma unroll
            for (int it = 0; it < WB_ITX; ++it) {
                if (ldso[it] >= 0) {
                    u32x4 w;
                    w[0] = wb_pack2(v[it][0], v[it][1]);
                    w[1] = wb_pack2(v[it][2], v[it][3]);
                    w[2] = wb_pack2(v[it][4], v[it][5]);
                    w[3] = wb_pack2(v[it][6], v[it][7]);
                    *(u32x4*)(s_x + ldso[it]) = w;
                    if (SPLIT) {
                        u32x4 l;
                        l[0] = wb_pack2_lo(v[it][0], v[it][1]);
                        l[1] = wb_pack2_lo(v[it][2], v[it][3]);
                        l[2] = wb_pack2_lo(v[it][4], v[it][5]);
                        l[3] = wb_pack2_lo(v[it][6], v[it][7]);
                        *(u32x4*)(s_x + a.lo_x + ldso[it]) = l;
                    }
                }
            }
        }
        __syncthreads();

        // ---- MFMA: this wave's k-steps (32 pixels = 2 tile rows each) ----
        for (int ks = wk; ks < 4; ks += WK) {
            bf16x8 af[MTW], afl[MTW];
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                af[mt] = *(const bf16x8*)(s_dz + (mt * 16 + li) * WB_DZP + ks * 64 + q * 16);
                if (SPLIT) afl[mt] = *(const bf16x8*)(s_dz + a.lo_dz + (mt * 16 + li) * WB_DZP + ks * 64 + q * 16);
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int rowoff = ((2 * ks * s + ky) * PWp) * 32;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int a0 = (bbase[0][kx] + rowoff) ^ bswz[0][kx];
                    const int a1 = (bbase[1][kx] + rowoff) ^ bswz[1][kx];
                    const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(s_x + a0));
                    const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(s_x + a1));
                    const s16x8 bb = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                    const bf16x8 bf = __builtin_bit_cast(bf16x8, bb);
                    bf16x8 bfl;
                    if (SPLIT) {
                        const s16x4 c0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(s_x + a.lo_x + a0));
                        const s16x4 c1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(s_x + a.lo_x + a1));
                        bfl = __builtin_bit_cast(bf16x8, __builtin_shufflevector(c0, c1, 0, 1, 2, 3, 4, 5, 6, 7));
                    }
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt) {
                        f32x4 c = acc[mt][ky * 3 + kx];
                        if (SPLIT) {        // small terms first
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afl[mt], bf, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bfl, c, 0, 0, 0);
                        }
                        acc[mt][ky * 3 + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bf, c, 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- epilogue: transpose through LDS, then atomics along dW's memory order ----
    // D layout: column lane&15 = input channel of the tile, row (lane>>4)*4 + reg = output channel.
    float* s_ep = (float*)smem + wave * (16 * WB_EP);
    const int cit = ci0 + wc * 16;                       // first input channel of this wave's tile
    const int nrem = (d.Cin - cit) * 9;                  // valid [ci][tap] entries of a row (may be <= 0)
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) s_ep[(q * 4 + j) * WB_EP + li * 9 + t] = acc[mt][t][j];
        __syncthreads();
        for (int e = lane; e < 16 * 144; e += 64) {
            const int row = e / 144;
            const int rem = e - row * 144;
            const int co = co0 + mt * 16 + row;
            if (co < d.Cout && rem < nrem) {
                float* p = a.dw + (((long)(g * d.Cout + co) * d.w_cin_tot) + d.w_cin_off + cit) * 9 + rem;
                atomicAdd(p, s_ep[row * WB_EP + rem]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static inline int rup_i(int v, int m) { return (v + m - 1) / m * m; }

int jafb_wgrad(hipStream_t s, const jaf_conv_desc* d, const float* src0, const float* src1,
               const float* src2, const float* dz, float* dw, int accumulate, void*, int64_t) {
    JAF_REQUIRE(d->KH == 3 && d->KW == 3 && d->dil_in == 1);
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(dw, 0, sizeof(float) * (size_t)d->G * d->Cout * d->w_cin_tot * 9, s);
        if (e != hipSuccess) return (int)e;
    }
    WgBArgs a;
    a.src[0] = src0;
    a.src[1] = src1;
    a.src[2] = src2;
    a.dz = dz;
    a.dw = dw;
    a.d = *d;
    int MTW = 1;
    long bestPad = 1L << 60;
    for (int mt = 4; mt >= 1; --mt) {
        long pad = (long)jaf_cdiv(d->Cout, 16 * mt) * 16 * mt;
        if (pad < bestPad) { bestPad = pad; MTW = mt; }
    }
    a.WC = d->Cin <= 16 ? 1 : (d->Cin <= 32 ? 2 : 4);
    {   // the split mode keeps two images of everything in LDS: narrower input-channel blocks where 160 KB would not hold them
        const int ph = (WB_TH - 1) * d->stride + 3, pwp = rup_i((WB_TW - 1) * d->stride + 3, 8);
        const int nimg_ = d->precision == JAF_PREC_BF16X3 ? 2 : 1;
        while (a.WC > 1 && nimg_ * (a.WC * ph * pwp * 32 + 16 * MTW * WB_DZP) + 16 * a.WC * 12 + 16 > 160 * 1024) a.WC >>= 1;
    }
    {   // a launch that cannot fill the chip (the discriminators' 8x8 ... 64x64 layers: 8-64 pixel tiles in all) is one long
        // serial chain per workgroup -- 25 us whatever the size; smaller output blocks give more, shorter workgroups
        const long items0 = (long)d->N * jaf_cdiv(d->OW, WB_TW) * jaf_cdiv(d->OH, WB_TH);
        const long sp = items0 < JAF_WGRAD_MAX_SPLIT ? items0 : JAF_WGRAD_MAX_SPLIT;
        while ((long)d->G * jaf_cdiv(d->Cout, 16 * MTW) * jaf_cdiv(d->Cin, 16 * a.WC) * sp < 512) {
            if (MTW > 1) MTW = (MTW == 4) ? 2 : 1;
            else if (a.WC > 1) a.WC >>= 1;
            else break;
        }
    }
    a.WK = 4 / a.WC;
    a.coblocks = jaf_cdiv(d->Cout, 16 * MTW);
    a.ciblocks = jaf_cdiv(d->Cin, 16 * a.WC);
    a.tiles_x = jaf_cdiv(d->OW, WB_TW);
    a.tiles_y = jaf_cdiv(d->OH, WB_TH);
    a.PH = (WB_TH - 1) * d->stride + 3;
    a.PW = (WB_TW - 1) * d->stride + 3;
    a.npos = a.PH * a.PW;
    a.PWp = rup_i(a.PW, 8);                              // the bit-7 swizzle stays inside a row
    const bool split = d->precision == JAF_PREC_BF16X3;
    const int nimg = split ? 2 : 1;
    a.xplane = a.PH * a.PWp * 32;
    a.lo_x = a.WC * a.xplane;
    a.lo_dz = 16 * MTW * WB_DZP;
    a.off_dz = nimg * a.WC * a.xplane;
    a.off_cptr = a.off_dz + nimg * 16 * MTW * WB_DZP;
    a.nitems_x = a.npos * 2 * a.WC;
    a.vec4 = (d->OW % 4 == 0) ? 1 : 0;
    a.inv_pw = 1.0f / (float)a.PW;
    a.inv_npos = 1.0f / (float)a.npos;
    int lds = a.off_cptr + 16 * a.WC * 12 + 16;
    const int lds_ep = 4 * 16 * WB_EP * 4;
    if (lds < lds_ep) lds = lds_ep;
    JAF_REQUIRE(lds <= 160 * 1024);
    const long items = (long)d->N * a.tiles_x * a.tiles_y;
    const long outblocks = (long)d->G * a.coblocks * a.ciblocks;
    // pixel splits from the kernel's true residency, in whole rounds (jaf_wgrad_nsplit_rounds; fp32-input staging: slower items)
#define JAF_WGB(MT_, SP_)                                                                              \
    do {                                                                                               \
        auto k = conv_wgrad_bf16_kernel<MT_, SP_>;                                                     \
        static int optin[JAF_MAX_DEVICES];                                                             \
        static JafOcc occ[JAF_MAX_DEVICES][8];                                                         \
        if (lds > 48 * 1024) {                                                                         \
            const int e = jaf_lds_optin((const void*)k, optin);                                        \
            if (e) return e;                                                                           \
        }                                                                                              \
        a.nsplit = (int)jaf_wgrad_nsplit_rounds(items, outblocks, (long)d->G * d->Cout * d->Cin * 9,    \
                                                (double)jaf_kernel_slots((const void*)k, lds, occ), 5e-6); \
        const long nblk = outblocks * a.nsplit;                                                        \
        if (nblk > 0x7fffffffL) return JAF_EINVAL;                                                     \
        JAF_NOTE_KERNEL("conv_wgrad_bf16_kernel<%d, %s>", MT_, (SP_) ? "true" : "false");               \
        hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(256), (size_t)lds, s, a);                     \
    } while (0)
    if (split) switch (MTW) {
        case 1: JAF_WGB(1, true); break;
        case 2: JAF_WGB(2, true); break;
        case 3: JAF_WGB(3, true); break;
        default: JAF_WGB(4, true); break;
    }
    else switch (MTW) {
        case 1: JAF_WGB(1, false); break;
        case 2: JAF_WGB(2, false); break;
        case 3: JAF_WGB(3, false); break;
        default: JAF_WGB(4, false); break;
    }
#undef JAF_WGB
    return jaf_launch_status();
}

int64_t jafb_wgrad_workspace(const jaf_conv_desc*) { return 0; }
