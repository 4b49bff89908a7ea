// Implicit-GEMM convolution on the bf16 matrix cores of gfx950 (v_mfma_f32_16x16x32_bf16),
// fp32 tensors in HBM, fp32 accumulation.
//
//   GEMM view:  D[M = Cout rows][N = output pixels] = W[M][K] * im2col(X)[K][N]
//
// * rows (M): 16-row tiles, MT per workgroup; columns (N): 16-pixel tiles on the MFMA lane,
//   NT per wave, 4 waves per workgroup -> 64*NT pixels per workgroup out of a TWIN-wide window;
// * K is walked in LDS chunks of NG 8-channel groups.  Inside a chunk the reduction index is the
//   flattened SLOT  s = tap*groups + group;  one MFMA (K = 32) consumes 4 consecutive slots,
//   the k-quarter q = lane>>4 of the instruction taking slot 4*step+q.  So a 3-channel 5x5
//   layer spends 7 MFMA steps (25 taps x 1 group) instead of 25, and Cin = 24 (3 groups x 9
//   taps = 27 slots) spends 7 instead of 9: channel counts are only padded to 8, never to 32;
// * the input patch of a chunk (halo, zero padding, x2 zero dilation for stride-2 dgrad, up to
//   three concatenated sources) is staged once as bf16 in LDS in the layout
//   [group][position][8 channels]: a B fragment (8 consecutive k of one pixel) is one
//   ds_read_b128, 16 consecutive pixels are 256 contiguous bytes and the group planes are
//   256-byte multiples apart, so operand reads are bank-conflict free at stride 1;
// * weights are pre-packed (jaf_conv2d_pack) into the exact A-fragment order
//   [chunk][step][row tile][q][row][8 channels]: staging is a linear copy and an A fragment is a
//   conflict-free ds_read_b128 at lane*16;
// * JAF_PREC_BF16X3 keeps a second (residual) image of both operands and issues
//   ah*bh + al*bh + ah*bl per step.
//
// Workgroups are numbered so that the blocks of one XCD (blockIdx % 8) walk a contiguous range
// of (row block fastest, then pixel tile): the row blocks that re-read one input patch run back
// to back on the same L2.
#include "conv_internal.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) float* gfptr;

#define CB_IT 6   // staged (position, 8-channel group) items per thread per chunk

struct ConvBArgs {
    const float* src[3];
    const unsigned char* wpk;
    const float* bias;
    float* out;
    const float* c_prev;
    float* c_out;
    float* h_out;
    float* gates_out;
    jaf_conv_desc d;
    jaf_conv_plan p;
    int off_w, off_tab, off_cptr;   // LDS byte offsets
    int ntiles;
    float inv_pw, inv_npos;
};

__device__ __forceinline__ unsigned int pack_bf16x2(float a, float b) {
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(unsigned int, r);
}
__device__ __forceinline__ float bf16_round(float a) { return (float)(__bf16)a; }

template <int MT, int NT, int NSPLIT, bool LSTM>
__global__ __launch_bounds__(256) void conv_bf16_kernel(const ConvBArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NIMG = (NSPLIT == 1) ? 1 : 2;
    const jaf_conv_desc& d = a.d;
    const jaf_conv_plan& P = a.p;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 15;
    const int q = lane >> 4;
    constexpr int MR = 16 * MT;
    const int NG = P.NG;
    const int CK = 8 * NG;
    const int npos = P.npos, plane = P.plane, PW = P.PW;

    unsigned char* s_patch = smem;
    unsigned char* s_w = smem + a.off_w;
    int* s_tab = (int*)(smem + a.off_tab);
    const float** s_cptr = (const float**)(smem + a.off_cptr);

    // ---- block -> (row block, pixel tile, image, group), XCD-contiguous ----
    int L;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int xcd = bid & 7, j = bid >> 3, qn = nblk >> 3, rn = nblk & 7;
        L = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + j;
    }
    const int mb = L % P.mblocks;
    L /= P.mblocks;
    const int tile = L % a.ntiles;
    const int ngi = L / a.ntiles;
    const int n = ngi / d.G;
    const int g = ngi - n * d.G;
    const int tx = tile % P.tiles_x;
    const int tb = tile / P.tiles_x;
    const int x0 = tx * P.TWIN;
    const int pbase = tb * (64 * NT);
    const int oy0 = pbase / P.TWIN;
    const int iy0 = oy0 * d.stride - d.pad_t;
    const int ix0 = x0 * d.stride - d.pad_l;
    const int OHW = d.OH * d.OW;
    const int HW = d.H * d.W;

    // ---- one-time tables: slot -> patch byte offset (full and last chunk), channel -> plane pointer
    {
        const int taps = d.KH * d.KW;
        for (int s = tid; s < 4 * P.nsteps; s += 256) {
            int v0 = 0, v1 = 0;
            if (s < taps * NG) {
                const int tap = s / NG, grp = s - tap * NG;
                const int ky = tap / d.KW, kx = tap - ky * d.KW;
                v0 = grp * plane + (ky * PW + kx) * 16;
            }
            if (s < taps * P.ng_last) {
                const int tap = s / P.ng_last, grp = s - tap * P.ng_last;
                const int ky = tap / d.KW, kx = tap - ky * d.KW;
                v1 = grp * plane + (ky * PW + kx) * 16;
            }
            s_tab[s] = v0;
            s_tab[4 * P.nsteps + s] = v1;
        }
        const int c0 = d.src_c[0];
        const int c01 = c0 + (d.nsrc > 1 ? d.src_c[1] : 0);
        for (int c = tid; c < P.nchunks * CK; c += 256) {
            const float* ptr = a.src[0];      // padding channels: any readable plane (value is discarded)
            if (c < d.Cin) {
                const int s = (c < c0) ? 0 : ((c < c01) ? 1 : 2);
                const int cl = (s == 0) ? c : ((s == 1) ? c - c0 : c - c01);
                const float* sp = (s == 0) ? a.src[0] : ((s == 1) ? a.src[1] : a.src[2]);
                ptr = sp + ((long)n * d.src_ctot[s] + d.src_coff[s] + g * d.src_gstride[s] + cl) * (long)HW;
            }
            s_cptr[c] = ptr;
        }
    }

    // ---- per-lane output pixels ----
    int boff[NT];
    int opix[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int p = pbase + (wave * NT + nt) * 16 + li;
        const int oy = p / P.TWIN;
        const int ox = x0 + (p - oy * P.TWIN);
        const bool valid = (oy < d.OH) && (ox < d.OW);
        boff[nt] = valid ? (((oy - oy0) * d.stride * PW + (ox - x0) * d.stride) * 16) : 0;
        opix[nt] = valid ? (oy * d.OW + ox) : -1;
    }

    // ---- per-thread staging items (chunk invariant): LDS byte offset, source element offset ----
    int it_lds[CB_IT], it_goff[CB_IT], it_c0[CB_IT];
    {
        const int dil = d.dil_in;
        const int Hd = (d.H - 1) * dil + 1;
        const int Wd = (d.W - 1) * dil + 1;
#pragma unroll
        for (int it = 0; it < CB_IT; ++it) {
            const int e = tid + 256 * it;
            const int grp = (int)(((float)e + 0.5f) * a.inv_npos);
            const int pos = e - grp * npos;
            const int r = (int)(((float)pos + 0.5f) * a.inv_pw);
            const int x = pos - r * PW;
            const int iyd = iy0 + r, ixd = ix0 + x;
            bool ok = (iyd >= 0) && (ixd >= 0) && (iyd < Hd) && (ixd < Wd);
            int iy = iyd, ix = ixd;
            if (dil == 2) {
                ok = ok && !((iyd | ixd) & 1);
                iy = iyd >> 1;
                ix = ixd >> 1;
            }
            const bool inr = e < npos * NG;
            it_lds[it] = inr ? (grp * plane + pos * 16) : -1;
            it_goff[it] = (ok && inr) ? (iy * d.W + ix) : -1;
            it_c0[it] = inr ? grp * 8 : 0;
        }
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const long wchunk_bytes = (long)NIMG * P.nsteps * MT * 1024;
    const unsigned char* wbase = a.wpk + ((long)(g * P.mblocks + mb) * P.nchunks) * wchunk_bytes;
    const int wimg = P.nsteps * MT * 1024;      // bytes between the hi and lo weight images
    const int pimg = NG * plane;                // bytes between the hi and lo patch images

    for (int chunk = 0; chunk < P.nchunks; ++chunk) {
        const bool last = (chunk == P.nchunks - 1);
        const int ngc = last ? P.ng_last : NG;
        const int nst = last ? P.nsteps_last : P.nsteps;
        __syncthreads();   // previous chunk consumed (first pass: tables visible)

        // ---- stage: issue every global load of the chunk, then convert and write ----
        // Loads are unconditional, from always-readable global addresses, and the value is masked
        // with integer arithmetic afterwards: any control flow here makes hipcc branch around each
        // load and wait for it (cdna_hip_programming.md 5, item 4(c)).
        float v[CB_IT][8];
#pragma unroll
        for (int it = 0; it < CB_IT; ++it) {
            const unsigned long long* cp = (const unsigned long long*)s_cptr + chunk * CK + it_c0[it];
            const int go = it_goff[it] < 0 ? 0 : it_goff[it];
            const int nvalid = (it_goff[it] >= 0) ? (d.Cin - chunk * CK - it_c0[it]) : 0;   // channels j < nvalid are real
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const gfptr p = (gfptr)cp[j];
                const unsigned int bits = __builtin_bit_cast(unsigned int, p[go]);
                const unsigned int m = (unsigned int)((j - nvalid) >> 31);      // all ones iff j < nvalid
                v[it][j] = __builtin_bit_cast(float, bits & m);
            }
        }
        {
            const int per = nst * MT * 64;          // 16-byte units per image actually used
            const u32x4* wsrc = (const u32x4*)(wbase + (long)chunk * wchunk_bytes);
            u32x4* wdst = (u32x4*)s_w;
#pragma unroll 4
            for (int e = tid; e < NIMG * per; e += 256) {
                const int img = (e >= per) ? 1 : 0;
                const int o = img * (wimg >> 4) + (e - img * per);
                wdst[o] = wsrc[o];
            }
        }
#pragma unroll
        for (int it = 0; it < CB_IT; ++it) {
            const bool act = (it_lds[it] >= 0) && (it_c0[it] < 8 * ngc);
            if (act) {
                u32x4 hi;
                hi[0] = pack_bf16x2(v[it][0], v[it][1]);
                hi[1] = pack_bf16x2(v[it][2], v[it][3]);
                hi[2] = pack_bf16x2(v[it][4], v[it][5]);
                hi[3] = pack_bf16x2(v[it][6], v[it][7]);
                *(u32x4*)(s_patch + it_lds[it]) = hi;
                if (NSPLIT == 3) {
                    float r[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) r[j] = v[it][j] - bf16_round(v[it][j]);
                    u32x4 lo;
                    lo[0] = pack_bf16x2(r[0], r[1]);
                    lo[1] = pack_bf16x2(r[2], r[3]);
                    lo[2] = pack_bf16x2(r[4], r[5]);
                    lo[3] = pack_bf16x2(r[6], r[7]);
                    *(u32x4*)(s_patch + pimg + it_lds[it]) = lo;
                }
            }
        }
        __syncthreads();

        // ---- MFMA over the chunk's steps ----
        const int* tab = s_tab + (last ? 4 * P.nsteps : 0);
        for (int st = 0; st < nst; ++st) {
            const int off = tab[4 * st + q];
            bf16x8 bh[NT], bl[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                bh[nt] = *(const bf16x8*)(s_patch + off + boff[nt]);
                if (NSPLIT == 3) bl[nt] = *(const bf16x8*)(s_patch + pimg + off + boff[nt]);
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned char* wp = s_w + (st * MT + mt) * 1024 + lane * 16;
                const bf16x8 ah = *(const bf16x8*)wp;
                if (NSPLIT == 3) {
                    const bf16x8 al = *(const bf16x8*)(wp + wimg);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[nt], acc[mt][nt], 0, 0, 0);
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[nt], acc[mt][nt], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[nt], acc[mt][nt], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue (D layout: column lane&15 = pixel, row (lane>>4)*4 + reg = output channel) ----
    if (!LSTM) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = mb * MR + mt * 16 + q * 4 + j;
                if (co >= d.Cout) continue;
                const float b = a.bias ? a.bias[g * d.Cout + co] : 0.f;
                float* op = a.out + ((long)n * d.out_ctot + d.out_coff + g * d.Cout + co) * OHW;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (opix[nt] < 0) continue;
                    op[opix[nt]] = jaf_act(acc[mt][nt][j] + b, d.act, d.slope);
                }
            }
        }
    } else {
        const int C = d.Cout >> 2;   // hidden channels per group (rows are gate-interleaved: 4c+gate)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int ch = ((mb * MR + mt * 16) >> 2) + q;
            if (ch >= C) continue;
            const float* bp = a.bias + g * d.Cout;
            const float bi = bp[ch], bf = bp[C + ch], bo = bp[2 * C + ch], bg = bp[3 * C + ch];
            const long hc = ((long)n * d.G + g) * C + ch;
            const long gc = ((long)n * d.G + g) * d.Cout;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (opix[nt] < 0) continue;
                const float gi = jaf_sigmoid(acc[mt][nt][0] + bi);
                const float gf = jaf_sigmoid(acc[mt][nt][1] + bf);
                const float go = jaf_sigmoid(acc[mt][nt][2] + bo);
                const float gg = jaf_tanh(acc[mt][nt][3] + bg);
                const float cp = a.c_prev ? a.c_prev[hc * OHW + opix[nt]] : 0.f;
                const float cc = gf * cp + gi * gg;
                a.c_out[hc * OHW + opix[nt]] = cc;
                a.h_out[hc * OHW + opix[nt]] = go * jaf_tanh(cc);
                if (a.gates_out) {
                    float* gp = a.gates_out + gc * OHW + opix[nt];
                    gp[(long)(ch)*OHW] = gi;
                    gp[(long)(C + ch) * OHW] = gf;
                    gp[(long)(2 * C + ch) * OHW] = go;
                    gp[(long)(3 * C + ch) * OHW] = gg;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// planning
// ---------------------------------------------------------------------------------------------
static inline int rup(int v, int m) { return (v + m - 1) / m * m; }

int jafb_plan(const jaf_conv_desc* d, int lstm, jaf_conv_plan* plan) {
    const int M = d->Cout;
    const int taps = d->KH * d->KW;
    const int NS = (d->precision == JAF_PREC_BF16X3) ? 3 : 1;
    const int NIMG = (NS == 1) ? 1 : 2;
    int bestMT = 1;
    long bestPad = 1L << 60;
    for (int mt = 4; mt >= 1; --mt) {
        long pad = (long)jaf_cdiv(M, 16 * mt) * 16 * mt;
        if (pad < bestPad) { bestPad = pad; bestMT = mt; }
    }
    int MT = bestMT;
    if (lstm) {
        MT = (M % 48 == 0) ? 3 : ((M % 64 == 0) ? 4 : ((M % 32 == 0) ? 2 : 1));
        JAF_REQUIRE(M % (16 * MT) == 0);
    }
    const int groups = jaf_cdiv(d->Cin, 8);
    const long OHW = (long)d->OH * d->OW;

    double bestCost = 1e300;
    int bTW = 0, bNT = 0, bNG = 0;
    const int cand_tw[4] = {16, 32, 64, d->OW};
    for (int ci = 0; ci < 4; ++ci) {
        const int TW = cand_tw[ci];
        if (ci < 3 && TW >= d->OW) continue;
        for (int NT = 4; NT >= 1; NT >>= 1) {
            const int Pn = 64 * NT;
            int rows_span, tiles_x, tiles_p;
            if (ci < 3) {
                if (Pn % TW) continue;
                rows_span = Pn / TW;
                tiles_x = jaf_cdiv(d->OW, TW);
                tiles_p = jaf_cdiv(d->OH, rows_span);
            } else {
                rows_span = (Pn % TW == 0) ? Pn / TW : (Pn + TW - 2) / TW + 1;
                if (rows_span > d->OH) rows_span = d->OH;
                tiles_x = 1;
                tiles_p = jaf_cdiv(OHW, Pn);
            }
            const int PH = (rows_span - 1) * d->stride + d->KH;
            const int PW = (TW - 1) * d->stride + d->KW;
            const int npos = PH * PW;
            const int plane = rup(npos * 16, 256);
            for (int NG = (groups < 4 ? groups : 4); NG >= 1; --NG) {
                if ((long)npos * NG > 256L * CB_IT) continue;
                const int nchunks = jaf_cdiv(groups, NG);
                const int ng_last = groups - (nchunks - 1) * NG;
                const int nsteps = jaf_cdiv(taps * NG, 4);
                const int nsteps_last = jaf_cdiv(taps * ng_last, 4);
                const long lds = (long)NIMG * ((long)NG * plane + (long)nsteps * MT * 1024) + 2L * 4 * nsteps * 4 +
                                 (long)nchunks * NG * 8 * 8 + 64;
                if (lds > 150 * 1024) continue;
                const double total_steps = (double)(nchunks - 1) * nsteps + nsteps_last;
                const double mfma = (double)MT * NT * NS * 16.0;
                const double ldsrd = 4.0 * (MT + NT) * NIMG * 4.0;
                const double t_step = mfma > ldsrd ? mfma : ldsrd;
                const double stage = ((double)npos * NG / 256.0) * 8.0 * 10.0 +
                                     (double)NIMG * ((double)npos * NG * 16.0 + (double)nsteps * MT * 1024.0) / 64.0 + 400.0;
                const int blocks_cu = (int)(160 * 1024 / lds);
                const double overlap = blocks_cu >= 2 ? 0.5 : 1.0;
                const double cost = (double)tiles_x * tiles_p * (total_steps * t_step + nchunks * stage * overlap);
                if (cost < bestCost) { bestCost = cost; bTW = TW; bNT = NT; bNG = NG; }
            }
        }
    }
    JAF_REQUIRE(bTW > 0);
    const int Pn = 64 * bNT;
    const bool linear = (bTW == d->OW);
    int rows_span;
    if (!linear) {
        rows_span = Pn / bTW;
        plan->tiles_x = jaf_cdiv(d->OW, bTW);
        plan->tiles_p = jaf_cdiv(d->OH, rows_span);
    } else {
        rows_span = (Pn % bTW == 0) ? Pn / bTW : (Pn + bTW - 2) / bTW + 1;
        if (rows_span > d->OH) rows_span = d->OH;
        plan->tiles_x = 1;
        plan->tiles_p = jaf_cdiv(OHW, Pn);
    }
    plan->precision = d->precision;
    plan->MT = MT;
    plan->NT = bNT;
    plan->NG = bNG;
    plan->CK = 8 * bNG;
    plan->TWIN = bTW;
    plan->PH = (rows_span - 1) * d->stride + d->KH;
    plan->PW = (bTW - 1) * d->stride + d->KW;
    plan->PWp = plan->PW;
    plan->npos = plan->PH * plan->PW;
    plan->plane = rup(plan->npos * 16, 256);
    plan->PS = plan->plane;
    plan->MRp = 16 * MT;
    plan->nchunks = jaf_cdiv(groups, bNG);
    plan->ng_last = groups - (plan->nchunks - 1) * bNG;
    plan->nsteps = jaf_cdiv(taps * bNG, 4);
    plan->nsteps_last = jaf_cdiv(taps * plan->ng_last, 4);
    plan->mblocks = jaf_cdiv(M, 16 * MT);
    plan->lds_bytes = (int)((long)NIMG * ((long)bNG * plan->plane + (long)plan->nsteps * MT * 1024) +
                            2L * 4 * plan->nsteps * 4 + (long)plan->nchunks * bNG * 8 * 8 + 64);
    plan->packed_floats = ((int64_t)d->G * plan->mblocks * plan->nchunks * NIMG * plan->nsteps * MT * 1024) / 4;
    return JAF_OK;
}

static bool planb_ok(const jaf_conv_desc* d, const jaf_conv_plan* p) {
    if (!p) return false;
    if (p->precision != d->precision) return false;
    if (p->MT < 1 || p->MT > 4) return false;
    if (p->NT != 1 && p->NT != 2 && p->NT != 4) return false;
    if (p->NG < 1 || p->NG > 4) return false;
    const int groups = jaf_cdiv(d->Cin, 8);
    if (p->nchunks != jaf_cdiv(groups, p->NG)) return false;
    if (p->ng_last != groups - (p->nchunks - 1) * p->NG) return false;
    const int taps = d->KH * d->KW;
    if (p->nsteps != jaf_cdiv(taps * p->NG, 4) || p->nsteps_last != jaf_cdiv(taps * p->ng_last, 4)) return false;
    if (p->mblocks != jaf_cdiv(d->Cout, 16 * p->MT)) return false;
    if (p->npos != p->PH * p->PW || p->plane < p->npos * 16 || (p->plane & 255)) return false;
    if ((long)p->npos * p->NG > 256L * CB_IT) return false;
    if (p->TWIN < 1 || p->tiles_x < 1 || p->tiles_p < 1) return false;
    // the patch must cover every tap of every pixel of a tile
    const int Pn = 64 * p->NT;
    int rows_span;
    if (p->TWIN == d->OW && p->tiles_x == 1) {
        rows_span = (Pn % p->TWIN == 0) ? Pn / p->TWIN : (Pn + p->TWIN - 2) / p->TWIN + 1;
        if (rows_span > d->OH) rows_span = d->OH;
    } else {
        if (Pn % p->TWIN) return false;
        rows_span = Pn / p->TWIN;
    }
    if (p->PH < (rows_span - 1) * d->stride + d->KH) return false;
    if (p->PW < (p->TWIN - 1) * d->stride + d->KW) return false;
    if (p->lds_bytes > 160 * 1024) return false;
    return true;
}

// ---------------------------------------------------------------------------------------------
// weight packing: fp32 reference layout -> bf16 A-fragment order
// ---------------------------------------------------------------------------------------------
struct PackBArgs {
    const float* w;
    unsigned short* out;
    long total;
    int G, M, Cred, taps, NG, ng_last, MT, nsteps, nchunks, mblocks, nimg;
    long sg, srow, sch, base;
    int flip, lstmC;
    int redC;              // JAF_PACK_DGRAD_LSTM: reduction channel 4 c + gate -> weight row gate * redC + c
};

// One 16-byte item (8 reduction channels of one (k-step, k-group, row)) per thread: the index decode -- eight integer
// divisions -- is paid once per item instead of once per element (element-wise the re-packing of a module's images after its
// optimiser step ran at 0.3 TB/s: 0.9 ms per train step).
__device__ __forceinline__ void jafb_pack_item8(const PackBArgs& a, long item) {
    long t = item;
    const int row = (int)(t & 15); t >>= 4;
    const int q = (int)(t & 3); t >>= 2;
    const int mt = (int)(t % a.MT); t /= a.MT;
    const int st = (int)(t % a.nsteps); t /= a.nsteps;
    const int img = (int)(t % a.nimg); t /= a.nimg;
    const int chunk = (int)(t % a.nchunks); t /= a.nchunks;
    const int mb = (int)(t % a.mblocks); t /= a.mblocks;
    const int g = (int)t;
    const int ngc = (chunk == a.nchunks - 1) ? a.ng_last : a.NG;
    const int s = 4 * st + q;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    const int r = mb * 16 * a.MT + mt * 16 + row;
    if (s < a.taps * ngc && r < a.M) {
        const int tap = s / ngc, grp = s - tap * ngc;
        int srow = r;
        if (a.lstmC > 0) srow = (r & 3) * a.lstmC + (r >> 2);
        const int stap = a.flip ? (a.taps - 1 - tap) : tap;
        const float* wp = a.w + a.base + g * a.sg + srow * a.srow + stap;
        const int ch0 = (chunk * a.NG + grp) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ch = ch0 + j;
            if (ch < a.Cred) {
                const int chs = a.redC > 0 ? (ch & 3) * a.redC + (ch >> 2) : ch;
                v[j] = wp[chs * a.sch];
            }
        }
    }
    typedef unsigned int pk_u32x4 __attribute__((ext_vector_type(4)));
    typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
    typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
    pk_u32x4 w;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float x0 = v[2 * u], x1 = v[2 * u + 1];
        if (img == 1) {          // residual image of the split-bf16 mode: v - bf16(v)
            x0 = x0 - (float)(__bf16)x0;
            x1 = x1 - (float)(__bf16)x1;
        }
        const pk_f32x2 p2 = {x0, x1};
        w[u] = __builtin_bit_cast(unsigned int, __builtin_convertvector(p2, pk_bf16x2));
    }
    *(pk_u32x4*)(a.out + item * 8) = w;
}

__global__ void conv_pack_bf16_kernel(const PackBArgs a) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (a.total >> 3); e += (long)gridDim.x * blockDim.x) {
        jafb_pack_item8(a, e);
    }
}

static int jafb_pack_args(const jaf_conv_desc* d, const jaf_conv_plan* plan, int mode, const float* w, int32_t w_rows_tot,
                          void* packed, PackBArgs& a);

int jafb_pack(hipStream_t s, const jaf_conv_desc* d, const jaf_conv_plan* plan, int mode, const float* w,
              int32_t w_rows_tot, void* packed) {
    PackBArgs a;
    const int e = jafb_pack_args(d, plan, mode, w, w_rows_tot, packed, a);
    if (e) return e;
    hipLaunchKernelGGL(conv_pack_bf16_kernel, dim3(jaf_ew_grid(a.total >> 3)), dim3(256), 0, s, a);
    return jaf_launch_status();
}

// Many weight images in ONE launch (the re-packing after an optimiser step: ~40 images per module): blockIdx.y picks
// the image's argument block out of a device-resident table.
__global__ void conv_pack_bf16_batch_kernel(const PackBArgs* __restrict__ table) {
    const PackBArgs a = table[blockIdx.y];
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (a.total >> 3); e += (long)gridDim.x * blockDim.x) {
        jafb_pack_item8(a, e);
    }
}

extern "C" int64_t jaf_conv2d_pack_item_bytes(void) { return (int64_t)sizeof(PackBArgs); }

extern "C" int jaf_conv2d_pack_item(const jaf_conv_desc* d, const jaf_conv_plan* plan, int mode, const float* w,
                                    int32_t w_rows_tot, void* packed, void* item_host, int64_t* total_out) {
    JAF_REQUIRE(d && item_host && d->precision != JAF_PREC_F32);
    PackBArgs a;
    const int e = jafb_pack_args(d, plan, mode, w, w_rows_tot, packed, a);
    if (e) return e;
    *(PackBArgs*)item_host = a;
    if (total_out) *total_out = a.total;
    return JAF_OK;
}

extern "C" int jaf_conv2d_pack_batch(jaf_stream_t s, const void* table_dev, int32_t n, int64_t max_total) {
    JAF_REQUIRE(table_dev && n >= 1 && n <= 65535 && max_total >= 1);
    long gx = jaf_cdiv(max_total, 256L * 8);          // ~8 elements per lane for the largest image; the others loop less
    if (gx < 1) gx = 1;
    if (gx > 512) gx = 512;
    hipLaunchKernelGGL(conv_pack_bf16_batch_kernel, dim3((unsigned)gx, (unsigned)n), dim3(256), 0, (hipStream_t)s,
                       (const PackBArgs*)table_dev);
    return jaf_launch_status();
}

static int jafb_pack_args(const jaf_conv_desc* d, const jaf_conv_plan* plan, int mode, const float* w, int32_t w_rows_tot,
                          void* packed, PackBArgs& a) {
    // the packed-input kernel (conv_dma.hip) shares this weight image: only the fields that shape it are checked
    JAF_REQUIRE(plan && w && packed && plan->precision == d->precision && plan->MT >= 1 && plan->MT <= 4 &&
                plan->NG >= 1 && plan->NG <= 4 && plan->nchunks == jaf_cdiv(jaf_cdiv(d->Cin, 8), plan->NG) &&
                plan->ng_last == jaf_cdiv(d->Cin, 8) - (plan->nchunks - 1) * plan->NG &&
                plan->nsteps == jaf_cdiv(d->KH * d->KW * plan->NG, 4) && plan->mblocks == jaf_cdiv(d->Cout, 16 * plan->MT));
    a.w = w;
    a.out = (unsigned short*)packed;
    a.G = d->G;
    a.M = d->Cout;
    a.Cred = d->Cin;
    a.taps = d->KH * d->KW;
    a.NG = plan->NG;
    a.ng_last = plan->ng_last;
    a.MT = plan->MT;
    a.nsteps = plan->nsteps;
    a.nchunks = plan->nchunks;
    a.mblocks = plan->mblocks;
    a.nimg = (d->precision == JAF_PREC_BF16X3) ? 2 : 1;
    a.total = plan->packed_floats * 2;
    a.flip = 0;
    a.lstmC = 0;
    a.redC = 0;
    const long khw = a.taps;
    if (mode == JAF_PACK_FWD || mode == JAF_PACK_LSTM) {
        JAF_REQUIRE(w_rows_tot >= d->Cout && d->w_cin_off + d->Cin <= d->w_cin_tot);
        a.sg = (long)w_rows_tot * d->w_cin_tot * khw;
        a.srow = (long)d->w_cin_tot * khw;
        a.sch = khw;
        a.base = (long)d->w_cin_off * khw;
        if (mode == JAF_PACK_LSTM) { JAF_REQUIRE((d->Cout & 3) == 0); a.lstmC = d->Cout >> 2; }
    } else if (mode == JAF_PACK_DGRAD || mode == JAF_PACK_DGRAD_LSTM) {
        JAF_REQUIRE(w_rows_tot >= d->Cin && d->w_cin_off + d->Cout <= d->w_cin_tot);
        if (mode == JAF_PACK_DGRAD_LSTM) { JAF_REQUIRE((d->Cin & 3) == 0 && d->precision != JAF_PREC_F32); a.redC = d->Cin >> 2; }
        a.sg = (long)w_rows_tot * d->w_cin_tot * khw;
        a.srow = khw;
        a.sch = (long)d->w_cin_tot * khw;
        a.base = (long)d->w_cin_off * khw;
        a.flip = 1;
    } else {
        return JAF_EINVAL;
    }
    return JAF_OK;
}

// ---------------------------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------------------------
template <int MT, int NT, int NS, bool LSTM>
static int launch_one(const ConvBArgs& a, hipStream_t s) {
    auto k = conv_bf16_kernel<MT, NT, NS, LSTM>;
    static int optin[JAF_MAX_DEVICES];
    const int lds = a.p.lds_bytes;
    if (lds > 48 * 1024) {
        const int e = jaf_lds_optin((const void*)k, optin);
        if (e) return e;
    }
    const long nblk = (long)a.ntiles * a.p.mblocks * a.d.N * a.d.G;
    if (nblk < 1 || nblk > 0x7fffffffL) return JAF_EINVAL;
    JAF_NOTE_KERNEL("conv_bf16_kernel<%d, %d, %d, %s>", MT, NT, NS, LSTM ? "true" : "false");
    hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(256), (size_t)lds, s, a);
    return jaf_launch_status();
}

template <int MT, int NS, bool LSTM>
static int launch_nt(const ConvBArgs& a, hipStream_t s) {
    switch (a.p.NT) {
        case 1: return launch_one<MT, 1, NS, LSTM>(a, s);
        case 2: return launch_one<MT, 2, NS, LSTM>(a, s);
        case 4: return launch_one<MT, 4, NS, LSTM>(a, s);
    }
    return JAF_EINVAL;
}

template <int NS, bool LSTM>
static int launch_mt(const ConvBArgs& a, hipStream_t s) {
    switch (a.p.MT) {
        case 1: return launch_nt<1, NS, LSTM>(a, s);
        case 2: return launch_nt<2, NS, LSTM>(a, s);
        case 3: return launch_nt<3, NS, LSTM>(a, s);
        case 4: return launch_nt<4, NS, LSTM>(a, s);
    }
    return JAF_EINVAL;
}

static void fill_args(ConvBArgs& a, const jaf_conv_desc* d, const jaf_conv_plan* plan) {
    const int NIMG = (d->precision == JAF_PREC_BF16X3) ? 2 : 1;
    a.d = *d;
    a.p = *plan;
    a.off_w = NIMG * plan->NG * plan->plane;
    a.off_tab = a.off_w + NIMG * plan->nsteps * plan->MT * 1024;
    a.off_cptr = (a.off_tab + 2 * 4 * plan->nsteps * 4 + 15) & ~15;
    a.ntiles = plan->tiles_x * plan->tiles_p;
    a.inv_pw = 1.0f / (float)plan->PW;
    a.inv_npos = 1.0f / (float)plan->npos;
    a.c_prev = nullptr;
    a.c_out = nullptr;
    a.h_out = nullptr;
    a.gates_out = nullptr;
}

int jafb_fwd(hipStream_t s, const jaf_conv_desc* d, const jaf_conv_plan* plan, const float* src0,
             const float* src1, const float* src2, const void* packed_w, const float* bias, float* out) {
    JAF_REQUIRE(planb_ok(d, plan) && src0 && packed_w && out);
    ConvBArgs a;
    fill_args(a, d, plan);
    JAF_REQUIRE(a.off_cptr + plan->nchunks * plan->NG * 64 <= plan->lds_bytes);
    a.src[0] = src0;
    a.src[1] = src1;
    a.src[2] = src2;
    a.wpk = (const unsigned char*)packed_w;
    a.bias = bias;
    a.out = out;
    if (d->precision == JAF_PREC_BF16X3) return launch_mt<3, false>(a, s);
    return launch_mt<1, false>(a, s);
}

int jafb_lstm(hipStream_t s, const jaf_conv_desc* d, const jaf_conv_plan* plan, const float* x,
              const float* h_prev, const void* packed_w, const float* bias, const float* c_prev,
              float* h_out, float* c_out, float* gates_out) {
    JAF_REQUIRE(planb_ok(d, plan) && x && packed_w && bias && h_out && c_out);
    ConvBArgs a;
    fill_args(a, d, plan);
    JAF_REQUIRE(a.off_cptr + plan->nchunks * plan->NG * 64 <= plan->lds_bytes);
    a.src[0] = x;
    a.src[1] = h_prev;
    a.src[2] = nullptr;
    a.wpk = (const unsigned char*)packed_w;
    a.bias = bias;
    a.out = nullptr;
    a.c_prev = c_prev;
    a.c_out = c_out;
    a.h_out = h_out;
    a.gates_out = gates_out;
    if (d->precision == JAF_PREC_BF16X3) return launch_mt<3, true>(a, s);
    return launch_mt<1, true>(a, s);
}
