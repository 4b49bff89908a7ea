// Implicit-GEMM convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_16x16x4_f32).
//
//   GEMM view:  D[M = Cout rows][N = output pixels] = W[M][K] * im2col(X)[K][N],  K = Cin*KH*KW
//
// * rows  (M) are output channels: 16-row tiles, MT tiles per workgroup;
// * cols  (N) are output pixels on the MFMA lane: 16-pixel tiles, NT tiles per wave, 4 waves
//   per workgroup -> 64*NT pixels per workgroup, taken from a window of the image that is
//   TWIN pixels wide (TWIN == OW walks the image linearly, which keeps 13x13 / 25x25 / 50x50
//   feature maps of the texture networks dense);
// * K is walked as (channel chunk of CK) x (tap) x (4 channels per MFMA): the chunk's input
//   patch (with halo, zero padding, optional x2 zero dilation, up to three concatenated
//   sources) and the matching slice of the pre-packed weights are staged in LDS, every tap
//   of the patch is then a shifted LDS read -- no im2col buffer ever exists in HBM.
//
// The ConvLSTM variant (src/convLSTM.py:41-56) packs the 4*C gate rows interleaved
// (row 4c+gate) so that the four accumulator registers of a lane are i,f,o,g of ONE hidden
// channel at ONE pixel; the cell update runs in the epilogue and the 4C-channel gate tensor
// is never round-tripped through HBM.
#include "conv_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
    const float* src[3];
    const float* wpk;
    const float* bias;
    float* out;
    // lstm
    const float* c_prev;
    float* c_out;
    float* h_out;
    float* gates_out;
    jaf_conv_desc d;
    jaf_conv_plan p;
    int sw_off;       // float offset of the weight image in LDS
    float inv_pw;     // 1/PW
    float inv_phpw;   // 1/(PH*PW)
};

template <int KS, int MT, int NT, bool LSTM>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;
    float* s_w = smem + a.sw_off;

    const jaf_conv_desc& d = a.d;
    const jaf_conv_plan& P = a.p;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 15;
    const int q = lane >> 4;
    const int KH = KS ? KS : d.KH;
    const int KW = KS ? KS : d.KW;
    const int KHW = KH * KW;
    constexpr int MR = 16 * MT;
    const int MRp = P.MRp;
    const int CK = P.CK;
    const int PS = P.PS, PWp = P.PWp, PH = P.PH, PW = P.PW;

    const int tx = blockIdx.x % P.tiles_x;
    const int tb = blockIdx.x / P.tiles_x;
    const int mb = blockIdx.y;
    const int n = blockIdx.z / d.G;
    const int g = blockIdx.z % d.G;
    const int x0 = tx * P.TWIN;
    const int pbase = tb * (64 * NT);
    const int oy0 = pbase / P.TWIN;
    const int iy0 = oy0 * d.stride - d.pad_t;
    const int ix0 = x0 * d.stride - d.pad_l;
    const int OHW = d.OH * d.OW;

    int boff[NT];
    int opix[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int p = pbase + (wave * NT + nt) * 16 + li;
        const int oy = p / P.TWIN;
        const int ox = x0 + (p - oy * P.TWIN);
        const bool valid = (oy < d.OH) && (ox < d.OW);
        boff[nt] = valid ? ((oy - oy0) * d.stride * PWp + (ox - x0) * d.stride) : 0;
        opix[nt] = valid ? (oy * d.OW + ox) : -1;
    }

    f32x4 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int c0 = d.src_c[0];
    const int c01 = d.src_c[0] + (d.nsrc > 1 ? d.src_c[1] : 0);
    const int patch_elems = CK * PH * PW;
    const int phpw = PH * PW;
    const int wchunk_floats = KHW * CK * MRp;
    const float* wbase = a.wpk + ((long)(g * P.mblocks + mb) * P.nchunks) * wchunk_floats;
    const int dil = d.dil_in;
    const int Hd = (d.H - 1) * dil + 1;   // extent of the (possibly dilated) source
    const int Wd = (d.W - 1) * dil + 1;

    for (int chunk = 0; chunk < P.nchunks; ++chunk) {
        __syncthreads();
        // ---- stage the input patch of this channel chunk ----
        for (int e = tid; e < patch_elems; e += 256) {
            int c = (int)(((float)e + 0.5f) * a.inv_phpw);
            int rem = e - c * phpw;
            int r = (int)(((float)rem + 0.5f) * a.inv_pw);
            int x = rem - r * PW;
            const int cg = chunk * CK + c;
            const int iyd = iy0 + r;
            const int ixd = ix0 + x;
            float v = 0.f;
            bool ok = (cg < d.Cin) && (iyd >= 0) && (ixd >= 0) && (iyd < Hd) && (ixd < Wd);
            int iy = iyd, ix = ixd;
            if (dil == 2) {
                ok = ok && !((iyd | ixd) & 1);
                iy = iyd >> 1;
                ix = ixd >> 1;
            }
            if (ok) {
                int s, cl;
                if (cg < c0) { s = 0; cl = cg; }
                else if (cg < c01) { s = 1; cl = cg - c0; }
                else { s = 2; cl = cg - c01; }
                const float* sp = a.src[s];
                const long ch = (long)n * d.src_ctot[s] + d.src_coff[s] + g * d.src_gstride[s] + cl;
                v = sp[(ch * d.H + iy) * d.W + ix];
            }
            s_in[c * PS + r * PWp + x] = v;
        }
        // ---- stage the packed weights of this chunk (linear float4 copy) ----
        {
            const f32x4* wsrc = (const f32x4*)(wbase + (long)chunk * wchunk_floats);
            f32x4* wdst = (f32x4*)s_w;
            const int nvec = wchunk_floats >> 2;
            for (int e = tid; e < nvec; e += 256) wdst[e] = wsrc[e];
        }
        __syncthreads();
        // ---- MFMA over taps x channel quads ----
        const int nq = CK >> 2;
        for (int ky = 0; ky < KH; ++ky) {
#pragma unroll
            for (int kx = 0; kx < (KS ? KS : 7); ++kx) {
                if (!KS && kx >= KW) break;
                const int tap = ky * KW + kx;
                const int toff = ky * PWp + kx;
                for (int cq = 0; cq < nq; ++cq) {
                    const float* wp = s_w + (tap * CK + cq * 4 + q) * MRp + li;
                    const float* ip = s_in + (cq * 4 + q) * PS + toff;
                    float av[MT], bv[NT];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) av[mt] = wp[mt * 16];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bv[nt] = ip[boff[nt]];
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv[nt], acc[mt][nt], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue ----
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
        const int C = d.Cout >> 2;   // hidden channels per group
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int ch = ((mb * MR + mt * 16) >> 2) + q;
            if (ch >= C) continue;
            const float* bp = a.bias + g * d.Cout;
            const float bi = bp[ch], bf = bp[C + ch], bo = bp[2 * C + ch], bg = bp[3 * C + ch];
            const long hc = ((long)n * d.G + g) * C + ch;       // channel in [N, G*C, H, W]
            const long gc = ((long)n * d.G + g) * d.Cout;        // gate block in [N, G*4C, H, W]
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
static int round_ps(int v) {   // channel pitch == 16 (mod 32) floats: the 4 k-lanes of an MFMA
    int r = v;                 // operand read hit disjoint LDS banks
    while ((r & 31) != 16) ++r;
    return r;
}

static bool desc_ok(const jaf_conv_desc* d) {
    if (!d) return false;
    if (d->N < 1 || d->G < 1 || d->Cin < 1 || d->Cout < 1) return false;
    if (d->H < 1 || d->W < 1 || d->OH < 1 || d->OW < 1) return false;
    if (d->KH < 1 || d->KW < 1 || d->KH > 7 || d->KW > 7) return false;
    if (d->stride < 1 || d->stride > 2) return false;
    if (d->dil_in != 1 && d->dil_in != 2) return false;
    if (d->dil_in == 2 && d->stride != 1) return false;
    if (d->nsrc < 1 || d->nsrc > 3) return false;
    int c = 0;
    for (int i = 0; i < d->nsrc; ++i) {
        if (d->src_c[i] < 1 || d->src_ctot[i] < 1 || d->src_coff[i] < 0 || d->src_gstride[i] < 0) return false;
        if (d->src_coff[i] + (d->G - 1) * d->src_gstride[i] + d->src_c[i] > d->src_ctot[i]) return false;
        c += d->src_c[i];
    }
    if (c != d->Cin) return false;
    if (d->w_cin_off < 0 || d->w_cin_tot < 1) return false;
    if (d->out_coff < 0 || d->out_coff + d->G * d->Cout > d->out_ctot) return false;
    if (d->pad_t < 0 || d->pad_l < 0) return false;
    if (d->precision < JAF_PREC_F32 || d->precision > JAF_PREC_BF16X3) return false;
    return true;
}

extern "C" int jaf_conv2d_plan(const jaf_conv_desc* d, int lstm, jaf_conv_plan* plan) {
    JAF_REQUIRE(desc_ok(d) && plan);
    if (lstm) JAF_REQUIRE((d->Cout & 3) == 0 && d->KH == 3 && d->KW == 3 && d->stride == 1);
    // the bf16 modes run on the packed-input kernels only (jaf_conv2d_plan_packed / jaf_conv2d_fwd_packed_io): the fp32-input
    // staging kernel of rounds 1-2 is gone
    if (d->precision != JAF_PREC_F32) return JAF_EUNSUPPORTED;
    plan->precision = JAF_PREC_F32;
    plan->NG = plan->ng_last = plan->nsteps = plan->nsteps_last = plan->npos = plan->plane = 0;
    const int M = d->Cout;
    // rows per workgroup: least padding, then the larger tile (more reuse of the patch)
    int bestMT = 1;
    long bestPad = 1L << 60;
    for (int mt = 4; mt >= 1; --mt) {
        long pad = (long)jaf_cdiv(M, 16 * mt) * 16 * mt;
        if (pad < bestPad) { bestPad = pad; bestMT = mt; }
    }
    int MT = bestMT;
    if (lstm) MT = (M % 48 == 0) ? 3 : ((M % 64 == 0) ? 4 : ((M % 32 == 0) ? 2 : 1));
    if (lstm) JAF_REQUIRE(M % (16 * MT) == 0);
    const int CKpref = (d->Cin <= 4) ? 4 : 8;
    const long OHW = (long)d->OH * d->OW;

    double bestCost = 1e30;
    int bTW = 0, bNT = 0, bCK = 0;
    const int cand_tw[4] = {16, 32, 64, d->OW};
    for (int ci = 0; ci < 4; ++ci) {
        const int TW = cand_tw[ci];
        if (ci < 3 && TW >= d->OW) continue;            // windows only narrower than the image
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
            const int PS = round_ps(PH * PW);
            int CK = CKpref;
            const int MRp = round_ps(16 * MT);
            long lds = ((long)CK * PS + (long)d->KH * d->KW * CK * MRp) * 4;
            if (lds > 64 * 1024 && CK == 8) { CK = 4; lds = ((long)CK * PS + (long)d->KH * d->KW * CK * MRp) * 4; }
            if (lds > 96 * 1024) continue;
            const double waste = (double)tiles_x * tiles_p * Pn / (double)OHW;
            const double halo = (double)PH * PW / (double)Pn;
            // small-pixel blocks amortise the weight staging worse
            const double wcost = (double)(d->KH * d->KW * MRp) / (double)(Pn * 8);
            const double cost = waste * (1.0 + 0.12 * halo + 0.10 * wcost);
            if (cost < bestCost) { bestCost = cost; bTW = TW; bNT = NT; bCK = CK; }
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
    plan->MT = MT;
    plan->NT = bNT;
    plan->CK = bCK;
    plan->TWIN = bTW;
    plan->PH = (rows_span - 1) * d->stride + d->KH;
    plan->PW = (bTW - 1) * d->stride + d->KW;
    plan->PWp = plan->PW;
    plan->PS = round_ps(plan->PH * plan->PWp);
    plan->MRp = round_ps(16 * MT);
    plan->nchunks = jaf_cdiv(d->Cin, bCK);
    plan->mblocks = jaf_cdiv(M, 16 * MT);
    plan->lds_bytes = (int)(((long)bCK * plan->PS + (long)d->KH * d->KW * bCK * plan->MRp) * 4);
    plan->packed_floats = (int64_t)d->G * plan->mblocks * plan->nchunks * d->KH * d->KW * bCK * plan->MRp;
    return JAF_OK;
}

// ---------------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------------
struct PackArgs {
    const float* w;
    float* out;
    long total;
    int G, M, Cred, KHW, CK, MR, MRp, nchunks, mblocks;
    long sg, srow, sch, base;
    int flip, lstmC;
};

__global__ void conv_pack_kernel(const PackArgs a) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < a.total; e += (long)gridDim.x * blockDim.x) {
        long t = e;
        const int m = (int)(t % a.MRp); t /= a.MRp;
        const int c = (int)(t % a.CK); t /= a.CK;
        const int tap = (int)(t % a.KHW); t /= a.KHW;
        const int chunk = (int)(t % a.nchunks); t /= a.nchunks;
        const int mb = (int)(t % a.mblocks); t /= a.mblocks;
        const int g = (int)t;
        float v = 0.f;
        const int row = mb * a.MR + m;
        const int ch = chunk * a.CK + c;
        if (m < a.MR && row < a.M && ch < a.Cred) {
            int srow = row;
            if (a.lstmC > 0) srow = (row & 3) * a.lstmC + (row >> 2);
            const int stap = a.flip ? (a.KHW - 1 - tap) : tap;
            v = a.w[a.base + g * a.sg + srow * a.srow + ch * a.sch + stap];
        }
        a.out[e] = v;
    }
}

extern "C" int jaf_conv2d_pack(jaf_stream_t s, const jaf_conv_desc* d, const jaf_conv_plan* plan, int mode,
                               const float* w, int32_t w_rows_tot, float* packed) {
    JAF_REQUIRE(desc_ok(d) && plan && w && packed);
    if (d->precision != JAF_PREC_F32) return jafb_pack((hipStream_t)s, d, plan, mode, w, w_rows_tot, packed);
    PackArgs a;
    a.w = w;
    a.out = packed;
    a.G = d->G;
    a.M = d->Cout;
    a.Cred = d->Cin;
    a.KHW = d->KH * d->KW;
    a.CK = plan->CK;
    a.MR = 16 * plan->MT;
    a.MRp = plan->MRp;
    a.nchunks = plan->nchunks;
    a.mblocks = plan->mblocks;
    a.total = plan->packed_floats;
    a.flip = 0;
    a.lstmC = 0;
    const long khw = a.KHW;
    if (mode == JAF_PACK_FWD || mode == JAF_PACK_LSTM) {
        // w: [G][w_rows_tot][w_cin_tot][KH][KW], rows = output channels
        JAF_REQUIRE(w_rows_tot >= d->Cout && d->w_cin_off + d->Cin <= d->w_cin_tot);
        a.sg = (long)w_rows_tot * d->w_cin_tot * khw;
        a.srow = (long)d->w_cin_tot * khw;
        a.sch = khw;
        a.base = (long)d->w_cin_off * khw;
        if (mode == JAF_PACK_LSTM) { JAF_REQUIRE((d->Cout & 3) == 0); a.lstmC = d->Cout >> 2; }
    } else if (mode == JAF_PACK_DGRAD) {
        // forward weight [G][w_rows_tot = fwd Cout][w_cin_tot = fwd Cin][KH][KW];
        // dgrad rows = forward input channels [w_cin_off, +d->Cout), reduction = forward Cout.
        JAF_REQUIRE(w_rows_tot >= d->Cin && d->w_cin_off + d->Cout <= d->w_cin_tot);
        a.sg = (long)w_rows_tot * d->w_cin_tot * khw;
        a.srow = khw;
        a.sch = (long)d->w_cin_tot * khw;
        a.base = (long)d->w_cin_off * khw;
        a.flip = 1;
    } else {
        return JAF_EINVAL;
    }
    hipLaunchKernelGGL(conv_pack_kernel, dim3(jaf_ew_grid(a.total)), dim3(256), 0, (hipStream_t)s, a);
    return jaf_launch_status();
}

// ---------------------------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------------------------
template <int KS, int MT, bool LSTM>
static int launch_nt(const ConvArgs& a, hipStream_t s) {
    dim3 grid(a.p.tiles_x * a.p.tiles_p, a.p.mblocks, a.d.N * a.d.G);
    dim3 block(256);
    const size_t lds = a.p.lds_bytes;
#define JAF_LAUNCH(NT_)                                                                              \
    do {                                                                                             \
        auto k = conv_mfma_kernel<KS, MT, NT_, LSTM>;                                                \
        if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        JAF_NOTE_KERNEL("conv_mfma_kernel<%d, %d, %d, %s>", KS, MT, NT_, LSTM ? "true" : "false");      \
        hipLaunchKernelGGL(k, grid, block, lds, s, a);                                               \
    } while (0)
    switch (a.p.NT) {
        case 1: JAF_LAUNCH(1); break;
        case 2: JAF_LAUNCH(2); break;
        case 4: JAF_LAUNCH(4); break;
        default: return JAF_EINVAL;
    }
#undef JAF_LAUNCH
    return jaf_launch_status();
}

template <int KS>
static int launch_mt(const ConvArgs& a, hipStream_t s) {
    switch (a.p.MT) {
        case 1: return launch_nt<KS, 1, false>(a, s);
        case 2: return launch_nt<KS, 2, false>(a, s);
        case 3: return launch_nt<KS, 3, false>(a, s);
        case 4: return launch_nt<KS, 4, false>(a, s);
    }
    return JAF_EINVAL;
}

static bool plan_ok(const jaf_conv_desc* d, const jaf_conv_plan* p) {
    if (!p) return false;
    if (p->precision != JAF_PREC_F32) return false;
    if (p->MT < 1 || p->MT > 4) return false;
    if (p->NT != 1 && p->NT != 2 && p->NT != 4) return false;
    if (p->CK < 4 || (p->CK & 3)) return false;
    if (p->nchunks != jaf_cdiv(d->Cin, p->CK)) return false;
    if (p->mblocks != jaf_cdiv(d->Cout, 16 * p->MT)) return false;
    if (p->MRp < 16 * p->MT || (p->MRp & 3)) return false;
    if (p->PS < p->PH * p->PWp || p->PWp < p->PW) return false;
    if (p->lds_bytes < (int)(((long)p->CK * p->PS + (long)d->KH * d->KW * p->CK * p->MRp) * 4)) return false;
    if (p->lds_bytes > 160 * 1024) return false;
    if (p->TWIN < 1 || p->tiles_x < 1 || p->tiles_p < 1) return false;
    return true;
}

static void fill_args(ConvArgs& a, const jaf_conv_desc* d, const jaf_conv_plan* plan) {
    a.d = *d;
    a.p = *plan;
    a.sw_off = plan->CK * plan->PS;
    a.inv_pw = 1.0f / (float)plan->PW;
    a.inv_phpw = 1.0f / (float)(plan->PH * plan->PW);
    a.c_prev = nullptr;
    a.c_out = nullptr;
    a.h_out = nullptr;
    a.gates_out = nullptr;
}

extern "C" int jaf_conv2d_fwd(jaf_stream_t s, const jaf_conv_desc* d, const jaf_conv_plan* plan,
                              const float* src0, const float* src1, const float* src2,
                              const float* packed_w, const float* bias, float* out) {
    JAF_REQUIRE(desc_ok(d) && plan && src0 && packed_w && out);
    JAF_REQUIRE(d->nsrc < 2 || src1);
    JAF_REQUIRE(d->nsrc < 3 || src2);
    if (d->precision != JAF_PREC_F32) return JAF_EUNSUPPORTED;          // bf16 modes: jaf_conv2d_fwd_packed_io
    JAF_REQUIRE(plan_ok(d, plan));
    ConvArgs a;
    fill_args(a, d, plan);
    a.src[0] = src0;
    a.src[1] = src1;
    a.src[2] = src2;
    a.wpk = packed_w;
    a.bias = bias;
    a.out = out;
    if (d->KH == 3 && d->KW == 3) return launch_mt<3>(a, (hipStream_t)s);
    return launch_mt<0>(a, (hipStream_t)s);
}

extern "C" int jaf_convlstm_cell_fwd(jaf_stream_t s, const jaf_conv_desc* d, const jaf_conv_plan* plan,
                                     const float* x, const float* h_prev, const float* packed_w,
                                     const float* bias, const float* c_prev,
                                     float* h_out, float* c_out, float* gates_out) {
    JAF_REQUIRE(desc_ok(d) && plan && x && packed_w && bias && h_out && c_out);
    JAF_REQUIRE(d->KH == 3 && d->KW == 3 && d->stride == 1 && (d->Cout & 3) == 0);
    JAF_REQUIRE(d->Cout % (16 * plan->MT) == 0);
    JAF_REQUIRE((d->nsrc == 2) == (h_prev != nullptr));
    JAF_REQUIRE(d->H == d->OH && d->W == d->OW);
    if (d->precision != JAF_PREC_F32) return JAF_EUNSUPPORTED;          // bf16 modes: jaf_convlstm_cell_fwd_packed_io
    JAF_REQUIRE(plan_ok(d, plan));
    ConvArgs a;
    fill_args(a, d, plan);
    a.src[0] = x;
    a.src[1] = h_prev;
    a.src[2] = nullptr;
    a.wpk = packed_w;
    a.bias = bias;
    a.out = nullptr;
    a.c_prev = c_prev;
    a.c_out = c_out;
    a.h_out = h_out;
    a.gates_out = gates_out;
    switch (plan->MT) {
        case 1: return launch_nt<3, 1, true>(a, (hipStream_t)s);
        case 2: return launch_nt<3, 2, true>(a, (hipStream_t)s);
        case 3: return launch_nt<3, 3, true>(a, (hipStream_t)s);
        case 4: return launch_nt<3, 4, true>(a, (hipStream_t)s);
    }
    return JAF_EINVAL;
}

// ---------------------------------------------------------------------------------------------
// direct convolution (one thread per output element): the in-library cross-check
// ---------------------------------------------------------------------------------------------
__global__ void conv_direct_kernel(const jaf_conv_desc d, const float* s0, const float* s1, const float* s2,
                                   const float* w, const float* bias, float* out) {
    const long total = (long)d.N * d.G * d.Cout * d.OH * d.OW;
    const float* srcs[3] = {s0, s1, s2};
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        long t = e;
        const int ox = (int)(t % d.OW); t /= d.OW;
        const int oy = (int)(t % d.OH); t /= d.OH;
        const int co = (int)(t % d.Cout); t /= d.Cout;
        const int g = (int)(t % d.G); t /= d.G;
        const int n = (int)t;
        float acc = 0.f;
        int cg = 0;
        for (int s = 0; s < d.nsrc; ++s) {
            for (int cl = 0; cl < d.src_c[s]; ++cl, ++cg) {
                const long ch = (long)n * d.src_ctot[s] + d.src_coff[s] + g * d.src_gstride[s] + cl;
                const float* wp = w + (((long)(g * d.Cout + co) * d.w_cin_tot) + d.w_cin_off + cg) * d.KH * d.KW;
                for (int ky = 0; ky < d.KH; ++ky) {
                    const int iyd = oy * d.stride - d.pad_t + ky;
                    if (iyd < 0) continue;
                    int iy = iyd;
                    if (d.dil_in == 2) { if (iyd & 1) continue; iy = iyd >> 1; }
                    if (iy >= d.H) continue;
                    for (int kx = 0; kx < d.KW; ++kx) {
                        const int ixd = ox * d.stride - d.pad_l + kx;
                        if (ixd < 0) continue;
                        int ix = ixd;
                        if (d.dil_in == 2) { if (ixd & 1) continue; ix = ixd >> 1; }
                        if (ix >= d.W) continue;
                        acc = fmaf(wp[ky * d.KW + kx], srcs[s][(ch * d.H + iy) * d.W + ix], acc);
                    }
                }
            }
        }
        if (bias) acc += bias[g * d.Cout + co];
        out[(((long)n * d.out_ctot + d.out_coff + g * d.Cout + co) * d.OH + oy) * d.OW + ox] = jaf_act(acc, d.act, d.slope);
    }
}

extern "C" int jaf_conv2d_fwd_direct(jaf_stream_t s, const jaf_conv_desc* d,
                                     const float* src0, const float* src1, const float* src2,
                                     const float* w, const float* bias, float* out) {
    JAF_REQUIRE(desc_ok(d) && src0 && w && out);
    JAF_REQUIRE(d->nsrc < 2 || src1);
    JAF_REQUIRE(d->nsrc < 3 || src2);
    JAF_REQUIRE(d->w_cin_off + d->Cin <= d->w_cin_tot);
    const long total = (long)d->N * d->G * d->Cout * d->OH * d->OW;
    hipLaunchKernelGGL(conv_direct_kernel, dim3(jaf_ew_grid(total)), dim3(256), 0, (hipStream_t)s, *d, src0, src1, src2, w, bias, out);
    return jaf_launch_status();
}

extern "C" int jaf_version(void) { return 100; }
