// Weight gradient of the convolution family on the fp32 matrix cores (v_mfma_f32_16x16x4_f32).
//
//   dW[co][(ci,ky,kx)] = sum over images n and output pixels p of  dz[n][co][p] * X[n][ci][p*s + tap]
//
// GEMM view: rows = output channels (A operand = dz tile in LDS), columns = the flattened
// (ci, ky, kx) index of the weight tensor (B operand = the forward input patch in LDS read at
// a per-lane (ci,tap) offset), reduction = pixels, 4 per MFMA.  A workgroup owns a
// [16*MTW rows] x [64*NPW columns] slice of dW for one group, walks a strided share of the
// (image, pixel-tile) list accumulating in registers, and finishes with fp32 atomics -- the
// column index is the memory order of the weight tensor, so each atomic wave-instruction
// covers 4 rows x 64 contiguous bytes.
#include "conv_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WG_P 128   // pixels per tile

struct WgradArgs {
    const float* src[3];
    const float* dz;
    float* dw;
    jaf_conv_desc d;
    int TWIN, tiles_x, tiles_p;
    int PH, PW, PWp, PS, DP;
    int CB;            // columns per block
    int ncolblocks, mblocks, nsplit;
    int sdz_off, spoff_off;
    float inv_pw, inv_phpw;
};

template <int MTW, int NPW>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* s_in = smem;
    float* s_dz = smem + a.sdz_off;
    int* s_poff = (int*)(smem + a.spoff_off);
    int* s_opix = s_poff + WG_P;

    const jaf_conv_desc& d = a.d;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 15;
    const int q = lane >> 4;
    const int KHW = d.KH * d.KW;
    constexpr int MRW = 16 * MTW;

    const int split = blockIdx.x;
    const int cb = blockIdx.y / a.mblocks;
    const int mb = blockIdx.y % a.mblocks;
    const int g = blockIdx.z;
    const int ncols = d.Cin * KHW;
    const int col0 = cb * a.CB;
    const int c_first = col0 / KHW;
    int c_last = (col0 + a.CB - 1) / KHW;
    if (c_last > d.Cin - 1) c_last = d.Cin - 1;
    const int nch = c_last - c_first + 1;
    const int OHW = d.OH * d.OW;
    const int PS = a.PS, PWp = a.PWp, PH = a.PH, PW = a.PW, DP = a.DP;

    int coloff[NPW];
    int colj[NPW];
#pragma unroll
    for (int np = 0; np < NPW; ++np) {
        const int j = col0 + (np * 4 + wave) * 16 + li;
        const bool valid = j < ncols;
        const int cg = valid ? j / KHW : c_first;
        const int tap = valid ? j - cg * KHW : 0;
        const int ky = tap / d.KW;
        const int kx = tap - ky * d.KW;
        coloff[np] = (cg - c_first) * PS + ky * PWp + kx;
        colj[np] = valid ? j : -1;
    }

    f32x4 acc[MTW][NPW];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int np = 0; np < NPW; ++np) acc[mt][np] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int c0 = d.src_c[0];
    const int c01 = d.src_c[0] + (d.nsrc > 1 ? d.src_c[1] : 0);
    const int tiles = a.tiles_x * a.tiles_p;
    const int items = d.N * tiles;
    const int phpw = PH * PW;
    const int patch_elems = nch * phpw;

    for (int item = split; item < items; item += a.nsplit) {
        const int n = item / tiles;
        const int tile = item - n * tiles;
        const int tx = tile % a.tiles_x;
        const int tb = tile / a.tiles_x;
        const int x0 = tx * a.TWIN;
        const int pbase = tb * WG_P;
        const int oy0 = pbase / a.TWIN;
        const int iy0 = oy0 * d.stride - d.pad_t;
        const int ix0 = x0 * d.stride - d.pad_l;

        __syncthreads();   // previous tile fully consumed
        if (tid < WG_P) {
            const int p = pbase + tid;
            const int oy = p / a.TWIN;
            const int ox = x0 + (p - oy * a.TWIN);
            const bool valid = (oy < d.OH) && (ox < d.OW);
            s_poff[tid] = valid ? ((oy - oy0) * d.stride * PWp + (ox - x0) * d.stride) : 0;
            s_opix[tid] = valid ? (oy * d.OW + ox) : -1;
        }
        // input patch (channels c_first .. c_last)
        for (int e = tid; e < patch_elems; e += 256) {
            int c = (int)(((float)e + 0.5f) * a.inv_phpw);
            int rem = e - c * phpw;
            int r = (int)(((float)rem + 0.5f) * a.inv_pw);
            int x = rem - r * PW;
            const int cg = c_first + c;
            const int iy = iy0 + r;
            const int ix = ix0 + x;
            float v = 0.f;
            if (iy >= 0 && ix >= 0 && iy < d.H && ix < d.W) {
                int s, cl;
                if (cg < c0) { s = 0; cl = cg; }
                else if (cg < c01) { s = 1; cl = cg - c0; }
                else { s = 2; cl = cg - c01; }
                const long ch = (long)n * d.src_ctot[s] + d.src_coff[s] + g * d.src_gstride[s] + cl;
                v = a.src[s][(ch * d.H + iy) * d.W + ix];
            }
            s_in[c * PS + r * PWp + x] = v;
        }
        __syncthreads();   // s_opix visible
        // dz tile: MRW rows x WG_P pixels
        for (int e = tid; e < MRW * WG_P; e += 256) {
            const int m = e / WG_P;
            const int p = e - m * WG_P;
            const int co = mb * MRW + m;
            const int op = s_opix[p];
            float v = 0.f;
            if (co < d.Cout && op >= 0)
                v = a.dz[((long)n * d.out_ctot + d.out_coff + g * d.Cout + co) * OHW + op];
            s_dz[m * DP + p] = v;
        }
        __syncthreads();
        for (int ks = 0; ks < WG_P / 4; ++ks) {
            const int pk = 4 * ks + q;
            const int pq = s_poff[pk];
            float av[MTW], bv[NPW];
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) av[mt] = s_dz[(mt * 16 + li) * DP + pk];
#pragma unroll
            for (int np = 0; np < NPW; ++np) bv[np] = s_in[coloff[np] + pq];
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
                for (int np = 0; np < NPW; ++np)
                    acc[mt][np] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mt], bv[np], acc[mt][np], 0, 0, 0);
        }
    }

    // D layout: column (lane&15) = weight column j, row (lane>>4)*4 + reg = output channel
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = mb * MRW + mt * 16 + q * 4 + r;
            if (co >= d.Cout) continue;
#pragma unroll
            for (int np = 0; np < NPW; ++np) {
                const int j = colj[np];
                if (j < 0) continue;
                const int cg = j / KHW;
                const int tap = j - cg * KHW;
                float* p = a.dw + (((long)(g * d.Cout + co) * d.w_cin_tot) + d.w_cin_off + cg) * KHW + tap;
                atomicAdd(p, acc[mt][np][r]);
            }
        }
    }
}

static int round16mod32(int v) { int r = v; while ((r & 31) != 16) ++r; return r; }

extern "C" int jaf_conv2d_wgrad(jaf_stream_t s_, const jaf_conv_desc* d,
                                const float* src0, const float* src1, const float* src2,
                                const float* dz, float* dw, int accumulate) {
    JAF_REQUIRE(d && src0 && dz && dw);
    JAF_REQUIRE(d->N >= 1 && d->G >= 1 && d->Cin >= 1 && d->Cout >= 1);
    JAF_REQUIRE(d->KH >= 1 && d->KW >= 1 && d->KH <= 7 && d->KW <= 7);
    JAF_REQUIRE(d->stride >= 1 && d->stride <= 2 && d->dil_in == 1);
    JAF_REQUIRE(d->nsrc >= 1 && d->nsrc <= 3);
    JAF_REQUIRE(d->nsrc < 2 || src1);
    JAF_REQUIRE(d->nsrc < 3 || src2);
    JAF_REQUIRE(d->w_cin_off >= 0 && d->w_cin_off + d->Cin <= d->w_cin_tot);
    JAF_REQUIRE(d->out_coff >= 0 && d->out_coff + d->G * d->Cout <= d->out_ctot);
    {
        int c = 0;
        for (int i = 0; i < d->nsrc; ++i) {
            JAF_REQUIRE(d->src_c[i] >= 1 && d->src_coff[i] >= 0 && d->src_gstride[i] >= 0);
            JAF_REQUIRE(d->src_coff[i] + (d->G - 1) * d->src_gstride[i] + d->src_c[i] <= d->src_ctot[i]);
            c += d->src_c[i];
        }
        JAF_REQUIRE(c == d->Cin);
    }
    hipStream_t s = (hipStream_t)s_;
    const int KHW = d->KH * d->KW;
    JAF_REQUIRE(d->precision >= JAF_PREC_F32 && d->precision <= JAF_PREC_BF16X3);
    // This fp32-input kernel computes in exact fp32 whatever d->precision says: in the bf16 modes it only serves the layers the
    // packed weight-gradient kernels do not cover (7 x 7 and 4 x 4 taps: the propagater's first / last layers, FlowNetSD).
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(dw, 0, sizeof(float) * (size_t)d->G * d->Cout * d->w_cin_tot * KHW, s);
        if (e != hipSuccess) return (int)e;
    }
    WgradArgs a;
    a.src[0] = src0; a.src[1] = src1; a.src[2] = src2;
    a.dz = dz; a.dw = dw; a.d = *d;
    // pixel tiling: 128 pixels, 16-wide windows for wide maps, linear walk otherwise
    const long OHW = (long)d->OH * d->OW;
    int rows_span;
    if (d->OW > 24) {
        a.TWIN = 16;
        rows_span = WG_P / 16;
        a.tiles_x = jaf_cdiv(d->OW, 16);
        a.tiles_p = jaf_cdiv(d->OH, rows_span);
    } else {
        a.TWIN = d->OW;
        rows_span = (WG_P % a.TWIN == 0) ? WG_P / a.TWIN : (WG_P + a.TWIN - 2) / a.TWIN + 1;
        if (rows_span > d->OH) rows_span = d->OH;
        a.tiles_x = 1;
        a.tiles_p = jaf_cdiv(OHW, WG_P);
    }
    a.PH = (rows_span - 1) * d->stride + d->KH;
    a.PW = (a.TWIN - 1) * d->stride + d->KW;
    a.PWp = a.PW;
    a.PS = round16mod32(a.PH * a.PWp);
    a.DP = WG_P + 2;
    a.inv_pw = 1.0f / (float)a.PW;
    a.inv_phpw = 1.0f / (float)(a.PH * a.PW);
    // rows per block
    int MTW = 1; long bestPad = 1L << 60;
    for (int mt = 4; mt >= 1; --mt) {
        long pad = (long)jaf_cdiv(d->Cout, 16 * mt) * 16 * mt;
        if (pad < bestPad) { bestPad = pad; MTW = mt; }
    }
    const int ncols = d->Cin * KHW;
    int NPW = ncols <= 64 ? 1 : (ncols <= 128 ? 2 : 3);
    a.CB = 64 * NPW;
    a.ncolblocks = jaf_cdiv(ncols, a.CB);
    a.mblocks = jaf_cdiv(d->Cout, 16 * MTW);
    const int nch_max = (a.CB + KHW - 2) / KHW + 1;
    a.sdz_off = nch_max * a.PS;
    a.spoff_off = a.sdz_off + 16 * MTW * a.DP;
    const size_t lds = ((size_t)a.spoff_off + 2 * WG_P) * 4;
    JAF_REQUIRE(lds <= 160 * 1024);
    const long items = (long)d->N * a.tiles_x * a.tiles_p;
    const long byz = (long)a.ncolblocks * a.mblocks * d->G;
    // fp32 tiles are small (one 16-column block per workgroup): more workgroup slots and up to 512 splits (96 / 256 /
    // 512 / 1024 splits measured 980 / 705 / 677 / 692 us over the two 7x7 layers of the propagater); deep layers
    // with a huge dW still want few splits (jaf_wgrad_nsplit)
    const long nsplit = jaf_wgrad_nsplit(items, byz, (long)d->G * d->Cout * d->Cin * KHW, 6e-6, 2048.0, 512);
    a.nsplit = (int)nsplit;
    dim3 grid((unsigned)nsplit, (unsigned)(a.ncolblocks * a.mblocks), (unsigned)d->G);
    JAF_REQUIRE((long)a.ncolblocks * a.mblocks <= 65535 && d->G <= 65535);
#define JAF_WG(MT_, NP_)                                                                            \
    do {                                                                                            \
        auto k = conv_wgrad_kernel<MT_, NP_>;                                                       \
        if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        JAF_NOTE_KERNEL("conv_wgrad_kernel<%d, %d>", MT_, NP_);                                     \
        hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);                                          \
    } while (0)
#define JAF_WG_NP(MT_)                          \
    switch (NPW) {                              \
        case 1: JAF_WG(MT_, 1); break;          \
        case 2: JAF_WG(MT_, 2); break;          \
        default: JAF_WG(MT_, 3); break;         \
    }
    switch (MTW) {
        case 1: JAF_WG_NP(1); break;
        case 2: JAF_WG_NP(2); break;
        case 3: JAF_WG_NP(3); break;
        default: JAF_WG_NP(4); break;
    }
#undef JAF_WG_NP
#undef JAF_WG
    return jaf_launch_status();
}

// ---------------------------------------------------------------------------------------------
// per-channel sum over (n, h, w): bias gradients.  grid (C, nsplit): a workgroup walks items =
// (image, 4096-element chunk of the plane) in strides of nsplit, 16 bytes per lane.
// ---------------------------------------------------------------------------------------------
#define CS_CHUNK 4096
template <int V>
__global__ void channel_sum_kernel(const float* x, int N, int ctot, int coff, int HW, float* out) {
    const int c = blockIdx.x;
    const int chunks = (HW + CS_CHUNK - 1) / CS_CHUNK;
    const int items = N * chunks;
    double acc = 0.0;
    for (int item = blockIdx.y; item < items; item += gridDim.y) {
        const int n = item / chunks;
        const int ch = item - n * chunks;
        const float* p = x + ((long)n * ctot + coff + c) * HW;
        const int lo = ch * CS_CHUNK;
        const int hi = lo + CS_CHUNK < HW ? lo + CS_CHUNK : HW;
        if (V == 4) {
            float part = 0.f;
            for (int i = lo + threadIdx.x * 4; i < hi; i += blockDim.x * 4) {
                const f32x4 v = *(const f32x4*)(p + i);
                part += (v[0] + v[1]) + (v[2] + v[3]);
            }
            acc += (double)part;      // at most 16 fp32 adds per lane before widening
        } else {
            for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) acc += (double)p[i];
        }
    }
    __shared__ double red[4];
    acc = jaf_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&out[c], (float)(red[0] + red[1] + red[2] + red[3]));
}

extern "C" int jaf_channel_sum(jaf_stream_t s_, const float* x, int32_t N, int32_t ctot, int32_t coff,
                               int32_t C, int32_t HW, float* out, int accumulate) {
    JAF_REQUIRE(x && out && N >= 1 && C >= 1 && HW >= 1 && coff >= 0 && coff + C <= ctot);
    hipStream_t s = (hipStream_t)s_;
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)C, s);
        if (e != hipSuccess) return (int)e;
    }
    const long items = (long)N * jaf_cdiv(HW, CS_CHUNK);
    long nsplit = (2048 + C - 1) / C;
    if (nsplit > items) nsplit = items;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > 65535) nsplit = 65535;
    if ((HW % 4 == 0) && (((uintptr_t)x) & 15) == 0)
        hipLaunchKernelGGL(channel_sum_kernel<4>, dim3(C, (unsigned)nsplit), dim3(256), 0, s, x, N, ctot, coff, HW, out);
    else
        hipLaunchKernelGGL(channel_sum_kernel<1>, dim3(C, (unsigned)nsplit), dim3(256), 0, s, x, N, ctot, coff, HW, out);
    return jaf_launch_status();
}
