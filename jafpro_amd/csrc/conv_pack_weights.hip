// Weight images of the bf16 matrix-core kernels (conv_dma.hip, conv_dma_split.hip): fp32 weights in the reference layout
// [G*Cout][Cin][KH][KW] -> bf16 A-fragment order [group][row block][chunk][hi, residual][k-step][MT][q][row][8 channels], so that
// staging a chunk is a linear DMA copy and an A fragment one conflict-free ds_read_b128 at lane*16.  The reduction index inside
// a chunk is the flattened SLOT s = tap*groups + group; one MFMA (K = 32) consumes 4 consecutive slots, the k-quarter
// q = lane>>4 taking slot 4*step+q.  JAF_PREC_BF16X3 keeps a second (residual, v - bf16(v)) image behind the first.
// (Until round 5 this file also held conv_bf16_kernel, the fp32-input staging form of the same arithmetic; every caller has
// used the packed-input kernels since round 3 and the kernel was removed.)
#include "conv_internal.h"

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

// The same image, tile by tile through LDS: a tile = 16 rows x one channel group of 8 x all taps of one (group, row block, chunk, MT
// slice).  Item by item (jafb_pack_item8) a lane reads 8 floats that lie `sch` apart and the other taps of the same cache lines belong
// to far-away threads: the re-packing of a module fetched every weight ~6 times (profiles/round5_g_pmc_hbm_traffic.txt: 230 MB per
// launch for ~38 MB of weights) and ran at a sixth of the HBM floor.  Here the tile's source floats are read ONCE along their memory
// order -- 8 x taps contiguous floats per row (forward / ConvLSTM images) or 16 x taps per channel (data-gradient images) -- and the
// 16-byte items leave in the image's order.  Writes every item jafb_pack_item8 writes (zeros for rows / channels / slots that do not
// exist), so the two kernels make identical images (tests/test_gpu_kernels.py::test_batched_weight_repack_equals_a_fresh_pack).
#define JAFB_TILE_FLOATS (16 * 8 * 49 + 16)  // 25 KB: one group of a 7 x 7 layer, or all four groups of a 3 x 3 chunk (+ an odd row pitch)
__device__ __forceinline__ void jafb_pack_tile(const PackBArgs& a, long tile, int GT, float* s_t /* [16][GT * 8][taps] */) {
    const int tid = threadIdx.x;
    const int gblocks = (a.NG + GT - 1) / GT;
    long t = tile;
    const int gb = (int)(t % gblocks); t /= gblocks;
    const int mt = (int)(t % a.MT); t /= a.MT;
    const int chunk = (int)(t % a.nchunks); t /= a.nchunks;
    const int mb = (int)(t % a.mblocks); t /= a.mblocks;
    const int g = (int)t;
    const int ngc = (chunk == a.nchunks - 1) ? a.ng_last : a.NG;
    const int grp0 = gb * GT;
    if (grp0 >= ngc) return;                             // (uniform)
    const int ng = ngc - grp0 < GT ? ngc - grp0 : GT;    // channel groups of this tile
    const int nch = 8 * ng;
    const int taps = a.taps;
    const int r0 = mb * 16 * a.MT + mt * 16;
    const int ch0 = (chunk * a.NG + grp0) * 8;
    const float* wg = a.w + a.base + g * a.sg;
    const int n = 16 * nch * taps;
    const int rs = (nch * taps) | 1;                     // row pitch in LDS, odd: the 16 rows of an item column fall into 16 banks
    __syncthreads();                                     // the previous tile's items have been read out of s_t
    const bool rowmajor = a.sch < a.srow;                // forward-like: a row's nch x taps floats are contiguous (channel stride = taps);
    const int span = rowmajor ? nch * taps : 16 * taps;  // data-gradient-like: for one channel the 16 rows x taps floats are
    // (index decoding with reciprocals: e < 6 272, exact for the + 0.5 form; two integer divisions per element were most of the loop)
    const float inv_span = 1.0f / (float)span, inv_taps = 1.0f / (float)taps;
    for (int e = tid; e < n; e += 256) {
        const int hi = (int)(((float)e + 0.5f) * inv_span), k = e - hi * span;
        const int lo = (int)(((float)k + 0.5f) * inv_taps), tp = k - lo * taps;
        const int row = rowmajor ? hi : lo, j = rowmajor ? lo : hi;
        const int r = r0 + row, ch = ch0 + j;
        float v = 0.f;
        if (r < a.M && ch < a.Cred) {
            const int srow = a.lstmC > 0 ? (r & 3) * a.lstmC + (r >> 2) : r;
            const int chs = a.redC > 0 ? (ch & 3) * a.redC + (ch >> 2) : ch;
            v = wg[(long)srow * a.srow + (long)chs * a.sch + tp];
        }
        s_t[row * rs + j * taps + tp] = v;
    }
    __syncthreads();
    typedef unsigned int pk_u32x4 __attribute__((ext_vector_type(4)));
    typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
    typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
    const long ibase = ((((long)g * a.mblocks + mb) * a.nchunks + chunk) * a.nimg) * a.nsteps;
    // items (tap, group of this tile, row), plus (first group block) the chunk's padding slots past taps x groups
    const int nslot = taps * ng;
    const float inv_ng = 1.0f / (float)ng;
    const int npad = gb == 0 ? 4 * a.nsteps - taps * ngc : 0;
    for (int e = tid; e < 16 * (nslot + npad); e += 256) {
        const int row = e & 15, si = e >> 4;
        const bool pad = si >= nslot;
        const int ti = pad ? 0 : (int)(((float)si + 0.5f) * inv_ng), gi = pad ? 0 : si - ti * ng;
        const int s = pad ? taps * ngc + (si - nslot) : ti * ngc + grp0 + gi;
        const int st = s >> 2, q = s & 3;
        float v[8];
        const int stap = a.flip ? (taps - 1 - ti) : ti;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = pad ? 0.f : s_t[row * rs + (gi * 8 + j) * taps + stap];
        for (int img = 0; img < a.nimg; ++img) {
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
            const long item = ((((ibase + (long)img * a.nsteps + st) * a.MT + mt) * 4 + q) << 4) + row;
            *(pk_u32x4*)(a.out + item * 8) = w;
        }
    }
}

// Many weight images in ONE launch (the re-packing after an optimiser step: ~40 images per module): blockIdx.y picks
// the image's argument block out of a device-resident table.
__global__ __launch_bounds__(256) void conv_pack_bf16_batch_kernel(const PackBArgs* __restrict__ table) {
    __shared__ float s_t[JAFB_TILE_FLOATS];
    const PackBArgs a = table[blockIdx.y];
    int GT = (JAFB_TILE_FLOATS - 16) / (16 * 8 * a.taps);       // channel groups per tile
    if (GT < 1) {                                        // (no such layer; kept correct)
        for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < (a.total >> 3); e += (long)gridDim.x * blockDim.x) jafb_pack_item8(a, e);
        return;
    }
    if (GT > a.NG) GT = a.NG;
    const long ntiles = (long)a.G * a.mblocks * a.nchunks * a.MT * ((a.NG + GT - 1) / GT);
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) jafb_pack_tile(a, tile, GT, s_t);
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

