// Weight gradient of 3x3 convolutions from PACKED bf16 operands (csrc/conv_dma.hip layout
// [n][g][group8][y][x][8 channels]): the kernel of wgrad_bf16.hip with both tiles staged by DMA.
//
//   dW[co][ci][tap] = sum over images n and output pixels p of  dz[n][co][p] * X[n][ci][p*s + tap]
//
// * the forward input's packed image is the one the forward convolution already made, the packed dz
//   is the one the data gradient already made: no conversion work is left in this kernel;
// * input patch in LDS: [ci tile][position][16 channels] = two packed 16-byte items side by side per
//   position; dz tile in LDS: [co tile][pixel][16 channels].  Both are consumed with transposed
//   reads (ds_read_b64_tr_b16): 8 consecutive PIXELS of one channel per lane for A and B alike;
// * bit 7 of an LDS byte address is XORed with bit 3 of the patch column (bit 3 of the pixel index
//   for dz) so that the two 16-lane groups of a half-wave, 8 rows apart, hit different banks.  The
//   DMA destination is lane-linear, so the swizzle is applied to the SOURCE: lane -> slot ->
//   unswizzled (position, half) -> global offset (cdna_hip_programming.md rule 21);
// * out-of-image positions / pixels get an out-of-range buffer offset and arrive as zeros.
#include "conv_internal.h"
#include "jaf_fdiv.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4;

#define WD_TW 16
#define WD_TH 8
#define WD_XI 11        // patch DMA instructions per wave per tile (upper bound: a stride-2 3 x 3 patch of 32 channels = 44 KB)
#define WD_EP 145
#define WD_OOB 0x7ffffff0

struct WgDArgs {
    const unsigned char* xp;
    const unsigned char* dzp;
    float* dw;
    jaf_conv_desc d;
    int WC, WK;
    int coblocks, ciblocks, nsplit;
    int tiles_x, tiles_y;
    int PH, PW, PWp;
    int xplane;            // bytes of one ci-tile plane: PH*PWp*32
    int nx;                // DMA instructions per ci-tile plane
    int off_dz;
    int bufsz;             // DB: bytes of one (patch, dz) tile buffer; the second buffer follows the first
    int ngin8, ngout8;
    int xng8;              // planes per (image, group) of the packed input image (>= ngin8)
    float inv_pwp;
    int lstmC;             // > 0: dz channels are the ConvLSTM's gate gradients, channel-major (4 c + gate): dW row gate * lstmC + c
    int step_n, step_ty, step_tx;      // conv_wgrad_fast_kernel: nsplit tiles further = (images, tile rows, tile columns)
    float* ws;             // conv_wgrad_fast_kernel: split-K partials [nsplit][dW layout] (plain stores) instead of atomics into dw; NULL: atomics
    long ws_stride;        // floats of one partial = G * Cout * w_cin_tot * 9
    int off_lo;            // SPLIT: byte offset of the lo tiles from the hi tiles inside a tile buffer (patch and dz alike)
    jaf_fdiv dv_tiles, dv_tiles_x;     // conv_wgrad_dma_kernel: tile index -> (image, tile row, tile column) without integer divisions
    int ky0, kyn;          // 7 x 7: the kernel rows [ky0, ky0 + kyn) this launch accumulates (49 accumulator tiles do not fit the
                           // register file: two launches of 4 + 3 rows); other sizes: 0, KS
    int xmul;              // !SPLIT: 2 when the x image is a split-bf16 image of which only the hi planes are read ("mixed" mode:
                           // forward in split-bf16, backward in bf16 -- a hi plane IS the bf16 image), else 1
};

// PAIR (Cin <= 8, i.e. one packed item per position): the 16 columns of an MFMA are TWO taps x 8 channels instead of one
// tap x 16 channels of which 8 are padding -- lanes p = 2, 3 of the transposed read point at the next tap's position --
// so a 5x5 layer issues 13 MFMAs per k-step instead of 25 and keeps 13 accumulators.
//
// DB (tiles of at most 40 KB, i.e. the 24-part networks and the other narrow layers, which are bound by HBM and not by the
// matrix cores): two tile buffers.  The DMA of tile i+1 is issued right after the barrier that publishes tile i and runs
// under tile i's matrix-core pass, so a workgroup keeps two tiles' worth of bytes in flight instead of alternating between
// "waiting for a tile" and "multiplying it"; one barrier per tile instead of two.  The wide layers (46 KB tiles) stay
// single-buffered: a second buffer would halve their residency (2 -> 1 workgroups per CU), which measured slower.
// XI: patch DMA instructions per wave and tile this instantiation provides for (4: stride 1 with up to 32 input channels per
// workgroup, 8: stride 1 with 64, 11: stride 2).  Their per-lane constants (x_rc, x_goff) stay live across the whole loop: at
// 11 for everybody the 48-row kernel took 175-181 registers (2 workgroups per CU), at 4 it fits 3.
// SPLIT (d.precision == JAF_PREC_BF16X3): both operands are split-bf16 images (hi plane of channel group cg at 2 cg, lo plane
// at 2 cg + 1: jaf_conv2d_pack_input / jaf_conv2d_pack_dz_prec), the tile buffer holds [x hi][dz hi][x lo][dz lo], and a product is
// dz_h x_h + dz_l x_h + dz_h x_l: three matrix-core instructions per fragment pair, fp32-grade gradients.
template <int MTW, int KS, bool PAIR, bool DB, int XI, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void conv_wgrad_dma_kernel(const WgDArgs a) {
    constexpr int NTAP = KS * KS;
    constexpr int KYN = KS == 7 ? 4 : KS;            // kernel rows per launch (WgDArgs.ky0 / kyn)
    constexpr int NACC = PAIR ? (NTAP + 1) / 2 : KYN * KS;
    static_assert(!(PAIR && KS == 7), "the paired form keeps all taps in one launch");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const jaf_conv_desc& d = a.d;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15;
    const int q = lane >> 4;
    const int WC = a.WC, WK = a.WK;
    const int wc = wave % WC;
    const int wk = wave / WC;
    const int s = d.stride;
    const int PWp = a.PWp;

    unsigned char* s_x = smem;
    unsigned char* s_dz = smem + a.off_dz;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;   // LDS byte address of smem

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

    // ---- DMA lane constants.  Patch: instruction i = wave + 4*j covers slots [64*(i % nx), +64) of ci
    // tile i / nx; a slot is 16 bytes: (position, half) after undoing the bit-7 swizzle. ----
    const int nxi = WC * a.nx;                 // patch DMA instructions per tile
    int x_rc[WD_XI], x_goff[WD_XI];        // (only the first XI entries are used)
#pragma unroll
    for (int j = 0; j < XI; ++j) {
        const int i = wave + 4 * j;
        const int tci = i / a.nx;
        const int slot = (i - tci * a.nx) * 64 + lane;
        const int addr = slot * 16;
        const int posq = addr >> 5;
        const int rq = (int)(((float)posq + 0.5f) * a.inv_pwp);
        const int cq = posq - rq * PWp;
        const int raw = addr ^ ((cq & 8) << 4);
        const int pos = raw >> 5;
        const int half = (raw >> 4) & 1;
        const int r = (int)(((float)pos + 0.5f) * a.inv_pwp);
        const int c = pos - r * PWp;
        const int grp8 = (ci0 >> 3) + 2 * tci + half;
        const bool live = (i < nxi) && (r < a.PH) && (c < a.PW) && (grp8 < a.ngin8);
        x_rc[j] = live ? ((r << 16) | c) : -1;
        x_goff[j] = ((SPLIT ? 2 : a.xmul) * grp8 * HW + r * d.W + c) * 16;
    }
    // dz: instruction (co tile j, chunk = wave) covers slots [64*wave, +64) of co tile j (256 slots)
    int z_yx, z_goff, z_half;
    {
        const int addr = (wave * 64 + lane) * 16;
        const int kq = addr >> 5;
        const int raw = addr ^ ((kq & 8) << 4);
        const int k = raw >> 5;
        z_half = (raw >> 4) & 1;
        const int y = k >> 4, x = k & 15;
        z_yx = (y << 16) | x;
        z_goff = ((SPLIT ? 2 * z_half : z_half) * OHW + y * d.OW + x) * 16;
    }

    // ---- per-lane constants of the transposed reads: lane = 16q + 4q' + p supplies the address of
    // row (pixel) 8q + 4h + q', channels 4p..4p+3 ----
    const int qp = (lane >> 2) & 3;
    const int pp = lane & 3;
    int bbase[2][KS], bswz[2][KS], abase[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c0h = (8 * (q & 1) + 4 * h + qp) * s;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
            bbase[h][kx] = ((q >> 1) * s * PWp + c0h + kx) * 32 + pp * 8 + wc * a.xplane;
            bswz[h][kx] = ((c0h + kx) & 8) << 4;
        }
        const int k = 8 * q + 4 * h + qp;                 // pixel within a 32-pixel k-step
        abase[h] = ((k * 32) ^ ((k & 8) << 4)) + pp * 8;
    }

    // PAIR (one input-channel tile per workgroup, so WK == 4 and a wave's k-step is always ks = wk): the 2 x NACC patch addresses of
    // the transposed reads do not depend on the tile.  They were recomputed for every tile (146 VALU instructions per 13 MFMAs,
    // profiles/experiments/round3_pmc_enc1.txt); now two 16-bit offsets per register, made once.
    unsigned padr[NACC];
    if constexpr (PAIR) {
        const int hb = pp >> 1;                  // which tap of the pair this lane's address belongs to
#pragma unroll
        for (int pr = 0; pr < NACC; ++pr) {
            const int tA = 2 * pr, tB = (2 * pr + 1 < NTAP) ? 2 * pr + 1 : 2 * pr;
            const int ky = hb ? tB / KS : tA / KS, kx = hb ? tB % KS : tA % KS;
            unsigned v = 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c0h = (8 * (q & 1) + 4 * h + qp) * s + kx;
                const int ad = ((((q >> 1) + 2 * wk) * s + ky) * PWp + c0h) * 32 + (pp & 1) * 8;
                v |= (unsigned)(ad ^ ((c0h & 8) << 4)) << (16 * h);
            }
            padr[pr] = v;
        }
    }

    f32x4 acc[MTW][NACC];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int t = 0; t < NACC; ++t) acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int tiles = a.tiles_x * a.tiles_y;
    const int items = d.N * tiles;
    const int xbytes = a.xng8 * HW * 16 * (SPLIT ? 2 : a.xmul);
    const int zbytes = a.ngout8 * OHW * 16 * (SPLIT ? 2 : 1);

    // DMA of one tile (image, pixel tile) into the buffer at byte offset `boff`
    auto issue = [&](int item, int boff) {
        const int n = (int)jaf_fdiv_q((unsigned)item, a.dv_tiles);
        const int tile = item - n * tiles;
        const int ty = (int)jaf_fdiv_q((unsigned)tile, a.dv_tiles_x);
        const int tx = tile - ty * a.tiles_x;
        const int oy0 = ty * WD_TH, ox0 = tx * WD_TW;
        const int iy0 = oy0 * s - d.pad_t;
        const int ix0 = ox0 * s - d.pad_l;
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.xp + ((long)n * d.G + g) * (long)xbytes), 0, xbytes, 0x00020000);
        const jaf_u32x4 rxa = jaf_make_rsrc(a.xp + ((long)n * d.G + g) * (long)xbytes, (unsigned)xbytes);
        const int tbase = (iy0 * d.W + ix0) * 16;
        // (uniform) the whole patch and the whole pixel tile lie inside the image: no per-lane bounds logic -- a dead slot's offset is the
        // out-of-range constant itself and stays out of range when the tile base is added (32-bit unsigned buffer offsets)
        const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + a.PH <= d.H && ix0 + a.PW <= d.W && oy0 + WD_TH <= d.OH && ox0 + WD_TW <= d.OW;
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            const int i = wave + 4 * j;
            if (i < nxi) {
                bool ok = x_rc[j] >= 0;
                if (!interior) {
                    const int r = x_rc[j] >> 16, c = x_rc[j] & 0xffff;
                    const int iy = iy0 + r, ix = ix0 + c;
                    ok = ok && (iy >= 0) && (ix >= 0) && (iy < d.H) && (ix < d.W);
                }
                const int tci = i / a.nx;
#pragma unroll
                for (int h = 0; h < (SPLIT ? 2 : 1); ++h) {
                    const int src = ok ? x_goff[j] + tbase + h * HW * 16 : WD_OOB;
                    if (DB)      // (from assembly: see jaf_dma16_async)
                        jaf_dma16_async(rxa, lds0 + boff + h * a.off_lo + tci * a.xplane + (i - tci * a.nx) * 1024, src);
                    else
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(
                            rx, (__attribute__((address_space(3))) void*)(s_x + boff + h * a.off_lo + tci * a.xplane + (i - tci * a.nx) * 1024), 16,
                            src, 0, 0, 0);
                }
            }
        }
        const __amdgpu_buffer_rsrc_t rz = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(a.dzp + ((long)n * d.G + g) * (long)zbytes), 0, zbytes, 0x00020000);
        const jaf_u32x4 rza = jaf_make_rsrc(a.dzp + ((long)n * d.G + g) * (long)zbytes, (unsigned)zbytes);
        const int y = z_yx >> 16, x = z_yx & 0xffff;
        const bool okp = interior || ((oy0 + y < d.OH) && (ox0 + x < d.OW));
        const int zb = z_goff + (oy0 * d.OW + ox0) * 16;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            const int grp8 = (co0 >> 3) + 2 * mt + z_half;
            const bool ok = okp && (grp8 < a.ngout8);
#pragma unroll
            for (int h = 0; h < (SPLIT ? 2 : 1); ++h) {
                const int src = ok ? zb + ((co0 >> 3) + 2 * mt) * (SPLIT ? 2 : 1) * OHW * 16 + h * OHW * 16 : WD_OOB;
                if (DB)
                    jaf_dma16_async(rza, lds0 + a.off_dz + boff + h * a.off_lo + mt * 4096 + wave * 1024, src);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(
                        rz, (__attribute__((address_space(3))) void*)(s_dz + boff + h * a.off_lo + mt * 4096 + wave * 1024), 16, src, 0, 0, 0);
            }
        }
    };

    int cur = 0;                 // DB: byte offset of the buffer that holds the tile being multiplied
    if (DB && split < items) issue(split, 0);
    for (int item = split; item < items; item += a.nsplit) {
        if (DB) {
            __builtin_amdgcn_s_waitcnt(0);      // this wave's share of tile `item` has landed
            __syncthreads();                    // ... everybody's has, and everybody is done with the other buffer
            if (item + a.nsplit < items) issue(item + a.nsplit, cur ^ a.bufsz);
        } else {
            __syncthreads();   // previous tile consumed
            issue(item, 0);
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
        }
        const unsigned char* c_x = s_x + cur;
        const unsigned char* c_dz = s_dz + cur;
        if (DB) cur ^= a.bufsz;

        for (int ks = wk; ks < 4; ks += WK) {
            bf16x8 af[MTW], al[SPLIT ? MTW : 1];
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                const unsigned char* ap = c_dz + mt * 4096 + ks * 1024;
                const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ap + abase[0]));
                const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ap + abase[1]));
                af[mt] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
                if constexpr (SPLIT) {
                    const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ap + a.off_lo + abase[0]));
                    const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ap + a.off_lo + abase[1]));
                    al[mt] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
                }
            }
            if constexpr (PAIR) {
#pragma unroll
                for (int pr = 0; pr < NACC; ++pr) {
                    s16x4 bb[2];
                    bb[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(c_x + (padr[pr] & 0xffffu)));
                    bb[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(c_x + (padr[pr] >> 16)));
                    const bf16x8 bf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(bb[0], bb[1], 0, 1, 2, 3, 4, 5, 6, 7));
                    if constexpr (SPLIT) {
                        const s16x4 c0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(c_x + a.off_lo + (padr[pr] & 0xffffu)));
                        const s16x4 c1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(c_x + a.off_lo + (padr[pr] >> 16)));
                        const bf16x8 bl = __builtin_bit_cast(bf16x8, __builtin_shufflevector(c0, c1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                        for (int mt = 0; mt < MTW; ++mt) {
                            acc[mt][pr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mt], bf, acc[mt][pr], 0, 0, 0);
                            acc[mt][pr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bl, acc[mt][pr], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt)
                        acc[mt][pr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bf, acc[mt][pr], 0, 0, 0);
                }
            } else
#pragma unroll
            for (int ky = 0; ky < KYN; ++ky) {
                if constexpr (KS == 7) { if (ky >= a.kyn) break; }
                const int rowoff = ((2 * ks * s + ky + (KS == 7 ? a.ky0 : 0)) * PWp) * 32;
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const int a0 = (bbase[0][kx] + rowoff) ^ bswz[0][kx];
                    const int a1 = (bbase[1][kx] + rowoff) ^ bswz[1][kx];
                    const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(c_x + a0));
                    const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(c_x + a1));
                    const bf16x8 bf = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
                    if constexpr (SPLIT) {
                        const s16x4 c0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(c_x + a.off_lo + a0));
                        const s16x4 c1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(c_x + a.off_lo + a1));
                        const bf16x8 bl = __builtin_bit_cast(bf16x8, __builtin_shufflevector(c0, c1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
                        for (int mt = 0; mt < MTW; ++mt) {
                            acc[mt][ky * KS + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mt], bf, acc[mt][ky * KS + kx], 0, 0, 0);
                            acc[mt][ky * KS + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bl, acc[mt][ky * KS + kx], 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt)
                        acc[mt][ky * KS + kx] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bf, acc[mt][ky * KS + kx], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue.  3x3: transpose through LDS, then atomics along dW's memory order.  5x5 / 7x7 (a few
    // tiny first/last layers): the gradient tensor is a few KB, direct atomics. ----
    const int cit = ci0 + wc * 16;
    if constexpr (KS == 3) {
        float* s_ep = (float*)smem + wave * (16 * WD_EP);
        const int nrem = (d.Cin - cit) * 9;
        __syncthreads();        // every wave is done with the tile buffers; from here on a wave only touches its own staging rows
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            // (same-wave LDS traffic is ordered: no barrier between the staging writes, the row reads and the next tile's writes)
#pragma unroll
            for (int t = 0; t < NTAP; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) s_ep[(q * 4 + j) * WD_EP + li * 9 + t] = acc[mt][t][j];
            // one dW row (16 input channels x 9 taps = 144 consecutive floats) at a time: the row's base address is wave-uniform
            // (scalar arithmetic), a lane adds rem = lane, lane + 64, lane + 128.  The flat e = lane + 64 i form cost a division
            // and a 64-bit address per element: ~40 VALU instructions per atomic, half of the kernel's VALU work on short launches
            // (profiles/round3_k_pmc_instmix.txt).
            for (int row = 0; row < 16; ++row) {
                const int co = co0 + mt * 16 + row;
                if (co >= d.Cout) break;
                const int cod = a.lstmC > 0 ? (co & 3) * a.lstmC + (co >> 2) : co;
                float* prow = a.dw + (((long)(g * d.Cout + cod) * d.w_cin_tot) + d.w_cin_off + cit) * 9;
                const float* srow = s_ep + row * WD_EP;
#pragma unroll
                for (int rem = lane; rem < 144; rem += 64)
                    if (rem < nrem) atomicAdd(prow + rem, srow[rem]);
            }
        }
    } else if constexpr (PAIR) {
        // enc_0 (5 x 5, 3 -> 12 channels per part): dW is 21 600 floats and 2 304 workgroups add to all of it; straight from the
        // accumulators a lane's taps are 100 bytes from its neighbour's (every lane its own cache line, 8.6 M lane-atomics per launch
        // at ~40 G/s = 0.2 ms of a kernel that nothing overlaps).  As for 7 x 7: the four waves' k-step partials are summed in LDS and
        // the workgroup walks its (row, channel, tap) block in memory order -- a quarter of the atomics, 25-float runs.
        static_assert(MTW == 1, "one 16-row tile per workgroup");
        const int ci = li & 7, hb = li >> 3;
        float* s_ep = (float*)smem;
        const int nrow = d.Cout - co0 < 16 ? d.Cout - co0 : 16;
        const int per = nrow * 8 * NTAP;
        __syncthreads();            // every wave is done with the tile buffers
        for (int e = tid; e < per; e += 256) s_ep[e] = 0.f;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (q * 4 + j < nrow && ci < d.Cin) {
#pragma unroll
                for (int pr = 0; pr < NACC; ++pr) {
                    const int t = 2 * pr + hb;
                    if (t < NTAP) atomicAdd(&s_ep[((q * 4 + j) * 8 + ci) * NTAP + t], acc[0][pr][j]);
                }
            }
        __syncthreads();
        for (int e = tid; e < per; e += 256) {
            const int row = e / (8 * NTAP), r1 = e - row * (8 * NTAP);
            const int cil = r1 / NTAP, t = r1 - cil * NTAP;
            if (cil < d.Cin)
                atomicAdd(a.dw + (((long)(g * d.Cout + co0 + row) * d.w_cin_tot) + d.w_cin_off + cil) * NTAP + t, s_ep[e]);
        }
    } else if constexpr (KS == 7) {
        // 7 x 7 (the propagater's first and last layers: dW is a few thousand floats, every workgroup adds to all of it).  A lane's
        // taps of one (output channel, input channel) pair are 196 bytes apart from its neighbour's: issued straight from the
        // accumulators, every lane of an atomic instruction hit its own cache line (0.38 ms per launch, ~40 G lane-atomics per
        // second).  The waves' k-step partials are summed in LDS first ([tile][row][channel][tap], ds_add_f32) and the
        // workgroup then walks dW in memory order: a quarter of the atomics, 16 lanes per cache line.
        static_assert(MTW == 1, "one 16-row tile per workgroup");
        // (only the rows that exist: the propagater's last layer has ONE output channel, its first layer 9 input channels)
        float* s_ep = (float*)smem;
        const int nrow = d.Cout - co0 < 16 ? d.Cout - co0 : 16;
        const int per_tile = nrow * 16 * NACC;
        __syncthreads();            // every wave is done with the tile buffers
        for (int e = tid; e < WC * per_tile; e += 256) s_ep[e] = 0.f;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (q * 4 + j < nrow && cit + li < d.Cin) {
#pragma unroll
                for (int t = 0; t < NACC; ++t) atomicAdd(&s_ep[((wc * nrow + q * 4 + j) * 16 + li) * NACC + t], acc[0][t][j]);
            }
        __syncthreads();
        const int ntap = a.kyn * KS;
        for (int e = tid; e < WC * per_tile; e += 256) {
            const int tci = e >= per_tile ? 1 : 0, r0 = e - tci * per_tile;            // (WC <= 2 for 7 x 7: wgd_core)
            const int row = r0 / (16 * NACC), r1 = r0 - row * (16 * NACC);
            const int cil = r1 / NACC, t = r1 - cil * NACC;
            const int co = co0 + row, ci = ci0 + tci * 16 + cil;
            if (ci < d.Cin && t < ntap)
                atomicAdd(a.dw + (((long)(g * d.Cout + co) * d.w_cin_tot) + d.w_cin_off + ci) * NTAP + a.ky0 * KS + t, s_ep[e]);
        }
    } else {
        const int ci = cit + li;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int co = co0 + mt * 16 + q * 4 + j;
                if (co < d.Cout && ci < d.Cin) {
                    float* p = a.dw + (((long)(g * d.Cout + co) * d.w_cin_tot) + d.w_cin_off + ci) * NTAP;
#pragma unroll
                    for (int t = 0; t < NTAP; ++t) atomicAdd(p + t, acc[mt][t][j]);
                }
            }
        }
    }
}

// The stride-1 3 x 3 layers on their own kernel (WC = 4 / 2 / 1 input-channel tiles of 16 per workgroup; a wave owns tile wave % WC
// and walks the k-steps wave / WC + (4 / WC) i).  Same tiles, same LDS images, same arithmetic and summation order as conv_wgrad_dma_kernel; what
// differs is the instruction count around the matrix-core steps (profiles/round3_k_pmc_instmix.txt: 3.6 VALU and 1.4 SALU
// instructions per MFMA, the SIMD's issue slots about as busy with them as with the MFMAs):
//  * stride, patch pitch and the channel tiling are compile-time constants, so every transposed LDS read is one of 8 per-lane bases
//    plus an immediate offset (the bit-7 swizzle commutes with the row offset, a multiple of 256): no address arithmetic in the loop;
//  * B fragments are requested three taps ahead of the MFMAs that consume them;
//  * the tile walk is incremental (no divisions per tile), and a tile whose patch lies inside the image skips the per-piece bounds
//    logic: one add per DMA piece.
template <int MTW, bool DB, int WC, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void conv_wgrad_fast_kernel(const WgDArgs a) {
    constexpr int NW = 4, WK = 4 / WC, XI = 2 * WC, PWP = 24, PH = 10, PW = 18, XPLANE = 8192;
    constexpr int LA = ((MTW == 4 && DB) || SPLIT) ? 2 : 3;      // taps the B fragments run ahead (registers: 64 rows double-buffered spill at 3)
    typedef __attribute__((address_space(3))) unsigned char* ldsb;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const jaf_conv_desc& d = a.d;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15;
    const int q = lane >> 4;
    const int wc = wave % WC, wk = wave / WC;
    const ldsb lb = (ldsb)smem;
    const unsigned lds0 = (unsigned)(uintptr_t)lb;

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

    // ---- DMA lane constants (as in conv_wgrad_dma_kernel; a dead slot carries the out-of-range offset itself) ----
    // (row, column) of the patch position behind slot (piece i, lane): needed again only by tiles that cross the image border
    auto slot_rc = [&](int i, int& r, int& c, int& hf) {
        const int addr = ((i & 7) * 64 + lane) * 16;
        const int posq = addr >> 5;
        const int rq = (int)(((float)posq + 0.5f) * (1.0f / PWP));
        const int cq = posq - rq * PWP;
        const int raw = addr ^ ((cq & 8) << 4);
        const int pos = raw >> 5;
        hf = (raw >> 4) & 1;
        r = (int)(((float)pos + 0.5f) * (1.0f / PWP));
        c = pos - r * PWP;
    };
    int x_goff[XI];
#pragma unroll
    for (int j = 0; j < XI; ++j) {
        const int i = wave + NW * j;
        int r, c, hf;
        slot_rc(i, r, c, hf);
        const int grp8 = (ci0 >> 3) + 2 * (i >> 3) + hf;
        const bool live = (r < PH) && (c < PW) && (grp8 < a.ngin8);
        x_goff[j] = live ? ((SPLIT ? 2 : a.xmul) * grp8 * HW + r * d.W + c) * 16 : WD_OOB;
    }
    int z_yx, z_goff, z_half;
    {
        const int addr = (wave * 64 + lane) * 16;
        const int kq = addr >> 5;
        const int raw = addr ^ ((kq & 8) << 4);
        const int k = raw >> 5;
        z_half = (raw >> 4) & 1;
        const int y = k >> 4, x = k & 15;
        z_yx = (y << 16) | x;
        z_goff = ((SPLIT ? 2 * z_half : z_half) * OHW + y * d.OW + x) * 16;
    }

    // ---- per-lane bases of the transposed reads (byte offsets inside a tile buffer) ----
    const int qp = (lane >> 2) & 3;
    const int pp = lane & 3;
    unsigned bx[2][3], ax[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int c0h = 8 * (q & 1) + 4 * h + qp;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int col = c0h + kx;
            bx[h][kx] = (unsigned)(((((q >> 1) * PWP + col) * 32 + pp * 8) ^ ((col & 8) << 4)) + wc * XPLANE + wk * (2 * PWP * 32));
        }
        const int k = 8 * q + 4 * h + qp;
        ax[h] = (unsigned)((((k * 32) ^ ((k & 8) << 4)) + pp * 8) + a.off_dz + wk * 1024);
    }

    f32x4 acc[MTW][9];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[mt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int tiles = a.tiles_x * a.tiles_y;
    const int items = d.N * tiles;
    const int xbytes = a.xng8 * HW * 16 * (SPLIT ? 2 : a.xmul);
    const int zbytes = a.ngout8 * OHW * 16 * (SPLIT ? 2 : 1);
    const int count = split < items ? (items - split + a.nsplit - 1) / a.nsplit : 0;       // tiles of this pixel split

    // the next tile to fetch: (image, tile row, tile column), advanced by nsplit tiles without divisions
    int in_n = split / tiles, in_ty, in_tx;
    {
        const int t0 = split - in_n * tiles;
        in_ty = t0 / a.tiles_x;
        in_tx = t0 - in_ty * a.tiles_x;
    }
    auto issue = [&](int boff) {
        const int oy0 = in_ty * WD_TH, ox0 = in_tx * WD_TW;
        const int iy0 = oy0 - d.pad_t, ix0 = ox0 - d.pad_l;
        const jaf_u32x4 rxa = jaf_make_rsrc(a.xp + ((long)in_n * d.G + g) * (long)xbytes, (unsigned)xbytes);
        const int tbase = (iy0 * d.W + ix0) * 16;
        const unsigned dstx = lds0 + boff + wave * 1024;
        if (iy0 >= 0 && ix0 >= 0 && iy0 + PH <= d.H && ix0 + PW <= d.W) {     // (uniform) the whole patch is inside the image
#pragma unroll
            for (int j = 0; j < XI; ++j) {
                // (unsigned: a dead slot's out-of-range offset plus the tile base passes 2^31; the buffer offset is a 32-bit unsigned quantity)
                const unsigned src = (unsigned)x_goff[j] + (unsigned)tbase;
                jaf_dma16_async(rxa, dstx + NW * j * 1024, (int)src);
                if constexpr (SPLIT) jaf_dma16_async(rxa, dstx + a.off_lo + NW * j * 1024, (int)(src + (unsigned)(HW * 16)));      // (the residual plane follows the hi plane)
            }
        } else {
#pragma unroll
            for (int j = 0; j < XI; ++j) {
                int r, c, hf;
                slot_rc(wave + NW * j, r, c, hf);
                const int iy = iy0 + r, ix = ix0 + c;
                const bool ok = (x_goff[j] != WD_OOB) && (iy >= 0) && (ix >= 0) && (iy < d.H) && (ix < d.W);
                jaf_dma16_async(rxa, dstx + NW * j * 1024, ok ? x_goff[j] + tbase : WD_OOB);
                if constexpr (SPLIT) jaf_dma16_async(rxa, dstx + a.off_lo + NW * j * 1024, ok ? x_goff[j] + tbase + HW * 16 : WD_OOB);
            }
        }
        const jaf_u32x4 rza = jaf_make_rsrc(a.dzp + ((long)in_n * d.G + g) * (long)zbytes, (unsigned)zbytes);
        const bool okp = (oy0 + (z_yx >> 16) < d.OH) && (ox0 + (z_yx & 0xffff) < d.OW);
        const int zb = z_goff + (oy0 * d.OW + ox0) * 16 + (co0 >> 3) * (SPLIT ? 2 : 1) * OHW * 16;
        const unsigned dstz = lds0 + a.off_dz + boff + wave * 1024;
#pragma unroll
        for (int mt = 0; mt < MTW; ++mt) {
            const bool ok = okp && ((co0 >> 3) + 2 * mt + z_half < a.ngout8);
            jaf_dma16_async(rza, dstz + mt * 4096, ok ? zb + 2 * mt * (SPLIT ? 2 : 1) * OHW * 16 : WD_OOB);
            if constexpr (SPLIT) jaf_dma16_async(rza, dstz + a.off_lo + mt * 4096, ok ? zb + 4 * mt * OHW * 16 + OHW * 16 : WD_OOB);
        }
        in_n += a.step_n;
        in_tx += a.step_tx;
        in_ty += a.step_ty;
        if (in_tx >= a.tiles_x) { in_tx -= a.tiles_x; ++in_ty; }
        if (in_ty >= a.tiles_y) { in_ty -= a.tiles_y; ++in_n; }
    };

    int cur = 0;
    if (DB && count > 0) issue(0);
    for (int it = 0; it < count; ++it) {
        if (DB) {
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (it + 1 < count) issue(cur ^ a.bufsz);
            __builtin_amdgcn_sched_barrier(0);      // (the DMA set-up interleaved with the first LDS reads of the tile spilled 36-60 registers at 64 rows)
        } else {
            __syncthreads();
            issue(0);
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
        }
        if (DB) asm volatile("" : "+s"(cur));      // (one add per base and tile; both buffers' bases kept live cost 8 registers and spills at 64 rows)
        const ldsb pa0 = lb + cur + ax[0], pa1 = lb + cur + ax[1];
        ldsb pb[2][3];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) pb[h][kx] = lb + cur + bx[h][kx];
        if (DB) cur ^= a.bufsz;
#pragma unroll
        for (int ks = 0; ks < 4; ks += WK) {          // (the wave's k-steps are wk + ks: wk sits in the bases)
            bf16x8 af[MTW], al[SPLIT ? MTW : 1];
#pragma unroll
            for (int mt = 0; mt < MTW; ++mt) {
                const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(pa0 + mt * 4096 + ks * 1024));
                const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(pa1 + mt * 4096 + ks * 1024));
                af[mt] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
                if constexpr (SPLIT) {
                    const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(pa0 + a.off_lo + mt * 4096 + ks * 1024));
                    const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(pa1 + a.off_lo + mt * 4096 + ks * 1024));
                    al[mt] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(l0, l1, 0, 1, 2, 3, 4, 5, 6, 7));
                }
            }
            bf16x8 bf[9], bl[SPLIT ? 9 : 1];
#pragma unroll
            for (int t = 0; t < 9 + LA; ++t) {
                if (t < 9) {
                    const int off = (2 * ks + t / 3) * (PWP * 32);
                    const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(pb[0][t % 3] + off));
                    const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(pb[1][t % 3] + off));
                    bf[t] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7));
                    if constexpr (SPLIT) {
                        const s16x4 c0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(pb[0][t % 3] + a.off_lo + off));
                        const s16x4 c1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(pb[1][t % 3] + a.off_lo + off));
                        bl[t] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(c0, c1, 0, 1, 2, 3, 4, 5, 6, 7));
                    }
                }
                if (t >= LA) {
#pragma unroll
                    for (int mt = 0; mt < MTW; ++mt) {
                        if constexpr (SPLIT) {      // dz_l x_h + dz_h x_l + dz_h x_h, in the general kernel's order
                            acc[mt][t - LA] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[mt], bf[t - LA], acc[mt][t - LA], 0, 0, 0);
                            acc[mt][t - LA] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bl[t - LA], acc[mt][t - LA], 0, 0, 0);
                        }
                        acc[mt][t - LA] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bf[t - LA], acc[mt][t - LA], 0, 0, 0);
                    }
                }
                // 64 rows double-buffered: keep the source order (the scheduler otherwise requests a whole k-step's fragments at once and
                // the register allocator spills the DMA constants, whose reloads then wait on the DMA in flight: vmcnt is shared)
                if constexpr ((MTW == 4 && DB) || SPLIT) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- epilogue: as conv_wgrad_dma_kernel's 3 x 3 branch ----
    const int cit = ci0 + wc * 16;
    float* s_ep = (float*)smem + wave * (16 * WD_EP);
    const int nrem = (d.Cin - cit) * 9;
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) s_ep[(q * 4 + j) * WD_EP + li * 9 + t] = acc[mt][t][j];
        for (int row = 0; row < 16; ++row) {
            const int co = co0 + mt * 16 + row;
            if (co >= d.Cout) break;
            const int cod = a.lstmC > 0 ? (co & 3) * a.lstmC + (co >> 2) : co;
            const long roff = (((long)(g * d.Cout + cod) * d.w_cin_tot) + d.w_cin_off + cit) * 9;
            const float* srow = s_ep + row * WD_EP;
            if (a.ws) {         // this (pixel split, k-step share)'s own copy of dW: plain stores, summed by wgrad_reduce_kernel
                float* prow = a.ws + (long)(split * WK + wk) * a.ws_stride + roff;
#pragma unroll
                for (int rem = lane; rem < 144; rem += 64)
                    if (rem < nrem) prow[rem] = srow[rem];
            } else if (WK == 1 && a.nsplit == 1) {      // the only writer of these elements: plain read-modify-write, no atomics
                float* prow = a.dw + roff;
#pragma unroll
                for (int rem = lane; rem < 144; rem += 64)
                    if (rem < nrem) prow[rem] += srow[rem];
            } else {
                float* prow = a.dw + roff;
#pragma unroll
                for (int rem = lane; rem < 144; rem += 64)
                    if (rem < nrem) atomicAdd(prow + rem, srow[rem]);
            }
        }
    }
}

// dW += sum over the pixel splits of their partial copies (rows of dW = (group, output channel), `seg` floats of each row starting
// at `off`: the input-channel slice this launch produced).  grid (column blocks, rows).
__global__ void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int seg, int pitch, int off, long stride, int nsplit) {
    const long base = (long)blockIdx.y * pitch + off;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < seg; c += gridDim.x * blockDim.x) {
        float acc = 0.f;
        for (int sidx = 0; sidx < nsplit; ++sidx) acc += ws[(long)sidx * stride + base + c];
        dw[base + c] += acc;
    }
}

// Split-K partials instead of atomics: where a launch's atomic traffic nsplit x dW is large -- the ConvLSTM's
// 25 x 25 / 13 x 13 levels (24 groups x 384 x 192 x 9 floats of dW, 6 splits: 127 MB of fp32 atomics at the memory side's ~1.3 TB/s =
// 98 us of a 123 us launch), the CRN's 512-channel layers at 4 x 4 .. 64 x 64 -- every pixel split stores its block to its own copy
// of dW and one reduction pass adds the copies: stores and loads at HBM rate, and the sum order is fixed (bit-stable gradients).
struct WgWs {
    float* ws;
    long ws_bytes;
    long* need;          // query: bytes the launch would use (0: it stays on atomics); nothing is launched
};

template <int MTW, bool DB, int WC, bool SPLIT = false>
static int wgd_launch_fast(WgDArgs& a, int lds, long items, long outblocks, long dw_floats, hipStream_t s, const WgWs* wr) {
    auto k = conv_wgrad_fast_kernel<MTW, DB, WC, SPLIT>;
    static int optin[JAF_MAX_DEVICES];
    static JafOcc occ[JAF_MAX_DEVICES][8];
    if (lds > 48 * 1024) {
        const int e = jaf_lds_optin((const void*)k, optin);
        if (e) return e;
    }
    const double slots_env = 0.0;
    const double slots = slots_env > 0.0 ? slots_env : (double)jaf_kernel_slots((const void*)k, lds, occ);
    a.nsplit = (int)(slots_env < 0.0 ? jaf_wgrad_nsplit(items, outblocks, dw_floats)
                                     : jaf_wgrad_nsplit_rounds(items, outblocks, dw_floats, slots));
    a.ws = nullptr;
    a.ws_stride = (long)a.d.G * a.d.Cout * a.d.w_cin_tot * 9;
    if (wr) {
        // a split costs a store + a load of dW at HBM rate instead of an atomic pass (~4x cheaper): the optimum has more splits
        const double part_rate = 1.0e12;
        const long part_min = 6000000L;      // floats of atomic traffic
        const int nsp = (int)jaf_wgrad_nsplit_rounds(items, outblocks, dw_floats, slots, 2.5e-6, JAF_WGRAD_MAX_SPLIT, part_rate);
        constexpr int WKC = 4 / WC;        // waves that share an input-channel tile hold partial sums over their own k-steps: one copy each
        const long need = (nsp >= 2 && (long)a.nsplit * dw_floats >= part_min) ? (long)nsp * WKC * a.ws_stride * 4 : 0;
        if (wr->need) { *wr->need = need; return JAF_OK; }
        if (need > 0 && wr->ws && need <= wr->ws_bytes) { a.nsplit = nsp; a.ws = wr->ws; }
    }
    const int tiles = a.tiles_x * a.tiles_y;
    a.step_n = a.nsplit / tiles;
    const int dt = a.nsplit - a.step_n * tiles;
    a.step_ty = dt / a.tiles_x;
    a.step_tx = dt - a.step_ty * a.tiles_x;
    const long nblk = outblocks * a.nsplit;
    if (nblk > 0x7fffffffL) return JAF_EINVAL;
    JAF_NOTE_KERNEL("conv_wgrad_fast_kernel<%d, %s, %d, %s>", MTW, DB ? "true" : "false", WC, SPLIT ? "true" : "false");
    hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(256), (size_t)lds, s, a);
    if (a.ws) {
        const int seg = a.d.Cin * 9, rows = a.d.G * a.d.Cout;
        int bx = jaf_cdiv(seg, 256);
        if (bx > 8) bx = 8;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(bx, rows), dim3(256), 0, s, a.ws, a.dw, seg, a.d.w_cin_tot * 9, a.d.w_cin_off * 9, a.ws_stride, a.nsplit * (4 / WC));
    }
    return jaf_launch_status();
}

static inline int rup_w(int v, int m) { return (v + m - 1) / m * m; }

template <int MTW, int KS, bool PAIR, bool DB, int XI, bool SPLIT>
static int wgd_launch_xi(WgDArgs& a, int lds, long items, long outblocks, long dw_floats, hipStream_t s) {
    auto k = conv_wgrad_dma_kernel<MTW, KS, PAIR, DB, XI, SPLIT>;
    static int optin[JAF_MAX_DEVICES];
    static JafOcc occ[JAF_MAX_DEVICES][8];
    if (lds > 48 * 1024) {
        const int e = jaf_lds_optin((const void*)k, optin);
        if (e) return e;
    }
    // every pixel split adds one fp32 atomic pass over dW (profiles/round1_b_pmc_hbm_traffic.txt: ~116 MB of atomic
    // traffic per launch at 1536 workgroups) and contends for the same addresses: see jaf_wgrad_nsplit
    const double slots_env = 0.0;
    const double slots = slots_env > 0.0 ? slots_env : (double)jaf_kernel_slots((const void*)k, lds, occ);
    a.nsplit = (int)(slots_env < 0.0 ? jaf_wgrad_nsplit(items, outblocks, dw_floats)
                     : (KS == 7 ? jaf_wgrad_nsplit_rounds(items, outblocks, dw_floats, slots, 1.0e-5, 1024)      // (49 taps: ~10 us per tile, measured)
                                : jaf_wgrad_nsplit_rounds(items, outblocks, dw_floats, slots)));
    a.ws = nullptr;
    const long nblk = outblocks * a.nsplit;
    if (nblk > 0x7fffffffL) return JAF_EINVAL;
    JAF_NOTE_KERNEL("conv_wgrad_dma_kernel<%d, %d, %s, %s, %d, %s>", MTW, KS, PAIR ? "true" : "false", DB ? "true" : "false", XI, SPLIT ? "true" : "false");
    hipLaunchKernelGGL(k, dim3((unsigned)nblk), dim3(256), (size_t)lds, s, a);
    return jaf_launch_status();
}

template <int MTW, int KS, bool PAIR, bool DB>
static int wgd_launch(WgDArgs& a, int lds, long items, long outblocks, long dw_floats, hipStream_t s) {
    const int need = jaf_cdiv(a.WC * a.nx, 4);
    if (a.d.precision == JAF_PREC_BF16X3) {
        if (need <= 4) return wgd_launch_xi<MTW, KS, PAIR, DB, 4, true>(a, lds, items, outblocks, dw_floats, s);
        if constexpr (!DB) {        // (double-buffered split tiles are at most 40 KB: one 16-channel patch tile, 4 pieces per wave)
            if (need <= 8) return wgd_launch_xi<MTW, KS, PAIR, false, 8, true>(a, lds, items, outblocks, dw_floats, s);
            return wgd_launch_xi<MTW, KS, PAIR, false, WD_XI, true>(a, lds, items, outblocks, dw_floats, s);
        } else {
            return JAF_EINVAL;
        }
    }
    if (need <= 4) return wgd_launch_xi<MTW, KS, PAIR, DB, 4, false>(a, lds, items, outblocks, dw_floats, s);
    if (need <= 8) return wgd_launch_xi<MTW, KS, PAIR, DB, 8, false>(a, lds, items, outblocks, dw_floats, s);
    return wgd_launch_xi<MTW, KS, PAIR, DB, WD_XI, false>(a, lds, items, outblocks, dw_floats, s);
}

extern "C" int jaf_conv2d_wgrad_packed(jaf_stream_t s_, const jaf_conv_desc* d, const void* packed_x,
                                       const void* packed_dz, float* dw, int accumulate) {
    return jaf_conv2d_wgrad_packed_ex(s_, d, packed_x, 0, packed_dz, dw, accumulate);
}

extern "C" int jaf_conv2d_wgrad_packed_ex(jaf_stream_t s_, const jaf_conv_desc* d, const void* packed_x, int32_t x_ng8_tot,
                                          const void* packed_dz, float* dw, int accumulate) {
    return jaf_conv2d_wgrad_packed_lstm(s_, d, packed_x, x_ng8_tot, packed_dz, dw, accumulate, 0);
}

static int wgd_core(jaf_stream_t s_, const jaf_conv_desc* d, const void* packed_x, int32_t x_ng8_tot, const void* packed_dz, float* dw,
                    int accumulate, int32_t hidden, const WgWs* wr, int x_split);

extern "C" int jaf_conv2d_wgrad_packed_lstm(jaf_stream_t s_, const jaf_conv_desc* d, const void* packed_x, int32_t x_ng8_tot,
                                            const void* packed_dz, float* dw, int accumulate, int32_t hidden) {
    JAF_REQUIRE(d && packed_x && packed_dz && dw);
    return wgd_core(s_, d, packed_x, x_ng8_tot, packed_dz, dw, accumulate, hidden, nullptr, 0);
}

extern "C" int64_t jaf_conv2d_wgrad_packed_ws_bytes(const jaf_conv_desc* d, int32_t hidden) {
    if (!d) return JAF_EINVAL;
    long need = 0;
    const WgWs wr = {nullptr, 0, &need};
    const int rc = wgd_core(nullptr, d, nullptr, 0, nullptr, nullptr, 1, hidden, &wr, 0);
    return rc == JAF_OK ? (int64_t)need : (int64_t)rc;
}

extern "C" int jaf_conv2d_wgrad_packed_ws(jaf_stream_t s_, const jaf_conv_desc* d, const void* packed_x, int32_t x_ng8_tot,
                                          const void* packed_dz, float* dw, int accumulate, int32_t hidden, void* workspace,
                                          int64_t workspace_bytes) {
    JAF_REQUIRE(d && packed_x && packed_dz && dw);
    const WgWs wr = {(float*)workspace, (long)workspace_bytes, nullptr};
    return wgd_core(s_, d, packed_x, x_ng8_tot, packed_dz, dw, accumulate, hidden, workspace ? &wr : nullptr, 0);
}

extern "C" int jaf_conv2d_wgrad_packed_ws_x(jaf_stream_t s_, const jaf_conv_desc* d, const void* packed_x, int32_t x_ng8_tot, int x_split,
                                            const void* packed_dz, float* dw, int accumulate, int32_t hidden, void* workspace,
                                            int64_t workspace_bytes) {
    JAF_REQUIRE(d && packed_x && packed_dz && dw);
    JAF_REQUIRE(!x_split || d->precision == JAF_PREC_BF16);
    const WgWs wr = {(float*)workspace, (long)workspace_bytes, nullptr};
    return wgd_core(s_, d, packed_x, x_ng8_tot, packed_dz, dw, accumulate, hidden, workspace ? &wr : nullptr, x_split ? 1 : 0);
}

static int wgd_core(jaf_stream_t s_, const jaf_conv_desc* d, const void* packed_x, int32_t x_ng8_tot, const void* packed_dz, float* dw,
                    int accumulate, int32_t hidden, const WgWs* wr, int x_split) {
    const bool query = wr && wr->need;
    JAF_REQUIRE(hidden == 0 || (hidden > 0 && d->KH == 3 && d->Cout == 4 * hidden));
    JAF_REQUIRE(x_ng8_tot == 0 || x_ng8_tot >= jaf_cdiv(d->Cin, 8));
    JAF_REQUIRE(d->KH == d->KW && (d->KH == 1 || d->KH == 3 || d->KH == 5 || d->KH == 7) && d->dil_in == 1 && d->stride >= 1 && d->stride <= 2);
    const int KS = d->KH;
    JAF_REQUIRE(d->N >= 1 && d->G >= 1 && d->Cin >= 1 && d->Cout >= 1 && d->w_cin_off >= 0 && d->w_cin_off + d->Cin <= d->w_cin_tot);
    hipStream_t s = (hipStream_t)s_;
    if (!accumulate && !query) {
        hipError_t e = hipMemsetAsync(dw, 0, sizeof(float) * (size_t)d->G * d->Cout * d->w_cin_tot * KS * KS, s);
        if (e != hipSuccess) return (int)e;
    }
    WgDArgs a;
    a.xp = (const unsigned char*)packed_x;
    a.dzp = (const unsigned char*)packed_dz;
    a.dw = dw;
    a.d = *d;
    a.lstmC = hidden;
    int MTW = 1;
    long bestPad = 1L << 60;
    for (int mt = (KS >= 5 ? 1 : 4); mt >= 1; --mt) {      // (5 x 5, 7 x 7: 25 / 49 accumulator tiles per 16 rows)
        long pad = (long)jaf_cdiv(d->Cout, 16 * mt) * 16 * mt;
        if (pad < bestPad) { bestPad = pad; MTW = mt; }
    }
    a.PH = (WD_TH - 1) * d->stride + KS;
    a.PW = (WD_TW - 1) * d->stride + KS;
    a.PWp = rup_w(a.PW, 8);
    a.xplane = rup_w(a.PH * a.PWp * 32, 1024);
    a.nx = a.xplane / 1024;
    const bool split = d->precision == JAF_PREC_BF16X3;
    JAF_REQUIRE(d->precision == JAF_PREC_BF16 || split);
    a.WC = d->Cin <= 16 ? 1 : (d->Cin <= 32 ? 2 : 4);
    // stride-2 patches are 22 KB per 16 channels: at most 32 channels per workgroup (more input-channel blocks instead)
    while (a.WC > 1 && jaf_cdiv(a.WC * a.nx, 4) > WD_XI) a.WC >>= 1;
    if (KS == 7 && a.WC > 2) a.WC = 2;          // (the 7 x 7 epilogue sums 16 x 16 x 28 floats per input-channel tile in LDS)
    // split-bf16: hi and lo tiles of both operands; keep two workgroups per CU (<= 80 KB) where the channel tiling allows
    while (split && a.WC > 1 && 2 * (a.WC * a.xplane + MTW * 4096) > 80 * 1024) a.WC >>= 1;
    // narrower input-channel tiles where that lets the split tile be double-buffered (<= 40 KB): bf16x3 step 116.2 -> 115.5 ms
    // (two alternating pairs)
    const int split_db = 1;
    if (split && split_db && d->stride == 1 && 2 * (a.xplane + MTW * 4096) <= 40 * 1024)
        while (a.WC > 1 && 2 * (a.WC * a.xplane + MTW * 4096) > 40 * 1024) a.WC >>= 1;
    {   // launches that cannot fill the chip: smaller output blocks = more, shorter workgroups (see jafb_wgrad)
        const long items0 = (long)d->N * jaf_cdiv(d->OW, WD_TW) * jaf_cdiv(d->OH, WD_TH);
        const long sp = items0 < JAF_WGRAD_MAX_SPLIT ? items0 : JAF_WGRAD_MAX_SPLIT;
        while (KS < 5 && (long)d->G * jaf_cdiv(d->Cout, 16 * MTW) * jaf_cdiv(d->Cin, 16 * a.WC) * sp < 512) {
            if (MTW > 1) MTW = (MTW == 4) ? 2 : 1;
            else if (a.WC > 1) a.WC >>= 1;
            else break;
        }
    }
    a.WK = 4 / a.WC;
    a.coblocks = jaf_cdiv(d->Cout, 16 * MTW);
    a.ciblocks = jaf_cdiv(d->Cin, 16 * a.WC);
    a.tiles_x = jaf_cdiv(d->OW, WD_TW);
    a.tiles_y = jaf_cdiv(d->OH, WD_TH);
    a.dv_tiles = jaf_fdiv_make((uint32_t)(a.tiles_x * a.tiles_y));
    a.dv_tiles_x = jaf_fdiv_make((uint32_t)a.tiles_x);
    if (jaf_cdiv(a.WC * a.nx, 4) > WD_XI) return JAF_EUNSUPPORTED;
    a.off_dz = a.WC * a.xplane;
    a.ngin8 = jaf_cdiv(d->Cin, 8);
    a.ngout8 = jaf_cdiv(d->Cout, 8);
    a.xng8 = x_ng8_tot ? x_ng8_tot : a.ngin8;
    a.inv_pwp = 1.0f / (float)a.PWp;
    a.xmul = (x_split && !split) ? 2 : 1;
    JAF_REQUIRE((long)a.xng8 * d->H * d->W * 16 * (split ? 2 : a.xmul) < WD_OOB && (long)a.ngout8 * d->OH * d->OW * 16 * (split ? 2 : 1) < WD_OOB);
    int lds = a.off_dz + MTW * 4096;
    a.off_lo = split ? lds : 0;
    if (split) lds *= 2;
    // two tile buffers when two workgroups per CU still fit (see the kernel's header)
    const int no_db = 0;
    const int db_max = 40;
    const bool db = (!split || split_db) && !no_db && lds <= db_max * 1024;
    a.bufsz = db ? lds : 0;
    if (db) lds *= 2;
    const int lds_ep = 4 * 16 * WD_EP * 4;
    if (KS == 3 && lds < lds_ep) lds = lds_ep;
    if (KS == 7 && lds < a.WC * 16 * 16 * 28 * 4) lds = a.WC * 16 * 16 * 28 * 4;
    JAF_REQUIRE(lds <= 160 * 1024);
    JAF_REQUIRE(!(KS == 5 && d->Cin <= 8) || (a.WK == 4 && a.xplane <= 65536));     // PAIR: see padr in the kernel
    const long items = (long)d->N * a.tiles_x * a.tiles_y;
    const long outblocks = (long)d->G * a.coblocks * a.ciblocks;
    const long dw_floats = (long)d->G * d->Cout * d->Cin * KS * KS;
    // the stride-1 3 x 3 layers: conv_wgrad_fast_kernel, in the buffering the rules above give each tiling (the general kernel
    // takes what they do not cover)
    const int fast_env = 1;
    if (fast_env && !split && KS == 3 && d->stride == 1 && a.nx == 8 && a.PWp == 24) {
#define JAF_WGF(MT_, DB_, WC_) if (MTW == MT_ && db == DB_ && a.WC == WC_) return wgd_launch_fast<MT_, DB_, WC_>(a, lds, items, outblocks, dw_floats, s, wr)
        JAF_WGF(1, true, 4); JAF_WGF(2, true, 4); JAF_WGF(3, false, 4); JAF_WGF(4, false, 4);
        JAF_WGF(1, true, 2); JAF_WGF(2, true, 2); JAF_WGF(3, true, 2); JAF_WGF(4, true, 2);
        JAF_WGF(1, true, 1); JAF_WGF(2, true, 1); JAF_WGF(3, true, 1); JAF_WGF(4, true, 1);
#undef JAF_WGF
    }
    // split-bf16: the tilings the LDS rules above produce (64 rows: 32 or 16 channels single-buffered; fewer rows: 16 channels double-buffered)
    const int fast_split_env = 1;
    if (fast_env && fast_split_env && split && KS == 3 && d->stride == 1 && a.nx == 8 && a.PWp == 24) {
#define JAF_WGF(MT_, DB_, WC_) if (MTW == MT_ && db == DB_ && a.WC == WC_) return wgd_launch_fast<MT_, DB_, WC_, true>(a, lds, items, outblocks, dw_floats, s, wr)
        JAF_WGF(4, false, 2); JAF_WGF(4, false, 1); JAF_WGF(3, true, 1); JAF_WGF(2, true, 1); JAF_WGF(1, true, 2); JAF_WGF(1, true, 1);
#undef JAF_WGF
    }
    if (query) { *wr->need = 0; return JAF_OK; }          // (the general kernel keeps its atomics)
#define JAF_WGD(MT_, KS_) JAF_WGDP(MT_, KS_, false)
#define JAF_WGDP(MT_, KS_, PAIR_) \
    return db ? wgd_launch<MT_, KS_, PAIR_, true>(a, lds, items, outblocks, dw_floats, s) \
              : wgd_launch<MT_, KS_, PAIR_, false>(a, lds, items, outblocks, dw_floats, s)
    a.ky0 = 0;
    a.kyn = KS;
    if (KS == 7) {          // kernel rows 0-3, then 4-6 (see WgDArgs.ky0)
        a.kyn = 4;
        const int rc = db ? wgd_launch<1, 7, false, true>(a, lds, items, outblocks, dw_floats, s)
                          : wgd_launch<1, 7, false, false>(a, lds, items, outblocks, dw_floats, s);
        if (rc != JAF_OK) return rc;
        a.ky0 = 4;
        a.kyn = 3;
        JAF_WGD(1, 7);
    }
    if (KS == 5 && d->Cin <= 8) JAF_WGDP(1, 5, true);
    else if (KS == 5) JAF_WGD(1, 5);
    else if (KS == 1) switch (MTW) {
        case 1: JAF_WGD(1, 1);
        case 2: JAF_WGD(2, 1);
        case 3: JAF_WGD(3, 1);
        default: JAF_WGD(4, 1);
    }
    else switch (MTW) {
        case 1: JAF_WGD(1, 3);
        case 2: JAF_WGD(2, 3);
        case 3: JAF_WGD(3, 3);
        default: JAF_WGD(4, 3);
    }
#undef JAF_WGD
#undef JAF_WGDP
}
