// Shared between conv_dma.hip (bf16 operands) and conv_dma_split.hip (split-bf16 operands: hi + lo images, three matrix-core
// instructions per product): the launch arguments of the packed-input convolution kernels and their epilogue.
#pragma once
#include "conv_internal.h"
#include "jaf_fdiv.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CD_OOB 0x40000000
#define CD_RPW 2          // patch DMA rounds per wave (4 waves x 2 rounds x 64 lanes = 512 slots max)

struct ConvDArgs {
    const unsigned char* xp;       // packed input
    const unsigned char* wpk;      // packed weights (jaf_conv2d_pack, bf16 image order)
    const float* bias;
    float* out;
    const float* c_prev;
    float* c_out;
    float* h_out;
    float* gates_out;
    jaf_conv_desc d;
    jaf_conv_plan p;
    int off_w, off_tab;
    int ntiles, ngroups8;
    float inv_pwp, inv_pwq, inv_twin;
    // the prologue's divisions of the block index by launch constants (jaf_fdiv.h) and the slot table's reciprocals
    jaf_fdiv dv_mblocks, dv_ntiles, dv_G, dv_tiles_x, dv_twin;
    float inv_kw, inv_ng, inv_ng_last;
    int ilv, vec;      // pixel interleave (a lane's NT tiles = NT consecutive pixels); vector epilogue allowed
    int gates_bf16;    // LSTM: gates_out is a bf16 tensor (halves the dominant epilogue traffic)
    double* stats;     // nullable: [N][slots][2] += (sum, sum of squares) of the image's outputs (act NONE, G == 1):
    int stat_slots;    // the statistics pass of the LayerNorm that follows, taken while the values are in registers
    // packed-image plumbing (jaf_packed_io): the input image may have more planes per (image, group) than this layer
    // reads, and the epilogue may write its (activated) outputs straight into the consumer's packed bf16 image
    int in_ng8;            // planes per (image, group) of the INPUT image (>= ngroups8)
    unsigned char* dst;    // destination packed image (nullable)
    int dst_ng8, dst_coff, dst_img_off, dst_pad_tail;
    int skip_f32;          // the fp32 output tensor is not written (nobody reads it)
    int acc_out;           // out += result (jaf_packed_io.accumulate_f32)
    float* out2;           // rows >= split of every group go here (jaf_packed_io.out2); nullptr: everything to `out`
    int split;
    // fused activation backward (jaf_packed_io.dz_mask): `dst` receives dz = (acc [+ *out]) * act'(x) of the PRODUCER layer
    const unsigned char* dz_mask;
    int dz_mask_ng8, dz_mask_coff;
    float dz_slope;
    float* dz_dbias;
    // bf16 STORAGE of the NCHW tensors this launch writes (jaf_packed_io.out_bf16 / out2_bf16 / state_bf16; bf16 arithmetic only):
    // `out` (and what acc_out reads), `out2`, and the ConvLSTM's cell state (c_prev read, c_out written) hold bf16 instead of fp32
    int out_bf16, out2_bf16, state_bf16;
    int dz_mask_split;     // the sign image of the dz mode is a split-bf16 image although this launch computes in bf16 ("mixed" arithmetic)
};

// Destination of channel `dc` (within a group) of pixel `pix` of (image, group) `ng` in a packed image with `ng8`
// planes: 2 bytes at ((ng*ng8 + dc/8)*HW + pix)*16 + (dc%8)*2.
__device__ __forceinline__ unsigned char* cd_dst_ptr(unsigned char* base, long ng, int ng8, int dc, int OHW, int pix) {
    return base + ((ng * ng8 + (dc >> 3)) * (long)OHW + pix) * 16 + (dc & 7) * 2;
}

// The same in a split-bf16 image (SP): channel group cg has its hi plane at 2 cg and its lo plane (`lo` = 1) right behind it.
template <bool SP>
__device__ __forceinline__ unsigned char* cd_dst_ptr_s(unsigned char* base, long ng, int ng8, int dc, int OHW, int pix, int lo) {
    if (SP) return base + (((ng * ng8 + (dc >> 3)) * 2 + lo) * (long)OHW + pix) * 16 + (dc & 7) * 2;
    return cd_dst_ptr(base, ng, ng8, dc, OHW, pix);
}

// v - bf16(v): what the lo plane of a split-bf16 image holds
__device__ __forceinline__ float cd_resid(float v) { return v - (float)(__bf16)v; }

#define cd_L2E 1.4426950408889634f

__device__ __forceinline__ unsigned int cd_pack2(float a, float b) {
    f32x2 v = {a, b};
    bf16x2 r = __builtin_convertvector(v, bf16x2);
    return __builtin_bit_cast(unsigned int, r);
}

// ---------------------------------------------------------------------------------------------
// epilogue
// ---------------------------------------------------------------------------------------------
// DZ: the fused activation backward of jaf_packed_io.dz_mask.  A template parameter, not a run-time branch: with the dz
// code in every instantiation the compiler kept its extra live ranges in ALL of them (conv_dma_kernel<4,4,false>: 160 -> 192
// VGPRs, 3 -> 2 waves per SIMD, 9.5 -> 11.0 ms per step over its 94 launches).
// SPD: the packed destination is a split-bf16 image (conv_dma_split.hip): every packed store is issued twice, the hi words
// into the group's first plane and the residual words into the plane behind it (DZ: the sign mask is read from the hi plane).
template <int MT, int NT, bool LSTM, bool DZ, bool PLAIN, bool SPD = false>
__device__ __forceinline__ void cd_epilogue(const ConvDArgs& a, const f32x4 (&acc)[MT][NT], const int (&opix)[NT],
                                            int n, int g, int mb, int q, int OHW, unsigned char* smem,
                                            const f32x4 (&cpre)[MT]) {
    const jaf_conv_desc& d = a.d;
    constexpr int MR = 16 * MT;
    if constexpr (PLAIN) {
        // the launch writes ONE fp32 tensor and nothing else (no packed image, statistics, accumulation or second output):
        // its own instantiation, so that the registers of those features are not carried through the matrix-core loop
        typedef float pvec __attribute__((ext_vector_type(NT == 1 ? 2 : NT)));
        typedef __bf16 phvec __attribute__((ext_vector_type(NT == 1 ? 2 : NT)));
        const bool pv = a.vec && (NT > 1);
        const bool hb = a.out_bf16 != 0;
        const long prow = ((long)n * d.out_ctot + d.out_coff + g * d.Cout + mb * MR + q * 4) * (long)OHW;      // the lane's row at mt = 0, j = 0
#define CD_EPILOGUE_PLAIN(ACT_)                                                                       \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                           \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                           \
                const int co = mb * MR + mt * 16 + q * 4 + j;                                         \
                if (co < d.Cout) {                                                                    \
                    const float b = a.bias ? a.bias[g * d.Cout + co] : 0.f;                           \
                    const long ooff = prow + (long)((mt * 16 + j) * OHW);                             \
                    float* op = a.out + ooff;                                                         \
                    __bf16* hp = (__bf16*)a.out + ooff;      /* out_bf16: the same tensor in bf16 */   \
                    if (pv) {                                                                         \
                        if (opix[0] >= 0) {                                                           \
                            pvec o;                                                                   \
                            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                         \
                                o[nt] = jaf_act(acc[mt][nt][j] + b, ACT_, d.slope);                   \
                            if (hb) *(phvec*)(hp + opix[0]) = __builtin_convertvector(o, phvec);      \
                            else *(pvec*)(op + opix[0]) = o;                                          \
                        }                                                                             \
                    } else {                                                                          \
                        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                             \
                            if (opix[nt] >= 0) {                                                      \
                                const float v = jaf_act(acc[mt][nt][j] + b, ACT_, d.slope);           \
                                if (hb) hp[opix[nt]] = (__bf16)v; else op[opix[nt]] = v;             \
                            }                                                                         \
                    }                                                                                 \
                }                                                                                     \
            }                                                                                         \
        }
        switch (d.act) {
            case JAF_ACT_LRELU: CD_EPILOGUE_PLAIN(JAF_ACT_LRELU) break;
            case JAF_ACT_RELU: CD_EPILOGUE_PLAIN(JAF_ACT_RELU) break;
            case JAF_ACT_SIGMOID: CD_EPILOGUE_PLAIN(JAF_ACT_SIGMOID) break;
            case JAF_ACT_TANH: CD_EPILOGUE_PLAIN(JAF_ACT_TANH) break;
            default: CD_EPILOGUE_PLAIN(JAF_ACT_NONE) break;
        }
#undef CD_EPILOGUE_PLAIN
        return;
    }
    // ---- epilogue (D layout: column lane&15 = pixel, row (lane>>4)*4 + reg = output channel).
    // With the pixel interleave a lane's NT tiles are NT consecutive pixels: one vector access. ----
    typedef float fvec __attribute__((ext_vector_type(NT == 1 ? 2 : NT)));
    typedef __bf16 hvec __attribute__((ext_vector_type(NT == 1 ? 2 : NT)));
    const bool vec = a.vec && (NT > 1);
    if (!LSTM) {
        float st1 = 0.f, st2 = 0.f;     // LayerNorm statistics of this lane's outputs (ACT NONE + a.stats only)
        if (a.dst) {
            // The lane's 4 rows of a tile are 4 consecutive output channels = half of a 16-byte packed item
            // (dst_coff % 4 == 0); lanes q and q^1 complete the item within the same store instruction.
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            const long ngd = ((long)(n + a.dst_img_off)) * d.G + g;
            // dz mode with a second fp32 output (the ConvLSTM's d[x_t, h_{t-1}] launch): only the rows below `split` (dx) are the
            // producer's dz; the rows from `split` on (dh) leave through out2 as always
            const int dC = (DZ && a.out2) ? a.split : d.Cout;
            // Address arithmetic, strength-reduced by hand (round 5; these launches are vector-instruction-issue bound and every
            // store used to recompute ((ng * ng8 + (dc >> 3)) * OHW + pix) * 16 with 64-bit multiplies, ~12 instructions of which 4
            // quarter-rate): the lane's FIRST destination channel is dcl (row tile 0); row tile mt is 16 channels = two planes
            // further, the residual plane of a split image one plane further, pixel p 16 p bytes further.
            const long plane16 = (long)OHW * 16;
            const int dcl = a.dst_coff + mb * MR + q * 4;
            unsigned char* const dlane = a.dst + ((ngd * a.dst_ng8 + (dcl >> 3)) * (SPD ? 2 : 1)) * plane16 + (dcl & 7) * 2;
            const long dmt = 2 * (SPD ? 2 : 1) * plane16;
            // the sign image of the dz mode, the same way (a split image under bf16 arithmetic in the mixed mode)
            const bool msplit = SPD || (DZ && a.dz_mask_split);
            const int mcl = a.dz_mask_coff + mb * MR + q * 4;
            const unsigned char* const mlane = DZ ? a.dz_mask + (((((long)n) * d.G + g) * a.dz_mask_ng8 + (mcl >> 3)) * (msplit ? 2 : 1)) * plane16 + (mcl & 7) * 2
                                                  : nullptr;
            const long mmt = 2 * (msplit ? 2 : 1) * plane16;
            // element offset of the lane's first row (row tile 0, j = 0) of the first consumer's gradient in `out` (dz mode + acc_out)
            const long prow0 = (a.out2 ? (((long)n * d.G + g) * a.split + mb * MR + q * 4)
                                       : ((long)n * d.out_ctot + d.out_coff + g * d.Cout + mb * MR + q * 4)) * (long)OHW;
            float wsum[MT][4];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) wsum[mt][j] = 0.f;
            const int cpad = a.dst_pad_tail ? ((a.dst_coff + dC + 7) & ~7) - a.dst_coff : dC;   // rows < cpad are written
#pragma unroll
            for (int sp = 0; sp < (SPD ? 2 : 1); ++sp)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int co0 = mb * MR + mt * 16 + q * 4;
                if (co0 >= cpad) continue;
                float bv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[j] = (a.bias && co0 + j < d.Cout) ? a.bias[g * d.Cout + co0 + j] : 0.f;
                float bsum[4] = {0.f, 0.f, 0.f, 0.f};     // dz mode: this lane's share of the producer's bias gradient
                // dz mode: the sign masks of the lane's NT pixels, requested together and ahead of the first consumer's gradient
                // (inside the pixel loop each 8-byte load sat behind the previous pixel's store: the compiler cannot move a
                // load above a store to a pointer that may alias it)
                unsigned int mk[NT][2];
                if (DZ) {
                    const unsigned char* const mrow = mlane + mt * mmt;
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        mk[nt][0] = mk[nt][1] = 0u;
                        if (opix[nt] >= 0) {
                            const unsigned int* xp2 = (const unsigned int*)(mrow + (unsigned long)(unsigned)opix[nt] * 16u);
                            mk[nt][0] = xp2[0];
                            mk[nt][1] = xp2[1];
                        }
                    }
                }
                // dz mode, second of two consumers: the first one's gradient, 4 channels x NT pixels (one vector load per
                // channel when the lane's pixels are consecutive)
                float part[4][NT];
                if (DZ && a.acc_out) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) part[j][nt] = 0.f;
                        if (co0 + j < dC) {
                            const long poff = prow0 + (long)((mt * 16 + j) * OHW);      // (lane's row at mt = 0, j = 0) + uniform rows
                            const float* pp = a.out + poff;
                            const __bf16* ph = (const __bf16*)a.out + poff;
                            if (vec) {
                                if (opix[0] >= 0) {
                                    if (a.out_bf16) {
                                        const hvec pv = *(const hvec*)(ph + opix[0]);
#pragma unroll
                                        for (int nt = 0; nt < NT; ++nt) part[j][nt] = (float)pv[nt];
                                    } else {
                                        const fvec pv = *(const fvec*)(pp + opix[0]);
#pragma unroll
                                        for (int nt = 0; nt < NT; ++nt) part[j][nt] = pv[nt];
                                    }
                                }
                            } else {
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt)
                                    if (opix[nt] >= 0) part[j][nt] = a.out_bf16 ? (float)ph[opix[nt]] : pp[opix[nt]];
                            }
                        }
                    }
                }
                // Whole 16-byte items: row groups q and q^1 hold the two halves (channels +0..3 / +4..7) of the same item for the same
                // NT pixels.  With 4 pixels per lane they exchange halves -- v_permlane16_swap: the even group keeps pixels 0, 1, the odd
                // one takes pixels 2, 3 -- and every lane issues two 16-byte stores instead of four 8-byte ones (the epilogue is
                // store-issue bound; measured on the fused-dz launches: the 8-byte stores were 8-25 % of the kernel).  Needs the
                // pair's 8 channels inside the written range and the slot on an item boundary; no lane of the pair may leave early.
                const bool pair_ok = (NT == 4) && ((a.dst_coff & 7) == 0) && ((co0 & ~4) + 8 <= cpad);
                unsigned int wl[NT], wh[NT];       // the lane's 8 bytes per pixel, as scalars (see the ConvLSTM transpose below)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    wl[nt] = wh[nt] = 0u;
                    const bool pv = opix[nt] >= 0;
                    if (!pv && !pair_ok) continue;
                    float v[4];
                    if (DZ) {
                        // producer's dz: (this data gradient [+ the first consumer's]) * act'(x), x from the packed image the
                        // consumer layer read (4 consecutive channels = 8 bytes of a 16-byte item)
                        const unsigned int x01 = mk[nt][0], x23 = mk[nt][1];
                        const float xs[4] = {__builtin_bit_cast(float, x01 << 16), __builtin_bit_cast(float, x01 & 0xffff0000u),
                                             __builtin_bit_cast(float, x23 << 16), __builtin_bit_cast(float, x23 & 0xffff0000u)};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float t = 0.f;
                            if (pv && co0 + j < dC) {
                                t = acc[mt][nt][j];
                                if (a.acc_out) t += part[j][nt];
                                t *= (xs[j] > 0.f) ? 1.f : a.dz_slope;
                            }
                            if (!SPD || sp == 0) bsum[j] += t;
                            v[j] = (SPD && sp == 1) ? cd_resid(t) : t;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = (co0 + j < d.Cout) ? jaf_act(acc[mt][nt][j] + bv[j], d.act, d.slope) : 0.f;
                        if (SPD && sp == 1) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = cd_resid(v[j]);
                        }
                    }
                    wl[nt] = cd_pack2(v[0], v[1]);
                    wh[nt] = cd_pack2(v[2], v[3]);
                    if (pair_ok) continue;
                    unsigned char* const dpx = dlane + mt * dmt + (SPD ? sp * plane16 : 0) + (unsigned long)(unsigned)opix[nt] * 16u;
                    if (co0 + 4 <= cpad || a.dst_pad_tail) {
                        const u32x2 w = {wl[nt], wh[nt]};
                        *(u32x2*)dpx = w;
                    } else {           // a 4-group that straddles the end of this source: channel by channel
                        unsigned short* hp = (unsigned short*)dpx;
                        const unsigned int ww[2] = {wl[nt], wh[nt]};
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (co0 + j < dC) hp[j] = (unsigned short)(ww[j >> 1] >> ((j & 1) * 16));
                    }
                }
                if (pair_ok) {
                    // (A, B) = (pixel 0, pixel 2) and (C, D) = (pixel 1, pixel 3): after the swaps an even group holds {own, partner's}
                    // halves of pixels 0 and 1, an odd group {partner's, own} halves of pixels 2 and 3
                    unsigned int a0 = wl[0], a1 = wh[0], b0 = wl[2 % NT], b1 = wh[2 % NT];
                    unsigned int c0 = wl[1 % NT], c1 = wh[1 % NT], e0 = wl[3 % NT], e1 = wh[3 % NT];
                    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %2\n\tv_permlane16_swap_b32 %1, %3\n\t"
                                 "v_permlane16_swap_b32 %4, %6\n\tv_permlane16_swap_b32 %5, %7\n\ts_nop 1"
                                 : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1), "+v"(c0), "+v"(c1), "+v"(e0), "+v"(e1));
                    const int pA = (q & 1) ? opix[2 % NT] : opix[0], pB = (q & 1) ? opix[3 % NT] : opix[1 % NT];
                    // the pair's item starts at channel co0 & ~4: for the odd row group that is 4 channels = 8 bytes below its own half
                    // (dst_coff % 8 == 0 here, so neither the plane nor the item changes)
                    unsigned char* const dpair = dlane + mt * dmt + (SPD ? sp * plane16 : 0) - ((q & 1) ? 8 : 0);
                    if (pA >= 0) {
                        const u32x4 w = {a0, a1, b0, b1};
                        *(u32x4*)(dpair + (unsigned long)(unsigned)pA * 16u) = w;
                    }
                    if (pB >= 0) {
                        const u32x4 w = {c0, c1, e0, e1};
                        *(u32x4*)(dpair + (unsigned long)(unsigned)pB * 16u) = w;
                    }
                }
                if (DZ && a.dz_dbias && (!SPD || sp == 0)) {       // the 16 lanes of a q-group hold the same 4 channels: fold them
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        float t = bsum[j];
                        t += __shfl_xor(t, 1); t += __shfl_xor(t, 2); t += __shfl_xor(t, 4); t += __shfl_xor(t, 8);
                        wsum[mt][j] = t;
                    }
                }
            }
            if (DZ && a.dz_dbias) {
                // workgroup sum of the four waves in LDS (the patch buffer is free once every wave has left the matrix-core
                // loop), then ONE atomic per channel and workgroup, spread over JAF_DZ_BIAS_SLOTS copies of the vector
                // (same-address fp32 atomics serialise: one per wave and channel made these launches 3x slower)
                __syncthreads();
                float* red = (float*)smem;                      // [4 waves][MT * 16 rows]
                if ((threadIdx.x & 15) == 0) {
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                        for (int j = 0; j < 4; ++j) red[(threadIdx.x >> 6) * (MT * 16) + mt * 16 + q * 4 + j] = wsum[mt][j];
                }
                __syncthreads();
                if (threadIdx.x < MT * 16) {
                    const int co = mb * MR + threadIdx.x;
                    const float t = (red[threadIdx.x] + red[MT * 16 + threadIdx.x]) + (red[2 * MT * 16 + threadIdx.x] + red[3 * MT * 16 + threadIdx.x]);
                    if (co < dC && t != 0.f)
                        atomicAdd(&a.dz_dbias[(long)(blockIdx.x % JAF_DZ_BIAS_SLOTS) * (d.G * dC) + g * dC + co], t);
                }
            }
        }
        if (a.skip_f32 || (DZ && !a.out2)) return;
        // element offsets of the lane's row (row tile 0, j = 0) in `out` / `out2`: every other row is a uniform number of planes further
        const long orow1 = (a.out2 ? (((long)n * d.G + g) * a.split + mb * MR + q * 4)
                                   : ((long)n * d.out_ctot + d.out_coff + g * d.Cout + mb * MR + q * 4)) * (long)OHW;
        const long orow2 = (((long)n * d.G + g) * (d.Cout - a.split) + (mb * MR + q * 4 - a.split)) * (long)OHW;
// one output row (channel `co`) of the lane's NT pixels: OP_ / HP_ = the row's base as fp32 / bf16 elements, HB_: bf16 storage,
// ACCP_: add what is there (GradSlot).  A macro, not a pointer select: selecting between the kernel's two output pointers at run
// time made the compiler park them in scratch memory.
#define CD_STORE_ROW(ACT_, ST_, OP_, HP_, HB_, ACCP_)                                                 \
                    if (vec) {                                                                        \
                        if (opix[0] >= 0) {                                                           \
                            fvec o;                                                                   \
                            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                         \
                                o[nt] = jaf_act(acc[mt][nt][j] + b, ACT_, d.slope);                   \
                            if (HB_) {                                                                \
                                if (ACCP_) o += __builtin_convertvector(*(const hvec*)((HP_) + opix[0]), fvec); \
                                *(hvec*)((HP_) + opix[0]) = __builtin_convertvector(o, hvec);         \
                            } else {                                                                  \
                                if (ACCP_) o += *(const fvec*)((OP_) + opix[0]);                      \
                                *(fvec*)((OP_) + opix[0]) = o;                                        \
                            }                                                                         \
                            if (ST_) {                                                                \
                                _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) { st1 += o[nt]; st2 += o[nt] * o[nt]; } \
                            }                                                                         \
                        }                                                                             \
                    } else {                                                                          \
                        _Pragma("unroll") for (int nt = 0; nt < NT; ++nt)                             \
                            if (opix[nt] >= 0) {                                                      \
                                float v = jaf_act(acc[mt][nt][j] + b, ACT_, d.slope);                 \
                                if (HB_) {                                                            \
                                    if (ACCP_) v += (float)(HP_)[opix[nt]];                           \
                                    (HP_)[opix[nt]] = (__bf16)v;                                      \
                                } else {                                                              \
                                    if (ACCP_) v += (OP_)[opix[nt]];                                  \
                                    (OP_)[opix[nt]] = v;                                              \
                                }                                                                     \
                                if (ST_) { st1 += v; st2 += v * v; }                                  \
                            }                                                                         \
                    }
#define CD_EPILOGUE(ACT_, ST_)                                                                        \
        _Pragma("unroll") for (int mt = 0; mt < MT; ++mt) {                                           \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                           \
                const int co = mb * MR + mt * 16 + q * 4 + j;                                         \
                if (co < d.Cout && !(DZ && co < a.split)) {      /* (dz rows went to `dst`) */  \
                    const float b = a.bias ? a.bias[g * d.Cout + co] : 0.f;                           \
                    if (a.out2 && co >= a.split) {                                                    \
                        const long ooff = orow2 + (long)((mt * 16 + j) * OHW);                        \
                        CD_STORE_ROW(ACT_, ST_, a.out2 + ooff, (__bf16*)a.out2 + ooff, a.out2_bf16, false)   \
                    } else {                                                                          \
                        const long ooff = orow1 + (long)((mt * 16 + j) * OHW);                        \
                        CD_STORE_ROW(ACT_, ST_, a.out + ooff, (__bf16*)a.out + ooff, a.out_bf16, a.acc_out)  \
                    }                                                                                 \
                }                                                                                     \
            }                                                                                         \
        }
        switch (d.act) {     // hoisted: one tight copy of the store loop per activation
            case JAF_ACT_LRELU: CD_EPILOGUE(JAF_ACT_LRELU, 0) break;
            case JAF_ACT_RELU: CD_EPILOGUE(JAF_ACT_RELU, 0) break;
            case JAF_ACT_SIGMOID: CD_EPILOGUE(JAF_ACT_SIGMOID, 0) break;
            case JAF_ACT_TANH: CD_EPILOGUE(JAF_ACT_TANH, 0) break;
            default:
                if (a.stats) { CD_EPILOGUE(JAF_ACT_NONE, 1) } else { CD_EPILOGUE(JAF_ACT_NONE, 0) }
                break;
        }
#undef CD_EPILOGUE
#undef CD_STORE_ROW
        if (a.stats) {      // uniform: wave sums -> LDS -> one pair of fp64 atomics per workgroup, spread over slots
            st1 = jaf_wave_sum(st1);
            st2 = jaf_wave_sum(st2);
            __syncthreads();                      // every wave has left the MFMA loop: the patch buffer is free
            float* red = (float*)smem;
            if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = st1; red[2 * (threadIdx.x >> 6) + 1] = st2; }
            __syncthreads();
            if (threadIdx.x == 0) {
                double* w = a.stats + ((long)n * a.stat_slots + (blockIdx.x % a.stat_slots)) * 2;
                atomicAdd(w, (double)((red[0] + red[2]) + (red[4] + red[6])));
                atomicAdd(w + 1, (double)((red[1] + red[3]) + (red[5] + red[7])));
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
            const long hc = (((long)n * d.G + g) * C + ch) * OHW;
            const long gc = ((long)n * d.G + g) * d.Cout * (long)OHW;
            if (vec) {
                // NT == 4 with a packed destination: no lane leaves before the 4 x 4 lane transpose of h below (a lane group's 4
                // consecutive pixels are valid or invalid together; dead lanes compute on zeros and store nothing).  Hidden channels
                // in whole groups of 4 (every real layer): the `ch >= C` exit above is then uniform over the four row groups.
                const bool live = opix[0] >= 0;
                const bool xpose = (NT == 4) && a.dst && (C & 3) == 0;
                if (!live && !xpose) continue;
                fvec cp, vi, vf, vo, vg, vc, vh;
                float hs4[4] = {0.f, 0.f, 0.f, 0.f};     // h of the lane's pixels as scalars (see the transpose below)
                if (a.c_prev) {
                    if (NT == 4) {                      // fetched before the matrix-core loop (conv_dma_kernel): no exposed latency here
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) cp[nt] = cpre[mt][nt & 3];
                    } else if (a.state_bf16) {
                        cp = __builtin_convertvector(*(const hvec*)((const __bf16*)a.c_prev + hc + opix[0]), fvec);
                    } else {
                        cp = *(const fvec*)(a.c_prev + hc + opix[0]);
                    }
                }
                // The launch is vector-instruction-issue bound, and the gate math is most of the epilogue: the biases are
                // folded into the exponent's fma (sigmoid(z + b) = 1 / (1 + 2^(-z L - b L)), L = log2 e) and tanh(x) is taken as
                // 1 - 2 / (1 + 2^(2 L x)) -- 5 instructions instead of 8, saturating cleanly through 2^(+-inf) -- : 28 instead of 40
                // vector instructions per hidden pixel, |error| ~1e-7 either way.
                const float nbi = -cd_L2E * bi, nbf = -cd_L2E * bf, nbo = -cd_L2E * bo, pbg = 2.f * cd_L2E * bg;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    vi[nt] = jaf_rcp(1.f + __builtin_amdgcn_exp2f(__builtin_fmaf(acc[mt][nt][0], -cd_L2E, nbi)));
                    vf[nt] = jaf_rcp(1.f + __builtin_amdgcn_exp2f(__builtin_fmaf(acc[mt][nt][1], -cd_L2E, nbf)));
                    vo[nt] = jaf_rcp(1.f + __builtin_amdgcn_exp2f(__builtin_fmaf(acc[mt][nt][2], -cd_L2E, nbo)));
                    vg[nt] = 1.f - 2.f * jaf_rcp(1.f + __builtin_amdgcn_exp2f(__builtin_fmaf(acc[mt][nt][3], 2.f * cd_L2E, pbg)));
                    const float c_old = a.c_prev ? cp[nt] : 0.f;
                    vc[nt] = vf[nt] * c_old + vi[nt] * vg[nt];
                    vh[nt] = vo[nt] * (1.f - 2.f * jaf_rcp(1.f + __builtin_amdgcn_exp2f(vc[nt] * (2.f * cd_L2E))));
                    hs4[nt & 3] = vh[nt];
                }
                if (live) {
                    if (a.state_bf16) *(hvec*)((__bf16*)a.c_out + hc + opix[0]) = __builtin_convertvector(vc, hvec);
                    else *(fvec*)(a.c_out + hc + opix[0]) = vc;
                    if (!a.skip_f32) *(fvec*)(a.h_out + hc + opix[0]) = vh;
                }
                if (a.dst) {     // h_t straight into the consumer's packed image (next step's [x, h] / the decoder's skip)
                    const long ngd = ((long)(n + a.dst_img_off)) * d.G + g;
                    if (xpose) {
                        // the 4 row groups (q) of a lane column hold channels cb .. cb+3 of the same 4 pixels: a 4 x 4 transpose over
                        // (q, pixel) -- two v_permlane32_swap + two v_permlane16_swap -- gives every lane 4 consecutive channels of
                        // ONE pixel = 8 contiguous bytes of its packed item: 1 store instead of 4 two-byte ones (12 -> 3 per lane).
                        // From assembly and from scalar copies of h: fed with elements of the float4 `vh`, this compiler passed ONE
                        // register as all four operands (builtins and assembly alike; the sequence itself is verified in
                        // profiles/experiments/swap_test*.hip).
                        unsigned t0 = __builtin_bit_cast(unsigned, hs4[0]), t1 = __builtin_bit_cast(unsigned, hs4[1]);
                        unsigned t2 = __builtin_bit_cast(unsigned, hs4[2]), t3 = __builtin_bit_cast(unsigned, hs4[3]);
                        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %1, %3\n\ts_nop 1\n\t"
                                     "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1"
                                     : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3));
                        if (live) {
                            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                            const float f0 = __builtin_bit_cast(float, t0), f1 = __builtin_bit_cast(float, t1);
                            const float f2 = __builtin_bit_cast(float, t2), f3 = __builtin_bit_cast(float, t3);
                            const u32x2 w = {cd_pack2(f0, f1), cd_pack2(f2, f3)};
                            *(u32x2*)cd_dst_ptr_s<SPD>(a.dst, ngd, a.dst_ng8, a.dst_coff + (ch - q), OHW, opix[0] + q, 0) = w;
                            if (SPD) {
                                const u32x2 wr = {cd_pack2(cd_resid(f0), cd_resid(f1)), cd_pack2(cd_resid(f2), cd_resid(f3))};
                                *(u32x2*)cd_dst_ptr_s<SPD>(a.dst, ngd, a.dst_ng8, a.dst_coff + (ch - q), OHW, opix[0] + q, 1) = wr;
                            }
                        }
                    } else {
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            *(__bf16*)cd_dst_ptr_s<SPD>(a.dst, ngd, a.dst_ng8, a.dst_coff + ch, OHW, opix[0] + nt, 0) = (__bf16)vh[nt];
                            if (SPD) *(__bf16*)cd_dst_ptr_s<SPD>(a.dst, ngd, a.dst_ng8, a.dst_coff + ch, OHW, opix[0] + nt, 1) = (__bf16)cd_resid(vh[nt]);
                        }
                    }
                }
                if (a.gates_out && live) {
                    if (a.gates_bf16) {
                        // bf16 gates are kept gate-innermost, [n][g][c][pixel][i, f, o, g]: the lane's NT pixels x 4 gates are
                        // 8 NT contiguous bytes (two 16-byte stores at NT = 4 instead of four 8-byte ones; the epilogue is
                        // store-issue bound) and the gate backward fetches a pixel pair's four gates with one 16-byte load
                        typedef __bf16 g4vec __attribute__((ext_vector_type(NT == 1 ? 8 : 4 * NT)));
                        typedef float f4vec __attribute__((ext_vector_type(NT == 1 ? 8 : 4 * NT)));
                        f4vec gv;
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            gv[4 * nt] = vi[nt]; gv[4 * nt + 1] = vf[nt]; gv[4 * nt + 2] = vo[nt]; gv[4 * nt + 3] = vg[nt];
                        }
                        *(g4vec*)((__bf16*)a.gates_out + gc + ((long)ch * OHW + opix[0]) * 4) = __builtin_convertvector(gv, g4vec);
                    } else {
                        float* gp = a.gates_out + gc + opix[0];
                        *(fvec*)(gp + (long)(ch)*OHW) = vi;
                        *(fvec*)(gp + (long)(C + ch) * OHW) = vf;
                        *(fvec*)(gp + (long)(2 * C + ch) * OHW) = vo;
                        *(fvec*)(gp + (long)(3 * C + ch) * OHW) = vg;
                    }
                }
            } else {
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    if (opix[nt] < 0) continue;
                    const float gi = jaf_sigmoid(acc[mt][nt][0] + bi);
                    const float gf = jaf_sigmoid(acc[mt][nt][1] + bf);
                    const float go = jaf_sigmoid(acc[mt][nt][2] + bo);
                    const float gg = jaf_tanh(acc[mt][nt][3] + bg);
                    const float cp = a.c_prev ? (a.state_bf16 ? (float)((const __bf16*)a.c_prev)[hc + opix[nt]] : a.c_prev[hc + opix[nt]]) : 0.f;
                    const float cc = gf * cp + gi * gg;
                    if (a.state_bf16) ((__bf16*)a.c_out)[hc + opix[nt]] = (__bf16)cc;
                    else a.c_out[hc + opix[nt]] = cc;
                    const float hv = go * jaf_tanh(cc);
                    if (!a.skip_f32) a.h_out[hc + opix[nt]] = hv;
                    if (a.dst) {
                        *(__bf16*)cd_dst_ptr_s<SPD>(a.dst, ((long)(n + a.dst_img_off)) * d.G + g, a.dst_ng8, a.dst_coff + ch, OHW, opix[nt], 0) = (__bf16)hv;
                        if (SPD) *(__bf16*)cd_dst_ptr_s<SPD>(a.dst, ((long)(n + a.dst_img_off)) * d.G + g, a.dst_ng8, a.dst_coff + ch, OHW, opix[nt], 1) = (__bf16)cd_resid(hv);
                    }
                    if (a.gates_out) {
                        if (a.gates_bf16) {
                            __bf16* gp = (__bf16*)a.gates_out + gc + ((long)ch * OHW + opix[nt]) * 4;
                            gp[0] = (__bf16)gi;
                            gp[1] = (__bf16)gf;
                            gp[2] = (__bf16)go;
                            gp[3] = (__bf16)gg;
                        } else {
                            float* gp = a.gates_out + gc + opix[nt];
                            gp[(long)(ch)*OHW] = gi;
                            gp[(long)(C + ch) * OHW] = gf;
                            gp[(long)(2 * C + ch) * OHW] = go;
                            gp[(long)(3 * C + ch) * OHW] = gg;
                        }
                    }
                }
            }
        }
    }
}

// conv_dma_split.hip: the split-bf16 launch of the same argument block (d.precision == JAF_PREC_BF16X3)
int cd_split_launch(const ConvDArgs& a, hipStream_t s, bool lstm);
