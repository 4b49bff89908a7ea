/*
 * jafpro_hip.h -- C ABI of libjafpro_hip.so: the MI355X (gfx950) kernels behind the JAFPro
 * stage-4 train step (BASELINE.json north_star).
 *
 * The reference has exactly one FFI on this path, the neural_renderer pybind module
 * (third_party/neural_renderer/neural_renderer/cuda/rasterize_cuda.cpp:70-95,194-200); every
 * other op reaches cuDNN/ATen through torch.nn.  This header is what a native replacement of
 * those call sites binds: plain pointers and sizes, no torch types.  Conventions (modelled on,
 * and stricter than, rasterize_cuda.cpp:66-95):
 *   - all buffers are caller-allocated device memory, dense fp32 NCHW unless stated
 *     (int32 face-index maps, uint8 HWC IUV maps), 16-byte aligned;
 *   - work is enqueued on the hipStream_t passed in (the reference kernels use the legacy
 *     default stream, rasterize_cuda_kernel.cu:616,630);
 *   - return 0 on success, JAF_E* (<0) for rejected arguments, a hipError_t (>0) for a failed
 *     launch; nothing throws; no hidden global state; re-entrant across devices/streams.
 */
#ifndef JAFPRO_HIP_H
#define JAFPRO_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* jaf_stream_t; /* == hipStream_t */

#define JAF_OK 0
#define JAF_EINVAL (-1)
#define JAF_EUNSUPPORTED (-2)

enum { JAF_ACT_NONE = 0, JAF_ACT_LRELU = 1, JAF_ACT_RELU = 2, JAF_ACT_SIGMOID = 3, JAF_ACT_TANH = 4 };

int jaf_version(void);

/* ------------------------------------------------------------------------------------------
 * Convolution family.  Replaces nn.Conv2d (+ fused bias / LeakyReLU / ReLU / sigmoid) at
 * src/networks.py:868-878,900-903 (Downsampler / Upsampler_SE), src/crn_model.py:98-100
 * (ConvBlock conv), src/flow_net.py:13-51, src/networks.py:361-417 (discriminators), VGG19
 * (src/networks.py:70-94), and the ConvLSTM gate conv (src/convLSTM.py:43-45).
 *
 * One call convolves G independent groups (the 24 body-part networks of
 * src/networks.py:1649-1652 run as ONE grouped launch) and reads its input as the channel
 * concatenation of up to three source tensors, so torch.cat at src/convLSTM.py:43,
 * src/networks.py:907,1164 and src/crn_model.py:276-299 is never materialised.
 * Source i is [N, src_ctot[i], H, W]; group g reads channels
 *   src_coff[i] + g*src_gstride[i] + [0, src_c[i]).
 * Weights are the reference tensor [G, Cout_w, w_cin_tot, KH, KW]; this call uses input
 * channels [w_cin_off, w_cin_off+Cin).  Output is [N, out_ctot, OH, OW], group g writes
 * channels out_coff + g*Cout + [0, Cout).
 * ------------------------------------------------------------------------------------------ */
typedef struct jaf_conv_desc {
    int32_t N, G;
    int32_t Cin, Cout;          /* per group; Cin == sum(src_c[0..nsrc)) */
    int32_t H, W, OH, OW;
    int32_t KH, KW, stride;
    int32_t pad_t, pad_l;       /* zero padding on top/left; bottom/right follow from OH/OW */
    int32_t dil_in;             /* 1; 2 = read the source as if zero-dilated by 2 (stride-2 dgrad) */
    int32_t nsrc;
    int32_t src_c[3], src_ctot[3], src_coff[3], src_gstride[3];
    int32_t w_cin_tot, w_cin_off;
    int32_t out_ctot, out_coff;
    int32_t act;                /* JAF_ACT_* applied after bias */
    float slope;                /* LeakyReLU negative slope */
    int32_t precision;          /* JAF_PREC_*: arithmetic of the matrix-core contraction */
} jaf_conv_desc;

/* Matrix-core arithmetic of the convolution family (storage is fp32 in every mode):
 *   JAF_PREC_F32     v_mfma_f32_16x16x4_f32, exact fp32 products (the <=1e-3 parity path);
 *   JAF_PREC_BF16    operands rounded to bf16 (RNE) when staged in LDS, fp32 accumulate
 *                    (v_mfma_f32_16x16x32_bf16; BASELINE.json configs[2] "bf16");
 *   JAF_PREC_BF16X3  operands split hi+lo in bf16, a*b ~= ah*bh + al*bh + ah*bl, fp32 accumulate:
 *                    ~2^-17 relative error per product at 3 bf16 MFMAs per k-step.          */
enum { JAF_PREC_F32 = 0, JAF_PREC_BF16 = 1, JAF_PREC_BF16X3 = 2 };

/* Tiling chosen by the library for a descriptor (jaf_conv2d_plan). */
typedef struct jaf_conv_plan {
    int32_t MT;                 /* 16-row MFMA tiles per workgroup along Cout */
    int32_t NT;                 /* 16-pixel MFMA tiles per wave */
    int32_t CK;                 /* input channels staged per LDS chunk (multiple of 4) */
    int32_t TWIN;               /* pixel-window width (== OW for small images) */
    int32_t tiles_x, tiles_p;   /* windows across / pixel blocks down */
    int32_t PH, PW, PWp, PS;    /* LDS patch rows, cols, row pitch, channel pitch (floats) */
    int32_t MRp;                /* LDS pitch of a weight row group */
    int32_t nchunks, mblocks;
    int32_t lds_bytes;
    int64_t packed_floats;      /* size of the packed-weight buffer for this plan, in 4-byte units */
    /* bf16 matrix-core path (precision != JAF_PREC_F32); CK == 8*NG there */
    int32_t precision;
    int32_t NG;                 /* 8-channel groups per LDS chunk (1..4) */
    int32_t ng_last;            /* groups in the last chunk */
    int32_t nsteps, nsteps_last;/* 32-deep MFMA k-steps per chunk: ceil(KH*KW*groups/4) */
    int32_t npos;               /* PH*PW patch positions */
    int32_t plane;              /* bytes of one (split, group) patch plane, multiple of 256 */
    int32_t PWp_slots_unused;   /* reserved */
    int32_t ilv;                /* packed path: a lane's NT tiles are NT consecutive pixels (vector epilogue) */
    int32_t pf;                 /* packed-input kernels: k-steps of a chunk's weights staged in LDS at a time (0: the whole chunk) */
} jaf_conv_plan;

enum { JAF_PACK_FWD = 0, JAF_PACK_DGRAD = 1, JAF_PACK_LSTM = 2, JAF_PACK_DGRAD_LSTM = 3 };

int jaf_conv2d_plan(const jaf_conv_desc* d, int lstm, jaf_conv_plan* plan);

/* Re-lays a weight tensor out in the LDS image order of the plan ([G][mblock][chunk][tap][c][row]).
 * JAF_PACK_FWD: rows = Cout.  JAF_PACK_LSTM: rows = 4*hidden, gate-interleaved so that one
 * accumulator tile holds i,f,o,g of a channel (src/convLSTM.py:46 split order i,f,o,g).
 * JAF_PACK_DGRAD_LSTM (bf16 plans): JAF_PACK_DGRAD for the ConvLSTM's gate gradients as jaf_convlstm_gates_bwd_packed
 * writes them, channel-major (reduction channel 4*c + gate reads weight row gate*hidden + c).
 * JAF_PACK_DGRAD: rows = forward input channels [w_cin_off, +Cout of the dgrad desc), reduction
 * over forward output channels, taps flipped (the transposed convolution).                     */
int jaf_conv2d_pack(jaf_stream_t s, const jaf_conv_desc* d, const jaf_conv_plan* plan, int mode,
                    const float* w, int32_t w_rows_tot, float* packed);

/* Many weight images in one launch (bf16 / bf16x3 plans): the images of a module are re-made right after its optimiser
 * step (the reference's counterpart is cuDNN's per-call filter transform).  jaf_conv2d_pack_item fills ONE entry
 * (jaf_conv2d_pack_item_bytes() bytes, opaque) of a HOST table from the arguments of jaf_conv2d_pack and reports the
 * image's element count; the caller uploads the table once and jaf_conv2d_pack_batch(table_dev, n, max element count)
 * re-packs all n images from the current weights. */
int64_t jaf_conv2d_pack_item_bytes(void);
int jaf_conv2d_pack_item(const jaf_conv_desc* d, const jaf_conv_plan* plan, int mode, const float* w, int32_t w_rows_tot,
                         void* packed, void* item_host, int64_t* total_out);
int jaf_conv2d_pack_batch(jaf_stream_t s, const void* table_dev, int32_t n, int64_t max_total);

int jaf_conv2d_fwd(jaf_stream_t s, const jaf_conv_desc* d, const jaf_conv_plan* plan,
                   const float* src0, const float* src1, const float* src2,
                   const float* packed_w, const float* bias, float* out);

/* Reference implementation of the same contract with one thread per output element and the
 * unpacked weights; used by the GPU tests to cross-check the MFMA kernel. */
int jaf_conv2d_fwd_direct(jaf_stream_t s, const jaf_conv_desc* d,
                          const float* src0, const float* src1, const float* src2,
                          const float* w, const float* bias, float* out);

/* Packed-input path of JAF_PREC_BF16 (csrc/conv_dma.hip).  jaf_conv2d_pack_input converts the
 * (concatenated, grouped) fp32 input of a descriptor ONCE into bf16
 * [N][G][ceil(Cin/8)][H][W][8 channels]; jaf_conv2d_fwd_packed / jaf_convlstm_cell_fwd_packed then
 * stage their LDS patches by DMA with no per-element work.  Plans come from jaf_conv2d_plan_packed;
 * weights are packed by jaf_conv2d_pack with that plan.  The forward conv and the weight gradient
 * of a layer share one packed input; dgrad and wgrad share one packed dz.                      */
int64_t jaf_conv2d_packed_input_bytes(const jaf_conv_desc* d);
int jaf_conv2d_pack_input(jaf_stream_t s, const jaf_conv_desc* d, const float* src0, const float* src1,
                          const float* src2, void* packed);
/* jaf_conv2d_pack_input with bilinear up-sampling fused in: source i with src_h[i] > 0 is given at its low resolution
 * [N][ctot][src_h[i]][src_w[i]] and sampled at the layer's H x W on the fly (ATen upsample_bilinear2d rules,
 * align_corners[i]); src_h[i] == 0: a plain source.  The decoders' `cat[up(x), skip]` (src/networks.py:896-909) and the
 * CRN's `cat[label, pool, up(net)]` (src/crn_model.py:276-299) then never exist at full resolution in fp32. */
int jaf_conv2d_pack_input_resized(jaf_stream_t s, const jaf_conv_desc* d, const float* src0, const float* src1,
                                  const float* src2, const int32_t* src_h, const int32_t* src_w,
                                  const int32_t* align_corners, void* packed);
/* Backward-pass companion of jaf_conv2d_pack_input for a conv output gradient dy [N, G*C, H, W]:
 * dz = dy * act'(y) (y = the activation output, nullable for JAF_ACT_NONE), written as the packed bf16
 * image; dbias[G*C] += per-channel sum of dz (nullable); dz (nullable) receives the fp32 dz for the
 * few layers whose weight gradient still runs on jaf_conv2d_wgrad.  One pass instead of
 * jaf_act_bwd + jaf_channel_sum + jaf_conv2d_pack_input.                                         */
int jaf_conv2d_pack_dz(jaf_stream_t s, const float* dy, const float* y, int32_t N, int32_t G, int32_t C,
                       int32_t H, int32_t W, int act, float slope, void* packed, float* dz, float* dbias);
/* Same with the activation output read from the packed bf16 image the forward epilogue wrote (y_packed: planes
 * y_ng8_tot per (image, group), this tensor's channels from y_coff, a multiple of 8) instead of an fp32 y -- for
 * layers whose fp32 output was never written (jaf_packed_io.skip_f32).  ReLU / LeakyReLU only (the sign survives bf16). */
int jaf_conv2d_pack_dz_ex(jaf_stream_t s, const float* dy, const float* y, const void* y_packed, int32_t y_ng8_tot,
                          int32_t y_coff, int32_t N, int32_t G, int32_t C, int32_t H, int32_t W, int act, float slope,
                          void* packed, float* dz, float* dbias);
/* Same with the image's arithmetic given: JAF_PREC_BF16 (= jaf_conv2d_pack_dz_ex) or JAF_PREC_BF16X3, the split image of
 * the parity-grade mode -- every group of 8 channels as TWO planes, hi = bf16(dz) and right behind it lo = bf16(dz - hi)
 * (twice the bytes; y_packed must be null) -- which jaf_conv2d_fwd_packed_io consumes when its descriptor says JAF_PREC_BF16X3
 * (csrc/conv_dma_split.hip: three matrix-core instructions per operand pair).  jaf_conv2d_pack_input /
 * jaf_conv2d_pack_input_resized / jaf_conv2d_packed_input_bytes take the same switch from d->precision.  No reference
 * counterpart: the reference computes in fp32 (src/networks.py:868-878 and every other nn.Conv2d of the path).           */
int jaf_conv2d_pack_dz_prec(jaf_stream_t s, const float* dy, const float* y, const void* y_packed, int32_t y_ng8_tot,
                            int32_t y_coff, int32_t N, int32_t G, int32_t C, int32_t H, int32_t W, int act, float slope,
                            void* packed, float* dz, float* dbias, int precision);
/* Same with dy's element type given: dy_bf16 = 1 -- dy holds bf16, the gradient of a tensor kept in bf16 storage (JAF_PREC_BF16
 * only; see jaf_packed_io.out_bf16). */
int jaf_conv2d_pack_dz_dt(jaf_stream_t s, const void* dy, int dy_bf16, const float* y, const void* y_packed, int32_t y_ng8_tot,
                          int32_t y_coff, int32_t N, int32_t G, int32_t C, int32_t H, int32_t W, int act, float slope,
                          void* packed, float* dz, float* dbias, int precision);
/* ... and with y_packed a SPLIT-bf16 image (y_split = 1) while the dz image written is plain bf16 (JAF_PREC_BF16): "mixed" arithmetic. */
int jaf_conv2d_pack_dz_dt2(jaf_stream_t s, const void* dy, int dy_bf16, const float* y, const void* y_packed, int32_t y_ng8_tot,
                           int32_t y_coff, int y_split, int32_t N, int32_t G, int32_t C, int32_t H, int32_t W, int act, float slope,
                           void* packed, float* dz, float* dbias, int precision);
/* jaf_convlstm_gates_bwd with the gate gradients written ONLY as the packed bf16 image
 * [N][G][4C/8][H*W][8], CHANNEL-MAJOR: packed channel 4*c + gate (one item = 2 hidden channels x i,f,o,g; its consumers:
 * jaf_conv2d_pack(JAF_PACK_DGRAD_LSTM) + jaf_conv2d_fwd_packed_io for d[x, h], jaf_conv2d_wgrad_packed_lstm for dW), and
 * their per-channel sums ADDED to dbias[G*4C] (reference order gate*C + c);
 * `gates` (as the forward cell wrote them: fp32 planes [N][G][gate][C][H*W], or bf16 gate-innermost
 * [N][G][C][H*W][i, f, o, g] from jaf_convlstm_cell_fwd_packed*) is read-only here.  JAF_EUNSUPPORTED when C % 4 != 0. */
int jaf_convlstm_gates_bwd_packed(jaf_stream_t s, int32_t N, int32_t G, int32_t C, int32_t HW, const float* dh,
                                  const float* dc_next, const void* gates, int gates_bf16, const float* c_prev,
                                  const float* c_cur, float* dc_prev, void* packed, float* dbias);
/* Same with the packed image's arithmetic given (JAF_PREC_BF16 or JAF_PREC_BF16X3: hi + lo planes per channel group, see
 * jaf_conv2d_pack_dz_prec); the saved gates may be fp32 (gates_bf16 = 0) in either. */
int jaf_convlstm_gates_bwd_packed_prec(jaf_stream_t s, int32_t N, int32_t G, int32_t C, int32_t HW, const float* dh,
                                       const float* dc_next, const void* gates, int gates_bf16, const float* c_prev,
                                       const float* c_cur, float* dc_prev, void* packed, float* dbias, int precision);
/* Same with the element types of the tensors given (bf16 STORAGE of BASELINE configs[2], JAF_PREC_BF16 with bf16 gates only):
 * dh_bf16: dh holds bf16 (the d h_{t-1} a data-gradient launch wrote with jaf_packed_io.out2_bf16); state_bf16: c_prev, c_cur,
 * dc_next and dc_prev hold bf16 (jaf_packed_io.state_bf16 of the forward cell).  26 instead of 36 bytes per hidden-channel pixel.
 * Reference: the adjoint of src/convLSTM.py:48-54 (fp32 there). */
int jaf_convlstm_gates_bwd_packed_dt(jaf_stream_t s, int32_t N, int32_t G, int32_t C, int32_t HW, const void* dh,
                                     int dh_bf16, const void* dc_next, const void* gates, int gates_bf16,
                                     const void* c_prev, const void* c_cur, void* dc_prev, int state_bf16, void* packed,
                                     float* dbias, int precision);
int jaf_conv2d_plan_packed(const jaf_conv_desc* d, int lstm, jaf_conv_plan* plan);
/* flags: JAF_PLAN_NO_INTERLEAVE keeps a lane's pixel tiles 16 pixels apart (no pixel interleave): the layout that makes
 * the 16-byte items of a packed bf16 OUTPUT image (jaf_packed_io) contiguous across the lanes of a store -- for launches
 * that write no fp32 output (skip_f32). */
#define JAF_PLAN_NO_INTERLEAVE 1
int jaf_conv2d_plan_packed_ex(const jaf_conv_desc* d, int lstm, int flags, jaf_conv_plan* plan);
int jaf_conv2d_fwd_packed(jaf_stream_t s, const jaf_conv_desc* d, const jaf_conv_plan* plan,
                          const void* packed_in, const void* packed_w, const float* bias, float* out);
/* Same, and the epilogue ADDS (sum, sum of squares) of every image's outputs to stats[n][slot][2] (fp64,
 * slot = workgroup % stat_slots, caller zeroes the buffer): the statistics pass of the CRN LayerNorm that
 * follows (src/crn_model.py:78-87), taken while the values are in registers.  act must be NONE, G == 1.
 * stats == NULL is jaf_conv2d_fwd_packed.                                                              */
int jaf_conv2d_fwd_packed_stats(jaf_stream_t s, const jaf_conv_desc* d, const jaf_conv_plan* plan,
                                const void* packed_in, const void* packed_w, const float* bias, float* out,
                                double* stats, int32_t stat_slots);
int jaf_convlstm_cell_fwd_packed(jaf_stream_t s, const jaf_conv_desc* d, const jaf_conv_plan* plan,
                                 const void* packed_in, const void* packed_w, const float* bias,
                                 const float* c_prev, float* h_out, float* c_out, void* gates_out,
                                 int gates_bf16 /* gates_out is bf16 [N, G*4C, H, W] instead of fp32 */);

/* Packed-image plumbing of the bf16 path: activations that only convolutions consume never exist as fp32 NCHW.
 * A producer (conv / ConvLSTM epilogue, LayerNorm, resize, pool) writes its result straight into the CONSUMER's packed
 * image [N][G][ng8_tot][H*W][8] bf16 at a channel offset -- the concatenation `cat[x, h]`, `cat[up, skip]` is laid out
 * by the producers -- and a consumer may read the leading planes of a larger image (enc_{i+1} reads x_i out of the
 * ConvLSTM's [x_i, h] image).  Values are the RNE bf16 roundings jaf_conv2d_pack_input would have produced.
 * All-zero / NULL = the plain behaviour. */
#define JAF_DZ_BIAS_SLOTS 16
typedef struct jaf_packed_io {
    int32_t in_ng8_tot;      /* planes per (image, group) of the packed INPUT image; 0 = ceil(Cin/8) */
    void* dst;               /* destination packed image of this launch's outputs (NULL: none) */
    int32_t dst_ng8_tot;     /* its planes per (image, group) */
    int32_t dst_coff;        /* first destination channel within the group; multiple of 4 */
    int32_t dst_img_off;     /* destination image index = n + dst_img_off */
    int32_t dst_pad_tail;    /* also write zeros to the channels up to the next multiple of 8 (last source of the image) */
    int32_t skip_f32;        /* do not write the fp32 output tensor (its pointer may be NULL) */
    int32_t accumulate_f32;  /* out += result instead of out = result (plain convolution launches only): the second data
                              * gradient of a tensor with two consumers lands in the first one's buffer, replacing the
                              * separate three-pass add of the autograd engine */
    float* out2;             /* second fp32 output (NULL: none): rows >= split_rows of every group go to out2 [N, G*(Cout-split_rows),
                              * OH, OW], rows below it to `out` taken as [N, G*split_rows, OH, OW] (the descriptor's out_ctot /
                              * out_coff then only have to be valid for Cout rows; they are not used).  The ConvLSTM's
                              * data gradient w.r.t. [x_t, h_{t-1}] in ONE launch: the gate gradients are read once */
    int32_t split_rows;
    /* Backward hand-over in bf16 (data-gradient launches): with `dz_mask` set, what goes to `dst` is not act(result + bias) but
     *   dz = (result [+ *out, when accumulate_f32]) * act'(x),   act' = 1 where x > 0, else dz_slope  (ReLU: 0, LeakyReLU: its slope),
     * i.e. the PRODUCER layer's packed dz -- the activation backward and the bf16 packing that jaf_conv2d_pack_dz would do in a
     * pass of its own, taken while the data gradient is in registers.  x = act(y) is read from the packed bf16 image this
     * launch's consumer layer read in its forward pass (`dz_mask`: dz_mask_ng8 planes per (image, group), the tensor's channels
     * from dz_mask_coff, a multiple of 8; only the sign is used).  With accumulate_f32 the first consumer's gradient is read
     * from `out` and NOT written back (use with skip_f32 = 0 only for that read; nothing is stored in fp32).
     * dz_dbias (nullable): [JAF_DZ_BIAS_SLOTS][G*Cout] += per-channel sums of dz -- the producer's bias gradient, spread over
     * JAF_DZ_BIAS_SLOTS copies that the caller sums (jaf_sum_slots): one atomic per channel and workgroup goes to copy
     * (workgroup index % JAF_DZ_BIAS_SLOTS). */
    const void* dz_mask;
    int32_t dz_mask_ng8, dz_mask_coff;
    float dz_slope;
    float* dz_dbias;
    /* bf16 STORAGE (JAF_PREC_BF16 launches only; BASELINE configs[2] names bf16): the NCHW tensors a launch writes hold bf16
     * instead of fp32 -- `out` (also what accumulate_f32 / the dz mode's first gradient read), `out2`, and for the ConvLSTM cell
     * the state tensors c_prev (read) and c_out (written).  Same shapes and strides in elements; the pointers are passed through
     * the float* / void* parameters.  The statistics of jaf_conv2d_fwd_packed_stats are taken from the unrounded values.
     * Reference: src/convLSTM.py:41-56 (c), src/crn_model.py:78-106 (pre-LayerNorm conv output), all fp32 there.          */
    int32_t out_bf16, out2_bf16, state_bf16;
    int32_t dz_mask_split;   /* JAF_PREC_BF16 launches: `dz_mask` is a SPLIT-bf16 image (the sign is read from its hi planes) -- the forward
                              * ran in JAF_PREC_BF16X3, this data gradient in bf16 ("mixed" arithmetic) */
} jaf_packed_io;
int jaf_conv2d_fwd_packed_io(jaf_stream_t s, const jaf_conv_desc* d, const jaf_conv_plan* plan,
                             const void* packed_in, const void* packed_w, const float* bias, float* out,
                             double* stats, int32_t stat_slots, const jaf_packed_io* io);
/* ConvLSTM cell: `io->dst` receives h_t (4 * hidden = Cout rows -> hidden channels); skip_f32 drops the fp32 h_out
 * (c_out and the saved gates are always written).  gates_bf16: the saved gates are bf16 and gate-innermost,
 * [N][G][hidden][H*W][i, f, o, g] (a lane's pixels x 4 gates are contiguous: 16-byte stores here, one 16-byte load per pixel
 * pair in jaf_convlstm_gates_bwd_packed); fp32 gates keep the planes [N][G][gate][hidden][H*W]. */
int jaf_convlstm_cell_fwd_packed_io(jaf_stream_t s, const jaf_conv_desc* d, const jaf_conv_plan* plan,
                                    const void* packed_in, const void* packed_w, const float* bias,
                                    const float* c_prev, float* h_out, float* c_out, void* gates_out,
                                    int gates_bf16, const jaf_packed_io* io);

/* 3x3 weight gradient from the packed input of the forward conv and the packed dz of the data
 * gradient (csrc/wgrad_dma.hip).  Returns JAF_EUNSUPPORTED for shapes it does not cover (stride-2
 * layers whose patch does not fit: none of the 1x1 / 3x3 / 5x5 shapes today); jaf_conv2d_wgrad covers 7x7.   */
int jaf_conv2d_wgrad_packed(jaf_stream_t s, const jaf_conv_desc* d, const void* packed_x,
                            const void* packed_dz, float* dw, int accumulate);
/* Same with packed_x an image of x_ng8_tot >= ceil(Cin/8) planes per (image, group) (see jaf_packed_io). */
int jaf_conv2d_wgrad_packed_ex(jaf_stream_t s, const jaf_conv_desc* d, const void* packed_x, int32_t x_ng8_tot,
                               const void* packed_dz, float* dw, int accumulate);
/* The same for the ConvLSTM's weight gradient, with packed_dz = the gate gradients as jaf_convlstm_gates_bwd_packed writes
 * them (channel 4*c + gate): row gate*hidden + c of dW receives channel 4*c + gate (hidden = 0: jaf_conv2d_wgrad_packed_ex). */
int jaf_conv2d_wgrad_packed_lstm(jaf_stream_t s, const jaf_conv_desc* d, const void* packed_x, int32_t x_ng8_tot,
                                 const void* packed_dz, float* dw, int accumulate, int32_t hidden);
/* The same with a caller-owned workspace for split-K partial sums: where the atomic traffic of a launch (pixel splits x dW) is
 * large, every pixel split stores its block of dW to its own copy in `workspace` and one reduction pass adds the copies to dW --
 * stores and loads at HBM rate instead of fp32 atomics at the memory side's ~1.3 TB/s, and a fixed summation order (bit-stable
 * weight gradients for those layers).  jaf_conv2d_wgrad_packed_ws_bytes: the bytes such a launch uses (0: the layer stays on atomics;
 * a smaller or NULL workspace also falls back to atomics).  The workspace is scratch: no contents survive the call's stream work. */
int64_t jaf_conv2d_wgrad_packed_ws_bytes(const jaf_conv_desc* d, int32_t hidden);
int jaf_conv2d_wgrad_packed_ws(jaf_stream_t s, const jaf_conv_desc* d, const void* packed_x, int32_t x_ng8_tot,
                               const void* packed_dz, float* dw, int accumulate, int32_t hidden, void* workspace,
                               int64_t workspace_bytes);
/* Same with the layout of packed_x given: x_split = 1 -- packed_x is a SPLIT-bf16 image (hi and lo plane per channel group, as the
 * JAF_PREC_BF16X3 forward made it) of which this JAF_PREC_BF16 launch reads the hi planes only (a hi plane is exactly the bf16
 * image): the "mixed" arithmetic of the host mirror -- parity-grade forward, bf16 backward. */
int jaf_conv2d_wgrad_packed_ws_x(jaf_stream_t s, const jaf_conv_desc* d, const void* packed_x, int32_t x_ng8_tot, int x_split,
                                 const void* packed_dz, float* dw, int accumulate, int32_t hidden, void* workspace,
                                 int64_t workspace_bytes);

/* dW[G][Cout][w_cin_tot][KH][KW] (+)= sum over n,pixels of dz * input patch; dz is laid out as
 * the forward output (out_ctot/out_coff).  accumulate=0 zeroes the touched slice first.        */
int jaf_conv2d_wgrad(jaf_stream_t s, const jaf_conv_desc* d,
                     const float* src0, const float* src1, const float* src2,
                     const float* dz, float* dw, int accumulate);

/* db[c] (+)= sum over n,h,w of x[n, coff + c, h, w] for c in [0, C). */
int jaf_channel_sum(jaf_stream_t s, const float* x, int32_t N, int32_t ctot, int32_t coff,
                    int32_t C, int32_t HW, float* out, int accumulate);

/* ConvLSTM cell, src/convLSTM.py:41-56 in one kernel: gates = conv3x3(cat[x, h_prev]) + b,
 * i,f,o = sigmoid, g = tanh, c = f*c_prev + i*g, h = o*tanh(c).  d->Cout == 4*hidden.
 * h_prev == NULL and c_prev == NULL mean the zero state of init_hidden (:58-63) and skip the
 * zero half of the reduction.  gates_out (nullable) receives the post-activation gates
 * [N, G*4*hidden, H, W] (order i,f,o,g per group) for the backward pass.                         */
int jaf_convlstm_cell_fwd(jaf_stream_t s, const jaf_conv_desc* d, const jaf_conv_plan* plan,
                          const float* x, const float* h_prev, const float* packed_w,
                          const float* bias, const float* c_prev,
                          float* h_out, float* c_out, float* gates_out);

/* Backward of the gate math: given dh, dc_next (nullable), gates (i,f,o,g), c_prev (nullable),
 * c_cur: writes the pre-activation gate gradients in place of `gates` and dc_prev.               */
int jaf_convlstm_gates_bwd(jaf_stream_t s, int32_t N, int32_t G, int32_t C, int32_t HW,
                           const float* dh, const float* dc_next, float* gates,
                           const float* c_prev, const float* c_cur, float* dc_prev);

/* ------------------------------------------------------------------------------------------
 * Activations and normalisation.
 * ------------------------------------------------------------------------------------------ */
/* dz = dy * act'(y) given the activation OUTPUT y (lrelu/relu/sigmoid/tanh). */
int jaf_act_bwd(jaf_stream_t s, const float* dy, const float* y, float* dz, int64_t n, int act,
                float slope);

/* CRN LayerNorm (src/crn_model.py:78-87) + LeakyReLU(0.01) (:100): per sample mean and
 * Bessel-corrected std over C*H*W, y = lrelu(gamma_c*(x-mean)/(std+eps)+beta_c).
 * stats[n] = {mean, 1/(std+eps)} (float2).                                                      */
int jaf_layernorm_stats(jaf_stream_t s, const float* x, int32_t N, int64_t chw, float eps,
                        double* workspace /* 2*N doubles */, float* stats /* 2*N */);
/* stats[n] from sums[n][slot][2] = (sum, sum of squares) accumulated by jaf_conv2d_fwd_packed_stats; the sums are
 * set back to zero, so a buffer zeroed once can be handed to the next jaf_conv2d_fwd_packed_stats as it is.       */
int jaf_layernorm_finalize(jaf_stream_t s, double* sums, int32_t N, int32_t slots, int64_t chw, float eps,
                           float* stats /* 2*N */);
int jaf_layernorm_lrelu_fwd(jaf_stream_t s, const float* x, const float* stats, const float* gamma,
                            const float* beta, float* y, int32_t N, int32_t C, int32_t HW,
                            float slope);
/* Same, and the result is ALSO written (RNE bf16) into channels [dst_coff, dst_coff + C) of the consumer convolution's
 * packed image (groups == 1; dst_coff a multiple of 8; the channels up to the next multiple of 8 are zeroed); y may be
 * NULL when nothing reads the fp32 result. */
int jaf_layernorm_lrelu_fwd_packed(jaf_stream_t s, const float* x, const float* stats, const float* gamma,
                                   const float* beta, float* y, void* dst, int32_t dst_ng8_tot, int32_t dst_coff,
                                   int32_t N, int32_t C, int32_t HW, float slope);
/* Same with the destination image's arithmetic given: JAF_PREC_BF16 (= jaf_layernorm_lrelu_fwd_packed) or JAF_PREC_BF16X3,
 * a split image -- hi = bf16(v) and lo = bf16(v - hi) planes per channel group (see jaf_conv2d_pack_dz_prec). */
int jaf_layernorm_lrelu_fwd_packed_prec(jaf_stream_t s, const float* x, const float* stats, const float* gamma,
                                        const float* beta, float* y, void* dst, int32_t dst_ng8_tot, int32_t dst_coff,
                                        int32_t N, int32_t C, int32_t HW, float slope, int precision);
/* Backward through lrelu + affine + normalisation.  x is the conv output (pre-norm).
 * dgamma/dbeta are accumulated (+=).                                                            */
int jaf_layernorm_lrelu_bwd(jaf_stream_t s, const float* dy, const float* x, const float* stats,
                            const float* gamma, const float* beta, float* dx, float* dgamma,
                            float* dbeta, double* workspace /* 32*N doubles */, int32_t N,
                            int32_t C, int32_t HW, float slope, float eps);

/* The same backward when x is the output of a convolution without activation whose only reader is this LayerNorm (the
 * CRN's conv -> LayerNorm -> LeakyReLU blocks, src/crn_model.py:90-106): dx leaves as that convolution's packed bf16 dz image
 * [n][ceil(C/8)][HW][8] (channels up to the next multiple of 8 zeroed) instead of an fp32 tensor, and its bias gradient
 * (sum of dx over images and pixels; conv_dbias nullable, `accumulate_dbias`: += instead of =) comes out of the reduction
 * pass.  scratch: 2*N*C floats. */
int jaf_layernorm_lrelu_bwd_packed(jaf_stream_t s, const float* dy, const float* x, const float* stats,
                                   const float* gamma, const float* beta, void* packed_dx, float* dgamma,
                                   float* dbeta, double* workspace /* 32*N doubles */, float* scratch, float* conv_dbias,
                                   int accumulate_dbias, int32_t N, int32_t C, int32_t HW, float slope, float eps);
/* Same with the packed dx image's arithmetic given (JAF_PREC_BF16 or JAF_PREC_BF16X3: hi + lo planes, see jaf_conv2d_pack_dz_prec). */
int jaf_layernorm_lrelu_bwd_packed_prec(jaf_stream_t s, const float* dy, const float* x, const float* stats,
                                        const float* gamma, const float* beta, void* packed_dx, float* dgamma,
                                        float* dbeta, double* workspace, float* scratch, float* conv_dbias,
                                        int accumulate_dbias, int32_t N, int32_t C, int32_t HW, float slope, float eps,
                                        int precision);
/* The LayerNorm entry points with the element types of their NCHW tensors given (bf16 STORAGE of BASELINE configs[2]; JAF_PREC_BF16
 * only): x_bf16 -- x, the pre-LayerNorm convolution output (jaf_packed_io.out_bf16), holds bf16; dy_bf16 -- the incoming gradient
 * holds bf16 (the consumer convolution's data gradient written with out_bf16); jaf_layernorm_lrelu_bwd_dt writes dx in x's type.
 * Statistics, gamma / beta and their gradients stay fp32 / fp64.  Reference: src/crn_model.py:67-87 (fp32 there). */
int jaf_layernorm_lrelu_fwd_dt(jaf_stream_t s, const void* x, int x_bf16, const float* stats, const float* gamma,
                               const float* beta, float* y, int32_t N, int32_t C, int32_t HW, float slope);
int jaf_layernorm_lrelu_fwd_packed_dt(jaf_stream_t s, const void* x, int x_bf16, const float* stats, const float* gamma,
                                      const float* beta, float* y, void* dst, int32_t dst_ng8_tot, int32_t dst_coff,
                                      int32_t N, int32_t C, int32_t HW, float slope, int precision);
int jaf_layernorm_lrelu_bwd_dt(jaf_stream_t s, const void* dy, int dy_bf16, const void* x, int x_bf16, const float* stats,
                               const float* gamma, const float* beta, void* dx, float* dgamma,
                               float* dbeta, double* workspace, int32_t N, int32_t C, int32_t HW,
                               float slope, float eps);
int jaf_layernorm_lrelu_bwd_packed_dt(jaf_stream_t s, const void* dy, int dy_bf16, const void* x, int x_bf16,
                                      const float* stats, const float* gamma, const float* beta, void* packed_dx,
                                      float* dgamma, float* dbeta, double* workspace, float* scratch, float* conv_dbias,
                                      int accumulate_dbias, int32_t N, int32_t C, int32_t HW, float slope, float eps,
                                      int precision);

/* BatchNorm2d in training mode (src/flow_net.py:13-51, src/networks.py:369-390; eps 1e-5,
 * momentum 0.1, biased var for normalisation, unbiased for running_var) + activation
 * (+ optional residual add: ResnetBlock, src/flow_net.py:139-141).
 * stats = {mean[C], rstd[C]} (2*C floats).  training=0 normalises with running stats.          */
int jaf_batchnorm_stats(jaf_stream_t s, const float* x, int32_t N, int32_t C, int32_t HW,
                        float eps, float momentum, float* running_mean, float* running_var,
                        float* stats, int training, double* workspace /* 2*C doubles */);
int jaf_batchnorm_act_fwd(jaf_stream_t s, const float* x, const float* stats, const float* weight,
                          const float* bias, const float* residual, float* y, int32_t N,
                          int32_t C, int32_t HW, int act, float slope);
/* jaf_batchnorm_stats + jaf_batchnorm_act_fwd as ONE call: feature maps of at most 65536 elements per channel (the
 * discriminators' 4 x 4 .. 64 x 64 levels) take the statistics and apply them in a single launch, one workgroup per channel;
 * larger ones run the two calls above.  Same arguments, same results.  jaf_batchnorm_act_bwd makes the same choice. */
int jaf_batchnorm_act_fwd_fused(jaf_stream_t s, const float* x, int32_t N, int32_t C, int32_t HW, float eps,
                                float momentum, float* running_mean, float* running_var, float* stats,
                                int training, double* workspace, const float* weight, const float* bias,
                                const float* residual, float* y, int act, float slope);
int jaf_batchnorm_act_bwd(jaf_stream_t s, const float* dy, const float* x, const float* y,
                          const float* stats, const float* weight, float* dx, float* dweight,
                          float* dbias, int32_t N, int32_t C, int32_t HW, int act, float slope,
                          int training, double* workspace /* 2*C doubles */,
                          int accumulate /* 1: dweight/dbias += (parameter .grad buffers) */);
/* `parts` (<= 4) equal chunks of the batch, each normalised with its own batch statistics, the running statistics updated chunk after
 * chunk -- what `parts` successive training-mode calls of the two functions above on the chunks compute (the reference runs its
 * discriminators on the real and on the generated images in separate calls, train/4.convLSTM_flowpro_interval.py:380-394) -- in ONE launch,
 * bit-identical to the per-chunk calls.  stats: [parts][2*C].  JAF_EUNSUPPORTED above 65536 elements per channel and chunk. */
int jaf_batchnorm_act_fwd_split(jaf_stream_t s, const float* x, int32_t N, int32_t C, int32_t HW, float eps, float momentum,
                                float* running_mean, float* running_var, float* stats, const float* weight, const float* bias,
                                float* y, int act, float slope, int32_t parts);
int jaf_batchnorm_act_bwd_split(jaf_stream_t s, const float* dy, const float* x, const float* y, const float* stats,
                                const float* weight, float* dx, float* dweight, float* dbias, int32_t N, int32_t C, int32_t HW,
                                int act, float slope, int training, int accumulate, int32_t parts);

/* ------------------------------------------------------------------------------------------
 * Resampling.
 * ------------------------------------------------------------------------------------------ */
/* F.avg_pool2d(3,stride 2,pad 1,count_include_pad) src/crn_model.py:268-273; k=2: VGG AvgPool2d(2,2)
 * src/networks.py:76-78. */
int jaf_avgpool_fwd(jaf_stream_t s, const float* x, float* y, int32_t NC, int32_t H, int32_t W,
                    int32_t OH, int32_t OW, int32_t k, int32_t stride, int32_t pad);
int jaf_avgpool_bwd(jaf_stream_t s, const float* dy, float* dx, int32_t NC, int32_t H, int32_t W,
                    int32_t OH, int32_t OW, int32_t k, int32_t stride, int32_t pad);

/* Bilinear resize of a crop window [y0,y0+ch) x [x0,x0+cw) of x[NC,H,W] to [NC,OH,OW];
 * align_corners per call site.  mode 1 = nearest (face IUV, train/4...py:350).
 * The output may be a channel slice of a larger tensor: y index = ((n*out_ctot + out_coff + c)).*/
int jaf_resize_fwd(jaf_stream_t s, const float* x, float* y, int32_t N, int32_t C, int32_t H,
                   int32_t W, int32_t y0, int32_t x0, int32_t ch, int32_t cw, int32_t OH,
                   int32_t OW, int align_corners, int nearest);
int jaf_resize_bwd(jaf_stream_t s, const float* dy, float* dx, int32_t N, int32_t C, int32_t H,
                   int32_t W, int32_t y0, int32_t x0, int32_t ch, int32_t cw, int32_t OH,
                   int32_t OW, int align_corners);
/* Same with dy's element type given (dy_bf16 = 1: bf16, the gradient of a tensor in bf16 storage); dx stays fp32. */
int jaf_resize_bwd_dt(jaf_stream_t s, const void* dy, int dy_bf16, float* dx, int32_t N, int32_t C, int32_t H, int32_t W,
                      int32_t y0, int32_t x0, int32_t ch, int32_t cw, int32_t OH, int32_t OW, int align_corners);

/* nn.ReflectionPad2d(p) (src/flow_net.py:13,51,113) and its adjoint. */
int jaf_reflect_pad_fwd(jaf_stream_t s, const float* x, float* y, int32_t NC, int32_t H, int32_t W,
                        int32_t p);
int jaf_reflect_pad_bwd(jaf_stream_t s, const float* dy, float* dx, int32_t NC, int32_t H,
                        int32_t W, int32_t p);

/* ------------------------------------------------------------------------------------------
 * Gathers and blends (HBM-bound wavefront kernels).
 * ------------------------------------------------------------------------------------------ */
/* texture_warp_pytorch, train/4.convLSTM_flowpro_interval.py:43-76, all 24 parts and the whole
 * batch in one pass.  tex: [B, 24*3, TH, TW] (part p = channels 3p..3p+2), iuv: uint8 [B,S,S,3]
 * HWC (I,U,V), out: [B,3,S,S].  x=((255-V)/255-.5)*2, y=(U/255-.5)*2, bilinear, zeros padding. */
int jaf_texture_warp_fwd(jaf_stream_t s, const float* tex, const uint8_t* iuv, float* out,
                         int32_t B, int32_t S, int32_t TH, int32_t TW, int align_corners);
/* dtex must be zero-filled by the caller; scatter-add of dout at the sampled taps. */
int jaf_texture_warp_bwd(jaf_stream_t s, const float* dout, const uint8_t* iuv, float* dtex,
                         int32_t B, int32_t S, int32_t TH, int32_t TW, int align_corners);

/* F.grid_sample(bilinear) of src[B,C,H,W] at grid[B,OH,OW,2]; padding 0 = zeros, 1 = border
 * (src/cal_flow.py:38). */
int jaf_grid_sample_fwd(jaf_stream_t s, const float* src, const float* grid, float* out,
                        int32_t B, int32_t C, int32_t H, int32_t W, int32_t OH, int32_t OW,
                        int padding_border, int align_corners);
/* Its adjoint (ATen grid_sampler_2d_backward): dsrc[B,C,H,W] += by scatter-add (nullable), dgrid[B,OH,OW,2]
 * (nullable); border clamping passes no gradient where it clipped.  Stage 4 never differentiates the flow warp
 * (SURVEY App. D); this is the "differentiable flow" of SURVEY 8(f1): gradient to the SMPL vertices through T. */
int jaf_grid_sample_bwd(jaf_stream_t s, const float* dout, const float* src, const float* grid, float* dsrc,
                        float* dgrid, int32_t B, int32_t C, int32_t H, int32_t W, int32_t OH, int32_t OW,
                        int padding_border, int align_corners);

/* out = a*m + b*(1-m), m broadcast over C (train/4...py:321; src/flow_net.py:98). */
int jaf_blend_fwd(jaf_stream_t s, const float* a, const float* b, const float* m, float* out,
                  int32_t N, int32_t C, int32_t HW);
/* da = dout*m, db = dout*(1-m) (either nullable), dm = sum_c dout*(a-b). */
int jaf_blend_bwd(jaf_stream_t s, const float* dout, const float* a, const float* b,
                  const float* m, float* da, float* db, float* dm, int32_t N, int32_t C,
                  int32_t HW);
/* out[n,c,i] = x[n,c,i] * m[n,(mc==1?0:c),i]  (mask premultiply src/flow_net.py:91; common-area
 * multiply train/4...py:295-298 through jaf_part_mask_mul). */
int jaf_mul_bcast(jaf_stream_t s, const float* x, const float* m, float* out, int32_t N, int32_t C,
                  int32_t MC, int32_t HW);
/* Common-area mask logic train/4...py:283-298: area = OR_t (mask[b,t] != 0 && used[t]) over the
 * 800x1200 atlas; out[b, 3p+c, y, x] = tex[b,3p+c,y,x] * area[b, (p/6)*200+y, (p%6)*200+x].
 * masks: float [B,T,AH,AW]; used: int32 [T].                                                   */
int jaf_part_mask_mul(jaf_stream_t s, const float* tex, const float* masks, const int32_t* used,
                      float* out, int32_t B, int32_t T, int32_t AH, int32_t AW, int32_t P,
                      int32_t PS_);
/* list[24] of (B,3,200,200) parts <-> atlas [B,T,3,800,1200] (train/4...py:269-276). */
int jaf_atlas_to_parts(jaf_stream_t s, const float* atlas, float* parts, int32_t B, int32_t T,
                       int32_t AH, int32_t AW, int32_t PSZ);
/* The same slicing (train/4...py:269-276) written as the packed input image of the first part-encoder convolution
 * (src/networks.py:1294 enc1): bf16 [T*B][24][1][PSZ*PSZ][8], channels 3..7 zero; JAF_PREC_BF16X3: hi plane + residual plane.
 * JAF_EUNSUPPORTED when PSZ or AW is not a multiple of 4 or a pointer is not 16-byte aligned (use jaf_atlas_to_parts +
 * jaf_conv2d_pack_input then).                                                                                            */
int jaf_atlas_to_parts_packed(jaf_stream_t s, const float* atlas, void* image, int32_t B, int32_t T, int32_t AH, int32_t AW,
                              int32_t PSZ, int32_t precision);

/* ------------------------------------------------------------------------------------------
 * Renderer path (src/cal_flow.py:28-35, src/nmr.py:263-278,617-659; NMR kernels
 * rasterize_cuda_kernel.cu:24-169).
 * ------------------------------------------------------------------------------------------ */
/* proj + y-flip + look_at (eye (0,0,-(1/tan30+1)), identity rotation) + vertices_to_faces:
 * verts[B,NV,3], cam[B,3], faces_idx int32[NF,3] -> faces[B,NF,3,3]. */
int jaf_project_faces(jaf_stream_t s, const float* verts, const float* cam,
                      const int32_t* faces_idx, float* faces, int32_t B, int32_t NV, int32_t NF,
                      float eye_z);
/* Face-index / weight map of `faces` at S x S, outputs vertically flipped as rasterize.py:334-338.
 * fim int32 [B,S,S] (-1 background), wim [B,S,S,3].  workspace: see jaf_rasterize_workspace. */
int64_t jaf_rasterize_workspace(int32_t B, int32_t NF, int32_t S);
int jaf_rasterize_fim_wim(jaf_stream_t s, const float* faces, int32_t* fim, float* wim,
                          void* workspace, int32_t B, int32_t NF, int32_t S, float near_, float far_);
/* The whole forward_face_index_map of the reference FFI (rasterize_cuda.cpp:70-95): besides fim / wim the
 * depth map (far where nothing is hit, rasterize.py:52), the per-pixel inverse of the winning face
 * (face_inv_map [B,S,S,3,3], return_depth) and the alpha map (rasterize.py:188-192); each of the three is
 * nullable.  flip = 0 writes the maps as RasterizeFunction saves them for backward, flip = 1 vertically flipped
 * as rasterize_rgbad returns them (rasterize.py:334-338). */
int jaf_rasterize_maps(jaf_stream_t s, const float* faces, int32_t* fim, float* wim, float* depth,
                       float* face_inv_map, float* alpha, void* workspace, int32_t B, int32_t NF, int32_t S,
                       float near_, float far_, int flip);
/* backward_pixel_map (rasterize_cuda.cpp:124-147, kernel rasterize_cuda_kernel.cu:245-491): grad_faces[B,NF,3,3]
 * (caller zeroes it; front faces are overwritten) from the UNFLIPPED maps; the rgb pair and the alpha pair are
 * each nullable (return_rgb / return_alpha), at least one must be given. */
int jaf_rasterize_bwd_pixel_map(jaf_stream_t s, const float* faces, const int32_t* face_index_map,
                                const float* rgb_map, const float* alpha_map, const float* grad_rgb_map,
                                const float* grad_alpha_map, float* grad_faces, int32_t B, int32_t NF, int32_t S,
                                float eps);
/* backward_depth_map (rasterize_cuda.cpp:169-190, kernel :537-593): ADDS to grad_faces. */
int jaf_rasterize_bwd_depth_map(jaf_stream_t s, const float* faces, const float* depth_map,
                                const int32_t* face_index_map, const float* face_inv_map, const float* weight_map,
                                const float* grad_depth_map, float* grad_faces, int32_t B, int32_t NF, int32_t S);
/* forward_texture_sampling of the reference FFI (rasterize_cuda.cpp:97-122, kernel rasterize_cuda_kernel.cu:171-243)
 * fused with forward_background (rasterize.py:194-202): rgb_map [B,S,S,3] = trilinear sample of the hit face's texture
 * block textures[B,NF,ts,ts,ts,3] at (weight_k * (ts-1) * depth / z_k), `background` ([3], or [B,3] when bg_per_image)
 * where no face is hit.  Maps UNFLIPPED.  sampling_index_map int32 [B,S,S,8] / sampling_weight_map [B,S,S,8] are the
 * reference's saved-for-backward maps; both NULLABLE here -- jaf_rasterize_texture_bwd_rebuild re-derives them. */
int jaf_rasterize_texture_fwd(jaf_stream_t s, const float* faces, const float* textures, const int32_t* face_index_map,
                              const float* weight_map, const float* depth_map, float* rgb_map,
                              int32_t* sampling_index_map, float* sampling_weight_map, const float* background,
                              int bg_per_image, int32_t B, int32_t NF, int32_t S, int32_t ts, float eps);
/* backward_textures (rasterize_cuda.cpp:149-167, kernel :506-541): grad_textures[B,NF,ts,ts,ts,3] += w * grad_rgb. */
int jaf_rasterize_texture_bwd(jaf_stream_t s, const int32_t* face_index_map, const float* sampling_weight_map,
                              const int32_t* sampling_index_map, const float* grad_rgb_map, float* grad_textures,
                              int32_t B, int32_t NF, int32_t S, int32_t ts);
/* The same adjoint without the two sampling maps: taps rebuilt from the face / weight / depth maps (saves 128 B/pixel). */
int jaf_rasterize_texture_bwd_rebuild(jaf_stream_t s, const float* faces, const int32_t* face_index_map,
                                      const float* weight_map, const float* depth_map, const float* grad_rgb_map,
                                      float* grad_textures, int32_t B, int32_t NF, int32_t S, int32_t ts, float eps);
/* neural_renderer.lighting (lighting.py:6-58; src/nmr.py:219-229): textures_out = textures_in * light with
 * light[b,f,c] = ia*ca[c] + id*cd[c]*relu(normal_f . direction), normal_f = normalize((v0-v1) x (v2-v1), eps 1e-5) of
 * faces[B,NF,3,3].  The three colour / direction arguments are HOST float[3]; in == out is allowed (the reference
 * multiplies in place); light_out [B,NF,3] nullable. */
int jaf_lighting_fwd(jaf_stream_t s, const float* faces, const float* textures_in, float* textures_out, float* light_out,
                     float intensity_ambient, float intensity_directional, const float* color_ambient,
                     const float* color_directional, const float* direction, int32_t B, int32_t NF, int32_t ts);
/* Its adjoint: grad_textures_in = grad_out * light (nullable), grad_faces[B,NF,3,3] (nullable; zero unless id != 0). */
int jaf_lighting_bwd(jaf_stream_t s, const float* faces, const float* textures_in, const float* grad_out,
                     float* grad_textures_in, float* grad_faces, float intensity_ambient, float intensity_directional,
                     const float* color_ambient, const float* color_directional, const float* direction, int32_t B,
                     int32_t NF, int32_t ts);
/* SMPLRenderer.dynamic_sampler (src/nmr.py:388-395, :445-477): sampler[B,NF,TT,2] = clamp(p2 + (p0-p2)*coords[0,j] +
 * (p1-p2)*coords[1,j], -1, 1), p_k = cam_s * (verts[faces_idx[f,k]].xy + cam_t); coords [2,TT] (create_coords, :479-495). */
int jaf_face_sampler_fwd(jaf_stream_t s, const float* verts, const float* cam, const int32_t* faces_idx,
                         const float* coords, float* sampler, int32_t B, int32_t NV, int32_t NF, int32_t TT);
/* Its adjoint: dverts[B,NV,3] += (x, y), dcam[B,3] += (nullable). */
int jaf_face_sampler_bwd(jaf_stream_t s, const float* verts, const float* cam, const int32_t* faces_idx,
                         const float* coords, const float* grad_sampler, float* dverts, float* dcam, int32_t B, int32_t NV,
                         int32_t NF, int32_t TT);
/* The layout half of SMPLRenderer.extract_tex (src/nmr.py:379-384): sampled[B,3,NF,T*T] -> tex[B,NF,T,T,T,3]. */
int jaf_tex_expand_fwd(jaf_stream_t s, const float* sampled, float* tex, int32_t B, int32_t NF, int32_t T);
int jaf_tex_expand_bwd(jaf_stream_t s, const float* grad_tex, float* grad_sampled, int32_t B, int32_t NF, int32_t T);
/* neural_renderer.vertices_to_faces (vertices_to_faces.py:4-22): faces[B,NF,3,3] = verts[b, faces_idx[f,k]]; the adjoint
 * adds into dverts[B,NV,3]. */
int jaf_vertices_to_faces(jaf_stream_t s, const float* verts, const int32_t* faces_idx, float* faces, int32_t B, int32_t NV,
                          int32_t NF);
int jaf_vertices_to_faces_bwd(jaf_stream_t s, const float* dfaces, const int32_t* faces_idx, float* dverts, int32_t B,
                              int32_t NV, int32_t NF);
/* Adjoint of jaf_project_faces (autograd of src/nmr.py:269-276 in the reference): dverts[B,NV,3] += , dcam[B,3] +=
 * (nullable). */
int jaf_project_faces_bwd(jaf_stream_t s, const float* dfaces, const float* verts, const float* cam,
                          const int32_t* faces_idx, float* dverts, float* dcam, int32_t B, int32_t NV, int32_t NF);
/* Adjoint of jaf_bc_transform w.r.t. the source faces (autograd of src/nmr.py:651-653): dsrc_faces[B,NF,3,3] +=. */
int jaf_bc_transform_bwd(jaf_stream_t s, const float* dT, const int32_t* fim, const float* wim, float* dsrc_faces,
                         int32_t B, int32_t NF, int32_t S);
/* float_estimate.forward fused (src/cal_flow.py:28-39): out[B,C,S,S] = grid_sample(src[B,C,H,W], T, border) with T the
 * barycentric flow of jaf_bc_transform computed per pixel and never stored; `mask` (nullable, [B,mask_c,S,S], mask_c 1 or C)
 * multiplies the result (src/flow_net.py:91).  Forward only. */
int jaf_flow_warp_fwd(jaf_stream_t s, const float* src, const float* src_faces, const int32_t* fim, const float* wim,
                      const float* mask, float* out, int32_t B, int32_t C, int32_t H, int32_t W, int32_t NF, int32_t S,
                      int32_t mask_c, int align_corners);
/* cal_bc_transform fused with the y re-flip of src/cal_flow.py:30-31: T[B,S,S,2]. */
int jaf_bc_transform(jaf_stream_t s, const float* src_faces /*[B,NF,3,3]*/, const int32_t* fim,
                     const float* wim, float* T, int32_t B, int32_t NF, int32_t S);

/* ------------------------------------------------------------------------------------------
 * Device-side input pipeline (src/data.py:640-773, src/utils.py:369-394, train/4...py:216-237): raw uint8 frames
 * in, the tensors of a stage-4 batch out.  The host keeps only what it needs as host integers (face boxes).
 * ------------------------------------------------------------------------------------------ */
/* in uint8 [N][HW][C] (cv2.imread layout, C = 1 or 3) -> out fp32 [N][C][HW]; mode 0: (x/255 - 0.5)*2
 * (src/data.py:746-750), mode 1: x/255 (:751, :739); evaluated in float64 then cast, like the reference. */
int jaf_u8_hwc_to_f32_chw(jaf_stream_t s, const uint8_t* in, float* out, int32_t N, int32_t HW, int32_t C, int mode);
/* out fp32 [N][3][S][S] = 1 where IUV part index (channel 0 of uint8 [N][S][S][3]) is 1..24: TransferTexture of an
 * all-ones atlas (src/data.py:690-695), i.e. src_mask_in_image / tgt_mask_in_image. */
int jaf_iuv_part_mask(jaf_stream_t s, const uint8_t* iuv, float* out, int32_t N, int32_t S);
/* TransferTexture (src/utils.py:369-394) on uint8: tex [N or 1][AH][AW][3] (4 x 6 cells), iuv [N][S][S][3], optional
 * background im [N][S][S][3] -> out [N][S][S][3]. */
int jaf_transfer_texture_u8(jaf_stream_t s, const uint8_t* tex, const uint8_t* iuv, const uint8_t* im, uint8_t* out,
                            int32_t N, int32_t S, int32_t AH, int32_t AW, int tex_batched);

/* ------------------------------------------------------------------------------------------
 * Evaluation metrics (test/video_evaluation.py:165-212; third-party arithmetic restated: OpenCV BGR2GRAY,
 * scikit-image 0.16.2 compare_ssim, scikit-video 1.1.11 psnr / msssim -- requirements.txt).
 * ------------------------------------------------------------------------------------------ */
/* cv2.cvtColor(img, COLOR_BGR2GRAY) on uint8 [npix][3] -> [npix] (video_evaluation.py:169-170). */
int jaf_bgr_to_gray_u8(jaf_stream_t s, const uint8_t* in, uint8_t* out, int64_t npix);
/* sums[n] += (sum of SSIM, sum of contrast-structure) over the valid win x win windows of the fp32 pair x, y [N][H][W];
 * weights[win*win] fp64, cov_norm = NP/(NP-1) for skimage's sample covariance or 1; caller zeroes sums[N][2]. */
int jaf_ssim_window_sums(jaf_stream_t s, const float* x, const float* y, const double* weights, double* sums,
                         int32_t N, int32_t H, int32_t W, int32_t win, double cov_norm, double C1, double C2);
/* sums[n] += (sum (a-b)^2, sum |a-b|) of uint8 frames [N][P] (exact); caller zeroes sums[N][2]. */
int jaf_frame_error_sums_u8(jaf_stream_t s, const uint8_t* a, const uint8_t* b, double* sums, int32_t N, int64_t P);

/* ------------------------------------------------------------------------------------------
 * Losses, classifier head, optimiser.
 * ------------------------------------------------------------------------------------------ */
/* vgg_preprocess src/networks.py:109-116: y = 255*(x+1)/2 - mean[c]. */
int jaf_vgg_preprocess(jaf_stream_t s, const float* x, float* y, int32_t N, int32_t HW);
/* loss[0] += w * mean|a-b|;  da (nullable) = w*sign(a-b)/n  (L1Loss, src/networks.py:99,122). */
int jaf_l1_loss_fwd(jaf_stream_t s, const float* a, const float* b, int64_t n, float w,
                    float* loss_accum);
int jaf_l1_loss_bwd(jaf_stream_t s, const float* a, const float* b, int64_t n, float w,
                    const float* dloss, float* da, int accumulate);
/* BCELoss(mean) on probabilities p vs constant target t (train/4...py:179,365-404);
 * log clamped at -100 like torch. */
int jaf_bce_fwd(jaf_stream_t s, const float* p, int32_t n, float target, float* loss);
int jaf_bce_bwd(jaf_stream_t s, const float* p, int32_t n, float target, const float* dloss,
                float* dp);
/* Two BCE terms of one vector (entries [0, n1) against t1, [n1, n) against t2: the discriminators' real / generated halves of one
 * batched pass, train/4...py:380-394) and their sum in one launch; backward for any of the three incoming gradients (NULL = absent). */
int jaf_bce_pair_fwd(jaf_stream_t s, const float* p, int32_t n1, int32_t n, float t1, float t2, float* loss1, float* loss2,
                     float* loss_sum);
int jaf_bce_pair_bwd(jaf_stream_t s, const float* p, int32_t n1, int32_t n, float t1, float t2, const float* g1, const float* g2,
                     const float* gsum, float* dp);
/* nn.Linear (+ act): y[N,O] = act(x[N,I] @ W[O,I]^T + b) (src/networks.py:408-410). */
int jaf_linear_fwd(jaf_stream_t s, const float* x, const float* w, const float* b, float* y,
                   int32_t N, int32_t I, int32_t O, int act, float slope);
int jaf_linear_bwd(jaf_stream_t s, const float* dz, const float* x, const float* w, float* dx,
                   float* dw, float* db, int32_t N, int32_t I, int32_t O);
/* The same with the activation backward folded in (dy and the layer's output y instead of dz) and, with `accumulate`, the parameter
 * gradients added to the caller's buffers (one add per element, as an accumulation pass would). */
int jaf_linear_bwd_fused(jaf_stream_t s, const float* dy, const float* y, const float* x, const float* w, float* dx, float* dw,
                         float* db, int32_t N, int32_t I, int32_t O, int act, float slope, int accumulate);
/* torch.optim.Adam defaults (betas .9/.999, eps 1e-8, no weight decay, train/4...py:169-175)
 * over one flat parameter buffer. step is the 1-based step count. */
int jaf_adam_step(jaf_stream_t s, float* p, const float* g, float* m, float* v, int64_t n,
                  float lr, float beta1, float beta2, float eps, int32_t step);
/* The same update with the step count on the device: state[0] (int32 bits) = steps taken so far, advanced by the call;
 * state[1..2] scratch.  For launches replayed from a captured hipGraph, whose arguments cannot change. */
int jaf_adam_step_dev(jaf_stream_t s, float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                      float beta2, float eps, float* state);
/* out[i] (+)= sum over s < slots of in[s*n + i]  (the slot copies of jaf_packed_io.dz_dbias). */
int jaf_sum_slots(jaf_stream_t s, const float* in, int32_t slots, int64_t n, float* out, int accumulate);
/* y = a*x + b*y elementwise (gradient accumulation, scaling). */
int jaf_axpby(jaf_stream_t s, float a, const float* x, float b, float* y, int64_t n);

/* ---------------------------------------------------------------------------------------------
 * Measured ceilings of the box ("confirm on the box"), used only by bench.py to report
 * roofline fractions against what this GPU delivers next to the nominal peaks.
 * ------------------------------------------------------------------------------------------ */
/* `blocks` workgroups of 4 waves, each wave issues iters*8 independent v_mfma_f32_16x16x32_bf16
 * (16384 FLOP each) from registers: the sustained dense bf16 matrix-core rate at the sustained clock. */
int jaf_ubench_mfma_bf16(jaf_stream_t s, int32_t blocks, int32_t iters, float* sink);
/* dst[i] = src[i], 16 bytes per lane, grid-stride: the streaming-copy HBM rate (n16 = 16-byte items). */
int jaf_ubench_copy(jaf_stream_t s, const void* src, void* dst, int64_t n16);
/* The same copy with `variant` choosing loads in flight per lane / nontemporal accesses and an explicit grid
 * (bench.py reports the best as this box's HBM ceiling): 0: 4 in flight, 1: 4 nt, 2: 8 nt, 3: 2 nt, 4: 1 nt, 5: 1. */
int jaf_ubench_copy_variant(jaf_stream_t s, const void* src, void* dst, int64_t n16, int32_t variant, int32_t blocks);

/* Profiling aid: while jaf_kernel_names(1) is on, every convolution-family launch (forward / data gradient, ConvLSTM cell,
 * weight gradients, the resizing pack) leaves the name of the kernel instantiation it picked -- as rocprofv3 prints it,
 * e.g. "conv_dma_kernel<4, 4, false, false, false>" -- in a thread-local buffer that jaf_last_kernel_name copies out (NUL-terminated, truncated to buflen).  The launch
 * code is the only place that knows which instantiation runs; bench.py's roofline rows and profiles/ take their names from
 * here.  jaf_kernel_names returns the previous setting.  No reference counterpart (the reference has no profiler hooks).     */
int jaf_kernel_names(int on);
int jaf_last_kernel_name(char* buf, int32_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* JAFPRO_HIP_H */
