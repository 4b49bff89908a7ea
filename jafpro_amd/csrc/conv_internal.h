// Internal (non-ABI) hooks between conv.hip (the public entry points) and the bf16 matrix-core
// implementation in conv_bf16.hip / wgrad_bf16.hip.
#pragma once
#include "jaf_common.h"

// Upper bound of the pixel splits of every weight-gradient kernel: all workgroups of one (co, ci) block add their
// partial sums to the same dW addresses with fp32 atomics, which serialise (~0.25 us per workgroup and address
// set); measured on the discriminator's 6->32 layer: 768 splits 197 us, 96 splits ~50 us.
#define JAF_WGRAD_MAX_SPLIT 96

int jafb_plan(const jaf_conv_desc* d, int lstm, jaf_conv_plan* plan);
int jafb_pack(hipStream_t s, const jaf_conv_desc* d, const jaf_conv_plan* plan, int mode, const float* w,
              int32_t w_rows_tot, void* packed);
int jafb_fwd(hipStream_t s, const jaf_conv_desc* d, const jaf_conv_plan* plan, const float* src0,
             const float* src1, const float* src2, const void* packed_w, const float* bias, float* out);
int jafb_lstm(hipStream_t s, const jaf_conv_desc* d, const jaf_conv_plan* plan, const float* x,
              const float* h_prev, const void* packed_w, const float* bias, const float* c_prev,
              float* h_out, float* c_out, float* gates_out);
int jafb_wgrad(hipStream_t s, const jaf_conv_desc* d, const float* src0, const float* src1,
               const float* src2, const float* dz, float* dw, int accumulate, void* workspace,
               int64_t workspace_bytes);
int64_t jafb_wgrad_workspace(const jaf_conv_desc* d);
