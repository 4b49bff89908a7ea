// Internal (non-ABI) hooks between conv.hip (the public entry points) and the bf16 matrix-core
// implementation in conv_bf16.hip / wgrad_bf16.hip.
#pragma once
#include "jaf_common.h"
#include <stdlib.h>

// Upper bound of the pixel splits of every weight-gradient kernel: all workgroups of one (co, ci) block add their
// partial sums to the same dW addresses with fp32 atomics, which serialise (~0.25 us per workgroup and address
// set); measured on the discriminator's 6->32 layer: 768 splits 197 us, 96 splits ~50 us.
#define JAF_WGRAD_MAX_SPLIT 96

// Pixel splits of a weight-gradient launch.  items = (image, pixel tile) pairs, outblocks = workgroups per split,
// dw_floats = size of the whole dW.  A split shortens every workgroup's serial walk over its items but adds one
// atomic pass over dW (device-wide ~3e11 float atomics/s): deep layers (512 channels at 4x4..32x32: 8-64 items,
// 2-4 M floats of dW) want 1-2 splits, the 256x256 layers (4096 items) want as many as fill the chip.
static inline long jaf_wgrad_nsplit(long items, long outblocks, long dw_floats, double t_item = 2.5e-6, double slots = 768.0,
                                    long max_split = JAF_WGRAD_MAX_SPLIT) {
    const double atomics_per_s = 3e11;
    // experiment hook: workgroup slots a weight-gradient launch is sized for (768 = the whole chip at 3 per CU; the kernels run
    // beside the data-gradient chain on their own stream)
    const double slots_env = 0.0;
    if (slots_env > 0.0 && slots == 768.0) slots = slots_env;
    long best = 1;
    double best_t = 1e30;
    const long hi = items < max_split ? items : max_split;
    for (long ns = 1; ns <= hi; ++ns) {
        const double waves = (double)(outblocks * ns) / slots;
        const double t = (double)((items + ns - 1) / ns) * t_item * (waves > 1.0 ? waves : 1.0) + (double)ns * (double)dw_floats / atomics_per_s;
        if (t < best_t * 0.999) { best_t = t; best = ns; }
    }
    return best;
}

// The same trade with the launch's TRUE residency: `slots` = workgroups the chip holds at once (occupancy x CUs, asked of
// the runtime for the very kernel and LDS size), and whole rounds -- 528 workgroups on 512 slots take two rounds, not 1.03:
// measured on the 24 -> 48 @ 200 x 200 ConvLSTM layer (2 workgroups per CU by registers), 22 splits (528 workgroups, what the
// continuous model picks for 512 slots) 0.59 ms, 32 splits (768 = 1.5 rounds) 0.46 ms, 21 splits (504, one round) 0.36 ms.
static inline long jaf_wgrad_nsplit_rounds(long items, long outblocks, long dw_floats, double slots, double t_item = 2.5e-6,
                                           long max_split = JAF_WGRAD_MAX_SPLIT, double atomics_per_s = 3e11) {
    long best = 1;
    double best_t = 1e30;
    const long hi = items < max_split ? items : max_split;
    for (long ns = 1; ns <= hi; ++ns) {
        const long blocks = outblocks * ns;
        const long rounds = (long)(((double)blocks + slots - 1.0) / slots);
        const double t = (double)((items + ns - 1) / ns) * t_item * (double)(rounds < 1 ? 1 : rounds) + (double)ns * (double)dw_floats / atomics_per_s;
        if (t < best_t * 0.999) { best_t = t; best = ns; }
    }
    return best;
}

// Workgroups of `kernel` (256 threads, `lds` bytes) the device holds at once, asked of the runtime once per
// (instantiation, LDS size, device): registers decide it for the 3 x 3 kernels (2 per CU at 16*3 and 16*4 rows, 3 at 16*2,
// 4 at 16), LDS for the stride-2 and double-buffered ones.
struct JafOcc { int lds, slots; };
static inline int jaf_kernel_slots(const void* kernel, int lds, JafOcc (*cache)[8]) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= JAF_MAX_DEVICES) return 512;
    JafOcc* c = cache[dev];
    for (int i = 0; i < 8; ++i)
        if (c[i].lds == lds && c[i].slots > 0) return c[i].slots;
    int per_cu = 0, cus = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, 256, (size_t)lds) != hipSuccess || per_cu < 1) per_cu = 2;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
    const int slots = per_cu * cus;
    for (int i = 0; i < 8; ++i)
        if (c[i].slots == 0) { c[i].lds = lds; c[i].slots = slots; break; }
    return slots;
}


// conv_pack_weights.hip: the bf16 / split-bf16 weight image of the packed-input kernels
int jafb_pack(hipStream_t s, const jaf_conv_desc* d, const jaf_conv_plan* plan, int mode, const float* w,
              int32_t w_rows_tot, void* packed);
