// The step's weight preparation as device functions, so that it can ride in another kernel's grid (conv_c1.hip pairs it with
// the first encoder layer's launch): the three-term bf16 split of the 32-channel conv weights in per-lane MFMA operand order
// (layout: conv32_common.h) and the latent block's matrix layouts (midprep.h).
#pragma once
#include "common.h"
#include "conv32_common.h"
#include "midprep.h"

namespace arvae {

__device__ __forceinline__ void split8x3(const float (&x)[8], bf16x8 &hi, bf16x8 &mid, bf16x8 &lo) {
    i32x4v h, m, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned a, b, c;
        split_pair3(x[2 * j], x[2 * j + 1], a, b, c);
        h[j] = (int)a; m[j] = (int)b; l[j] = (int)c;
    }
    hi = __builtin_bit_cast(bf16x8, h); mid = __builtin_bit_cast(bf16x8, m); lo = __builtin_bit_cast(bf16x8, l);
}

constexpr int PREP_MAX_LAYERS = 8;
struct PrepArgs {
    const float *wt[PREP_MAX_LAYERS];
    uint4 *out[PREP_MAX_LAYERS];
};

// 16 workgroups per layer: items 0..2047 build the DOWN part, 2048..4095 the UP part; an item = 8 weights -> 3 x 16 bytes
__device__ __forceinline__ void conv32_prep_block(const PrepArgs &p, const int block) {
    const int layer = block >> 4, item = (block & 15) * 256 + threadIdx.x;
    const float *wt = nullptr;
    uint4 *out = nullptr;
#pragma unroll
    for (int q = 0; q < PREP_MAX_LAYERS; ++q)                    // constant indices into the by-value argument block
        if (q == layer) { wt = p.wt[q]; out = p.out[q]; }
    const int lane = item & 63, half = lane >> 5, rc = lane & 31;
    float x[8];
    uint4 *dst;
    if (item < 2048) {                                           // DOWN: (kh, tap = kyl*4 + kx, c)
        const int c = (item >> 6) & 1, tap = (item >> 7) & 7, kh = item >> 10;
        const int ky = 2 * kh + (tap >> 2), kx = tap & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = wt[((rc * C32) + c * 16 + half * 8 + j) * 16 + ky * 4 + kx];
        dst = out + (kh * PREP_DOWN_SLOTS + (tap * 2 + c) * 3) * 64 + lane;
    } else {                                                     // UP: (class = wave, ty, tx, c)
        const int u = item - 2048;
        const int c = (u >> 6) & 1, tx = (u >> 7) & 1, ty = (u >> 8) & 1, cls = u >> 9;
        const int ky = 1 - (cls >> 1) + 2 * ty, kx = 1 - (cls & 1) + 2 * tx;
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = wt[((c * 16 + half * 8 + j) * C32 + rc) * 16 + ky * 4 + kx];
        dst = out + PREP_DOWN_UINT4 + (cls * PREP_UP_SLOTS + ((ty * 2 + tx) * 2 + c) * 3) * 64 + lane;
    }
    bf16x8 h, m, l;
    split8x3(x, h, m, l);
    dst[0] = __builtin_bit_cast(uint4, h);
    dst[64] = __builtin_bit_cast(uint4, m);
    dst[128] = __builtin_bit_cast(uint4, l);
}


// one block of the combined prep grid: blocks [0, conv_blocks) split conv weights, the rest lay out the latent block's matrices
__device__ __forceinline__ void prep_all_block(const PrepArgs &p, const MidPrepArgs &mid, int conv_blocks, int block) {
    if (block < conv_blocks) conv32_prep_block(p, block);
    else mid_prep_block(mid, block - conv_blocks);
}

}  // namespace arvae
