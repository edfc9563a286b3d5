// The step's weight preparation as device functions, so that it can ride in another kernel's grid (conv_c1.hip pairs it with
// the first encoder layer's launch): the scaled two-term fp16 split of the 32-channel conv weights in per-lane MFMA operand order
// (layout: conv32_common.h) and the latent block's matrix layouts (midprep.h).
#pragma once
#include "common.h"
#include "conv32_common.h"
#include "midprep.h"

namespace arvae {

// eight scaled values -> their (h, l) operand registers
__device__ __forceinline__ void split8_h2(const float (&x)[8], float s, f16x8 &hi, f16x8 &lo) {
    i32x4v h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned a, b;
        split_pair_h2(x[2 * j], x[2 * j + 1], s, a, b);
        h[j] = (int)a; l[j] = (int)b;
    }
    hi = __builtin_bit_cast(f16x8, h); lo = __builtin_bit_cast(f16x8, l);
}

constexpr int PREP_MAX_LAYERS = 8;
struct PrepArgs {
    const float *wt[PREP_MAX_LAYERS];
    uint4 *out[PREP_MAX_LAYERS];
};

// 16 workgroups (of 256 threads) per layer: items 0..2047 build the DOWN part, 2048..4095 the UP part; an item = 8 weights ->
// 2 x 16 bytes.  Every workgroup first takes the maximum magnitude of the layer's 16384 weights itself (64 KB from L2, sixteen
// 16-byte loads per thread: cheaper than a second launch or a grid-wide exchange), which fixes the layer's scale.
__device__ __forceinline__ void conv32_prep_block(const PrepArgs &p, const int block) {
    __shared__ float wmax[4];
    const int layer = block >> 4, item = (block & 15) * 256 + threadIdx.x;
    const float *wt = nullptr;
    uint4 *out = nullptr;
#pragma unroll
    for (int q = 0; q < PREP_MAX_LAYERS; ++q)                    // constant indices into the by-value argument block
        if (q == layer) { wt = p.wt[q]; out = p.out[q]; }
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) m = fmaxf(m, amax4(*reinterpret_cast<const float4 *>(wt + (i * 256 + threadIdx.x) * 4)));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    const Pow2 sc = pow2_for(__builtin_bit_cast(unsigned, m));
    if (item == 0) out[PREP_DOWN_UINT4 + PREP_UP_UINT4] = make_uint4(__builtin_bit_cast(unsigned, sc.inv), 0u, 0u, 0u);
    const int lane = item & 63, half = lane >> 5, rc = lane & 31;
    float x[8];
    uint4 *dst;
    if (item < 2048) {                                           // DOWN: (kh, tap = kyl*4 + kx, c)
        const int c = (item >> 6) & 1, tap = (item >> 7) & 7, kh = item >> 10;
        const int ky = 2 * kh + (tap >> 2), kx = tap & 3;
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = wt[((rc * C32) + c * 16 + half * 8 + j) * 16 + ky * 4 + kx];
        dst = out + (kh * PREP_DOWN_SLOTS + (tap * 2 + c) * 2) * 64 + lane;
    } else {                                                     // UP: (class = wave, ty, tx, c)
        const int u = item - 2048;
        const int c = (u >> 6) & 1, tx = (u >> 7) & 1, ty = (u >> 8) & 1, cls = u >> 9;
        const int ky = 1 - (cls >> 1) + 2 * ty, kx = 1 - (cls & 1) + 2 * tx;
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = wt[((c * 16 + half * 8 + j) * C32 + rc) * 16 + ky * 4 + kx];
        dst = out + PREP_DOWN_UINT4 + (cls * PREP_UP_SLOTS + ((ty * 2 + tx) * 2 + c) * 2) * 64 + lane;
    }
    f16x8 h, l;
    split8_h2(x, sc.s, h, l);
    dst[0] = __builtin_bit_cast(uint4, h);
    dst[64] = __builtin_bit_cast(uint4, l);
}


// one block of the combined prep grid: blocks [0, conv_blocks) split conv weights, the rest lay out the latent block's matrices
__device__ __forceinline__ void prep_all_block(const PrepArgs &p, const MidPrepArgs &mid, int conv_blocks, int block) {
    if (block < conv_blocks) conv32_prep_block(p, block);
    else mid_prep_block(mid, block - conv_blocks);
}

}  // namespace arvae
