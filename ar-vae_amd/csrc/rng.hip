#include <algorithm>
// Standalone draws of the Philox generator (rng.h): reparameterisation noise and dropout keep-masks as ONE library launch
// each, for the callers that do not generate them inside a consuming kernel (heads.hip does for the conv VAEs' eps).
#include "rng.h"

namespace arvae {

__global__ __launch_bounds__(256) void philox_normal_kernel(float *__restrict__ out, int64_t count, RngStream s) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) out[i] = rng_normal(s, (uint64_t)i);
}

// sixteen mask bytes per Philox block: byte j of the block's 128 bits keeps its element when the byte is < threshold
// (keep probability quantised to 1/256: exact for the reference's p = 0.5)
__global__ __launch_bounds__(256) void philox_keep_mask_kernel(uint8_t *__restrict__ out, int64_t count, uint32_t threshold, RngStream s) {
    const int64_t blocks = (count + 15) / 16;
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < blocks; b += (int64_t)gridDim.x * 256) {
        const uint4 r = rng_block(s, (uint64_t)b);
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
        uint32_t m[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            m[q] = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) m[q] |= (((w[q] >> (8 * j)) & 0xffu) < threshold ? 1u : 0u) << (8 * j);
        }
        if (16 * b + 16 <= count) {
            *reinterpret_cast<uint4 *>(out + 16 * b) = make_uint4(m[0], m[1], m[2], m[3]);
        } else {
            for (int64_t e = 16 * b; e < count; ++e) out[e] = (uint8_t)((m[(e - 16 * b) >> 2] >> (8 * ((e - 16 * b) & 3))) & 0xffu);
        }
    }
}

// several draws of a step as ONE launch (the MeasureVAE executor's encoder keep-mask, eps and decoder keep-masks were three ~5 us
// launches in a row): blockIdx.x walks the jobs' block ranges; a draw is a pure function of (stream, element index), so the values
// are those of the separate launches
constexpr int RNG_BATCH_MAX = 8;
struct RngBatch {
    int count;
    int first_block[RNG_BATCH_MAX + 1];
    int kind[RNG_BATCH_MAX];            // 0: standard normals (float), 1: keep-mask bytes
    void *out[RNG_BATCH_MAX];
    int64_t n[RNG_BATCH_MAX];
    uint32_t threshold[RNG_BATCH_MAX];
    RngStream stream[RNG_BATCH_MAX];
};
__global__ __launch_bounds__(256) void philox_batch_kernel(RngBatch b) {
    int j = 0;
    while (j + 1 < b.count && (int)blockIdx.x >= b.first_block[j + 1]) ++j;
    const int64_t blk = (int)blockIdx.x - b.first_block[j], nblk = b.first_block[j + 1] - b.first_block[j];
    const RngStream s = b.stream[j];
    const int64_t count = b.n[j];
    if (b.kind[j] == 0) {
        float *out = static_cast<float *>(b.out[j]);
        for (int64_t i = blk * 256 + threadIdx.x; i < count; i += nblk * 256) out[i] = rng_normal(s, (uint64_t)i);
        return;
    }
    uint8_t *out = static_cast<uint8_t *>(b.out[j]);
    const uint32_t threshold = b.threshold[j];
    const int64_t blocks = (count + 15) / 16;
    for (int64_t q = blk * 256 + threadIdx.x; q < blocks; q += nblk * 256) {
        const uint4 r = rng_block(s, (uint64_t)q);
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
        uint32_t m[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            m[k] = 0;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) m[k] |= (((w[k] >> (8 * jj)) & 0xffu) < threshold ? 1u : 0u) << (8 * jj);
        }
        if (16 * q + 16 <= count) {
            *reinterpret_cast<uint4 *>(out + 16 * q) = make_uint4(m[0], m[1], m[2], m[3]);
        } else {
            for (int64_t e = 16 * q; e < count; ++e) out[e] = (uint8_t)((m[(e - 16 * q) >> 2] >> (8 * ((e - 16 * q) & 3))) & 0xffu);
        }
    }
}

// kinds / outs / counts / keep probabilities (masks) / call offsets of up to RNG_BATCH_MAX draws that share seed, step and device step
int philox_draws(int n_draws, const int *kind, void *const *out, const int64_t *count, const float *keep_prob, const uint32_t *offset,
                 uint64_t seed, uint32_t step, const uint32_t *dev_step, hipStream_t s) {
    ARVAE_REQUIRE(n_draws >= 1 && n_draws <= RNG_BATCH_MAX, "philox_draws: 1..%d draws per launch", RNG_BATCH_MAX);
    RngBatch b{};
    b.count = n_draws;
    int total = 0;
    for (int j = 0; j < n_draws; ++j) {
        ARVAE_REQUIRE(out[j] != nullptr && count[j] > 0, "philox_draws: bad argument");
        int64_t blocks;
        if (kind[j] == 0) {
            blocks = std::min<int64_t>((count[j] + 255) / 256, 2048);
        } else {
            ARVAE_REQUIRE(keep_prob[j] > 0.f && keep_prob[j] <= 1.f, "philox_draws: keep probability %f outside (0, 1]", keep_prob[j]);
            ARVAE_REQUIRE((reinterpret_cast<uintptr_t>(out[j]) & 15) == 0, "philox_draws: mask output must be 16-byte aligned");
            b.threshold[j] = (uint32_t)(keep_prob[j] * 256.0f + 0.5f);
            blocks = std::min<int64_t>(((count[j] + 15) / 16 + 255) / 256, 4096);
        }
        b.first_block[j] = total;
        total += (int)blocks;
        b.kind[j] = kind[j]; b.out[j] = out[j]; b.n[j] = count[j];
        b.stream[j] = RngStream{seed, offset[j], dev_step, step};
    }
    b.first_block[n_draws] = total;
    ARVAE_LAUNCH(philox_batch_kernel, dim3((unsigned)total), dim3(256), 0, s, b);
    return check_launch("philox_batch_kernel");
}

}  // namespace arvae

using namespace arvae;

extern "C" int arvae_philox_keep_masks(int32_t n_masks, uint8_t *const *outs, const int64_t *counts, float keep_prob, uint64_t seed,
                                       const uint32_t *offsets, uint32_t step, const uint32_t *dev_step, arvae_stream_t stream) {
    ARVAE_REQUIRE(n_masks >= 1 && n_masks <= RNG_BATCH_MAX && outs != nullptr && counts != nullptr && offsets != nullptr,
                  "philox_keep_masks: 1..%d masks per launch", RNG_BATCH_MAX);
    int kind[RNG_BATCH_MAX];
    void *out[RNG_BATCH_MAX];
    float keep[RNG_BATCH_MAX];
    for (int j = 0; j < n_masks; ++j) { kind[j] = 1; out[j] = outs[j]; keep[j] = keep_prob; }
    return philox_draws(n_masks, kind, out, counts, keep, offsets, seed, step, dev_step, as_stream(stream));
}

extern "C" int arvae_philox_normal(float *out, int64_t count, uint64_t seed, uint32_t offset, uint32_t step, const uint32_t *dev_step,
                                   arvae_stream_t stream) {
    ARVAE_REQUIRE(out != nullptr && count > 0, "philox_normal: bad argument");
    int64_t blocks = (count + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    ARVAE_LAUNCH(philox_normal_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), out, count, RngStream{seed, offset, dev_step, step});
    return check_launch("philox_normal_kernel");
}

extern "C" int arvae_philox_keep_mask(uint8_t *out, int64_t count, float keep_prob, uint64_t seed, uint32_t offset, uint32_t step,
                                      const uint32_t *dev_step, arvae_stream_t stream) {
    ARVAE_REQUIRE(out != nullptr && count > 0, "philox_keep_mask: bad argument");
    ARVAE_REQUIRE(keep_prob > 0.f && keep_prob <= 1.f, "philox_keep_mask: keep probability %f outside (0, 1]", keep_prob);
    ARVAE_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "philox_keep_mask: output must be 16-byte aligned");
    const uint32_t threshold = (uint32_t)(keep_prob * 256.0f + 0.5f);
    int64_t blocks = ((count + 15) / 16 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    ARVAE_LAUNCH(philox_keep_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), out, count, threshold,
                 RngStream{seed, offset, dev_step, step});
    return check_launch("philox_keep_mask_kernel");
}
