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

}  // namespace arvae

using namespace arvae;

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
