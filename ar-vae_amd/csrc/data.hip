// Input side of the training step: minibatches are cut on the device from a dataset that stays in HBM in its
// on-disk precision (dSprites / MNIST images are uint8: 4x less HBM and no host round trip per batch).
#include "common.h"

namespace arvae {

// out[b][j] = scale * src[idx[b]][j], four bytes per thread
__global__ __launch_bounds__(256) void gather_rows_u8_kernel(const uint8_t *__restrict__ src, int64_t n_rows, int64_t row4,
                                                             const int64_t *__restrict__ idx, int64_t count, float scale,
                                                             float *__restrict__ out) {
    const int64_t total = count * row4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / row4, j = i - b * row4;
        const int64_t r = idx[b];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r >= 0 && r < n_rows) {
            const uchar4 u = reinterpret_cast<const uchar4 *>(src + r * row4 * 4)[j];
            v = make_float4(scale * u.x, scale * u.y, scale * u.z, scale * u.w);
        }
        reinterpret_cast<float4 *>(out)[i] = v;
    }
}

}  // namespace arvae

using namespace arvae;

extern "C" int arvae_gather_rows_u8(const uint8_t *src, int64_t n_rows, int64_t row_elems, const int64_t *idx, int64_t count,
                                    float scale, float *out, arvae_stream_t stream) {
    ARVAE_REQUIRE(src && idx && out && n_rows > 0 && count > 0, "gather_rows_u8: bad argument");
    ARVAE_REQUIRE(row_elems > 0 && row_elems % 4 == 0, "gather_rows_u8: the row length must be a multiple of 4 bytes");
    ARVAE_REQUIRE((((uintptr_t)src | (uintptr_t)out) & 3) == 0, "gather_rows_u8: src must be 4-byte, out 16-byte aligned");
    const int64_t total = count * (row_elems / 4);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    ARVAE_LAUNCH(gather_rows_u8_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), src, n_rows, row_elems / 4,
                       idx, count, scale, out);
    return check_launch("gather_rows_u8");
}
