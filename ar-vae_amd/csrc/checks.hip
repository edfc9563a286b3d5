// Debug-mode failure checks of the reference, done on the device and off by default (ARVAE_CHECK=1 on the Python side):
//   * the NaN scan of every weight at the top of Encoder.forward / Decoder.forward (measurevae/encoder.py:101-106,
//     decoder.py:420-425; the reference runs it on the host with one sync per parameter),
//   * Decoder.check_index (decoder.py:30-41): every fed-back note index must lie in [0, num_notes).
// Both count offending elements into a caller-provided device word (the caller zeroes it and decides when to read it).
#include "common.h"

namespace arvae {

__global__ __launch_bounds__(256) void count_nonfinite_kernel(const float *__restrict__ p, int64_t count, int32_t *__restrict__ flag) {
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        const unsigned u = __builtin_bit_cast(unsigned, p[i]);
        bad += (u & 0x7f800000u) == 0x7f800000u ? 1 : 0;          // exponent all ones: NaN or infinity
    }
    bad = (int)wave_sum((float)bad);
    if ((threadIdx.x & 63) == 0 && bad > 0) atomicAdd(flag, bad);
}

__global__ __launch_bounds__(256) void count_out_of_range_kernel(const int64_t *__restrict__ idx, int64_t count, int64_t lo, int64_t hi,
                                                                  int32_t *__restrict__ flag) {
    int bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) bad += (idx[i] < lo || idx[i] >= hi) ? 1 : 0;
    bad = (int)wave_sum((float)bad);
    if ((threadIdx.x & 63) == 0 && bad > 0) atomicAdd(flag, bad);
}

}  // namespace arvae

using namespace arvae;

extern "C" int arvae_count_nonfinite(const float *values, int64_t count, int32_t *flag, arvae_stream_t stream) {
    ARVAE_REQUIRE(values && flag && count > 0, "count_nonfinite: bad argument");
    int64_t blocks = (count + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    ARVAE_LAUNCH(count_nonfinite_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), values, count, flag);
    return check_launch("count_nonfinite_kernel");
}

extern "C" int arvae_count_out_of_range(const int64_t *indices, int64_t count, int64_t lo, int64_t hi, int32_t *flag,
                                        arvae_stream_t stream) {
    ARVAE_REQUIRE(indices && flag && count > 0, "count_out_of_range: bad argument");
    int64_t blocks = (count + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    ARVAE_LAUNCH(count_out_of_range_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), indices, count, lo, hi, flag);
    return check_launch("count_out_of_range_kernel");
}
