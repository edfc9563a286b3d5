// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on this box for the access patterns of this library (round 6).
// MI355X_MICROARCH.md (HBM section) prescribes 2 x FETCH_SIZE for wide coalesced reads and says to calibrate other patterns on a
// known byte count.  Each kernel here moves a KNOWN number of bytes exactly once:
//   rd_global16   plain global loads, 16 B per lane, contiguous (1 KB per wave instruction)
//   rd_buffer16   raw buffer loads (the conv kernels' form: conv32_common.h buf_load4), same addresses
//   rd_pixels     the producers' pattern of down32p.h / wgrad32r.h: thread = (pixel, 16-byte channel chunk), a 256-thread
//                 group covers 4 KB per instruction, SLOTS instructions 4 KB apart in flight
//   rd_global4    4 B per lane, contiguous
//   wr_global16   16 B per lane stores
//   rd_twice_l2   the same 32 MB read twice by workgroups of the same XCD a few microseconds apart (what the aligned pair does)
//     hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib tools/probes/fetch_calib.hip
//     rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/fc_f -o p -- /tmp/fetch_calib     (and a second pass with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef int i32x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const f32x4v v = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float sum4(float4 v) { return v.x + v.y + v.z + v.w; }

__global__ __launch_bounds__(256) void rd_global16(const float4 *in, float *out, int64_t n16) {
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) acc += sum4(in[i]);
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void rd_buffer16(const float4 *in, float *out, int64_t n16) {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(in, n16 * 16);
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) acc += sum4(buf_load4(rs, (unsigned)(i * 16)));
    if (acc == 12345.678f) out[0] = acc;
}
// 512-thread workgroups, threads 256.. load (as the producer waves do), 10 slots of 4 KB in flight, contiguous 40 KB per round
__global__ __launch_bounds__(512) void rd_pixels(const float4 *in, float *out, int64_t n16) {
    if (threadIdx.x < 256) return;
    const int pt = threadIdx.x - 256;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(in, n16 * 16);
    const int64_t rounds = n16 / (10 * 256);
    float acc = 0.f;
    for (int64_t r = blockIdx.x; r < rounds; r += gridDim.x) {
        float4 v[10];
#pragma unroll
        for (int s = 0; s < 10; ++s) v[s] = buf_load4(rs, (unsigned)((r * 10 + s) * 4096 + pt * 16));
#pragma unroll
        for (int s = 0; s < 10; ++s) acc += sum4(v[s]);
    }
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void rd_global4(const float *in, float *out, int64_t n4) {
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) acc += in[i];
    if (acc == 12345.678f) out[0] = acc;
}
__global__ __launch_bounds__(256) void wr_global16(float4 *o, int64_t n16) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) o[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
// 256 workgroups; workgroup w and workgroup w + 128 (same XCD when the dispatcher deals workgroups round robin) read the SAME
// 256 KB range, the second one `lag` rounds of 4 KB behind the first
__global__ __launch_bounds__(256) void rd_twice_l2(const float4 *in, float *out, int64_t n16) {
    const int w = blockIdx.x & 127;
    const int64_t per = n16 / 128;
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(in, n16 * 16);
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < per; i += 256) acc += sum4(buf_load4(rs, (unsigned)((w * per + i) * 16)));
    if (acc == 12345.678f) out[0] = acc;
}

int main() {
    const int64_t bytes = 256ll << 20, small = 32ll << 20;
    float4 *buf;
    float *out;
    hipMalloc(&buf, bytes);
    hipMalloc(&out, 64);
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        rd_global16<<<2048, 256>>>(buf, out, bytes / 16);
        rd_buffer16<<<2048, 256>>>(buf, out, bytes / 16);
        rd_pixels<<<256, 512>>>(buf, out, bytes / 16);
        rd_global4<<<2048, 256>>>(reinterpret_cast<float *>(buf), out, bytes / 4);
        wr_global16<<<2048, 256>>>(buf, bytes / 16);
        rd_twice_l2<<<256, 256>>>(buf, out, small / 16);
        hipDeviceSynchronize();
    }
    printf("bytes per launch: rd_global16 %lld rd_buffer16 %lld rd_pixels %lld rd_global4 %lld wr_global16 %lld rd_twice_l2 %lld (x2 issued)\n",
           (long long)bytes, (long long)bytes, (long long)(bytes / 40960 * 40960), (long long)bytes, (long long)bytes, (long long)small);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
