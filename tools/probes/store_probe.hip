// Store-pattern probe: 67 MB written either as contiguous 1 KB per wave instruction or as 32-byte pieces at a 128-byte
// stride (the epilogue pattern of the transposed-MFMA conv kernels: 4 instructions fill a pixel's 128-byte line).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void st_contig(float4 *out, int64_t n16) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256)
        out[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
// lane (px = lane&31, half = lane>>5), 4 instructions g: address = pixel*128 + 32 g + 16 half
__global__ __launch_bounds__(256) void st_piece(float4 *out, int64_t npix) {
    const int lane = threadIdx.x & 63, half = lane >> 5, rc = lane & 31;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t p0 = wave * 32; p0 < npix; p0 += nw * 32) {
        const int64_t pix = p0 + rc;
#pragma unroll
        for (int g = 0; g < 4; ++g) out[pix * 8 + 2 * g + half] = make_float4(1.f, 2.f, 3.f, (float)g);
    }
}
// same bytes per lane, but a pixel's line is written by one instruction of 8 lanes: lane l -> pixel 8*i + l/8, piece l%8
__global__ __launch_bounds__(256) void st_line(float4 *out, int64_t npix) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * 256) >> 6;
    for (int64_t p0 = wave * 32; p0 < npix; p0 += nw * 32) {
#pragma unroll
        for (int g = 0; g < 4; ++g) out[(p0 + 8 * g + (lane >> 3)) * 8 + (lane & 7)] = make_float4(1.f, 2.f, 3.f, (float)g);
    }
}
typedef float f32x16 __attribute__((ext_vector_type(16)));
// the piece pattern with the down_c1 work added level by level: LV 1 = + 8 fp32 MFMAs per 32-pixel row, 2 = + the 8 image
// loads per lane per row (stride-2 4x4 patch reads of a 64x64 fp32 image), 3 = + the uint16 sign-bit store per lane
template <int LV>
__global__ __launch_bounds__(256) void st_work(float4 *out, const float *img, uint16_t *bits, int n_rows) {
    const int lane = threadIdx.x & 63, half = lane >> 5, rc = lane & 31;
    const int wave0 = (blockIdx.x * 256 + threadIdx.x) >> 6, nw = (gridDim.x * 256) >> 6;
    float w8[8];
    for (int s = 0; s < 8; ++s) w8[s] = 0.01f * (rc + s + half);
    for (int row = wave0; row < n_rows; row += nw) {
        const int n = row >> 5, r = row & 31;
        float a[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (LV >= 2) {
                const int gy = 2 * r - 1 + (s >> 1), gx = 2 * rc - 1 + 2 * (s & 1) + half;
                const bool ok = (unsigned)gy < 64u && (unsigned)gx < 64u;
                const float v = img[ok ? (n * 64 + gy) * 64 + gx : 0];
                a[s] = ok ? v : 0.f;
            } else a[s] = (float)(row + s);
        }
        f32x16 acc;
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w8[s], a[s], acc, 0, 0, 0);
        const int64_t pix = (int64_t)row * 32 + rc;
        unsigned b = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[4];
            for (int j = 0; j < 4; ++j) { v[j] = fmaxf(acc[4 * g + j], 0.f); b |= (v[j] > 0.f ? 1u : 0u) << (4 * g + j); }
            out[pix * 8 + 2 * g + half] = make_float4(v[0], v[1], v[2], v[3]);
        }
        if (LV >= 3) bits[pix * 2 + half] = (uint16_t)b;
    }
}
int main() {
    const int64_t bytes = 67108864, n16 = bytes / 16, npix = bytes / 128;
    float4 *buf; hipMalloc(&buf, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {256, 512, 1024, 2048, 4096}) {
        for (int k = 0; k < 3; ++k) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0);
                if (k == 0) hipLaunchKernelGGL(st_contig, dim3(grid), dim3(256), 0, 0, buf, n16);
                else if (k == 1) hipLaunchKernelGGL(st_piece, dim3(grid), dim3(256), 0, 0, buf, npix);
                else hipLaunchKernelGGL(st_line, dim3(grid), dim3(256), 0, 0, buf, npix);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            printf("grid %5d %-8s %7.1f us  %5.2f TB/s\n", grid, k == 0 ? "contig" : k == 1 ? "piece" : "line", best * 1e3, bytes / best / 1e9);
        }
    }
    float *img; hipMalloc(&img, 512 * 4096 * 4); hipMemset(img, 0, 512 * 4096 * 4);
    uint16_t *bits; hipMalloc(&bits, npix * 4);
    for (int grid : {512, 2048, 4096}) {
        for (int lv = 1; lv <= 3; ++lv) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; ++rep) {
                hipEventRecord(e0);
                if (lv == 1) hipLaunchKernelGGL(st_work<1>, dim3(grid), dim3(256), 0, 0, buf, img, bits, (int)(npix / 32));
                else if (lv == 2) hipLaunchKernelGGL(st_work<2>, dim3(grid), dim3(256), 0, 0, buf, img, bits, (int)(npix / 32));
                else hipLaunchKernelGGL(st_work<3>, dim3(grid), dim3(256), 0, 0, buf, img, bits, (int)(npix / 32));
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep > 0 && ms < best) best = ms;
            }
            printf("grid %5d work level %d %7.1f us  %5.2f TB/s\n", grid, lv, best * 1e3, bytes / best / 1e9);
        }
    }
    return 0;
}
