// hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
// Issue rate of v_mfma_f32_32x32x16_bf16 / 16x16x32 in s_memtime ticks and wall time, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k32(float *out, unsigned long long *st, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f - threadIdx.x * 0.002f); }
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 24 / NACC; ++r)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { st[blockIdx.x * 2] = t1 - t0; st[blockIdx.x * 2 + 1] = w1 - w0; }
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float *out, unsigned long long *st, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f - threadIdx.x * 0.002f); }
    f32x4 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int i = 0; i < 4; ++i) acc[j][i] = 0.f;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 24 / NACC; ++r)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int i = 0; i < 4; ++i) s += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { st[blockIdx.x * 2] = t1 - t0; st[blockIdx.x * 2 + 1] = w1 - w0; }
}
template <class K> void run(const char *name, K kern, int grid, int threads, int iters) {
    float *out; unsigned long long *st;
    hipMalloc(&out, grid * threads * 4); hipMalloc(&st, grid * 16);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, out, st, iters);
    hipDeviceSynchronize();
    unsigned long long h[2];
    hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
    const double n = 24.0 * iters;
    printf("%-34s grid %4d: %.1f ticks / MFMA, %.2f ns / MFMA, tick clock %.0f MHz\n", name, grid, h[0] / n, h[1] * 10.0 / n, h[0] * 100.0 / h[1]);
    hipFree(out); hipFree(st);
}
int main() {
    for (int grid : {8, 256}) {
        run("32x32x16 bf16, 1 acc, 4 waves", k32<1, 4>, grid, 256, 2000);
        run("32x32x16 bf16, 4 acc, 4 waves", k32<4, 4>, grid, 256, 2000);
        run("32x32x16 bf16, 4 acc, 8 waves", k32<4, 8>, grid, 512, 2000);
        run("32x32x16 bf16, 4 acc, 1 wave", k32<4, 1>, grid, 64, 2000);
        run("16x16x32 bf16, 1 acc, 4 waves", k16<1>, grid, 256, 2000);
        run("16x16x32 bf16, 4 acc, 4 waves", k16<4>, grid, 256, 2000);
    }
    return 0;
}
