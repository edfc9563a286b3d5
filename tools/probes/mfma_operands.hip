// hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_operands.hip -o /tmp/mfma_operands && /tmp/mfma_operands
// Does the issue rate of v_mfma_f32_32x32x16_bf16 depend on which registers feed it?  24 MFMAs per iteration as in the
// weight-gradient kernels: 4 accumulators x 6 products, A operands a[kx][t], B operands b[t] (all distinct registers),
// against the same count with ONE a and ONE b.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256) void kern(float *out, unsigned long long *st, int iters) {
    bf16x8 a[4][3], b[3];
    for (int k = 0; k < 4; ++k) for (int t = 0; t < 3; ++t) for (int i = 0; i < 8; ++i) a[k][t][i] = (__bf16)(threadIdx.x * 0.001f + i + k + 0.1f * t);
    for (int t = 0; t < 3; ++t) for (int i = 0; i < 8; ++i) b[t][i] = (__bf16)(i * 0.5f + t);
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {            // one a, one b
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][0], b[0], acc[j], 0, 0, 0);
        } else {
#define P(TA, TB) _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j][TA], b[TB], acc[j], 0, 0, 0);
            P(2, 0) P(0, 2) P(1, 1) P(1, 0) P(0, 1) P(0, 0)
#undef P
        }
        if (MODE == 2) {            // operands change every iteration (as after an LDS read)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int t = 0; t < 3; ++t) asm volatile("" : "+v"(a[k][t]));
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter(), w1 = wall_clock64();
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { st[blockIdx.x * 2] = t1 - t0; st[blockIdx.x * 2 + 1] = w1 - w0; }
}
template <class K> void run(const char *name, K k, int grid) {
    float *out; unsigned long long *st;
    (void)hipMalloc(&out, grid * 256 * 4); (void)hipMalloc(&st, grid * 16);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, st, 2000);
    (void)hipDeviceSynchronize();
    unsigned long long h[2];
    (void)hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
    printf("%-40s grid %3d: %.1f ticks / MFMA, %.2f ns / MFMA\n", name, grid, h[0] / 48000.0, h[1] * 10.0 / 48000.0);
    (void)hipFree(out); (void)hipFree(st);
}
int main() {
    for (int grid : {8, 256}) {
        run("one A, one B register set", kern<0>, grid);
        run("a[kx][t], b[t]: 15 register sets", kern<1>, grid);
        run("same, operands redefined each iteration", kern<2>, grid);
    }
    return 0;
}
