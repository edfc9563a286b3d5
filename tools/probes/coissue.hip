// hipcc --offload-arch=gfx950 -O3 tools/probes/coissue.hip -o /tmp/coissue && /tmp/coissue
// What a partner wave's instructions cost an MFMA wave on the same SIMD: 8 waves per workgroup, waves 0-3 issue
// v_mfma_f32_32x32x16_bf16 back to back, waves 4-7 run a stream of one kind of instruction.  Prints the MFMA waves' ticks per
// MFMA and the partner's ticks per instruction, alone and together.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

enum { K_NONE, K_FMA, K_CVT, K_PKADD, K_AND, K_DSW64, K_DSR64, K_PERM, K_SUB };
template <int KIND, bool MFMA_ON>
__global__ __launch_bounds__(512, 2) void kern(float *out, unsigned long long *st, int iters) {
    __shared__ unsigned lds[8192];
    const int wave = threadIdx.x >> 6;
    unsigned long long t0 = 0, t1 = 0;
    if (wave < 4) {
        bf16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
        f32x16 acc[4];
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
        __syncthreads();
        t0 = __builtin_readcyclecounter();
        if (MFMA_ON)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
            }
        t1 = __builtin_readcyclecounter();
        float s = 0.f;
        for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) s += acc[j][i];
        out[blockIdx.x * 512 + threadIdx.x] = s;
        if (threadIdx.x == 0) st[blockIdx.x * 2] = t1 - t0;
    } else {
        float x[8];
        unsigned u[8];
        for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.37f + i; u[i] = threadIdx.x * 77u + i; }
        __syncthreads();
        t0 = __builtin_readcyclecounter();
        // partner runs ~ as long as the MFMA waves: 24 MFMAs x 32 cycles = 768 cycles per iteration; 96 partner ops per iteration
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 12; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (KIND == K_FMA) x[i] = __builtin_fmaf(x[i], 1.0001f, 0.5f);
                    if (KIND == K_SUB) x[i] = x[i] - __builtin_bit_cast(float, u[i]);
                    if (KIND == K_CVT) { f32x2 v = {x[i], x[(i + 1) & 7]}; u[i] ^= __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2)); }
                    if (KIND == K_PKADD) { f32x2 v = {x[i], x[(i + 1) & 7]}; f32x2 w = {1.5f, 2.5f}; v = v + w; x[i] = v.x; x[(i + 1) & 7] = v.y; }
                    if (KIND == K_AND) u[i] = (u[i] & 0xffff0000u) + 3u * (unsigned)r;
                    if (KIND == K_PERM) u[i] = __builtin_amdgcn_perm(u[i], u[(i + 1) & 7], 0x07060302u);
                    if (KIND == K_DSW64) *reinterpret_cast<uint2 *>(lds + ((threadIdx.x * 2 + 24 * i) & 8190)) = make_uint2(u[i], u[(i + 1) & 7]);
                    if (KIND == K_DSR64) { const unsigned long long q = *reinterpret_cast<volatile unsigned long long *>(lds + ((threadIdx.x * 2 + 24 * i) & 8190)); u[i] ^= (unsigned)q + (unsigned)(q >> 32); }
                }
        }
        t1 = __builtin_readcyclecounter();
        float s = 0.f;
        for (int i = 0; i < 8; ++i) s += x[i] + (float)u[i];
        out[blockIdx.x * 512 + threadIdx.x] = s;
        if (threadIdx.x == 256) st[blockIdx.x * 2 + 1] = t1 - t0;
    }
}
template <class K> void run(const char *name, K kern, int iters) {
    float *out; unsigned long long *st;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&st, 256 * 16);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(256), dim3(512), 0, 0, out, st, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[2];
    (void)hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
    printf("%-28s MFMA wave %6.1f ticks / MFMA   partner %6.2f ticks / op (96 ops per 24 MFMAs)\n", name, h[0] / (24.0 * iters), h[1] / (96.0 * iters));
    (void)hipFree(out); (void)hipFree(st);
}
#define BOTH(KIND, NAME) run(NAME " alone", kern<KIND, false>, 2000); run(NAME " + MFMA", kern<KIND, true>, 2000);
int main() {
    run("MFMA alone (partner idle)", kern<K_NONE, true>, 2000);
    BOTH(K_FMA, "v_fma_f32")
    BOTH(K_SUB, "v_sub_f32")
    BOTH(K_AND, "v_and+v_add (2 ops)")
    BOTH(K_PERM, "v_perm_b32")
    BOTH(K_CVT, "v_cvt_pk_bf16+xor")
    BOTH(K_PKADD, "v_pk_add_f32")
    BOTH(K_DSW64, "ds_write_b64")
    BOTH(K_DSR64, "ds_read_b64+xor")
    return 0;
}
