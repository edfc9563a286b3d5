// hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_barrier.hip -o /tmp/mb && /tmp/mb
// 48 MFMAs (32x32x16 bf16, 4 accumulators) per step and one workgroup barrier per step: what does the barrier cost the MFMA
// waves, with 4 waves (all computing) and with 8 waves (4 computing, 4 only at the barrier)?  MODE 2 also issues 60 LDS
// reads per step ahead of their use, MODE 3 puts a small VALU block (24 v_cndmask) in front of each 24-MFMA block.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int WAVES, int MODE>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void kern(float *out, unsigned long long *st, int steps) {
    __shared__ __attribute__((aligned(16))) unsigned lds[16384];
    for (int i = threadIdx.x; i < 16384; i += WAVES * 64) lds[i] = i * 2654435761u;
    const int wave = threadIdx.x >> 6;
    bf16x8 a[4][3], b[3];
    for (int k = 0; k < 4; ++k) for (int t = 0; t < 3; ++t) for (int i = 0; i < 8; ++i) a[k][t][i] = (__bf16)(threadIdx.x * 0.001f + i + k + 0.1f * t);
    for (int t = 0; t < 3; ++t) for (int i = 0; i < 8; ++i) b[t][i] = (__bf16)(i * 0.5f + t);
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < 4) {
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                if (MODE == 2 || MODE == 4) {            // 30 reads for the next block, issued before this block's MFMAs
                    i32x4 nx[15];
#pragma unroll
                    for (int r = 0; r < 15; ++r) nx[r] = *reinterpret_cast<const i32x4 *>(lds + ((threadIdx.x * 4 + 260 * r + 64 * s) & 16380));
                    if (MODE == 2) __builtin_amdgcn_sched_barrier(0);
#define P(TA, TB) _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j][TA], b[TB], acc[j], 0, 0, 0);
                    P(2, 0) P(0, 2) P(1, 1) P(1, 0) P(0, 1) P(0, 0)
                    if (MODE == 4) {        // one MFMA, then (at most) one LDS read, 24 times: the reads ride in the MFMAs' shadows
#define G1 __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        G1 G1 G1 G1 G1 G1 G1 G1 G1 G1 G1 G1 G1 G1 G1
#undef G1
                        __builtin_amdgcn_sched_group_barrier(0x008, 9, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int t = 0; t < 3; ++t) a[k][t] = __builtin_bit_cast(bf16x8, nx[k * 3 + t]);
#pragma unroll
                    for (int t = 0; t < 3; ++t) b[t] = __builtin_bit_cast(bf16x8, nx[12 + t]);
                } else {
                    if (MODE == 3) {
#pragma unroll
                        for (int t = 0; t < 3; ++t) {
                            i32x4 v = __builtin_bit_cast(i32x4, b[t]);
                            const bool z = ((s + threadIdx.x) & 63) == 7;
                            v.x = z ? 0 : v.x; v.y = z ? 0 : v.y; v.z = z ? 0 : v.z; v.w = z ? 0 : v.w;
                            b[t] = __builtin_bit_cast(bf16x8, v);
                        }
                    }
                    P(2, 0) P(0, 2) P(1, 1) P(1, 0) P(0, 1) P(0, 0)
                }
                if (blk == 0 && MODE != 1) __syncthreads();
            }
        }
    } else {
        for (int s = 0; s < steps; ++s)
            if (MODE != 1) __syncthreads();
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 16; ++i) sum += acc[j][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0) st[blockIdx.x] = t1 - t0;
}
template <class K> void run(const char *name, K k, int threads) {
    float *out; unsigned long long *st;
    (void)hipMalloc(&out, 256 * threads * 4); (void)hipMalloc(&st, 256 * 8);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, st, 1000);
    (void)hipDeviceSynchronize();
    unsigned long long h;
    (void)hipMemcpy(&h, st, 8, hipMemcpyDeviceToHost);
    printf("%-64s %.0f ticks per step (48 MFMAs = 1536)\n", name, h / 1000.0);
    (void)hipFree(out); (void)hipFree(st);
}
int main() {
    run("4 waves, no barrier", kern<4, 1>, 256);
    run("4 waves, barrier per step", kern<4, 0>, 256);
    run("8 waves (4 idle at the barrier), barrier per step", kern<8, 0>, 512);
    run("8 waves, barrier per step, 60 LDS reads (b128) per step", kern<8, 2>, 512);
    run("8 waves, barrier per step, 24 v_cndmask in front of each block", kern<8, 3>, 512);
    run("8 waves, barrier, 60 LDS reads interleaved 1 per MFMA (sched_group_barrier)", kern<8, 4>, 512);
    return 0;
}
