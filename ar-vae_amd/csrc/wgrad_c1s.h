// The single-channel links' weight gradient as a device body (conv_c1.hip launches it alone or beside the layer's data gradient;
// dense.hip beside the grouped Linear weight gradients): see the comment at the body.
#pragma once
#include "common.h"
#include "reduce.h"

namespace arvae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int HI1 = 64, LO1 = 32, CC = 32;  // image side, feature-map side, channels of the 1 <-> 32 channel links

// ================================================================================================
// slab per workgroup: [32 clo][16 taps] + 32 lo sums + 1 image sum
constexpr int WG1_SLAB = SLAB_C1_FLOATS;

// wgrad_c1 on the 16x16x4 MFMA: D[clo][tap] += sum over 4 positions of lo[pos][clo] * img[pos @ tap]; M = 32 channels
// (two 16-row tiles), N = the 16 taps (every column useful), K = positions.  A lane is (row/col index lane & 15,
// position-in-quad lane >> 4).

// ================================================================================================
// wgrad_c1, streaming form (the tiled first generation, eight waves marching between two barriers per tile at 2.9 TB/s, was
// removed in round 3): a WAVE owns one lo row (32 positions x 32
// channels = 4 KB) at a time and stages it, with the four image rows it meets, in LDS of its own: no workgroup barrier in the
// loop, five fully coalesced 1 KB loads per row (four of lo, one of the image) issued one row ahead, and enough independent
// waves per CU (two workgroups of eight) that somebody is always loading.  The tiled kernel's eight waves march between two
// barriers per tile and reach 2.9 TB/s; the loads are all this kernel has to do (7 FLOP/B).
constexpr int WS1 = 48;                     // LDS position stride of the lo row: the A reads of the four position slots
                                            // (g * 48 + li) fall on four disjoint bank groups
constexpr int IMS = 72;                     // LDS row stride of the image rows: pixel gx at 4 + gx, zero pads at 3 and 68
constexpr int WGS_WAVES = 8;

__device__ __forceinline__ void wgrad_c1s_body(Operand lo, Operand img, float *__restrict__ slab, int n_rows, const int BID, const int NBLK,
                                               float *lo_s, float *im_s) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
    const int ky = li >> 2, kx = li & 3;
    float *lw = lo_s + wave * (LO1 * WS1), *iw = im_s + wave * (4 * IMS);
    if (lane < 8) iw[(lane >> 1) * IMS + ((lane & 1) ? 68 : 3)] = 0.f;            // pixels -1 and 64
    f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float lo_sum[2] = {0.f, 0.f}, img_sum = 0.f;
    const float gs = img.scale != nullptr ? img.scale[0] : 1.f;
    const int wave0 = BID * WGS_WAVES + wave, n_waves = NBLK * WGS_WAVES;
    float4 lr[4], ir;
    // lo row `row` as 4 x 1 KB; image rows 2 r - 1 .. 2 r + 2 (lane >> 4), pixels 4 (lane & 15) .. + 3
    auto issue = [&](int row) __attribute__((always_inline)) {
        const bool in = row < n_rows;
        const int n = row >> 5, r = row & 31;
#pragma unroll
        for (int i = 0; i < 4; ++i) lr[i] = lo.at4(in ? (int64_t)row * (LO1 * CC) + (64 * i + lane) * 4 : 0);
        const int gy = 2 * r - 1 + (lane >> 4);
        const bool ok = in && (unsigned)gy < (unsigned)HI1;
        const float4 v = img.at4(ok ? ((int64_t)n * HI1 + gy) * HI1 + 4 * (lane & 15) : 0);       // unconditional load, clamped index
        ir = ok ? make_float4(gs * v.x, gs * v.y, gs * v.z, gs * v.w) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    issue(wave0);
    for (int row = wave0; row < n_rows; row += n_waves) {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4 *>(lw + (8 * i + (lane >> 3)) * WS1 + 4 * (lane & 7)) = lr[i];
        *reinterpret_cast<float4 *>(iw + (lane >> 4) * IMS + 4 + 4 * (lane & 15)) = ir;
        if (((lane >> 4) + 1) & 2) img_sum += (ir.x + ir.y) + (ir.z + ir.w);       // image rows 2 r and 2 r + 1 belong to lo row r
        // the wave's LDS operations execute in order: no barrier, only keep the compiler from moving them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        issue(row + n_waves);
#pragma unroll
        for (int q = 0; q < 8; ++q) {                  // positions 4 q .. 4 q + 3, this lane's slot: c = 4 q + g
            const int c = 4 * q + g;
            const float b = iw[ky * IMS + 3 + 2 * c + kx];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const float a = lw[c * WS1 + 16 * mt + li];
                lo_sum[mt] += a;
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[mt], 0, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // reduce the waves' tiles: red[wave][mt*4 + r][lane]; D row = 4g + r -> clo = 16 mt + 4g + r, column = tap li
    float *red = lo_s;
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * 8 + mt * 4 + r) * 64 + lane] = acc[mt][r];
    __syncthreads();
    float *out = slab + (int64_t)BID * WG1_SLAB;
    {                                                   // wave w finishes register w of the 8
        const int reg = wave, mt = reg >> 2, r = reg & 3;
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < WGS_WAVES; ++w) tot += red[(w * 8 + reg) * 64 + lane];
        out[(16 * mt + 4 * g + r) * 16 + li] = tot;
    }
    // bias sums: lo per channel (lane li of tile mt, over the 4 position slots and the waves), image total
    __syncthreads();
    constexpr int T = 64 * WGS_WAVES;
    red[threadIdx.x] = lo_sum[0];
    red[T + threadIdx.x] = lo_sum[1];
    red[2 * T + threadIdx.x] = img_sum;
    __syncthreads();
    if (threadIdx.x < CC) {
        const int mt = threadIdx.x >> 4, i = threadIdx.x & 15;
        float tot = 0.f;
        for (int j = 0; j < T / 16; ++j) tot += red[mt * T + j * 16 + i];
        out[CC * 16 + threadIdx.x] = tot;
    } else if (threadIdx.x >= 64 && threadIdx.x < 128) {
        float tot = 0.f;
        for (int j = lane; j < T; j += 64) tot += red[2 * T + j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
        if (lane == 0) out[CC * 16 + CC] = tot;
    }
}

}  // namespace arvae
