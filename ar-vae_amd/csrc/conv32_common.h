// Shared pieces of the 32-channel conv kernels (conv32.hip, conv32k.hip, conv32r.hip): vector types, LDS pitches, raw buffer
// access, the scaled fp16 two-term split with its per-tensor maxima, and the transposing LDS read.
#pragma once
#include "common.h"

#include <type_traits>

namespace arvae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef int i32x4v __attribute__((ext_vector_type(4)));

constexpr int C32 = 32;
constexpr int PSB2 = 36;                // LDS pixel stride in dwords of the packed two-term image (terms at +0, +16 dwords, 4 pad):
                                        // conflict-free 16-byte reads for pixel walks of stride 1 and 2 (eight lanes per clock:
                                        // 36 rc and 72 rc mod 64 are eight disjoint groups of four banks)
constexpr int PSB = 20;                 // LDS pixel stride in dwords of one 16-bit plane (16 payload + 4 pad: conflict-free
                                        // 16-byte reads for pixel walks of stride 1 and 2)
#ifndef ARVAE_WGRAD_PSB_H
#define ARVAE_WGRAD_PSB_H 24
#endif
#ifndef ARVAE_WGRAD_PSB_L
#define ARVAE_WGRAD_PSB_L 16
#endif
constexpr int WGRAD_PSB_H = ARVAE_WGRAD_PSB_H, WGRAD_PSB_L = ARVAE_WGRAD_PSB_L;   // plane pitches of wgrad32x_kernel (see there)
constexpr int PIXB = C32 * 4;           // bytes of one 32-channel pixel
constexpr unsigned OOB = 0x7fffffffu;   // byte offset beyond any tensor here: loads return 0, stores are dropped

// tile geometry per lo-resolution size LO (hi = 2*LO): PX lo pixels = PX/32 MFMA tiles of full-width rows (TC == LO):
//   LO 16, PX 128: 8 rows of one image      LO 8, PX 128: two whole images      LO 4, PX 128: eight whole images
//   LO 4, PX 32: two whole images (the small-problem variant: 4x more tiles when 128-pixel tiles leave CUs idle)
template <int LO, int PX = 128> struct Tile {
    static constexpr int ROWS = PX / LO;
    static constexpr int TI = ROWS > LO ? ROWS / LO : 1, TR = ROWS > LO ? LO : ROWS, TC = LO;
};

// lo pixel p of a tile -> (image, row, col) inside the tile; in memory pixel p sits p*PIXB after the tile start
template <int LO, int PX = 128> __device__ __forceinline__ constexpr void tile_pixel(int p, int &img, int &r, int &c) {
    using T = Tile<LO, PX>;
    c = p % T::TC;
    r = (p / T::TC) % T::TR;
    img = p / (T::TC * T::TR);
}
template <int LO, int PX = 128> __device__ __forceinline__ void tile_origin(int tile, int &img0, int &r0) {
    using T = Tile<LO, PX>;
    constexpr int TILES_PER_IMG = LO / T::TR;
    img0 = (T::TI == 1) ? tile / TILES_PER_IMG : tile * T::TI;
    r0 = (T::TI == 1) ? (tile % TILES_PER_IMG) * T::TR : 0;
}
// compile-time loop: f(std::integral_constant<int, I>) for I in [0, N); keeps every register-array index static
template <int I, int N, class F> __device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// epilogue modes (one template parameter: a ReLU layer is never gated and vice versa)
enum { EP_PLAIN = 0, EP_RELU = 1, EP_GATE_F = 2, EP_GATE_B = 3 };
struct Ep32 {
    const float *bias;          // per output channel or null
    const float *gate;          // EP_GATE_F: saved activation of the OUTPUT location: result *= (gate > 0)
    const uint16_t *gate_bits;  // EP_GATE_B: the same as sign bits (common.h: relu_bits16)
    uint16_t *bits_out;         // EP_RELU: sign bits of the result for a later gated kernel, may be null
    float *out;
    const uint4 *wprep;         // this layer's weights split and laid out per lane (conv32_prep_block), with their inverse scale
    const unsigned *amax_in;    // AMAX array of the input tensor
    unsigned *amax_out;         // AMAX array of `out`, or null (nobody multiplies it on the matrix pipe)
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, int64_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const f32x4v v = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void buf_store4(float4 v, __amdgpu_buffer_rsrc_t r, unsigned off) {
    f32x4v q;
    q.x = v.x; q.y = v.y; q.z = v.z; q.w = v.w;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4v, q), r, (int)off, 0, 0);
}

__device__ __forceinline__ unsigned buf_load_u16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(r, (int)off, 0, 0);
}
__device__ __forceinline__ void buf_store_u16(unsigned v, __amdgpu_buffer_rsrc_t r, unsigned off) {
    __builtin_amdgcn_raw_buffer_store_b16((unsigned short)v, r, (int)off, 0, 0);
}
// ---- the arithmetic of these kernels: fp16 MFMA on SCALED TWO-TERM operands ---------------------------------------------------
// Every fp32 operand tensor X enters the matrix pipe as s X = h + l, h = fp16(s X) and l = fp16(s X - h), both round-to-nearest,
// where s is the power of two that brings max |X| into [2^14, 2^15) (an fp16 below 2^16; the per-tensor maximum comes with the
// tensor, see AMAX below).  h + l reproduces s X to 2^-22 relative for every element within 2^16 of the tensor's maximum and to
// 2^-39 of that maximum below; a product is the THREE partial products l h', h l', h h' (l l' <= 2^-22 is dropped) on
// v_mfma_f32_32x32x16_f16 with fp32 accumulation, smallest first, and the epilogue multiplies by the two inverse scales (exact).
// Measured against float64 the results sit at 0.7e-7 relative L2 for K = 512 dot products -- the three-term bf16 split these
// kernels used through round 3 (six partial products) measured 0.6e-7, an fp32 FMA chain 2-3e-7 -- at half the MFMAs and two
// thirds of the LDS operand bytes (same-box: -46 us per dSprites step before anything was re-tuned).
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
#define MFMA_H(ACC, W, A) ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(W, A, ACC, 0, 0, 0)

struct Pow2 { float s, inv; };           // a tensor's scale and its inverse
// from the bit pattern of max |X| (0: an all-zero tensor, any scale will do)
__host__ __device__ __forceinline__ Pow2 pow2_for(unsigned amax_bits) {
    int e = (int)((amax_bits >> 23) & 0xffu);                    // biased exponent: 2^(e - 127) <= max |X| < 2^(e - 126)
    e = e < 16 ? 16 : e;                                         // (denormal / zero maxima: scale 2^125, nothing overflows)
    Pow2 r;
    const unsigned sb = (unsigned)(268 - e) << 23, ib = (unsigned)(e - 14) << 23;     // 2^(14 - (e - 127)) and its inverse
#if defined(__HIP_DEVICE_COMPILE__)
    r.s = __builtin_bit_cast(float, sb); r.inv = __builtin_bit_cast(float, ib);
#else
    memcpy(&r.s, &sb, 4); memcpy(&r.inv, &ib, 4);
#endif
    return r;
}

// two values -> packed (h, l) pairs, low half of a dword = first value; eight single-issue instructions per pair (no packed-f32
// instruction: a loader wave runs beside an MFMA wave, and a packed-f32 instruction costs the partner three of the ~3.5 issue
// slots it gets per MFMA, tools/probes/coissue.hip).  A five-instruction form on the mixed-precision FMA (v_fma_mixlo / mixhi_f16
// for h, v_fma_mix_f32 with h as its fp16 addend for the residual) passed every test and measured SLOWER in the loader waves
// (1.3 against 1.1 us per tile of down32p_kernel): those encodings do not co-issue beside the partner's MFMAs either.
__device__ __forceinline__ void split_pair_h2(float x0, float x1, float s, unsigned &hi, unsigned &lo) {
    float y0 = x0 * s, y1 = x1 * s;
    asm volatile("" : "+v"(y0), "+v"(y1));
    const f32x2v y = {y0, y1};
    const f16x2v h = __builtin_convertvector(y, f16x2v);         // v_cvt_pk_f16_f32, round to nearest even
    hi = __builtin_bit_cast(unsigned, h);
    float r0 = y0 - (float)h.x, r1 = y1 - (float)h.y;            // exact
    asm volatile("" : "+v"(r0), "+v"(r1));
    const f32x2v r = {r0, r1};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2v));
}

// ---- AMAX: a tensor's maximum magnitude travels with it as AMAX_N partial maxima (bit patterns of non-negative floats) ------------
// The kernel that WRITES a tensor publishes the maxima of what its G writer units stored (unit u -> entry u, and zeros into the
// entries u + G, u + 2 G, ... no unit owns: the whole array is rewritten by every launch -- no atomics, nothing to clear, the
// same array every time a captured graph replays); the kernels that READ it take the maximum of all AMAX_N entries (4 KB, four
// 16-byte loads per lane, from L2).  G <= AMAX_N is the launcher's business.
constexpr int AMAX_N = 1024;
__device__ __forceinline__ float wave_max(float m) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    return m;
}
// m: this lane's maximum; every lane of the wave calls
__device__ __forceinline__ void amax_publish(unsigned *p, int unit, int units, float m) {
    if (p == nullptr) return;
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0 && unit < AMAX_N) {              // (units <= AMAX_N is the launcher's promise; never write past the array)
        p[unit] = __builtin_bit_cast(unsigned, m);
        for (int e = unit + units; e < AMAX_N; e += units) p[e] = 0u;
    }
}
__device__ __forceinline__ float amax4(const float4 &v) { return fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))); }
// Reading one: the loads are issued FIRST in a kernel (amax_issue) -- memory returns loads in order, so whatever is requested
// after them (several tiles of prefetch) does not stand between them and their use -- and reduced where the scale is first
// needed (amax_scale); every lane of the wave calls both, the result is wave-uniform.
struct AmaxLoad { uint4 v[AMAX_N / 256]; };
__device__ __forceinline__ AmaxLoad amax_issue(const unsigned *p) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p) + (threadIdx.x & 63);
    AmaxLoad a;
#pragma unroll
    for (int i = 0; i < AMAX_N / 256; ++i) a.v[i] = q[64 * i];
    return a;
}
__device__ __forceinline__ Pow2 amax_scale(const AmaxLoad &a) {
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < AMAX_N / 256; ++i) m = max(max(m, a.v[i].x), max(max(a.v[i].y, a.v[i].z), a.v[i].w));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    return pow2_for((unsigned)__builtin_amdgcn_readfirstlane((int)m));
}

// Per-layer prepared weights (conv32_prep_block, prep32.h, once per training step): the scaled two-term split of wt in per-lane
// MFMA operand order, 16 bytes per (slot, lane) with lanes contiguous, followed by the inverse of the layer's weight scale.
//   DOWN part: [kh 2][slot 32 = (tap 8 = kyl*4 + kx, c 2, term 2)][lane 64], ky = 2 kh + kyl: lane (rc, half) holds the input
//              channels c*16 + half*8 + j of wt[clo = rc][.][ky][kx]
//   UP part:   [class 4][slot 16 = (ty, tx, c, term)][lane 64]
//   tail:      one uint4 whose first dword is the inverse weight scale (float)
constexpr int PREP_DOWN_SLOTS = 32, PREP_UP_SLOTS = 16;
constexpr int PREP_DOWN_UINT4 = 2 * PREP_DOWN_SLOTS * 64, PREP_UP_UINT4 = 4 * PREP_UP_SLOTS * 64;
constexpr int PREP_FLOATS = (PREP_DOWN_UINT4 + PREP_UP_UINT4 + 1) * 4;
__device__ __forceinline__ float prep_inv_scale(const uint4 *wprep) {
    return __builtin_bit_cast(float, wprep[PREP_DOWN_UINT4 + PREP_UP_UINT4].x);
}

// byte offset of a lane's (pixel, half) entry in a relu_bits16 array, from its byte offset pixel*128 + half*16
__device__ __forceinline__ unsigned bits_off(unsigned out_off, int half) { return (out_off >> 7) * 4 + half * 2; }

typedef short s16x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f16x8 lds_tr_f16x8(const unsigned *p0, const unsigned *p1) {
    typedef __attribute__((address_space(3))) s16x4v *lds_ptr;
    const s16x4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p0);
    const s16x4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p1);
    typedef short s16x8v __attribute__((ext_vector_type(8)));
    const s16x8v v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(f16x8, v);
}


}  // namespace arvae
