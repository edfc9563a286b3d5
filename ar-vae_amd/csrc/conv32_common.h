// Shared pieces of the 32-channel conv kernels (conv32.hip, conv32r.hip): vector types, LDS pitches, raw buffer access,
// the bf16 three-term split and the transposing LDS read.
#pragma once
#include "common.h"

#include <type_traits>

namespace arvae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef int i32x4v __attribute__((ext_vector_type(4)));

constexpr int C32 = 32;
constexpr int PS = 36;                  // LDS pixel stride in floats
constexpr int PSB3 = 52;                // LDS pixel stride in dwords of the packed three-term image (terms at +0, +16, +32 dwords,
                                        // 4 pad): conflict-free 16-byte reads for pixel walks of stride 1 and 2, 13 % smaller than
                                        // three padded planes, which lets a second buffer fit
constexpr int PSB = 20;                 // LDS pixel stride in dwords of one bf16 plane (16 payload + 4 pad: conflict-free
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
// byte offset of hi pixel (2r, 2c) of lo pixel p inside a [*, 2LO, 2LO, 32] tensor, relative to hi (img0, 2*r0, 0)
template <int LO, int PX = 128> __device__ __forceinline__ constexpr int hi_rel(int p) {
    int img = 0, r = 0, c = 0;
    tile_pixel<LO, PX>(p, img, r, c);
    return ((img * 2 * LO + 2 * r) * 2 * LO + 2 * c) * PIXB;
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
    const uint4 *wprep;         // null, or this layer's weights already split and laid out per lane (conv32_weight_prep)
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
// fp32 pair -> two packed bf16 pairs (hi, mid), both round-to-nearest-even (v_cvt_pk_bf16_f32): x = hi + mid + e with
// |e| <= 2^-18 |x|.  Low half of a dword = first value.
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned &hi, unsigned &mid) {
    const f32x2v x = {x0, x1};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2v));
    const f32x2v r = {x0 - __builtin_bit_cast(float, hi << 16), x1 - __builtin_bit_cast(float, hi & 0xffff0000u)};
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2v));
}

// three-term version: x = hi + mid + lo with a residual <= 2^-26 |x| (exact for all but the last bit or two of x)
__device__ __forceinline__ void split_pair3(float x0, float x1, unsigned &hi, unsigned &mid, unsigned &lo) {
    const f32x2v x = {x0, x1};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2v));
    const f32x2v r = {x0 - __builtin_bit_cast(float, hi << 16), x1 - __builtin_bit_cast(float, hi & 0xffff0000u)};
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2v));
    const f32x2v q = {r.x - __builtin_bit_cast(float, mid << 16), r.y - __builtin_bit_cast(float, mid & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16x2v));
}

// x = hi + mid + lo EXACTLY by truncation (8 + 8 + 8 significant bits; every subtraction is exact), packed in pairs with
// v_perm_b32: 11 single-issue vector instructions per two values.  What a PRODUCER wave beside an MFMA wave uses: the
// round-to-nearest split above compiles to v_cvt_pk_bf16_f32 + v_pk_add_f32, and a packed-f32 instruction costs the partner
// wave of an MFMA wave three of the ~3.5 issue slots it gets per MFMA (tools/probes/coissue.hip).
__device__ __forceinline__ void trunc_pair3(float x0, float x1, unsigned &hi, unsigned &mid, unsigned &lo) {
    const unsigned u0 = __builtin_bit_cast(unsigned, x0), u1 = __builtin_bit_cast(unsigned, x1);
    float r0 = x0 - __builtin_bit_cast(float, u0 & 0xffff0000u), r1 = x1 - __builtin_bit_cast(float, u1 & 0xffff0000u);
    asm volatile("" : "+v"(r0), "+v"(r1));                // keep the two subtractions scalar (no v_pk_add_f32)
    const unsigned m0 = __builtin_bit_cast(unsigned, r0), m1 = __builtin_bit_cast(unsigned, r1);
    float q0 = r0 - __builtin_bit_cast(float, m0 & 0xffff0000u), q1 = r1 - __builtin_bit_cast(float, m1 & 0xffff0000u);
    asm volatile("" : "+v"(q0), "+v"(q1));
    hi = __builtin_amdgcn_perm(u1, u0, 0x07060302u);     // upper halves: {x1.hi16, x0.hi16}
    mid = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
    lo = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, q1), __builtin_bit_cast(unsigned, q0), 0x07060302u);
}

// Per-layer prepared weights (conv32_weight_prep_kernel, conv32.hip, once per training step): the three-term split of wt in
// per-lane MFMA operand order, 16 bytes per (slot, lane) with lanes contiguous.
//   DOWN part: [kh 2][slot 48 = (tap 8 = kyl*4 + kx, c 2, term 3)][lane 64], ky = 2 kh + kyl: lane (rc, half) holds the input
//              channels c*16 + half*8 + j of wt[clo = rc][.][ky][kx]
//   UP part:   [class 4][slot 24 = (ty, tx, c, term)][lane 64]
constexpr int PREP_DOWN_SLOTS = 48, PREP_UP_SLOTS = 24;
constexpr int PREP_DOWN_UINT4 = 2 * PREP_DOWN_SLOTS * 64, PREP_UP_UINT4 = 4 * PREP_UP_SLOTS * 64;
constexpr int PREP_FLOATS = (PREP_DOWN_UINT4 + PREP_UP_UINT4) * 4;

// byte offset of a lane's (pixel, half) entry in a relu_bits16 array, from its byte offset pixel*128 + half*16
__device__ __forceinline__ unsigned bits_off(unsigned out_off, int half) { return (out_off >> 7) * 4 + half * 2; }

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA_B(ACC, W, A) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W, A, ACC, 0, 0, 0)

typedef short s16x4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 lds_tr_bf16x8(const unsigned *p0, const unsigned *p1) {
    typedef __attribute__((address_space(3))) s16x4v *lds_ptr;
    const s16x4v a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p0);
    const s16x4v b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p1);
    typedef short s16x8v __attribute__((ext_vector_type(8)));
    const s16x8v v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
}


}  // namespace arvae
