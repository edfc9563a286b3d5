// Tiles of the split-bf16 ("x3") MFMA GEMM kernels: an fp32 operand is carried as three bf16 terms (hi + mid + lo, exact
// to 2^-26) in three LDS planes, and a multiply-add is six partial products on v_mfma_f32_32x32x16_bf16, smallest first
// (fp32-accurate, 2.7x fewer MFMA cycles than the fp32 MFMA).  Shared by dense.hip (rows GEMM) and conv64.hip.
#pragma once
#include "common.h"

namespace arvae {

constexpr int RG_R = 32;                     // reduction indices per chunk
enum { RG_ROWSK = 0, RG_KROWS = 1 };         // operand memory order: reduction index contiguous / output index contiguous

// fp32 pair -> three bf16 terms each (hi + mid + lo exact to 2^-26), packed (first value in the low half)
typedef __bf16 rg_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 rg_bf16x8 __attribute__((ext_vector_type(8)));
typedef float rg_f32x2 __attribute__((ext_vector_type(2)));
typedef int rg_i32x4 __attribute__((ext_vector_type(4)));
constexpr int RG_XP = RG_R + 8;              // bf16 row pitch of the split planes (16 bytes of padding)
constexpr int RG_TRP = 96;                   // bf16 row pitch of a "K x rows" image (64 columns + padding: 192 bytes)
__device__ __forceinline__ void rg_split3(float x0, float x1, unsigned &hi, unsigned &mid, unsigned &lo) {
    const rg_f32x2 x = {x0, x1};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x, rg_bf16x2));
    const rg_f32x2 r = {x0 - __builtin_bit_cast(float, hi << 16), x1 - __builtin_bit_cast(float, hi & 0xffff0000u)};
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r, rg_bf16x2));
    const rg_f32x2 q = {r.x - __builtin_bit_cast(float, mid << 16), r.y - __builtin_bit_cast(float, mid & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(q, rg_bf16x2));
}
__device__ __forceinline__ rg_bf16x8 rg_lds_x8(const unsigned short *p) {
    return __builtin_bit_cast(rg_bf16x8, *reinterpret_cast<const rg_i32x4 *>(p));
}

// A matrix [rows][K] as bf16 planes in memory for the wide tile GEMMs (dense.hip WidePlanes): TILED [K / 32][rows][32] -- the
// (32 reduction indices) x (all rows) block of one chunk is contiguous, whichever axis a product reduces along.  Element index
// of (row, k) inside a plane:
__host__ __device__ __forceinline__ int64_t x3_tiled_index(int64_t rows, int64_t row, int64_t k) { return ((k >> 5) * rows + row) * 32 + (k & 31); }

// One operand tile (TP output indices x RG_R reduction indices) as three bf16 planes.
//   "rows x K" (RG_ROWSK): [TP][RG_XP], reduction index contiguous; the MFMA operand (8 consecutive r per lane) is one
//   ds_read_b128.   "K x rows" (RG_KROWS): memory order kept, [RG_R][RG_TRP] with the output index contiguous (one 8-byte
//   write per plane) and read through ds_read_b64_tr_b16, the transposing read; the 192-byte row pitch puts the four rows
//   of a transposed block on disjoint banks.
template <int LAY, int TP>
struct X3Plane {
    // row pitch of a "K x rows" image: TP columns + padding that keeps the four rows of a transposed block on disjoint bank
    // quarters (192 bytes for 64 columns, 320 for 128: both 16 banks past a multiple of 32)
    static constexpr int TRP = (LAY == RG_KROWS && TP > 64) ? TP + 32 : RG_TRP;
    static constexpr int PLANE = LAY == RG_ROWSK ? TP * RG_XP : RG_R * TRP;
    // element idx of the tile's float4 grid (TP * RG_R / 4 of them): 4 values consecutive along the contiguous axis
    __device__ static __forceinline__ void commit(unsigned short *lds, int idx, const float4 &v) {
        unsigned h0, m0, l0, h1, m1, l1;
        rg_split3(v.x, v.y, h0, m0, l0);
        rg_split3(v.z, v.w, h1, m1, l1);
        unsigned short *d;
        if (LAY == RG_ROWSK) d = lds + (idx / (RG_R / 4)) * RG_XP + 4 * (idx % (RG_R / 4));
        else d = lds + (idx / (TP / 4)) * TRP + 4 * (idx % (TP / 4));
        *reinterpret_cast<uint2 *>(d) = uint2{h0, h1};
        *reinterpret_cast<uint2 *>(d + PLANE) = uint2{m0, m1};
        *reinterpret_cast<uint2 *>(d + 2 * PLANE) = uint2{l0, l1};
    }
    // this lane's MFMA operand (8 consecutive reduction indices 16 s + 8 (lane >> 5) .. of output index 32 w + (lane & 31))
    // of plane t; `base` = lane_base(w)
    __device__ static __forceinline__ int lane_base(int w) {
        const int lane = threadIdx.x & 63;
        if (LAY == RG_ROWSK) return (32 * w + (lane & 31)) * RG_XP + 8 * (lane >> 5);
        const int g16 = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;      // block row q, columns 4 pp .. 4 pp + 3
        return (8 * (g16 >> 1) + q) * TRP + 32 * w + 16 * (g16 & 1) + 4 * pp;
    }
    __device__ static __forceinline__ rg_bf16x8 operand(const unsigned short *lds, int base, int t, int s) {
        if (LAY == RG_ROWSK) return rg_lds_x8(lds + t * PLANE + base + 16 * s);
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        typedef __attribute__((address_space(3))) s16x4 *lds_ptr;
        const unsigned short *p = lds + t * PLANE + base + 16 * s * TRP;
        const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p);
        const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 4 * TRP));
        return __builtin_bit_cast(rg_bf16x8, (s16x8)__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
    }
};

// acc += A . B for one 16-deep k-step: the six products, smallest first
#define X3_MFMA6(ACC, AH, AM, AL, BH, BM, BL)                                      \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AL, BH, ACC, 0, 0, 0);           \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, BL, ACC, 0, 0, 0);           \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AM, BM, ACC, 0, 0, 0);           \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AM, BH, ACC, 0, 0, 0);           \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, BM, ACC, 0, 0, 0);           \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AH, BH, ACC, 0, 0, 0)

}  // namespace arvae
