// The three-term bf16 arithmetic of the wide (64-channel) convolution kernels: x = hi + mid + lo with a residual <= 2^-26 |x|,
// a product = the six partial products of weight >= 2^-18 on v_mfma_f32_32x32x16_bf16, fp32 accumulation, smallest first.
// (The 32-channel kernels moved to scaled two-term fp16 operands in round 3, conv32_common.h: half the MFMAs at the same
// measured accuracy; these kernels have not been moved yet.)
#pragma once
#include "common.h"

namespace arvae {

typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA_B(ACC, W, A) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W, A, ACC, 0, 0, 0)

// fp32 pair -> three packed bf16 pairs, every step round-to-nearest-even (v_cvt_pk_bf16_f32); low half of a dword = first value
__device__ __forceinline__ void split_pair3(float x0, float x1, unsigned &hi, unsigned &mid, unsigned &lo) {
    typedef float f32x2q __attribute__((ext_vector_type(2)));
    const f32x2q x = {x0, x1};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2v));
    const f32x2q r = {x0 - __builtin_bit_cast(float, hi << 16), x1 - __builtin_bit_cast(float, hi & 0xffff0000u)};
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2v));
    const f32x2q q = {r.x - __builtin_bit_cast(float, mid << 16), r.y - __builtin_bit_cast(float, mid & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16x2v));
}

}  // namespace arvae
