// Whole-sequence GRU kernels of the MeasureVAE path: all T time steps of one GRU layer (any number of independent
// directions / parameter sets) in ONE launch, forward and backward-through-time.
//
// Reference: nn.GRU inside measurevae/encoder.py:27-34,113-118 (2-layer bidirectional, 24 ticks) and
// measurevae/decoder.py:338-368,436-525 (beat RNN 4 steps, tick RNN 6 steps per beat), gate order r | z | n:
//     r = sigmoid(gi_r + gh_r);  z = sigmoid(gi_z + gh_z);  n = tanh(gi_n + r * gh_n);  h' = (1-z)*n + z*h
// with gi = W_ih x + b_ih computed for all time steps by one dense launch beforehand and gh = W_hh h + b_hh here.
//
// Every batch row is an independent recurrence, so a workgroup owns 16 rows for the whole sequence and never talks
// to another workgroup: H/16 waves, wave w owns hidden units [16w, 16w+16) of all three gates and keeps its
// 3 x 16 x H slice of W_hh in registers (96 VGPRs at H = 128) for all T steps.  Per step: h (16 x H, LDS, double
// buffered) times the slice on v_mfma_f32_16x16x4_f32 (the 16x16 result tile = 4 rows x 1 unit per lane for each
// gate, so the gate math is lane-local), the new h goes back to LDS, one barrier.  gi of the next step is prefetched
// while the MFMAs run.  The backward kernel keeps W_hh^T the same way and carries dL/dh in registers.
//
// The weight gradients (dW_hh = dgh^T h_prev, dW_ih = dgi^T x) and the input gradients are ordinary dense launches
// over all T*R rows afterwards (ops.py).
#include <algorithm>
#include "diag.h"
#include "common.h"
#include "gru_mask.h"

namespace arvae {

constexpr int GRU_SEQ_MAX = 4;

struct GruSeq {
    // forward
    const float *gi;        // [T][R][3H]  (gi_tstride floats between steps; 0 = the same block every step)
    int64_t gi_tstride;
    const float *w_hh;      // [3H][H]
    const float *b_hh;      // [3H]
    const float *h0;        // [R][H] or null (zeros)
    float *h_all;           // h of step t, row r, unit j at h_all[(t*R + r) * h_stride + j]
    int64_t h_stride;
    float *saved;           // [T][R][H][4] : r, z, n, gh_n
    int reverse;            // process t = T-1 .. 0
    // backward
    const float *dh_all;    // gradient w.r.t. h_all, same addressing with dh_stride; may be null
    int64_t dh_stride;
    float *dgi;             // [T][R][3H]
    float *dgh;             // [T][R][3H]
    float *dh0;             // [R][H] or null
    const float *dh_last;   // gradient w.r.t. the final state (h of the last processed step), [R] rows of dh_last_stride; may be null
    int64_t dh_last_stride;
    float *h_prev_out;      // [T][R][H]: h entering step t (the operand of the W_hh weight gradient); may be null
    // merged projections (both directions of a layer as ONE GEMM write / read [T][R][ndir * 3H]): floats between two rows of gi /
    // of dgi (fill_batch sets 3H when the caller leaves them 0)
    int64_t gi_rstride, dgi_rstride;
    float *h_fin;           // forward, optional: the state after the last processed step, row r at h_fin + r * h_fin_stride
    int64_t h_fin_stride;
    int64_t h0_stride, dh0_stride;   // floats between two rows of h0 / dh0 (fill_batch: H when the caller leaves them 0)
    // dropout on the sequence's output (gru_mask.h; the fp16 two-term kernels only): mask null = none
    const uint8_t *mask;
    float keep;
    float *h_masked;
    int hm_stride, mk_tstride, mk_rstride, mk_gstride, mk_group;
};
struct GruSeqBatch {
    GruSeq seq[GRU_SEQ_MAX];
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

// reciprocals on v_rcp_f32 (1 ulp): __frcp_rn is a correctly rounded division -- v_div_scale x 2, v_rcp, four fused steps,
// v_div_fmas, v_div_fixup -- and three of them per hidden unit and step were ~10 % of the forward recurrence's instructions
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) {
    // 1 - 2 / (1 + e^{2x}); saturates correctly at both ends (e -> inf gives 1, e -> 0 gives -1)
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x));
}

// Addressing of the recurrences' per-step memory operations: raw buffer operations, byte offset = a SCALAR part (the step's
// t * R * row pitch: one s_mul per array and step) + a per-lane part (row * pitch + unit: one v_mad_u32_u24 per operation).  With
// 64-bit pointer arithmetic each of a step's ~40 loads and stores cost 6-10 vector instructions -- half of what a wave executes
// per step in kernels that are bound by exactly that (16-32 workgroups on the chip, every step a chain of dependent phases).
// A null array is an empty range: its loads return zero and its stores are dropped, no branch; a lane drops a store with the
// per-lane offset GRU_DEAD (the hardware checks the per-lane offset + the instruction's immediate against the range; the scalar
// offset is NOT checked -- a step without an operation selects the empty range instead).  The entry points bound the arrays at
// GRU_RANGE bytes.
constexpr int GRU_RANGE = 0x7fff0000, GRU_DEAD = 0x7fff0000;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t gru_rsrc(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, p != nullptr ? GRU_RANGE : 0, 0x00020000);
}
__device__ __forceinline__ float gru_ld(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ f32x4 gru_ld4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void gru_st(float v, __amdgpu_buffer_rsrc_t r, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, voff, soff, 0);
}
__device__ __forceinline__ int gru_off(int row, int pitch_bytes, int base_bytes) { return (int)__umul24(row, pitch_bytes) + base_bytes; }

template <int H>
__global__ __launch_bounds__(H * 4) void gru_seq_fwd_kernel(GruSeqBatch batch, int T, int R) {
    constexpr int KQ = H / 16;             // groups of 16 k values (4 MFMAs each)
    constexpr int HS = H + 4;              // LDS row stride (floats)
    __shared__ float hbuf[2][16][HS];
    const GruSeq &s = batch.seq[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, quad = lane >> 4;
    const int unit = 16 * w + col;
    const int row0 = blockIdx.x * 16;

    f32x4 wreg[3][KQ];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq)
            wreg[g][kq] = *reinterpret_cast<const f32x4 *>(s.w_hh + (int64_t)(g * H + unit) * H + 16 * kq + 4 * quad);
    const float bh_r = s.b_hh[unit], bh_z = s.b_hh[H + unit], bh_n = s.b_hh[2 * H + unit];

    int rows[4];
    bool live[4];
    float h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + 4 * quad + i;
        live[i] = r < R;
        rows[i] = live[i] ? r : R - 1;
        h[i] = (s.h0 != nullptr && live[i]) ? s.h0[(int64_t)rows[i] * s.h0_stride + unit] : 0.f;
        hbuf[0][4 * quad + i][unit] = h[i];
    }
    float gi_next[4][3];
    {
        const int t0 = s.reverse ? T - 1 : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float *p = s.gi + t0 * s.gi_tstride + (int64_t)rows[i] * s.gi_rstride + unit;
            gi_next[i][0] = p[0]; gi_next[i][1] = p[H]; gi_next[i][2] = p[2 * H];
        }
    }
    lds_barrier();

    for (int step = 0; step < T; ++step) {
        const int t = s.reverse ? T - 1 - step : step;
        const int cur = step & 1;
        float gi[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 3; ++g) gi[i][g] = gi_next[i][g];
        if (step + 1 < T) {
            const int tn = s.reverse ? t - 1 : t + 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float *p = s.gi + tn * s.gi_tstride + (int64_t)rows[i] * s.gi_rstride + unit;
                gi_next[i][0] = p[0]; gi_next[i][1] = p[H]; gi_next[i][2] = p[2 * H];
            }
        }
        f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(&hbuf[cur][col][16 * kq + 4 * quad]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], wreg[g][kq][j], acc[g], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float r = fast_sigmoid(gi[i][0] + acc[0][i] + bh_r);
            const float z = fast_sigmoid(gi[i][1] + acc[1][i] + bh_z);
            const float ghn = acc[2][i] + bh_n;
            const float n = fast_tanh(gi[i][2] + r * ghn);
            const float hn = (1.f - z) * n + z * h[i];
            h[i] = hn;
            hbuf[cur ^ 1][4 * quad + i][unit] = hn;
            if (live[i]) {
                const int64_t tr = (int64_t)t * R + rows[i];
                s.h_all[tr * s.h_stride + unit] = hn;
                *reinterpret_cast<f32x4 *>(s.saved + (tr * H + unit) * 4) = f32x4{r, z, n, ghn};
                if (step == T - 1 && s.h_fin != nullptr) s.h_fin[(int64_t)rows[i] * s.h_fin_stride + unit] = hn;
            }
        }
        lds_barrier();
    }
}

// Backward through time.  Per step (in the reverse of the forward's processing order):
//   g = dh_all[t] + carry;  dpn = g (1-z)(1-n^2);  dpz = g (h_prev - n) z (1-z);  dpr = dpn gh_n r (1-r)
//   dgi = [dpr, dpz, dpn];  dgh = [dpr, dpz, dpn r];  carry = g z + dgh . W_hh
template <int H>
__global__ __launch_bounds__(H * 4) void gru_seq_bwd_kernel(GruSeqBatch batch, int T, int R) {
    constexpr int KQ = 3 * H / 16;
    constexpr int DS = 3 * H + 4;
    __shared__ float dbuf[2][16][DS];
    const GruSeq &s = batch.seq[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, quad = lane >> 4;
    const int unit = 16 * w + col;
    const int row0 = blockIdx.x * 16;

    // B[k = c][n = unit] = W_hh[c][unit], c = 16 kq + 4 quad + j
    f32x4 wreg[KQ];
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq)
#pragma unroll
        for (int j = 0; j < 4; ++j) wreg[kq][j] = s.w_hh[(int64_t)(16 * kq + 4 * quad + j) * H + unit];

    int rows[4];
    bool live[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + 4 * quad + i;
        live[i] = r < R;
        rows[i] = live[i] ? r : R - 1;
    }
    float carry[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        carry[i] = (s.dh_last != nullptr && live[i]) ? s.dh_last[(int64_t)rows[i] * s.dh_last_stride + unit] : 0.f;

    // operands of one step: dh, r, z, n, gh_n, h_prev
    float nx[4][6];
    auto fetch = [&](int step) {
        const int t = s.reverse ? step : T - 1 - step;
        const bool has_prev = step + 1 < T;
        const int tp = s.reverse ? t + 1 : t - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t tr = (int64_t)t * R + rows[i];
            nx[i][0] = s.dh_all != nullptr ? s.dh_all[tr * s.dh_stride + unit] : 0.f;
            const f32x4 sv = *reinterpret_cast<const f32x4 *>(s.saved + (tr * H + unit) * 4);
            nx[i][1] = sv[0]; nx[i][2] = sv[1]; nx[i][3] = sv[2]; nx[i][4] = sv[3];
            if (has_prev) nx[i][5] = s.h_all[((int64_t)tp * R + rows[i]) * s.h_stride + unit];
            else nx[i][5] = s.h0 != nullptr ? s.h0[(int64_t)rows[i] * s.h0_stride + unit] : 0.f;
        }
    };
    fetch(0);

    for (int step = 0; step < T; ++step) {
        const int t = s.reverse ? step : T - 1 - step;
        const int cur = step & 1;
        float gz[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float g = live[i] ? nx[i][0] + carry[i] : 0.f;
            const float r = nx[i][1], z = nx[i][2], n = nx[i][3], ghn = nx[i][4], hp = nx[i][5];
            const float dpn = g * (1.f - z) * (1.f - n * n);
            const float dpz = g * (hp - n) * z * (1.f - z);
            const float dpr = dpn * ghn * r * (1.f - r);
            const float dhn = dpn * r;
            gz[i] = g * z;
            float *d = &dbuf[cur][4 * quad + i][unit];
            d[0] = dpr; d[H] = dpz; d[2 * H] = dhn;
            if (live[i]) {
                const int64_t o = ((int64_t)t * R + rows[i]) * 3 * H + unit;
                const int64_t og = ((int64_t)t * R + rows[i]) * s.dgi_rstride + unit;
                s.dgi[og] = dpr; s.dgi[og + H] = dpz; s.dgi[og + 2 * H] = dpn;
                s.dgh[o] = dpr; s.dgh[o + H] = dpz; s.dgh[o + 2 * H] = dhn;
                if (s.h_prev_out != nullptr) s.h_prev_out[((int64_t)t * R + rows[i]) * H + unit] = hp;
            }
        }
        if (step + 1 < T) fetch(step + 1);
        lds_barrier();
        f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(&dbuf[cur][col][16 * kq + 4 * quad]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[kq % 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], wreg[kq][j], acc[kq % 3], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) carry[i] = gz[i] + (acc[0][i] + acc[1][i] + acc[2][i]);
    }
    if (s.dh0 != nullptr)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (live[i]) s.dh0[(int64_t)rows[i] * s.dh0_stride + unit] = carry[i];
}

// ------------------------------------------------------------------------------------------------------------------
// The same two kernels on the bf16 MFMA at fp32 accuracy (the three-term split of conv32.hip): W_hh is split once into
// hi + mid + lo bf16 terms per lane (144 VGPRs at H = 128), h (or dgh) is split when it is written to LDS as three
// bf16 planes, and a multiply-add is the six partial products >= 2^-18 on v_mfma_f32_16x16x32_bf16 (16 cycles each
// instead of 8 x 32 for the fp32 16x16x4), smallest first: 72 instead of 96 MFMAs per wave and step at half the
// cycles each.  Results are within one fp32 rounding of the fp32-MFMA kernels' (same tests, same tolerances).
typedef __bf16 bf16x8g __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2g __attribute__((ext_vector_type(2)));
typedef float f32x2g __attribute__((ext_vector_type(2)));
typedef int i32x4g __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned &hi, unsigned &mid, unsigned &lo) {
    const f32x2g x = {x0, x1};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2g));
    const f32x2g r = {x0 - __builtin_bit_cast(float, hi << 16), x1 - __builtin_bit_cast(float, hi & 0xffff0000u)};
    mid = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2g));
    const f32x2g q = {r.x - __builtin_bit_cast(float, mid << 16), r.y - __builtin_bit_cast(float, mid & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(q, bf16x2g));
}
__device__ __forceinline__ void split3_x8(const float (&x)[8], bf16x8g &hi, bf16x8g &mid, bf16x8g &lo) {
    i32x4g h, m, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned a, b, c;
        split3_pair(x[2 * j], x[2 * j + 1], a, b, c);
        h[j] = (int)a; m[j] = (int)b; l[j] = (int)c;
    }
    hi = __builtin_bit_cast(bf16x8g, h); mid = __builtin_bit_cast(bf16x8g, m); lo = __builtin_bit_cast(bf16x8g, l);
}
// one value -> its three bf16 terms into the three LDS planes (plane stride in ushorts)
__device__ __forceinline__ void store_split3(unsigned short *p, int plane, float x) {
    unsigned a, b, c;
    split3_pair(x, 0.f, a, b, c);
    p[0] = (unsigned short)a; p[plane] = (unsigned short)b; p[2 * plane] = (unsigned short)c;
}
// two values (rows `rowpitch` apart in every plane) for the price of one split
__device__ __forceinline__ void store_split3_pair(unsigned short *p, int rowpitch, int plane, float x0, float x1) {
    unsigned a, b, c;
    split3_pair(x0, x1, a, b, c);
    p[0] = (unsigned short)a; p[rowpitch] = (unsigned short)(a >> 16);
    p[plane] = (unsigned short)b; p[plane + rowpitch] = (unsigned short)(b >> 16);
    p[2 * plane] = (unsigned short)c; p[2 * plane + rowpitch] = (unsigned short)(c >> 16);
}
__device__ __forceinline__ bf16x8g lds_x8(const unsigned short *p) {
    return __builtin_bit_cast(bf16x8g, *reinterpret_cast<const i32x4g *>(p));
}
// acc += a . w with a = (ah, am, al), w = (wh, wm, wl): the six products, smallest first
#define GRU_MFMA6(ACC, AH, AM, AL, WH, WM, WL)                                         \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AL, WH, ACC, 0, 0, 0);               \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AH, WL, ACC, 0, 0, 0);               \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AM, WM, ACC, 0, 0, 0);               \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AM, WH, ACC, 0, 0, 0);               \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AH, WM, ACC, 0, 0, 0);               \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_bf16(AH, WH, ACC, 0, 0, 0)

// three independent accumulators, product-major: consecutive MFMAs never hit the same accumulator (a dependent 16x16x32 MFMA
// waits for its predecessor's result, about twice the issue interval), every accumulator still sees its six products in order
#define GRU_MFMA6X3(A0, A1, A2, H0, M0, L0, H1, M1, L1, H2, M2, L2, WH0, WM0, WL0, WH1, WM1, WL1, WH2, WM2, WL2)                 \
    A0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(L0, WH0, A0, 0, 0, 0); A1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(L1, WH1, A1, 0, 0, 0); \
    A2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(L2, WH2, A2, 0, 0, 0);                                                          \
    A0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(H0, WL0, A0, 0, 0, 0); A1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(H1, WL1, A1, 0, 0, 0); \
    A2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(H2, WL2, A2, 0, 0, 0);                                                          \
    A0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(M0, WM0, A0, 0, 0, 0); A1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(M1, WM1, A1, 0, 0, 0); \
    A2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(M2, WM2, A2, 0, 0, 0);                                                          \
    A0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(M0, WH0, A0, 0, 0, 0); A1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(M1, WH1, A1, 0, 0, 0); \
    A2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(M2, WH2, A2, 0, 0, 0);                                                          \
    A0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(H0, WM0, A0, 0, 0, 0); A1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(H1, WM1, A1, 0, 0, 0); \
    A2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(H2, WM2, A2, 0, 0, 0);                                                          \
    A0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(H0, WH0, A0, 0, 0, 0); A1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(H1, WH1, A1, 0, 0, 0); \
    A2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(H2, WH2, A2, 0, 0, 0)

#ifdef ARVAE_GRU_STAMPS
__device__ unsigned long long g_gru_stamps[8];
__device__ unsigned long long g_tick_stamps[9];
#endif
template <int H>
__global__ __launch_bounds__(H * 4) void gru_seq_fwd_x3_kernel(GruSeqBatch batch, int T, int R) {
    constexpr int KS = H / 32;             // MFMA k-steps of 32
    constexpr int HP = H + 8;              // LDS row pitch in bf16 elements (16 bytes of padding)
    constexpr int PLANE = 16 * HP;
    __shared__ __attribute__((aligned(16))) unsigned short hbuf[2][3 * PLANE];
    const GruSeq &s = batch.seq[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, quad = lane >> 4;
    const int unit = 16 * w + col;
    const int row0 = blockIdx.x * 16;

    bf16x8g wh[3][KS], wm[3][KS], wl[3][KS];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const float *src = s.w_hh + (int64_t)(g * H + unit) * H + 32 * ks + 8 * quad;
            const f32x4 v0 = *reinterpret_cast<const f32x4 *>(src), v1 = *reinterpret_cast<const f32x4 *>(src + 4);
            const float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            split3_x8(x, wh[g][ks], wm[g][ks], wl[g][ks]);
        }
    const float bh_r = s.b_hh[unit], bh_z = s.b_hh[H + unit], bh_n = s.b_hh[2 * H + unit];

    int rows[4];
    bool live[4];
    float h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + 4 * quad + i;
        live[i] = r < R;
        rows[i] = live[i] ? r : R - 1;
        h[i] = (s.h0 != nullptr && live[i]) ? s.h0[(int64_t)rows[i] * s.h0_stride + unit] : 0.f;
        store_split3(&hbuf[0][(4 * quad + i) * HP + unit], PLANE, h[i]);
    }
    // running per-row pointers, advanced by a signed stride every step (the address arithmetic of 12 loads and 8 stores
    // per step was a quarter of the step: the kernel is vector-ALU bound around its MFMAs)
    const int t0 = s.reverse ? T - 1 : 0;
    const int64_t dir = s.reverse ? -1 : 1;
    const int64_t gi_step = dir * s.gi_tstride, h_step = dir * (int64_t)R * s.h_stride, sv_step = dir * (int64_t)R * H * 4;
    const float *gi_p[4];
    float *h_p[4], *sv_p[4];
    float gi_next[4][3];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        gi_p[i] = s.gi + t0 * s.gi_tstride + (int64_t)rows[i] * s.gi_rstride + unit;
        h_p[i] = s.h_all + ((int64_t)t0 * R + rows[i]) * s.h_stride + unit;
        sv_p[i] = s.saved + (((int64_t)t0 * R + rows[i]) * H + unit) * 4;
        gi_next[i][0] = gi_p[i][0]; gi_next[i][1] = gi_p[i][H]; gi_next[i][2] = gi_p[i][2 * H];
        gi_p[i] += gi_step;
    }
    lds_barrier();
    f32x4 keep_sv[4];                        // results of the previous step, stored after the barrier
    float keep_h[4];
    int keep_t = -1;
#ifdef ARVAE_GRU_STAMPS
    unsigned long long ph[4] = {0, 0, 0, 0}, tc = __builtin_readcyclecounter();
#define GSTAMP(k) { const unsigned long long now = __builtin_readcyclecounter(); ph[k] += now - tc; tc = now; }
#else
#define GSTAMP(k)
#endif

    for (int step = 0; step < T; ++step) {
        const int t = s.reverse ? T - 1 - step : step;
        const int cur = step & 1;
        float gi[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 3; ++g) gi[i][g] = gi_next[i][g];
        GSTAMP(0);
        f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const unsigned short *hb = &hbuf[cur][col * HP + 8 * quad];
        // Row i's share of the step's memory traffic (next step's three input projections in, the previous step's h and saved
        // gates out) is issued BEHIND the MFMAs of k-step i, with a scheduling barrier pinning it there: as one block in front
        // of the MFMAs it was 1200 of the step's 6000 cycles (tools/stamp_gru.py), all of it issue time of an in-order wave
        // while the matrix pipe sat idle.
        auto row_traffic = [&](int i) __attribute__((always_inline)) {
            if (step + 1 < T) {
                gi_next[i][0] = gi_p[i][0]; gi_next[i][1] = gi_p[i][H]; gi_next[i][2] = gi_p[i][2 * H];
                gi_p[i] += gi_step;
            }
            if (keep_t >= 0) {
                if (live[i]) {
                    *h_p[i] = keep_h[i];
                    *reinterpret_cast<f32x4 *>(sv_p[i]) = keep_sv[i];
                }
                h_p[i] += h_step; sv_p[i] += sv_step;
            }
        };
        static_assert(KS <= 4, "one row's traffic per k-step; rows left over go last");
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8g ah = lds_x8(hb + 32 * ks), am = lds_x8(hb + PLANE + 32 * ks), al = lds_x8(hb + 2 * PLANE + 32 * ks);
            GRU_MFMA6X3(acc[0], acc[1], acc[2], ah, am, al, ah, am, al, ah, am, al, wh[0][ks], wm[0][ks], wl[0][ks], wh[1][ks], wm[1][ks],
                        wl[1][ks], wh[2][ks], wm[2][ks], wl[2][ks]);
            __builtin_amdgcn_sched_barrier(0);
            row_traffic(ks);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = KS; i < 4; ++i) row_traffic(i);
#ifdef ARVAE_GRU_STAMPS
        { float dep = acc[0][0] + acc[1][0] + acc[2][3]; asm volatile("" :: "v"(dep)); __builtin_amdgcn_s_waitcnt(0); }
#endif
        GSTAMP(1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float r = fast_sigmoid(gi[i][0] + acc[0][i] + bh_r);
            const float z = fast_sigmoid(gi[i][1] + acc[1][i] + bh_z);
            const float ghn = acc[2][i] + bh_n;
            const float n = fast_tanh(gi[i][2] + r * ghn);
            const float hn = (1.f - z) * n + z * h[i];
            h[i] = hn;
            keep_h[i] = hn;
            keep_sv[i] = f32x4{r, z, n, ghn};
        }
        store_split3_pair(&hbuf[cur ^ 1][(4 * quad) * HP + unit], HP, PLANE, h[0], h[1]);
        store_split3_pair(&hbuf[cur ^ 1][(4 * quad + 2) * HP + unit], HP, PLANE, h[2], h[3]);
        keep_t = t;
#ifdef ARVAE_GRU_STAMPS
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the LDS writes are done
#endif
        GSTAMP(2);
        lds_barrier();
        GSTAMP(3);
    }
#ifdef ARVAE_GRU_STAMPS
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        for (int k = 0; k < 4; ++k) g_gru_stamps[k] = ph[k];
        g_gru_stamps[4] = T;
    }
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (live[i]) {
            *h_p[i] = keep_h[i];
            *reinterpret_cast<f32x4 *>(sv_p[i]) = keep_sv[i];
            if (s.h_fin != nullptr) s.h_fin[(int64_t)rows[i] * s.h_fin_stride + unit] = keep_h[i];
        }
}

// ------------------------------------------------------------------------------------------------------------------
// The forward recurrence on the fp16 MFMA with SCALED TWO-TERM operands (the arithmetic of conv32_common.h): s x = h + l with
// h = fp16(s x), l = fp16(s x - h), a product = the three partial products l h', h l', h h' on v_mfma_f32_16x16x32_f16, smallest
// first -- half the MFMAs and two thirds of the LDS operand bytes of the three-term bf16 split at the same accuracy (2^-22 per
// product; measured against float64: conv32_common.h).  fp16 has 5 exponent bits, so the scales must place the operands.
// The SEQUENCE kernels take every scale from the data (round 5): W_hh's slice of a wave its own power of two (a column's scale
// factors out of the dot product), the state the workgroup's max(1, max |h0|) -- a bound for the whole sequence, h_t being a convex
// combination of a tanh output and h_(t-1) --, the backward pass's gradients a scale per batch row and step (gru_seq_bwd_h2_kernel):
// nothing can overflow.  The FREE-RUNNING decoder (tick_free_run_h2_kernel) takes its scales from the data as well: the three matrices'
// from their maxima (tick_weight_amax_kernel, one launch in front of the weight prep; W_ih1 and W_hh1 share a scale because their
// products share accumulators), the states' per beat from the workgroup's rows (a beat's states are convex combinations of tanh
// outputs and the beat's initial state; the layer-1 input is a layer-0 state times 0 or the keep scale).
typedef _Float16 f16x8g __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2g __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split2_pair(float x0, float x1, float s, unsigned &hi, unsigned &lo) {
    const float y0 = x0 * s, y1 = x1 * s;
    const f32x2g y = {y0, y1};
    const f16x2g h = __builtin_convertvector(y, f16x2g);
    hi = __builtin_bit_cast(unsigned, h);
    const f32x2g r = {y0 - (float)h.x, y1 - (float)h.y};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2g));
}
__device__ __forceinline__ void split2_x8(const float (&x)[8], float s, f16x8g &hi, f16x8g &lo) {
    i32x4g h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned a, b;
        split2_pair(x[2 * j], x[2 * j + 1], s, a, b);
        h[j] = (int)a; l[j] = (int)b;
    }
    hi = __builtin_bit_cast(f16x8g, h); lo = __builtin_bit_cast(f16x8g, l);
}
__device__ __forceinline__ f16x8g lds_h8(const unsigned short *p) {
    return __builtin_bit_cast(f16x8g, *reinterpret_cast<const i32x4g *>(p));
}
// three accumulators (the gates), product-major as GRU_MFMA6X3: l h', h l', h h'
#define GRU_MFMA3X3(A0, A1, A2, AH, AL, WH0, WL0, WH1, WL1, WH2, WL2)                                                             \
    A0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(AL, WH0, A0, 0, 0, 0); A1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(AL, WH1, A1, 0, 0, 0); \
    A2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(AL, WH2, A2, 0, 0, 0);                                                            \
    A0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(AH, WL0, A0, 0, 0, 0); A1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(AH, WL1, A1, 0, 0, 0); \
    A2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(AH, WL2, A2, 0, 0, 0);                                                            \
    A0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(AH, WH0, A0, 0, 0, 0); A1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(AH, WH1, A1, 0, 0, 0); \
    A2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(AH, WH2, A2, 0, 0, 0)

// power-of-two scales found at run time (the backward recurrence's per-row scales, both recurrences' per-wave weight scales, the
// forward recurrence's state scale)
struct GruPow2 { float s, inv; };
__device__ __forceinline__ GruPow2 gru_pow2(float amax) {        // 2^k with amax * 2^k in [2^14, 2^15), and 2^-k (amax 0: 2^125)
    int e = (int)((__builtin_bit_cast(unsigned, amax) >> 23) & 0xffu);
    e = e < 16 ? 16 : e;
    return GruPow2{__builtin_bit_cast(float, (unsigned)(268 - e) << 23), __builtin_bit_cast(float, (unsigned)(e - 14) << 23)};
}
template <int CTRL> __device__ __forceinline__ float dpp_max(float v) {
    const float o = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
    return fmaxf(v, o);
}
// maximum over the 16 lanes of a quad (a DPP row), in every lane: quad permutes, the half row and the row mirrored
__device__ __forceinline__ float row16_max(float v) {
    v = dpp_max<0xB1>(v);
    v = dpp_max<0x4E>(v);
    v = dpp_max<0x141>(v);
    return dpp_max<0x140>(v);
}
__device__ __forceinline__ void store_split2_pair_s(unsigned short *p, int rowpitch, int plane, float x0, float x1, float sc) {
    unsigned a, b;
    split2_pair(x0, x1, sc, a, b);
    p[0] = (unsigned short)a; p[rowpitch] = (unsigned short)(a >> 16);
    p[plane] = (unsigned short)b; p[plane + rowpitch] = (unsigned short)(b >> 16);
}
__device__ __forceinline__ void store_split2_s(unsigned short *p, int plane, float x, float sc) {
    unsigned a, b;
    split2_pair(x, 0.f, sc, a, b);
    p[0] = (unsigned short)a; p[plane] = (unsigned short)b;
}
// Quad q's lanes receive element q of the f32x4 their column's lane in quad 0 holds (three swaps of register halves / quarters): the
// four live rows of a 16 x 16 MFMA result, one per lane.
// (inline assembly: through __builtin_amdgcn_permlane16_swap / _permlane32_swap this compiler fed the first swap the SAME register
// twice when only one half of the builtin's result pair was used, and declared the other three accumulator registers dead -- quads 1-3
// then received element 0; tools/probes/permlane_swap.hip shows the instructions themselves do what the ISA says.  The s_nop in front
// covers an MFMA result read by a vector instruction the compiler's hazard recogniser does not see: 8 passes + 2.)
__device__ __forceinline__ float spread_rows(const f32x4 &a) {
    float x0 = a[0], x1 = a[1], x2 = a[2], x3 = a[3];
    asm volatile("s_nop 15\n\t"
                 "v_permlane16_swap_b32 %0, %1\n\t"              // x0 = [a0.q0 | a1.q0 | a0.q2 | a1.q2]
                 "v_permlane16_swap_b32 %2, %3\n\t"              // x2 = [a2.q0 | a3.q0 | a2.q2 | a3.q2]
                 "s_nop 1\n\t"
                 "v_permlane32_swap_b32 %0, %2\n\t"              // x0 = [a0.q0 | a1.q0 | a2.q0 | a3.q0]
                 "s_nop 1"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    return x0;
}

// RW 8: quads 0 and 1 hold the eight live rows; their elements 2 and 3 go to quads 2 and 3 (one swap of register halves each)
__device__ __forceinline__ void spread_pairs(const f32x4 &a, float &o0, float &o1) {
    float x0 = a[0], x1 = a[1], x2 = a[2], x3 = a[3];
    asm volatile("s_nop 15\n\t"
                 "v_permlane32_swap_b32 %0, %2\n\t"              // x0 = [a0.q0 | a0.q1 | a2.q0 | a2.q1]
                 "v_permlane32_swap_b32 %1, %3\n\t"              // x1 = [a1.q0 | a1.q1 | a3.q0 | a3.q1]
                 "s_nop 1"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));
    o0 = x0;
    o1 = x1;
}
// E = RW / 4 elements per lane: the tile row of a lane's element i, the tile row a lane reads its A operand from (rows past the live
// ones repeat them), a 16 x 16 result's values for the lane's elements
template <int E> __device__ __forceinline__ int gru_lrow(int quad, int i) {
    return E == 4 ? 4 * quad + i : E == 2 ? 4 * (quad & 1) + 2 * (quad >> 1) + i : quad;
}
template <int E> __device__ __forceinline__ int gru_arow(int col) { return E == 4 ? col : E == 2 ? (col & 7) : (col & 3); }
template <int E> __device__ __forceinline__ void gru_elems(const f32x4 &acc, float (&out)[E]) {
    if constexpr (E == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = acc[i];
    } else if constexpr (E == 2) {
        spread_pairs(acc, out[0], out[1]);
    } else {
        out[0] = spread_rows(acc);
    }
}

// RW: batch rows per workgroup, 16, 8 or 4.  A recurrence is a chain of T dependent steps whose length is the instruction stream of one
// wave between two barriers, and most of that stream is per (row, hidden unit) ELEMENT work: projections in,
// gates, the state's split and its LDS writes, h and the saved gates out -- four elements per lane when a workgroup owns 16 rows.
// With 4 rows per workgroup the 16 x 16 MFMA tile is three quarters empty (the matrix pipe was idle anyway), the four live rows of
// a result go out to the four quads (spread_rows) and every lane does ONE element per step; four times the workgroups, on a chip
// that the recurrences of a 256-measure batch fill to an eighth.  The host picks the smallest of 4, 8, 16 whose workgroups are all on
// the chip at once (8: two elements per lane, the live rows in quads 0 and 1).
template <int H, int RW>
__global__ __launch_bounds__(H * 4) void gru_seq_fwd_h2_kernel(GruSeqBatch batch, int T, int R) {
    static_assert(RW == 16 || RW == 8 || RW == 4, "16, 8 or 4 rows: four, two or one per lane");
    constexpr int E = RW / 4;              // elements (rows) per lane
    constexpr int KS = H / 32;             // MFMA k-steps of 32
    constexpr int HP = H + 8;              // LDS row pitch in bf16 elements (16 bytes of padding)
    constexpr int PLANE = 16 * HP;
    __shared__ __attribute__((aligned(16))) unsigned short hbuf[2][2 * PLANE];
    __shared__ float h0max[H / 16];
    const GruSeq &s = batch.seq[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, quad = lane >> 4;
    const int unit = 16 * w + col;
    const int row0 = blockIdx.x * RW;

    // W_hh's slice of this wave as two fp16 terms at the wave's own scale (round 5: the slice's largest magnitude just below 2^15;
    // through round 4 a fixed 2^8, which overflowed fp16 for |w| >= 255)
    f16x8g wh[3][KS], wl[3][KS];
    float w_inv;
    {
        float x[3][KS][8], m = 0.f;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const float *src = s.w_hh + (int64_t)(g * H + unit) * H + 32 * ks + 8 * quad;
                const f32x4 v0 = *reinterpret_cast<const f32x4 *>(src), v1 = *reinterpret_cast<const f32x4 *>(src + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    x[g][ks][j] = v0[j]; x[g][ks][4 + j] = v1[j];
                    m = fmaxf(m, fmaxf(fabsf(v0[j]), fabsf(v1[j])));
                }
            }
        m = row16_max(m);
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        const GruPow2 sw = gru_pow2(m);
        w_inv = sw.inv;
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) split2_x8(x[g][ks], sw.s, wh[g][ks], wl[g][ks]);
    }
    const float bh_r = s.b_hh[unit], bh_z = s.b_hh[H + unit], bh_n = s.b_hh[2 * H + unit];

    int rows[E];
    bool live[E];
    float h[E];
    auto lrow = [&](int i) { return gru_lrow<E>(quad, i); };                  // the tile row of this lane's element i
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int r = row0 + lrow(i);
        live[i] = r < R;
        rows[i] = live[i] ? r : R - 1;
        h[i] = (s.h0 != nullptr && live[i]) ? s.h0[(int64_t)rows[i] * s.h0_stride + unit] : 0.f;
    }
    // The state's scale: every h_t is a convex combination of a tanh output and h_(t-1), so |h_t| <= max(1, max |h0|) for the whole
    // sequence -- the workgroup's rows' largest |h0| (or 1) goes just below 2^15 (through round 4 a fixed 2^4: fp16 overflow for an
    // initial state beyond 4094, and the decoder's comes out of a SELU layer)
    float h_s, unscale;
    {
        float m = 1.f;
#pragma unroll
        for (int i = 0; i < E; ++i) m = fmaxf(m, fabsf(h[i]));
        m = row16_max(m);
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        if (lane == 0) h0max[w] = m;
        lds_barrier();
#pragma unroll
        for (int q = 0; q < H / 16; ++q) m = fmaxf(m, h0max[q]);
        const GruPow2 sh = gru_pow2(m);
        h_s = sh.s;
        unscale = sh.inv * w_inv;
    }
#pragma unroll
    for (int i = 0; i < E; ++i) store_split2_s(&hbuf[0][lrow(i) * HP + unit], PLANE, h[i], h_s);
    // per-step memory operations as raw buffer operations (gru_rsrc): a scalar step offset + one per-lane offset per array and row
    // (the running 64-bit pointers this kernel had cost 24 registers and a 64-bit add each per step; the first and last step's
    // missing operations were branches).  A dead row's stores go beyond the range.
    const __amdgpu_buffer_rsrc_t rs_gi = gru_rsrc(s.gi), rs_h = gru_rsrc(s.h_all), rs_sv = gru_rsrc(s.saved), rs_none = gru_rsrc(nullptr);
    const int reverse = s.reverse, unit4 = 4 * unit;
    const int gi_tp = 4 * (int)s.gi_tstride, h_tp = 4 * R * (int)s.h_stride;
    // dropout on the output (gru_mask.h): the keep byte of the CURRENT step is requested with the step's other traffic, the masked
    // copy of a step's h leaves one step later beside h.  No mask: an empty range (loads return 0, the stores are dropped).
    const __amdgpu_buffer_rsrc_t rs_mk = gru_rsrc(s.mask), rs_hm = gru_rsrc(s.mask != nullptr ? s.h_masked : nullptr);
    const int mk_tp = s.mk_tstride, hm_tp = 4 * R * s.hm_stride;
    const float keep_scale = s.keep;
    int mk_o[E], hm_o[E];
    unsigned mk_cur[E];
    float keep_hm[E];
    int gi_o[E], h_o[E], sv_o[E];
    float gi_next[E][3];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        gi_o[i] = gru_off(rows[i], 4 * (int)s.gi_rstride, unit4);
        h_o[i] = live[i] ? gru_off(rows[i], 4 * (int)s.h_stride, unit4) : GRU_DEAD;
        sv_o[i] = live[i] ? gru_off(rows[i], 16 * H, 4 * unit4) : GRU_DEAD;
        mk_o[i] = s.mk_group > 0 ? (rows[i] / s.mk_group) * s.mk_gstride + (rows[i] % s.mk_group) * s.mk_rstride + unit : rows[i] * s.mk_rstride + unit;
        hm_o[i] = live[i] ? gru_off(rows[i], 4 * s.hm_stride, unit4) : GRU_DEAD;
        mk_cur[i] = 0;
        keep_hm[i] = 0.f;
        const int so = (reverse ? T - 1 : 0) * gi_tp;
        gi_next[i][0] = gru_ld(rs_gi, gi_o[i], so); gi_next[i][1] = gru_ld(rs_gi, gi_o[i] + 4 * H, so); gi_next[i][2] = gru_ld(rs_gi, gi_o[i] + 8 * H, so);
    }
    lds_barrier();
    f32x4 keep_sv[E];                        // results of the previous step, stored after the barrier
    float keep_h[E];
    int keep_t = -1;
#ifdef ARVAE_GRU_STAMPS
    unsigned long long ph[4] = {0, 0, 0, 0}, tc = __builtin_readcyclecounter();
#define GSTAMP(k) { const unsigned long long now = __builtin_readcyclecounter(); ph[k] += now - tc; tc = now; }
#else
#define GSTAMP(k)
#endif

    for (int step = 0; step < T; ++step) {
        const int t = s.reverse ? T - 1 - step : step;
        const int cur = step & 1;
        float gi[E][3];
#pragma unroll
        for (int i = 0; i < E; ++i)
#pragma unroll
            for (int g = 0; g < 3; ++g) gi[i][g] = gi_next[i][g];
        GSTAMP(0);
        f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const unsigned short *hb = &hbuf[cur][gru_arow<E>(col) * HP + 8 * quad];
        // Row i's share of the step's memory traffic (next step's three input projections in, the previous step's h and saved
        // gates out) is issued BEHIND the MFMAs of k-step i, with a scheduling barrier pinning it there: as one block in front
        // of the MFMAs it was 1200 of the step's 6000 cycles (tools/stamp_gru.py), all of it issue time of an in-order wave
        // while the matrix pipe sat idle.
        // next step's projections (the last step reads its own again); the previous step's results (none in step 0: the empty range)
        const int tn = step + 1 < T ? (reverse ? t - 1 : t + 1) : t;
        const int kt = keep_t >= 0 ? keep_t : 0;
        const int so_gi = tn * gi_tp, so_h = kt * h_tp, so_sv = kt * R * (16 * H);
        const __amdgpu_buffer_rsrc_t rs_hs = keep_t >= 0 ? rs_h : rs_none, rs_svs = keep_t >= 0 ? rs_sv : rs_none;
        const __amdgpu_buffer_rsrc_t rs_hms = keep_t >= 0 ? rs_hm : rs_none;
        const int so_mk = t * mk_tp, so_hm = kt * hm_tp;
        auto row_traffic = [&](int i) __attribute__((always_inline)) {
            gi_next[i][0] = gru_ld(rs_gi, gi_o[i], so_gi); gi_next[i][1] = gru_ld(rs_gi, gi_o[i] + 4 * H, so_gi);
            gi_next[i][2] = gru_ld(rs_gi, gi_o[i] + 8 * H, so_gi);
            gru_st(keep_h[i], rs_hs, h_o[i], so_h);
            gru_st(keep_hm[i], rs_hms, hm_o[i], so_hm);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4g, keep_sv[i]), rs_svs, sv_o[i], so_sv, 0);
            mk_cur[i] = __builtin_amdgcn_raw_buffer_load_b8(rs_mk, mk_o[i], so_mk, 0);
        };
        static_assert(KS <= 4, "one row's traffic per k-step; rows left over go last");
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f16x8g ah = lds_h8(hb + 32 * ks), al = lds_h8(hb + PLANE + 32 * ks);
            GRU_MFMA3X3(acc[0], acc[1], acc[2], ah, al, wh[0][ks], wl[0][ks], wh[1][ks], wl[1][ks], wh[2][ks], wl[2][ks]);
            __builtin_amdgcn_sched_barrier(0);
            if (ks < E) row_traffic(ks);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = KS; i < E; ++i) row_traffic(i);
#ifdef ARVAE_GRU_STAMPS
        { float dep = acc[0][0] + acc[1][0] + acc[2][3]; asm volatile("" :: "v"(dep)); __builtin_amdgcn_s_waitcnt(0); }
#endif
        GSTAMP(1);
        float av[3][E];                      // the gates' products of this lane's elements
#pragma unroll
        for (int g = 0; g < 3; ++g) gru_elems<E>(acc[g], av[g]);
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const float r = fast_sigmoid(gi[i][0] + av[0][i] * unscale + bh_r);
            const float z = fast_sigmoid(gi[i][1] + av[1][i] * unscale + bh_z);
            const float ghn = av[2][i] * unscale + bh_n;
            const float n = fast_tanh(gi[i][2] + r * ghn);
            const float hn = (1.f - z) * n + z * h[i];
            h[i] = hn;
            keep_h[i] = hn;
            keep_hm[i] = mk_cur[i] != 0 ? keep_scale * hn : 0.f;
            keep_sv[i] = f32x4{r, z, n, ghn};
        }
        if constexpr (E >= 2) {
#pragma unroll
            for (int i = 0; i < E; i += 2) store_split2_pair_s(&hbuf[cur ^ 1][lrow(i) * HP + unit], HP, PLANE, h[i], h[i + 1], h_s);
        } else {
            store_split2_s(&hbuf[cur ^ 1][quad * HP + unit], PLANE, h[0], h_s);
        }
        keep_t = t;
#ifdef ARVAE_GRU_STAMPS
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the LDS writes are done
#endif
        GSTAMP(2);
        lds_barrier();
        GSTAMP(3);
    }
#ifdef ARVAE_GRU_STAMPS
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) {
        for (int k = 0; k < 4; ++k) g_gru_stamps[k] = ph[k];
        g_gru_stamps[4] = T;
    }
#endif
#pragma unroll
    for (int i = 0; i < E; ++i)
        if (live[i]) {
            gru_st(keep_h[i], rs_h, h_o[i], keep_t * h_tp);
            gru_st(keep_hm[i], rs_hm, hm_o[i], keep_t * hm_tp);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4g, keep_sv[i]), rs_sv, sv_o[i], keep_t * R * (16 * H), 0);
            if (s.h_fin != nullptr) s.h_fin[(int64_t)rows[i] * s.h_fin_stride + unit] = keep_h[i];
        }
}

// (RW: batch rows per workgroup, 16 or 4 -- see gru_seq_fwd_h2_kernel)
template <int H, int RW>
__global__ __launch_bounds__(H * 4) void gru_seq_bwd_x3_kernel(GruSeqBatch batch, int T, int R) {
    static_assert(RW == 16 || RW == 8 || RW == 4, "16, 8 or 4 rows: four, two or one per lane");
    constexpr int E = RW / 4;              // elements (rows) per lane
    constexpr int KS = 3 * H / 32;
    constexpr int DP = 3 * H + 8;           // LDS row pitch in bf16 elements
    constexpr int PLANE = 16 * DP;
    __shared__ __attribute__((aligned(16))) unsigned short dbuf[2][3 * PLANE];
    const GruSeq &s = batch.seq[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, quad = lane >> 4;
    const int unit = 16 * w + col;
    const int row0 = blockIdx.x * RW;

    // B[k = c][n = unit] = W_hh[c][unit], c = 32 ks + 8 quad + j
    bf16x8g wh[KS], wm[KS], wl[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = s.w_hh[(int64_t)(32 * ks + 8 * quad + j) * H + unit];
        split3_x8(x, wh[ks], wm[ks], wl[ks]);
    }

    int rows[E];
    bool live[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int r = row0 + gru_lrow<E>(quad, i);
        live[i] = r < R;
        rows[i] = live[i] ? r : R - 1;
    }
    float carry[E];
#pragma unroll
    for (int i = 0; i < E; ++i)
        carry[i] = (s.dh_last != nullptr && live[i]) ? s.dh_last[(int64_t)rows[i] * s.dh_last_stride + unit] : 0.f;

    // the step's arrays as buffer resources (gru_rsrc), their row pitches in bytes
    const __amdgpu_buffer_rsrc_t rs_dh = gru_rsrc(s.dh_all), rs_sv = gru_rsrc(s.saved), rs_hall = gru_rsrc(s.h_all), rs_h0 = gru_rsrc(s.h0);
    const __amdgpu_buffer_rsrc_t rs_dgi = gru_rsrc(s.dgi), rs_dgh = gru_rsrc(s.dgh), rs_hpo = gru_rsrc(s.h_prev_out);
    const int reverse = s.reverse, unit4 = 4 * unit;
    const int dh_p = 4 * (int)s.dh_stride, h_p = 4 * (int)s.h_stride, h0_p = 4 * (int)s.h0_stride, dgi_p = 4 * (int)s.dgi_rstride;
    float nx[E][6];                          // dh, r, z, n, gh_n, h_prev of the next step
    auto fetch = [&](int step) {
        const int t = reverse ? step : T - 1 - step;
        const bool has_prev = step + 1 < T;
        const int tp = reverse ? t + 1 : t - 1;
        const __amdgpu_buffer_rsrc_t rs_hp = has_prev ? rs_hall : rs_h0;
        const int hp_p = has_prev ? h_p : h0_p, hp_s = has_prev ? tp * R * h_p : 0;
#pragma unroll
        for (int i = 0; i < E; ++i) {
            nx[i][0] = gru_ld(rs_dh, gru_off(rows[i], dh_p, unit4), t * R * dh_p);
            const f32x4 sv = gru_ld4(rs_sv, gru_off(rows[i], 16 * H, 4 * unit4), t * R * (16 * H));
            nx[i][1] = sv[0]; nx[i][2] = sv[1]; nx[i][3] = sv[2]; nx[i][4] = sv[3];
            nx[i][5] = gru_ld(rs_hp, gru_off(rows[i], hp_p, unit4), hp_s);
        }
    };
    fetch(0);
#ifdef ARVAE_GRU_STAMPS
    unsigned long long ph[4] = {0, 0, 0, 0}, tc = __builtin_readcyclecounter();
#endif

    for (int step = 0; step < T; ++step) {
        const int t = reverse ? step : T - 1 - step;
        const int cur = step & 1;
        float gz[E], o_gi[E][3], o_hn[E], o_hp[E];
#ifdef ARVAE_GRU_STAMPS
        { float dep = nx[0][0] + nx[E - 1][5] + nx[E / 2][3]; asm volatile("" :: "v"(dep)); __builtin_amdgcn_s_waitcnt(0); }
#endif
        GSTAMP(0);
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const float g = live[i] ? nx[i][0] + carry[i] : 0.f;
            const float r = nx[i][1], z = nx[i][2], n = nx[i][3], ghn = nx[i][4], hp = nx[i][5];
            const float dpn = g * (1.f - z) * (1.f - n * n);
            const float dpz = g * (hp - n) * z * (1.f - z);
            const float dpr = dpn * ghn * r * (1.f - r);
            const float dhn = dpn * r;
            gz[i] = g * z;
            o_gi[i][0] = dpr; o_gi[i][1] = dpz; o_gi[i][2] = dpn; o_hn[i] = dhn; o_hp[i] = hp;
        }
        if constexpr (E >= 2) {
#pragma unroll
            for (int i = 0; i < E; i += 2) {
                unsigned short *d = &dbuf[cur][gru_lrow<E>(quad, i) * DP + unit];
                store_split3_pair(d, DP, PLANE, o_gi[i][0], o_gi[i + 1][0]);
                store_split3_pair(d + H, DP, PLANE, o_gi[i][1], o_gi[i + 1][1]);
                store_split3_pair(d + 2 * H, DP, PLANE, o_hn[i], o_hn[i + 1]);
            }
        } else {
            unsigned short *d = &dbuf[cur][quad * DP + unit];
            store_split3(d, PLANE, o_gi[0][0]);
            store_split3(d + H, PLANE, o_gi[0][1]);
            store_split3(d + 2 * H, PLANE, o_hn[0]);
        }
#ifdef ARVAE_GRU_STAMPS
        __builtin_amdgcn_s_waitcnt(0xc07f);
#endif
        GSTAMP(1);
        if (step + 1 < T) fetch(step + 1);
        lds_barrier();
        GSTAMP(2);
        f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const unsigned short *db = &dbuf[cur][gru_arow<E>(col) * DP + 8 * quad];
        static_assert(KS % 3 == 0 && KS / 3 <= 4, "three k-steps at a time, one per accumulator; one row's stores behind each group");
        // row i's gradients of this step leave BEHIND the MFMAs of k-step group i (pinned: see gru_seq_fwd_x3_kernel)
        auto row_stores = [&](int i) __attribute__((always_inline)) {
            if (live[i]) {
                const int og = gru_off(rows[i], dgi_p, unit4), sg = t * R * dgi_p;
                const int o = gru_off(rows[i], 12 * H, unit4), so = t * R * (12 * H);
                gru_st(o_gi[i][0], rs_dgi, og, sg); gru_st(o_gi[i][1], rs_dgi, og + 4 * H, sg); gru_st(o_gi[i][2], rs_dgi, og + 8 * H, sg);
                gru_st(o_gi[i][0], rs_dgh, o, so); gru_st(o_gi[i][1], rs_dgh, o + 4 * H, so); gru_st(o_hn[i], rs_dgh, o + 8 * H, so);
                gru_st(o_hp[i], rs_hpo, gru_off(rows[i], 4 * H, unit4), t * R * (4 * H));
            }
        };
#pragma unroll
        for (int ks = 0; ks < KS; ks += 3) {
            const bf16x8g ah0 = lds_x8(db + 32 * ks), am0 = lds_x8(db + PLANE + 32 * ks), al0 = lds_x8(db + 2 * PLANE + 32 * ks);
            const bf16x8g ah1 = lds_x8(db + 32 * (ks + 1)), am1 = lds_x8(db + PLANE + 32 * (ks + 1)), al1 = lds_x8(db + 2 * PLANE + 32 * (ks + 1));
            const bf16x8g ah2 = lds_x8(db + 32 * (ks + 2)), am2 = lds_x8(db + PLANE + 32 * (ks + 2)), al2 = lds_x8(db + 2 * PLANE + 32 * (ks + 2));
            GRU_MFMA6X3(acc[0], acc[1], acc[2], ah0, am0, al0, ah1, am1, al1, ah2, am2, al2, wh[ks], wm[ks], wl[ks], wh[ks + 1], wm[ks + 1],
                        wl[ks + 1], wh[ks + 2], wm[ks + 2], wl[ks + 2]);
            __builtin_amdgcn_sched_barrier(0);
            if (ks / 3 < E) row_stores(ks / 3);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = KS / 3; i < E; ++i) row_stores(i);
        {
            const f32x4 sum = {acc[0][0] + acc[1][0] + acc[2][0], acc[0][1] + acc[1][1] + acc[2][1], acc[0][2] + acc[1][2] + acc[2][2],
                               acc[0][3] + acc[1][3] + acc[2][3]};
            float cs[E];
            gru_elems<E>(sum, cs);
#pragma unroll
            for (int i = 0; i < E; ++i) carry[i] = gz[i] + cs[i];
        }
#ifdef ARVAE_GRU_STAMPS
        { float dep = carry[0] + carry[E - 1]; asm volatile("" :: "v"(dep)); }
#endif
        GSTAMP(3);
    }
#ifdef ARVAE_GRU_STAMPS
#ifndef GRU_STAMP_WAVE
#define GRU_STAMP_WAVE 0
#endif
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 64 * GRU_STAMP_WAVE) {
        for (int q = 0; q < 4; ++q) g_gru_stamps[q] = ph[q];
        g_gru_stamps[4] = T;
    }
#endif
    if (s.dh0 != nullptr)
#pragma unroll
        for (int i = 0; i < E; ++i)
            if (live[i]) s.dh0[(int64_t)rows[i] * s.dh0_stride + unit] = carry[i];
}

// The backward recurrence on scaled two-term fp16 (round 5).  Its operand -- a step's (dpr, dpz, dhn) per batch row -- is a gradient:
// no magnitude known beforehand, and it moves over the time steps (what kept this kernel on the three-term bf16 split in round 4).
// With one element per lane (RW 4) the recurrence became MFMA-bound -- 72 dependent-free 16x16x32 MFMAs per wave and step at 32
// cycles, two waves per SIMD: 4600 of a step's cycles -- so half the products are worth a second barrier: every batch ROW gets its own
// power-of-two scale each step (a row's scale factors out of its dot products), the largest magnitude of the row's 3 H values brought
// to [2^14, 2^15): a 16-lane butterfly per wave, one LDS slot per (row, wave), a barrier, NW slots read back.  Nothing can overflow
// (the scaled maximum is below 2^15 by construction), a row whose gradient is 1e-9 keeps the same 22 bits as one at 1e+3, and W_hh^T
// gets a per-wave scale from its own slice's maximum in the prologue (a column's scale factors out as well) instead of the fixed 2^8.
// three accumulators, one k-step each, product-major (l h', h l', h h'): consecutive MFMAs never hit the same accumulator
#define GRU_MFMA3K3(A0, A1, A2, H0, L0, H1, L1, H2, L2, WH0, WL0, WH1, WL1, WH2, WL2)                                              \
    A0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(L0, WH0, A0, 0, 0, 0); A1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(L1, WH1, A1, 0, 0, 0); \
    A2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(L2, WH2, A2, 0, 0, 0);                                                            \
    A0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(H0, WL0, A0, 0, 0, 0); A1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(H1, WL1, A1, 0, 0, 0); \
    A2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(H2, WL2, A2, 0, 0, 0);                                                            \
    A0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(H0, WH0, A0, 0, 0, 0); A1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(H1, WH1, A1, 0, 0, 0); \
    A2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(H2, WH2, A2, 0, 0, 0)

template <int H, int RW>
__global__ __launch_bounds__(H * 4) void gru_seq_bwd_h2_kernel(GruSeqBatch batch, int T, int R) {
    static_assert(RW == 16 || RW == 8 || RW == 4, "16, 8 or 4 rows: four, two or one per lane");
    constexpr int E = RW / 4;              // elements (rows) per lane
    constexpr int NW = H / 16;             // waves
    constexpr int KS = 3 * H / 32;
    constexpr int DP = 3 * H + 8;           // LDS row pitch in fp16 elements
    constexpr int PLANE = 16 * DP;
    constexpr int MW = NW < 4 ? 4 : NW;     // slots per row of the maxima (16-byte reads)
    __shared__ __attribute__((aligned(16))) unsigned short dbuf[2 * PLANE];
    __shared__ __attribute__((aligned(16))) float rmax[16][MW];
    const GruSeq &s = batch.seq[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, quad = lane >> 4;
    const int unit = 16 * w + col;
    const int row0 = blockIdx.x * RW;

    // B[k = c][n = unit] = W_hh[c][unit], c = 32 ks + 8 quad + j: two fp16 terms at this wave's own scale
    f16x8g wh[KS], wl[KS];
    float w_inv;
    {
        float x[KS][8], m = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                x[ks][j] = s.w_hh[(int64_t)(32 * ks + 8 * quad + j) * H + unit];
                m = fmaxf(m, fabsf(x[ks][j]));
            }
        m = row16_max(m);
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        const GruPow2 sw = gru_pow2(m);
        w_inv = sw.inv;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) split2_x8(x[ks], sw.s, wh[ks], wl[ks]);
    }
    if (MW > NW && threadIdx.x < 16 * (MW - NW)) rmax[threadIdx.x / (MW - NW)][NW + threadIdx.x % (MW - NW)] = 0.f;   // (slots no wave writes)

    auto lrow = [&](int i) { return gru_lrow<E>(quad, i); };                  // the tile row of this lane's element i
    int rows[E];
    bool live[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int r = row0 + lrow(i);
        live[i] = r < R;
        rows[i] = live[i] ? r : R - 1;
    }
    float carry[E];
#pragma unroll
    for (int i = 0; i < E; ++i)
        carry[i] = (s.dh_last != nullptr && live[i]) ? s.dh_last[(int64_t)rows[i] * s.dh_last_stride + unit] : 0.f;

    // the step's arrays as buffer resources (gru_rsrc), their row pitches in bytes
    const __amdgpu_buffer_rsrc_t rs_dh = gru_rsrc(s.dh_all), rs_sv = gru_rsrc(s.saved), rs_hall = gru_rsrc(s.h_all), rs_h0 = gru_rsrc(s.h0);
    const __amdgpu_buffer_rsrc_t rs_dgi = gru_rsrc(s.dgi), rs_dgh = gru_rsrc(s.dgh), rs_hpo = gru_rsrc(s.h_prev_out);
    const int reverse = s.reverse, unit4 = 4 * unit;
    const int dh_p = 4 * (int)s.dh_stride, h_p = 4 * (int)s.h_stride, h0_p = 4 * (int)s.h0_stride, dgi_p = 4 * (int)s.dgi_rstride;
    float nx[E][6];                          // dh, r, z, n, gh_n, h_prev of the next step
    // dropout on the sequence's output (gru_mask.h): dh_all is then the gradient w.r.t. keep * mask * h
    const bool masked = s.mask != nullptr;
    const __amdgpu_buffer_rsrc_t rs_mk = gru_rsrc(s.mask);
    const float keep_scale = s.keep;
    int mk_o[E];
    unsigned mk_nx[E];
#pragma unroll
    for (int i = 0; i < E; ++i)
        mk_o[i] = s.mk_group > 0 ? (rows[i] / s.mk_group) * s.mk_gstride + (rows[i] % s.mk_group) * s.mk_rstride + unit : rows[i] * s.mk_rstride + unit;
    auto fetch = [&](int step) {
        const int t = reverse ? step : T - 1 - step;
        const bool has_prev = step + 1 < T;
        const int tp = reverse ? t + 1 : t - 1;
        const __amdgpu_buffer_rsrc_t rs_hp = has_prev ? rs_hall : rs_h0;
        const int hp_p = has_prev ? h_p : h0_p, hp_s = has_prev ? tp * R * h_p : 0;
#pragma unroll
        for (int i = 0; i < E; ++i) {
            nx[i][0] = gru_ld(rs_dh, gru_off(rows[i], dh_p, unit4), t * R * dh_p);
            mk_nx[i] = __builtin_amdgcn_raw_buffer_load_b8(rs_mk, mk_o[i], t * s.mk_tstride, 0);
            const f32x4 sv = gru_ld4(rs_sv, gru_off(rows[i], 16 * H, 4 * unit4), t * R * (16 * H));
            nx[i][1] = sv[0]; nx[i][2] = sv[1]; nx[i][3] = sv[2]; nx[i][4] = sv[3];
            nx[i][5] = gru_ld(rs_hp, gru_off(rows[i], hp_p, unit4), hp_s);
        }
    };
    fetch(0);

    for (int step = 0; step < T; ++step) {
        const int t = reverse ? step : T - 1 - step;
        float gz[E], o_gi[E][3], o_hn[E], o_hp[E];
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const float dh = masked ? (mk_nx[i] != 0 ? keep_scale * nx[i][0] : 0.f) : nx[i][0];
            const float g = live[i] ? dh + carry[i] : 0.f;
            const float r = nx[i][1], z = nx[i][2], n = nx[i][3], ghn = nx[i][4], hp = nx[i][5];
            const float dpn = g * (1.f - z) * (1.f - n * n);
            const float dpz = g * (hp - n) * z * (1.f - z);
            const float dpr = dpn * ghn * r * (1.f - r);
            const float dhn = dpn * r;
            gz[i] = g * z;
            o_gi[i][0] = dpr; o_gi[i][1] = dpz; o_gi[i][2] = dpn; o_hn[i] = dhn; o_hp[i] = hp;
            const float m = row16_max(fmaxf(fmaxf(fabsf(dpr), fabsf(dpz)), fabsf(dhn)));
            if (col == 0) rmax[lrow(i)][w] = m;
        }
        if (step + 1 < T) fetch(step + 1);
        lds_barrier();                       // the rows' maxima are in LDS; every read of the previous step's operand image is done
        float unscale[E];
#pragma unroll
        for (int i = 0; i < E; ++i) {
            float m = 0.f;
#pragma unroll
            for (int q = 0; q < MW; q += 4) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(&rmax[lrow(i)][q]);
                m = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
            }
            const GruPow2 sc = gru_pow2(m);
            unscale[i] = sc.inv * w_inv;
            unsigned short *d = &dbuf[lrow(i) * DP + unit];
            store_split2_s(d, PLANE, o_gi[i][0], sc.s);
            store_split2_s(d + H, PLANE, o_gi[i][1], sc.s);
            store_split2_s(d + 2 * H, PLANE, o_hn[i], sc.s);
        }
        lds_barrier();                       // the operand image is written; the maxima have been read
        f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const unsigned short *db = &dbuf[gru_arow<E>(col) * DP + 8 * quad];
        static_assert(KS % 3 == 0 && KS / 3 <= 4, "three k-steps at a time, one per accumulator; one row's stores behind each group");
        // row i's gradients of this step leave BEHIND the MFMAs of k-step group i (pinned: see gru_seq_fwd_x3_kernel)
        auto row_stores = [&](int i) __attribute__((always_inline)) {
            if (live[i]) {
                const int og = gru_off(rows[i], dgi_p, unit4), sg = t * R * dgi_p;
                const int o = gru_off(rows[i], 12 * H, unit4), so = t * R * (12 * H);
                gru_st(o_gi[i][0], rs_dgi, og, sg); gru_st(o_gi[i][1], rs_dgi, og + 4 * H, sg); gru_st(o_gi[i][2], rs_dgi, og + 8 * H, sg);
                gru_st(o_gi[i][0], rs_dgh, o, so); gru_st(o_gi[i][1], rs_dgh, o + 4 * H, so); gru_st(o_hn[i], rs_dgh, o + 8 * H, so);
                gru_st(o_hp[i], rs_hpo, gru_off(rows[i], 4 * H, unit4), t * R * (4 * H));
            }
        };
#pragma unroll
        for (int ks = 0; ks < KS; ks += 3) {
            const f16x8g ah0 = lds_h8(db + 32 * ks), al0 = lds_h8(db + PLANE + 32 * ks);
            const f16x8g ah1 = lds_h8(db + 32 * (ks + 1)), al1 = lds_h8(db + PLANE + 32 * (ks + 1));
            const f16x8g ah2 = lds_h8(db + 32 * (ks + 2)), al2 = lds_h8(db + PLANE + 32 * (ks + 2));
            GRU_MFMA3K3(acc[0], acc[1], acc[2], ah0, al0, ah1, al1, ah2, al2, wh[ks], wl[ks], wh[ks + 1], wl[ks + 1], wh[ks + 2], wl[ks + 2]);
            __builtin_amdgcn_sched_barrier(0);
            if (ks / 3 < E) row_stores(ks / 3);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = KS / 3; i < E; ++i) row_stores(i);
        {
            const f32x4 sum = {acc[0][0] + acc[1][0] + acc[2][0], acc[0][1] + acc[1][1] + acc[2][1], acc[0][2] + acc[1][2] + acc[2][2],
                               acc[0][3] + acc[1][3] + acc[2][3]};
            float cs[E];
            gru_elems<E>(sum, cs);
#pragma unroll
            for (int i = 0; i < E; ++i) carry[i] = gz[i] + cs[i] * unscale[i];
        }
    }
    if (s.dh0 != nullptr)
#pragma unroll
        for (int i = 0; i < E; ++i)
            if (live[i]) s.dh0[(int64_t)rows[i] * s.dh0_stride + unit] = carry[i];
}

static int fill_batch(GruSeqBatch *b, const arvae_gru_seq_t *seqs, int nseq, int hidden, const GruSeqMask *masks = nullptr) {
    for (int i = 0; i < nseq; ++i) {
        const arvae_gru_seq_t &q = seqs[i];
        GruSeq &s = b->seq[i];
        if (masks != nullptr && masks[i].mask != nullptr) {
            const GruSeqMask &k = masks[i];
            s.mask = k.mask; s.keep = k.keep; s.h_masked = k.h_masked; s.hm_stride = (int)k.hm_stride;
            s.mk_tstride = (int)k.tstride; s.mk_rstride = (int)k.rstride; s.mk_gstride = (int)k.gstride; s.mk_group = k.group;
        }
        s.gi = q.gi; s.gi_tstride = q.gi_tstride; s.w_hh = q.w_hh; s.b_hh = q.b_hh; s.h0 = q.h0;
        s.h_all = q.h_all; s.h_stride = q.h_stride; s.saved = q.saved; s.reverse = q.reverse;
        s.dh_all = q.dh_all; s.dh_stride = q.dh_stride; s.dgi = q.dgi; s.dgh = q.dgh; s.dh0 = q.dh0;
        s.dh_last = q.dh_last; s.dh_last_stride = q.dh_last_stride; s.h_prev_out = q.h_prev_out;
        s.gi_rstride = q.gi_rstride != 0 ? q.gi_rstride : 3 * hidden;
        s.dgi_rstride = q.dgi_rstride != 0 ? q.dgi_rstride : 3 * hidden;
        s.h_fin = q.h_fin; s.h_fin_stride = q.h_fin_stride;
        s.h0_stride = q.h0_stride != 0 ? q.h0_stride : hidden;
        s.dh0_stride = q.dh0_stride != 0 ? q.dh0_stride : hidden;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Free-running tick decoder: the argmax-feedback pass of the hierarchical decoder (measurevae/decoder.py:459-525 with
// teacher forcing off: the embedding of the previous tick's top-1 note is the next input) in ONE launch that returns
// only the tokens.  The differentiable graph is then evaluated on those tokens by the whole-sequence kernels above
// (argmax is not differentiated), so nothing else has to be saved here.
//
// Every batch row is an independent 24-tick recurrence: a workgroup owns 16 rows, H/16 waves, wave w owns hidden units
// [16w, 16w+16).  Per tick:   gi0 = gib[beat][row] + ptab[previous token]      (both precomputed by dense launches)
//   layer 0: gh0 = W_hh0 h0 (W_hh0 slice register-resident) -> gates -> h0' ; mid = h0' * keep-mask * scale
//   layer 1: r,z: W_ih1 mid + W_hh1 h1 in one accumulator each; n: the two products apart (weights streamed from L2,
//            two k-groups ahead) -> gates -> h1'
//   logits = relu(W_out h1' + b_out) on waves 0..ceil(V/16)-1, row argmax (lowest index on ties) through lane shuffles
//            and LDS -> token, fed back.
struct TickFreeRun {
    const float *w_hh0, *b_hh0, *w_ih1, *b_ih1, *w_hh1, *b_hh1, *w_out, *b_out;
    const float *h0_l0, *h0_l1;    // [beats*B] rows of H values h0_stride floats apart, row = beat*B + b
    int64_t h0_stride;
    const float *gib;              // [beats*B][3H]
    const float *ptab;             // [V+1][3H]; row V = the start token
    const uint8_t *mask;           // [beats*tpb][B][H] or null
    float keep_scale;
    int batch, beats, tpb, vocab;
    int64_t *tokens;               // [B][beats*tpb]
};

template <int H, int TICK_PF, bool MASKED>
__global__ __launch_bounds__(H * 4) void tick_free_run_kernel(TickFreeRun p) {
    constexpr int KQ = H / 16, HS = H + 4;
    __shared__ float hA0[2][16][HS];
    __shared__ float hA1[2][16][HS];
    __shared__ float mid[16][HS];
    __shared__ float cand_v[4][16];
    __shared__ int cand_i[4][16];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, quad = lane >> 4;
    const int unit = 16 * w + col;
    const int row0 = blockIdx.x * 16;
    const int B = p.batch;
    const int ntile = (p.vocab + 15) / 16;          // waves that compute logits (<= 4)

    f32x4 whh0[3][KQ];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq)
            whh0[g][kq] = *reinterpret_cast<const f32x4 *>(p.w_hh0 + (int64_t)(g * H + unit) * H + 16 * kq + 4 * quad);
    const float b0r = p.b_hh0[unit], b0z = p.b_hh0[H + unit], b0n = p.b_hh0[2 * H + unit];
    const float b1r = p.b_ih1[unit] + p.b_hh1[unit], b1z = p.b_ih1[H + unit] + p.b_hh1[H + unit];
    const float b1in = p.b_ih1[2 * H + unit], b1hn = p.b_hh1[2 * H + unit];
    // logits: wave w < ntile owns notes [16w, 16w+16)
    const int note = 16 * w + col;
    const bool note_ok = w < ntile && note < p.vocab;
    f32x4 wout[KQ];
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq) {
        wout[kq] = *reinterpret_cast<const f32x4 *>(p.w_out + (int64_t)(note_ok ? note : 0) * H + 16 * kq + 4 * quad);
        if (!note_ok) wout[kq] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float bout = note_ok ? p.b_out[note] : 0.f;

    int rows[4];
    bool live[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + 4 * quad + i;
        live[i] = r < B;
        rows[i] = live[i] ? r : B - 1;
    }
    float h0[4], h1[4], gb[4][3];
    int tok[4] = {p.vocab, p.vocab, p.vocab, p.vocab};        // start token row of ptab
    const int ticks = p.beats * p.tpb;

    for (int t = 0; t < ticks; ++t) {
        const int cur = t & 1;
        const int beat = t / p.tpb;
        if (t % p.tpb == 0) {                                 // the hidden state restarts at every beat
            lds_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t br = (int64_t)beat * B + rows[i];
                h0[i] = p.h0_l0[br * p.h0_stride + unit];
                h1[i] = p.h0_l1[br * p.h0_stride + unit];
                hA0[cur][4 * quad + i][unit] = h0[i];
                hA1[cur][4 * quad + i][unit] = h1[i];
                const float *g = p.gib + br * 3 * H + unit;
                gb[i][0] = g[0]; gb[i][1] = g[H]; gb[i][2] = g[2 * H];
            }
            lds_barrier();
        }
        // input projection of this tick: beat part + previous-token part
        float gi[4][3];
        float keep[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float *pt = p.ptab + (int64_t)tok[i] * 3 * H + unit;
            gi[i][0] = gb[i][0] + pt[0]; gi[i][1] = gb[i][1] + pt[H]; gi[i][2] = gb[i][2] + pt[2 * H];
            keep[i] = MASKED ? p.keep_scale * (float)p.mask[((int64_t)t * B + rows[i]) * H + unit] : 1.f;
        }
        // ---- layer 0
        {
            f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int kq = 0; kq < KQ; ++kq) {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(&hA0[cur][col][16 * kq + 4 * quad]);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int g = 0; g < 3; ++g)
                        acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], whh0[g][kq][j], acc[g], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float r = fast_sigmoid(gi[i][0] + acc[0][i] + b0r);
                const float z = fast_sigmoid(gi[i][1] + acc[1][i] + b0z);
                const float n = fast_tanh(gi[i][2] + r * (acc[2][i] + b0n));
                h0[i] = (1.f - z) * n + z * h0[i];
                hA0[cur ^ 1][4 * quad + i][unit] = h0[i];
                mid[4 * quad + i][unit] = h0[i] * keep[i];
            }
        }
        lds_barrier();
        // ---- layer 1: weights streamed, PF k-groups ahead
        {
            constexpr int PF = TICK_PF < KQ ? TICK_PF : KQ;
            f32x4 wi[PF + 1][3], wh[PF + 1][3];
            auto fetch = [&](int kq, int slot) {
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const int64_t o = (int64_t)(g * H + unit) * H + 16 * kq + 4 * quad;
                    wi[slot][g] = *reinterpret_cast<const f32x4 *>(p.w_ih1 + o);
                    wh[slot][g] = *reinterpret_cast<const f32x4 *>(p.w_hh1 + o);
                }
            };
            f32x4 ar = {0.f, 0.f, 0.f, 0.f}, az = ar, ain = ar, ahn = ar;
            static_assert(KQ >= PF, "tick_free_run: hidden size too small for the prefetch depth");
#pragma unroll
            for (int k = 0; k < PF; ++k) fetch(k, k);
#pragma unroll
            for (int kq = 0; kq < KQ; ++kq) {
                if (kq + PF < KQ) fetch(kq + PF, (kq + PF) % (PF + 1));
                const int sl = kq % (PF + 1);
                const f32x4 am = *reinterpret_cast<const f32x4 *>(&mid[col][16 * kq + 4 * quad]);
                const f32x4 ah = *reinterpret_cast<const f32x4 *>(&hA1[cur][col][16 * kq + 4 * quad]);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    ar = __builtin_amdgcn_mfma_f32_16x16x4f32(am[j], wi[sl][0][j], ar, 0, 0, 0);
                    az = __builtin_amdgcn_mfma_f32_16x16x4f32(am[j], wi[sl][1][j], az, 0, 0, 0);
                    ain = __builtin_amdgcn_mfma_f32_16x16x4f32(am[j], wi[sl][2][j], ain, 0, 0, 0);
                    ar = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[j], wh[sl][0][j], ar, 0, 0, 0);
                    az = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[j], wh[sl][1][j], az, 0, 0, 0);
                    ahn = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[j], wh[sl][2][j], ahn, 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float r = fast_sigmoid(ar[i] + b1r);
                const float z = fast_sigmoid(az[i] + b1z);
                const float n = fast_tanh(ain[i] + b1in + r * (ahn[i] + b1hn));
                h1[i] = (1.f - z) * n + z * h1[i];
                hA1[cur ^ 1][4 * quad + i][unit] = h1[i];
            }
        }
        lds_barrier();
        // ---- logits + row argmax
        if (w < ntile) {
            f32x4 lg = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kq = 0; kq < KQ; ++kq) {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(&hA1[cur ^ 1][col][16 * kq + 4 * quad]);
#pragma unroll
                for (int j = 0; j < 4; ++j) lg = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], wout[kq][j], lg, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = note_ok ? fmaxf(lg[i] + bout, 0.f) : -1.f;
                int ix = note;
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    const float ov = __shfl_xor(v, off, 64);
                    const int oi = __shfl_xor(ix, off, 64);
                    if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; }
                }
                if (col == 0) { cand_v[w][4 * quad + i] = v; cand_i[w][4 * quad + i] = ix; }
            }
        }
        lds_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * quad + i;
            float v = cand_v[0][r];
            int ix = cand_i[0][r];
            for (int c = 1; c < ntile; ++c) {
                const float ov = cand_v[c][r];
                if (ov > v) { v = ov; ix = cand_i[c][r]; }          // later tiles hold larger indices: ties keep the earlier
            }
            tok[i] = ix;
            if (w == 0 && col == 0 && live[i]) p.tokens[(int64_t)rows[i] * ticks + t] = ix;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The free-running tick decoder on the bf16 MFMA (three-term split, fp32-accurate).  Three terms of the three
// recurrent matrices do not fit any on-chip store (885 KB), so they are streamed: tick_weight_prep_kernel writes them
// once per call in the exact per-lane register order (group = (matrix, k-step), 9 x 16 bytes per lane and group, lanes
// contiguous), and every wave keeps a three-slot ring of groups in registers, always two groups (and across the tick
// boundary) ahead of the MFMAs.  Per tick a workgroup pulls 885 KB from L2 against 3 x 72 MFMAs of 16 cycles per wave:
// the kernel is L2-bandwidth bound (the fp32-MFMA version above was bound by its 288 x 32-cycle MFMAs and the exposed
// latency of its unpipelined weight loads).
struct TickPrep {
    const float *w[3];           // w_hh0, w_ih1, w_hh1
    uint4 *out;
};

template <int H>
__global__ __launch_bounds__(256) void tick_weight_prep_kernel(TickPrep p) {
    constexpr int NW = H / 16, KS = H / 32;
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const int lane = tid & 63;
    int rest = tid >> 6;
    const int g = rest % 3; rest /= 3;
    const int w = rest % NW; rest /= NW;
    const int ks = rest % KS;
    const int m = rest / KS;
    if (m >= 3) return;
    const int col = lane & 15, quad = lane >> 4;
    const float *src = (m == 0 ? p.w[0] : m == 1 ? p.w[1] : p.w[2]) + (int64_t)(g * H + 16 * w + col) * H + 32 * ks + 8 * quad;
    const f32x4 v0 = *reinterpret_cast<const f32x4 *>(src), v1 = *reinterpret_cast<const f32x4 *>(src + 4);
    const float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    bf16x8g hi, mid, lo;
    split3_x8(x, hi, mid, lo);
    uint4 *dst = p.out + ((int64_t)((m * KS + ks) * NW + w) * 9 + g * 3) * 64 + lane;
    dst[0] = __builtin_bit_cast(uint4, hi);
    dst[64] = __builtin_bit_cast(uint4, mid);
    dst[128] = __builtin_bit_cast(uint4, lo);
}

template <int H, bool MASKED>
__global__ __launch_bounds__(H * 4) void tick_free_run_x3_kernel(TickFreeRun p, const uint4 *__restrict__ packed) {
    constexpr int NW = H / 16, KS = H / 32, KQ = H / 16;
    constexpr int NGG = 9 * KS;                    // weight groups per tick: (matrix, k-step, gate), 3 x 16 bytes per lane each
    constexpr int RS = NGG % 6 == 0 ? 6 : 3;       // register ring of groups; RS - 1 groups are in flight
    constexpr int PFD = RS - 1;
    constexpr int HP = H + 8, PLANE = 16 * HP, HS = H + 4;
    __shared__ __attribute__((aligned(16))) unsigned short hA0[2][3 * PLANE];
    __shared__ __attribute__((aligned(16))) unsigned short hA1[2][3 * PLANE];
    __shared__ __attribute__((aligned(16))) unsigned short midp[3 * PLANE];
    __shared__ __attribute__((aligned(16))) float h1f[16][HS];
    __shared__ __attribute__((aligned(16))) float wout_s[64][HS];
    __shared__ float cand_v[4][16];
    __shared__ int cand_i[4][16];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, quad = lane >> 4;
    const int unit = 16 * w + col;
    const int row0 = blockIdx.x * 16;
    const int B = p.batch;
    const int ntile = (p.vocab + 15) / 16;

    // weight stream: one buffer resource, one per-lane byte offset, the group's offset as the scalar offset of each
    // load -- per-load 64-bit addresses would be hoisted out of the tick loop into 200+ VGPRs
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(packed), 0, 3 * KS * NW * 9 * 64 * 16, 0x00020000);
    const int wlane = (w * 9 * 64 + lane) * 16;
    bf16x8g wb[RS][3];
    auto fetch = [&](int gg) {                     // gg = (matrix * KS + ks) * 3 + gate, compile-time at every call site
        const int g = gg / 3, gate = gg % 3;
#pragma unroll
        for (int term = 0; term < 3; ++term)
            wb[gg % RS][term] = __builtin_bit_cast(bf16x8g, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlane, (g * NW * 9 + gate * 3 + term) * 64 * 16, 0));
    };
#pragma unroll
    for (int d = 0; d < PFD; ++d) fetch(d % NGG);

    for (int e = threadIdx.x; e < 64 * H; e += H * 4) {       // note projection weights -> LDS (rows >= vocab: zeros)
        const int n = e / H, k = e - n * H;
        wout_s[n][k] = n < p.vocab ? p.w_out[(int64_t)n * H + k] : 0.f;
    }
    const float b0r = p.b_hh0[unit], b0z = p.b_hh0[H + unit], b0n = p.b_hh0[2 * H + unit];
    const float b1r = p.b_ih1[unit] + p.b_hh1[unit], b1z = p.b_ih1[H + unit] + p.b_hh1[H + unit];
    const float b1in = p.b_ih1[2 * H + unit], b1hn = p.b_hh1[2 * H + unit];
    const int note = 16 * w + col;
    const bool note_ok = w < ntile && note < p.vocab;
    const float bout = note_ok ? p.b_out[note] : 0.f;

    int rows[4];
    bool live[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + 4 * quad + i;
        live[i] = r < B;
        rows[i] = live[i] ? r : B - 1;
    }
    float h0[4], h1[4], gb[4][3];
    int tok[4] = {p.vocab, p.vocab, p.vocab, p.vocab};
    const int ticks = p.beats * p.tpb;
    const int aoff = col * HP + 8 * quad;                     // this lane's A-operand offset inside a plane

    for (int t = 0; t < ticks; ++t) {
        const int cur = t & 1;
        const int beat = t / p.tpb;
        if (t % p.tpb == 0) {
            lds_barrier();
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t br = (int64_t)beat * B + rows[i];
                h0[i] = p.h0_l0[br * p.h0_stride + unit];
                h1[i] = p.h0_l1[br * p.h0_stride + unit];
                store_split3(&hA0[cur][(4 * quad + i) * HP + unit], PLANE, h0[i]);
                store_split3(&hA1[cur][(4 * quad + i) * HP + unit], PLANE, h1[i]);
                const float *g = p.gib + br * 3 * H + unit;
                gb[i][0] = g[0]; gb[i][1] = g[H]; gb[i][2] = g[2 * H];
            }
            lds_barrier();
        }
        float gi[4][3], keep[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float *pt = p.ptab + (int64_t)tok[i] * 3 * H + unit;
            gi[i][0] = gb[i][0] + pt[0]; gi[i][1] = gb[i][1] + pt[H]; gi[i][2] = gb[i][2] + pt[2 * H];
            keep[i] = MASKED ? p.keep_scale * (float)p.mask[((int64_t)t * B + rows[i]) * H + unit] : 1.f;
        }
        // ---- layer 0: matrix 0
        {
            f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            const unsigned short *ab = &hA0[cur][aoff];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8g ah = lds_x8(ab + 32 * ks), am = lds_x8(ab + PLANE + 32 * ks), al = lds_x8(ab + 2 * PLANE + 32 * ks);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int gg = (0 * KS + ks) * 3 + q;
                    fetch((gg + PFD) % NGG);
                    __builtin_amdgcn_sched_barrier(0);
                    GRU_MFMA6(acc[q], ah, am, al, wb[gg % RS][0], wb[gg % RS][1], wb[gg % RS][2]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float r = fast_sigmoid(gi[i][0] + acc[0][i] + b0r);
                const float z = fast_sigmoid(gi[i][1] + acc[1][i] + b0z);
                const float n = fast_tanh(gi[i][2] + r * (acc[2][i] + b0n));
                h0[i] = (1.f - z) * n + z * h0[i];
                store_split3(&hA0[cur ^ 1][(4 * quad + i) * HP + unit], PLANE, h0[i]);
                store_split3(&midp[(4 * quad + i) * HP + unit], PLANE, h0[i] * keep[i]);
            }
        }
        lds_barrier();
        // ---- layer 1: matrix 1 (W_ih1 on mid), matrix 2 (W_hh1 on h1); r and z share an accumulator
        {
            f32x4 a1[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // r, z, i_n, h_n
#pragma unroll
            for (int m = 1; m <= 2; ++m) {
                const unsigned short *ab = m == 1 ? &midp[aoff] : &hA1[cur][aoff];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8g ah = lds_x8(ab + 32 * ks), am = lds_x8(ab + PLANE + 32 * ks), al = lds_x8(ab + 2 * PLANE + 32 * ks);
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const int gg = (m * KS + ks) * 3 + q;
                        fetch((gg + PFD) % NGG);
                        __builtin_amdgcn_sched_barrier(0);
                        GRU_MFMA6(a1[q == 2 ? m + 1 : q], ah, am, al, wb[gg % RS][0], wb[gg % RS][1], wb[gg % RS][2]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float r = fast_sigmoid(a1[0][i] + b1r);
                const float z = fast_sigmoid(a1[1][i] + b1z);
                const float n = fast_tanh(a1[2][i] + b1in + r * (a1[3][i] + b1hn));
                h1[i] = (1.f - z) * n + z * h1[i];
                store_split3(&hA1[cur ^ 1][(4 * quad + i) * HP + unit], PLANE, h1[i]);
                h1f[4 * quad + i][unit] = h1[i];
            }
        }
        lds_barrier();
        // ---- logits (fp32 MFMA, weights in LDS) + row argmax
        if (w < ntile) {
            f32x4 lg = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kq = 0; kq < KQ; ++kq) {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(&h1f[col][16 * kq + 4 * quad]);
                const f32x4 b = *reinterpret_cast<const f32x4 *>(&wout_s[note][16 * kq + 4 * quad]);
#pragma unroll
                for (int j = 0; j < 4; ++j) lg = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], lg, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = note_ok ? fmaxf(lg[i] + bout, 0.f) : -1.f;
                int ix = note;
#pragma unroll
                for (int off = 1; off < 16; off <<= 1) {
                    const float ov = __shfl_xor(v, off, 64);
                    const int oi = __shfl_xor(ix, off, 64);
                    const bool take = ov > v || (ov == v && oi < ix);
                    v = take ? ov : v;
                    ix = take ? oi : ix;
                }
                if (col == 0) { cand_v[w][4 * quad + i] = v; cand_i[w][4 * quad + i] = ix; }
            }
        }
        lds_barrier();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * quad + i;
            float v = cand_v[0][r];
            int ix = cand_i[0][r];
            for (int c = 1; c < ntile; ++c) {
                const float ov = cand_v[c][r];
                const int oi = cand_i[c][r];
                const bool take = ov > v;                      // later tiles hold larger indices: ties keep the earlier
                v = take ? ov : v;
                ix = take ? oi : ix;
            }
            tok[i] = ix;
            if (w == 0 && col == 0 && live[i]) p.tokens[(int64_t)rows[i] * ticks + t] = ix;
        }
    }
}

// one accumulator: l h', h l', h h'
#define GRU_MFMA3(ACC, AH, AL, WH, WL)                                                 \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(AL, WH, ACC, 0, 0, 0);                \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(AH, WL, ACC, 0, 0, 0);                \
    ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(AH, WH, ACC, 0, 0, 0)

// the same free-running pass on the scaled two-term fp16 operands of gru_seq_fwd_h2_kernel: two thirds of the weight stream
// (590 instead of 885 KB per tick and workgroup, the kernel's bound) and half the MFMAs
// largest magnitude of each of the three matrices (one workgroup per matrix) -> wmax[0 .. 3): the free-running kernel's weight scales
// (round 5; a fixed 2^8 before, which overflowed fp16 for |w| >= 255)
template <int H>
__global__ __launch_bounds__(1024) void tick_weight_amax_kernel(TickPrep p, float *__restrict__ wmax) {
    __shared__ float red[16];
    const float *w = blockIdx.x == 0 ? p.w[0] : blockIdx.x == 1 ? p.w[1] : p.w[2];
    float m = 0.f;
    for (int i = threadIdx.x; i < 3 * H * H / 4; i += 1024) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(w + 4 * i);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int q = 0; q < 16; ++q) t = fmaxf(t, red[q]);
        wmax[blockIdx.x] = t;
    }
}
// the matrices' scales: W_hh0 its own, W_ih1 and W_hh1 one between them (their products with the layer-1 operands share accumulators)
__device__ __forceinline__ GruPow2 tick_weight_scale(const float *wmax, int m) {
    return gru_pow2(m == 0 ? wmax[0] : fmaxf(wmax[1], wmax[2]));
}

template <int H>
__global__ __launch_bounds__(256) void tick_weight_prep_h2_kernel(TickPrep p, const float *__restrict__ wmax) {
    constexpr int NW = H / 16, KS = H / 32;
    const int tid = blockIdx.x * 256 + threadIdx.x;
    const int lane = tid & 63;
    int rest = tid >> 6;
    const int g = rest % 3; rest /= 3;
    const int w = rest % NW; rest /= NW;
    const int ks = rest % KS;
    const int m = rest / KS;
    if (m >= 3) return;
    const int col = lane & 15, quad = lane >> 4;
    const float *src = (m == 0 ? p.w[0] : m == 1 ? p.w[1] : p.w[2]) + (int64_t)(g * H + 16 * w + col) * H + 32 * ks + 8 * quad;
    const f32x4 v0 = *reinterpret_cast<const f32x4 *>(src), v1 = *reinterpret_cast<const f32x4 *>(src + 4);
    const float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    f16x8g hi, lo;
    split2_x8(x, tick_weight_scale(wmax, m).s, hi, lo);
    uint4 *dst = p.out + ((int64_t)((m * KS + ks) * NW + w) * 6 + g * 2) * 64 + lane;
    dst[0] = __builtin_bit_cast(uint4, hi);
    dst[64] = __builtin_bit_cast(uint4, lo);
}

// one stage of a (value, index) argmax inside a 16-lane DPP row: the partner lane's pair through a DPP move, larger value wins,
// equal values keep the lower index (commutative and associative: any sequence of pairings that connects the 16 lanes gives the
// row's maximum with its lowest index in every lane)
template <int CTRL>
__device__ __forceinline__ void tick_argmax_stage(float &v, int &ix) {
    const float ov = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
    const int oi = __builtin_amdgcn_update_dpp(0, ix, CTRL, 0xf, 0xf, true);
    const bool take = ov > v || (ov == v && oi < ix);
    v = take ? ov : v;
    ix = take ? oi : ix;
}

// (RW: batch rows per workgroup, 16 or 4 -- one element per lane and four times the workgroups, as gru_seq_fwd_h2_kernel: the gates,
// the token's projections, the state splits and the rows' argmax are per-element work; the weight stream per workgroup is unchanged)
template <int H, bool MASKED, int RW>
__global__ __launch_bounds__(H * 4) void tick_free_run_h2_kernel(TickFreeRun p, const uint4 *__restrict__ packed) {
    static_assert(RW == 16 || RW == 8 || RW == 4, "16, 8 or 4 rows: four, two or one per lane");
    constexpr int E = RW / 4;
    constexpr int NW = H / 16, KS = H / 32, KQ = H / 16;
    constexpr int NGG = 9 * KS;                    // weight groups per tick: (matrix, k-step, gate), 3 x 16 bytes per lane each
    constexpr int RS = NGG % 6 == 0 ? 6 : 3;       // register ring of groups; RS - 1 groups are in flight
    constexpr int PFD = RS - 1;
    constexpr int HP = H + 8, PLANE = 16 * HP, HS = H + 4;
    __shared__ __attribute__((aligned(16))) unsigned short hA0[2][2 * PLANE];
    __shared__ __attribute__((aligned(16))) unsigned short hA1[2][2 * PLANE];
    __shared__ __attribute__((aligned(16))) unsigned short midp[2 * PLANE];
    __shared__ __attribute__((aligned(16))) float h1f[16][HS];
    __shared__ __attribute__((aligned(16))) float wout_s[64][HS];
    __shared__ float cand_v[4][16];
    __shared__ float hmax[H / 16];
    __shared__ int cand_i[4][16];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, quad = lane >> 4;
    const int unit = 16 * w + col;
    const int row0 = blockIdx.x * RW;
    const int B = p.batch;
    const int ntile = (p.vocab + 15) / 16;

    // weight stream: one buffer resource, one per-lane byte offset, the group's offset as the scalar offset of each
    // load -- per-load 64-bit addresses would be hoisted out of the tick loop into 200+ VGPRs
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4 *>(packed), 0, 3 * KS * NW * 6 * 64 * 16, 0x00020000);
    const int wlane = (w * 6 * 64 + lane) * 16;
    f16x8g wb[RS][2];
    auto fetch = [&](int gg) {                     // gg = (matrix * KS + ks) * 3 + gate, compile-time at every call site
        const int g = gg / 3, gate = gg % 3;
#pragma unroll
        for (int term = 0; term < 2; ++term)
            wb[gg % RS][term] = __builtin_bit_cast(f16x8g, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlane, (g * NW * 6 + gate * 2 + term) * 64 * 16, 0));
    };
#pragma unroll
    for (int d = 0; d < PFD; ++d) fetch(d % NGG);

    for (int e = threadIdx.x; e < 64 * H; e += H * 4) {       // note projection weights -> LDS (rows >= vocab: zeros)
        const int n = e / H, k = e - n * H;
        wout_s[n][k] = n < p.vocab ? p.w_out[(int64_t)n * H + k] : 0.f;
    }
    const float b0r = p.b_hh0[unit], b0z = p.b_hh0[H + unit], b0n = p.b_hh0[2 * H + unit];
    const float b1r = p.b_ih1[unit] + p.b_hh1[unit], b1z = p.b_ih1[H + unit] + p.b_hh1[H + unit];
    const float b1in = p.b_ih1[2 * H + unit], b1hn = p.b_hh1[2 * H + unit];
    const int note = 16 * w + col;
    const bool note_ok = w < ntile && note < p.vocab;
    const float bout = note_ok ? p.b_out[note] : 0.f;

    auto lrow = [&](int i) { return gru_lrow<E>(quad, i); };                  // the tile row of this lane's element i
    int rows[E];
    bool live[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int r = row0 + lrow(i);
        live[i] = r < B;
        rows[i] = live[i] ? r : B - 1;
    }
    float h0[E], h1[E], gb[E][3];
    int tok[E];
#pragma unroll
    for (int i = 0; i < E; ++i) tok[i] = p.vocab;
    const int ticks = p.beats * p.tpb;
    // operand scales from the data (round 5): the matrices' from their maxima (tick_weight_amax_kernel, behind the packed weights), the
    // states' per beat from the workgroup's rows -- every state of a beat is a convex combination of tanh outputs and the beat's
    // initial state, the layer-1 input is a layer-0 state times a keep byte's 0 or keep_scale
    const float *wmax = reinterpret_cast<const float *>(packed) + 9 * H * H;
    const float w0_inv = tick_weight_scale(wmax, 0).inv, w12_inv = tick_weight_scale(wmax, 1).inv;
    const float keep_bound = MASKED ? fmaxf(p.keep_scale, 1.f) : 1.f;
    float h_s = 1.f, us0 = 1.f, us12 = 1.f;                   // (set at every beat's start: tick 0 starts one)
    const int arow = gru_arow<E>(col);
    const int aoff = arow * HP + 8 * quad;                    // this lane's A-operand offset inside a plane
    auto elems = [&](const f32x4 &acc, float (&out)[E]) __attribute__((always_inline)) { gru_elems<E>(acc, out); };
    // layer 0's recurrent product W_hh0 h0 of a tick does not wait for the tick's token: it is multiplied at the END of the previous
    // tick, under the logits and the argmax (three waves' latency chain of ~3500 cycles, during which the workgroup's weight
    // stream -- what bounds the layers: 590 KB per tick at the CU's 64 bytes per clock -- stood still; tools/stamp_tick.py).
    // A beat's first tick starts from the beat's own state and multiplies at its top, as every tick did.
    f32x4 acc0[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    // the 3 KS weight groups of matrix 0 against the state image `ab`; piece(g) runs behind group g
    auto layer0 = [&](const unsigned short *ab, auto piece) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 3; ++q) acc0[q] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const f16x8g ah = lds_h8(ab + 32 * ks), al = lds_h8(ab + PLANE + 32 * ks);
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int gg = (0 * KS + ks) * 3 + q;
                fetch((gg + PFD) % NGG);
                __builtin_amdgcn_sched_barrier(0);
                GRU_MFMA3(acc0[q], ah, al, wb[gg % RS][0], wb[gg % RS][1]);
                __builtin_amdgcn_sched_barrier(0);
                piece(gg);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto no_piece = [](int) __attribute__((always_inline)) {};
#ifdef ARVAE_GRU_STAMPS
    // diagnostic build (tools/stamp_tick.py): cycles per phase of a tick, wave GRU_STAMP_WAVE of workgroup 0
    unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ttc = __builtin_readcyclecounter();
#define TSTAMP(k) { __builtin_amdgcn_s_waitcnt(0xc07f); const unsigned long long now = __builtin_readcyclecounter(); tph[k] += now - ttc; ttc = now; }
#else
#define TSTAMP(k)
#endif
    for (int t = 0; t < ticks; ++t) {
        const int cur = t & 1;
        const int beat = t / p.tpb;
        const bool beat_start = t % p.tpb == 0;
        if (beat_start) {
            lds_barrier();
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const int64_t br = (int64_t)beat * B + rows[i];
                h0[i] = p.h0_l0[br * p.h0_stride + unit];
                h1[i] = p.h0_l1[br * p.h0_stride + unit];
                const float *g = p.gib + br * 3 * H + unit;
                gb[i][0] = g[0]; gb[i][1] = g[H]; gb[i][2] = g[2 * H];
            }
            {   // the beat's state scale: max(keep bound, keep bound * |h0|, |h1|) over the workgroup's rows just below 2^15
                float mx = keep_bound;
#pragma unroll
                for (int i = 0; i < E; ++i) mx = fmaxf(mx, fmaxf(keep_bound * fabsf(h0[i]), fabsf(h1[i])));
                mx = row16_max(mx);
                mx = fmaxf(mx, __shfl_xor(mx, 16));
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                if (lane == 0) hmax[w] = mx;
                lds_barrier();
#pragma unroll
                for (int q = 0; q < H / 16; ++q) mx = fmaxf(mx, hmax[q]);
                const GruPow2 sh = gru_pow2(mx);
                h_s = sh.s;
                us0 = sh.inv * w0_inv;
                us12 = sh.inv * w12_inv;
            }
#pragma unroll
            for (int i = 0; i < E; ++i) {
                store_split2_s(&hA0[cur][lrow(i) * HP + unit], PLANE, h0[i], h_s);
                store_split2_s(&hA1[cur][lrow(i) * HP + unit], PLANE, h1[i], h_s);
            }
            lds_barrier();
        }
        float gi[E][3], keep[E];
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const float *pt = p.ptab + (int64_t)tok[i] * 3 * H + unit;
            gi[i][0] = gb[i][0] + pt[0]; gi[i][1] = gb[i][1] + pt[H]; gi[i][2] = gb[i][2] + pt[2 * H];
            keep[i] = MASKED ? p.keep_scale * (float)p.mask[((int64_t)t * B + rows[i]) * H + unit] : 1.f;
        }
        TSTAMP(0);                                             // tick top: a beat's state; the token's projections requested
        // ---- layer 0: matrix 0 (multiplied at the end of the previous tick unless a beat starts)
        {
            if (beat_start) layer0(&hA0[cur][aoff], no_piece);
#ifdef ARVAE_GRU_STAMPS
            { float dep = acc0[0][0] + acc0[1][1] + acc0[2][3]; asm volatile("" :: "v"(dep)); }
#endif
            TSTAMP(1);                                         // layer 0 at the top (a beat's first tick only)
            float ar[E], az[E], an[E];
            elems(acc0[0], ar); elems(acc0[1], az); elems(acc0[2], an);
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const float r = fast_sigmoid(gi[i][0] + ar[i] * us0 + b0r);
                const float z = fast_sigmoid(gi[i][1] + az[i] * us0 + b0z);
                const float n = fast_tanh(gi[i][2] + r * (an[i] * us0 + b0n));
                h0[i] = (1.f - z) * n + z * h0[i];
                store_split2_s(&hA0[cur ^ 1][lrow(i) * HP + unit], PLANE, h0[i], h_s);
                store_split2_s(&midp[lrow(i) * HP + unit], PLANE, h0[i] * keep[i], h_s);
            }
        }
        TSTAMP(2);                                             // layer 0 gates (wait for the projections) + LDS writes
        lds_barrier();
        TSTAMP(3);
        // ---- layer 1: matrix 1 (W_ih1 on mid), matrix 2 (W_hh1 on h1); r and z share an accumulator
        {
            f32x4 a1[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};   // r, z, i_n, h_n
#pragma unroll
            for (int m = 1; m <= 2; ++m) {
                const unsigned short *ab = m == 1 ? &midp[aoff] : &hA1[cur][aoff];
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const f16x8g ah = lds_h8(ab + 32 * ks), al = lds_h8(ab + PLANE + 32 * ks);
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const int gg = (m * KS + ks) * 3 + q;
                        fetch((gg + PFD) % NGG);
                        __builtin_amdgcn_sched_barrier(0);
                        GRU_MFMA3(a1[q == 2 ? m + 1 : q], ah, al, wb[gg % RS][0], wb[gg % RS][1]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
#ifdef ARVAE_GRU_STAMPS
            { float dep = a1[0][0] + a1[1][1] + a1[2][3] + a1[3][2]; asm volatile("" :: "v"(dep)); }
#endif
            TSTAMP(4);                                         // layer 1: operand reads + MFMAs behind the weight stream
            float ar[E], az[E], ai[E], ah[E];
            elems(a1[0], ar); elems(a1[1], az); elems(a1[2], ai); elems(a1[3], ah);
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const float r = fast_sigmoid(ar[i] * us12 + b1r);
                const float z = fast_sigmoid(az[i] * us12 + b1z);
                const float n = fast_tanh(ai[i] * us12 + b1in + r * (ah[i] * us12 + b1hn));
                h1[i] = (1.f - z) * n + z * h1[i];
                store_split2_s(&hA1[cur ^ 1][lrow(i) * HP + unit], PLANE, h1[i], h_s);
                h1f[lrow(i)][unit] = h1[i];
            }
        }
        TSTAMP(5);                                             // layer 1 gates + LDS writes
        lds_barrier();
        // ---- logits (fp32 MFMA, weights in LDS) + row argmax on the first waves; on every wave the NEXT tick's layer 0
        {
            const bool pre = t + 1 < ticks && (t + 1) % p.tpb != 0;
            const unsigned short *ab_next = &hA0[cur ^ 1][aoff];
            if (w < ntile) {
                f32x4 lg = {0.f, 0.f, 0.f, 0.f};
                auto logits_step = [&](int kq) __attribute__((always_inline)) {
                    const f32x4 a = *reinterpret_cast<const f32x4 *>(&h1f[arow][16 * kq + 4 * quad]);
                    const f32x4 b = *reinterpret_cast<const f32x4 *>(&wout_s[note][16 * kq + 4 * quad]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) lg = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], lg, 0, 0, 0);
                };
                // the four rows' (value, lowest index) maxima over the tile's 16 notes = one DPP row each: lane pairings on the
                // vector ALU (quad permutes, then the half-row and the row mirrored) instead of four ds_bpermute round trips per
                // row; stage by stage over the four rows (four independent chains), one block of candidate writes
                auto argmax_rows = [&]() __attribute__((always_inline)) {
                    float v[E], lv[E];
                    int ix[E];
                    elems(lg, lv);
#pragma unroll
                    for (int i = 0; i < E; ++i) { v[i] = note_ok ? fmaxf(lv[i] + bout, 0.f) : -1.f; ix[i] = note; }
#pragma unroll
                    for (int i = 0; i < E; ++i) tick_argmax_stage<0xB1>(v[i], ix[i]);
#pragma unroll
                    for (int i = 0; i < E; ++i) tick_argmax_stage<0x4E>(v[i], ix[i]);
#pragma unroll
                    for (int i = 0; i < E; ++i) tick_argmax_stage<0x141>(v[i], ix[i]);
#pragma unroll
                    for (int i = 0; i < E; ++i) tick_argmax_stage<0x140>(v[i], ix[i]);
                    if (col == 0) {
#pragma unroll
                        for (int i = 0; i < E; ++i) { cand_v[w][lrow(i)] = v[i]; cand_i[w][lrow(i)] = ix[i]; }
                    }
                };
                constexpr int NG0 = 3 * KS;
                if (pre) {
                    // the logits' k-steps behind matrix 0's first groups, the rows' argmax behind the next one
                    layer0(ab_next, [&](int g) __attribute__((always_inline)) {
                        if (g < KQ) logits_step(g);
                        else if (g == KQ) argmax_rows();
                    });
                    if (NG0 <= KQ) argmax_rows();
                } else {
#pragma unroll
                    for (int kq = 0; kq < KQ; ++kq) logits_step(kq);
                    argmax_rows();
                }
            } else if (pre) {
                layer0(ab_next, no_piece);
            }
#ifdef ARVAE_GRU_STAMPS
            { float dep = acc0[0][0] + acc0[1][1] + acc0[2][3]; asm volatile("" :: "v"(dep)); }
#endif
        }
        TSTAMP(6);                                             // barrier + logits / argmax (first waves) + the next tick's layer 0
        lds_barrier();
#pragma unroll
        for (int i = 0; i < E; ++i) {
            const int r = lrow(i);
            float v = cand_v[0][r];
            int ix = cand_i[0][r];
            for (int c = 1; c < ntile; ++c) {
                const float ov = cand_v[c][r];
                const int oi = cand_i[c][r];
                const bool take = ov > v;                      // later tiles hold larger indices: ties keep the earlier
                v = take ? ov : v;
                ix = take ? oi : ix;
            }
            tok[i] = ix;
            if (w == 0 && col == 0 && live[i]) p.tokens[(int64_t)rows[i] * ticks + t] = ix;
        }
        TSTAMP(7);                                             // barrier + the tiles' candidates -> token
    }
#ifdef ARVAE_GRU_STAMPS
#ifndef GRU_STAMP_WAVE
#define GRU_STAMP_WAVE 0
#endif
    if (blockIdx.x == 0 && threadIdx.x == 64 * GRU_STAMP_WAVE) {
        for (int q = 0; q < 8; ++q) g_tick_stamps[q] = tph[q];
        g_tick_stamps[8] = ticks;
    }
#endif
}

}  // namespace arvae

using namespace arvae;

// ARVAE_GRU_FP32=1: the fp32-MFMA kernels (A/B measurements; the default is the three-term bf16 split)
static bool gru_fp32_mfma() {
    static const bool on = diag_env("ARVAE_GRU_FP32") != nullptr;
    return on;
}

// ARVAE_GRU_BF16_FWD=1 (diagnostic build): the forward recurrence on the three-term bf16 split, as through round 3
static bool gru_bf16_forward() {
    static const bool on = diag_env("ARVAE_GRU_BF16_FWD") != nullptr;
    return on;
}

// Batch rows per workgroup (gru_seq_fwd_h2_kernel): the smallest of 4, 8, 16 whose workgroups are all on the chip at once.
// ARVAE_GRU_WIDE=1 (diagnostic build): always sixteen, as through round 4; ARVAE_GRU_ROWS=4|8|16: that many
static int gru_rows_per_wg(int rows, int nseq) {
    static const bool wide = diag_env("ARVAE_GRU_WIDE") != nullptr;
    static const int forced = diag_env("ARVAE_GRU_ROWS") != nullptr ? atoi(diag_env("ARVAE_GRU_ROWS")) : 0;
    if (wide) return 16;
    if (forced == 4 || forced == 8 || forced == 16) return forced;
    for (int rw = 4; rw < 16; rw *= 2)
        if ((int64_t)((rows + rw - 1) / rw) * nseq <= device_cu_count()) return rw;
    return 16;
}
// launch KERNEL<hidden, ..., rows per workgroup> for the hidden size and row width at hand
#define GRU_LAUNCH_RW(KERNEL, HH, RWV, ...)                                                                                       \
    {                                                                                                                            \
        const dim3 g_((rows + (RWV) - 1) / (RWV), nseq);                                                                          \
        if ((RWV) == 4) ARVAE_LAUNCH((KERNEL<HH, 4>), g_, dim3(4 * HH), 0, st, __VA_ARGS__);                                      \
        else if ((RWV) == 8) ARVAE_LAUNCH((KERNEL<HH, 8>), g_, dim3(4 * HH), 0, st, __VA_ARGS__);                                 \
        else ARVAE_LAUNCH((KERNEL<HH, 16>), g_, dim3(4 * HH), 0, st, __VA_ARGS__);                                                \
    }
#define GRU_LAUNCH(KERNEL, RWV, ...)                                                                                             \
    {                                                                                                                            \
        if (hidden == 128) GRU_LAUNCH_RW(KERNEL, 128, RWV, __VA_ARGS__)                                                           \
        else if (hidden == 64) GRU_LAUNCH_RW(KERNEL, 64, RWV, __VA_ARGS__)                                                        \
        else GRU_LAUNCH_RW(KERNEL, 32, RWV, __VA_ARGS__)                                                                          \
    }

// ARVAE_GRU_BF16_BWD=1 (diagnostic build): the backward recurrence on the three-term bf16 split, as through round 4
static bool gru_bf16_backward() {
    static const bool on = diag_env("ARVAE_GRU_BF16_BWD") != nullptr;
    return on;
}

extern "C" int arvae_gru_seq_supported(int32_t hidden) { return hidden == 32 || hidden == 64 || hidden == 128; }

static bool any_mask(const GruSeqMask *masks, int nseq) {
    for (int i = 0; masks != nullptr && i < nseq; ++i)
        if (masks[i].mask != nullptr) return true;
    return false;
}
namespace arvae {
bool gru_seq_masks_supported() { return !gru_fp32_mfma() && !gru_bf16_forward() && !gru_bf16_backward(); }
}  // namespace arvae

extern "C" int arvae_gru_seq_fwd(const arvae_gru_seq_t *seqs, int32_t nseq, int32_t steps, int32_t rows, int32_t hidden,
                                 arvae_stream_t stream) {
    return gru_seq_fwd_masked(seqs, nullptr, nseq, steps, rows, hidden, stream);
}

int arvae::gru_seq_fwd_masked(const arvae_gru_seq_t *seqs, const GruSeqMask *masks, int32_t nseq, int32_t steps, int32_t rows, int32_t hidden,
                              arvae_stream_t stream) {
    ARVAE_REQUIRE(seqs != nullptr && nseq >= 1 && nseq <= GRU_SEQ_MAX, "gru_seq_fwd: 1..%d sequences per launch", GRU_SEQ_MAX);
    ARVAE_REQUIRE(!any_mask(masks, nseq) || gru_seq_masks_supported(), "gru_seq_fwd: output dropout needs the fp16 two-term kernels");
    for (int i = 0; masks != nullptr && i < nseq; ++i)
        ARVAE_REQUIRE(masks[i].mask == nullptr || (masks[i].h_masked != nullptr && masks[i].hm_stride >= hidden && masks[i].group >= 0),
                      "gru_seq_fwd: a masked sequence needs its output buffer");
    ARVAE_REQUIRE(steps >= 1 && rows >= 1, "gru_seq_fwd: empty sequence");
    ARVAE_REQUIRE(arvae_gru_seq_supported(hidden), "gru_seq_fwd: hidden size %d is not built (32, 64, 128)", hidden);
    for (int i = 0; i < nseq; ++i)
        ARVAE_REQUIRE(seqs[i].gi && seqs[i].w_hh && seqs[i].b_hh && seqs[i].h_all && seqs[i].saved, "gru_seq_fwd: null pointer");
    GruSeqBatch b{};
    fill_batch(&b, seqs, nseq, hidden, masks);
    // (the recurrence addresses its arrays with 32-bit byte offsets: gru_rsrc)
    for (int i = 0; i < nseq; ++i) {
        const GruSeq &q = b.seq[i];
        const int64_t gi_bytes = 4 * ((int64_t)steps * q.gi_tstride + (int64_t)rows * q.gi_rstride);
        const int64_t h_bytes = 4 * (int64_t)steps * rows * std::max<int64_t>(q.h_stride, 4 * (int64_t)hidden);
        ARVAE_REQUIRE(gi_bytes < GRU_RANGE && h_bytes < GRU_RANGE, "gru_seq_fwd: %d steps x %d rows do not fit 2 GB per array", steps, rows);
    }
    hipStream_t st = as_stream(stream);
    const dim3 grid((rows + 15) / 16, nseq);
    if (gru_fp32_mfma()) {
        if (hidden == 128) ARVAE_LAUNCH(gru_seq_fwd_kernel<128>, grid, dim3(512), 0, st, b, steps, rows);
        else if (hidden == 64) ARVAE_LAUNCH(gru_seq_fwd_kernel<64>, grid, dim3(256), 0, st, b, steps, rows);
        else ARVAE_LAUNCH(gru_seq_fwd_kernel<32>, grid, dim3(128), 0, st, b, steps, rows);
    } else if (gru_bf16_forward()) {
        if (hidden == 128) ARVAE_LAUNCH(gru_seq_fwd_x3_kernel<128>, grid, dim3(512), 0, st, b, steps, rows);
        else if (hidden == 64) ARVAE_LAUNCH(gru_seq_fwd_x3_kernel<64>, grid, dim3(256), 0, st, b, steps, rows);
        else ARVAE_LAUNCH(gru_seq_fwd_x3_kernel<32>, grid, dim3(128), 0, st, b, steps, rows);
    } else {
        const int rw = gru_rows_per_wg(rows, nseq);
        GRU_LAUNCH(gru_seq_fwd_h2_kernel, rw, b, steps, rows)
    }
    return check_launch("gru_seq_fwd_kernel");
}

extern "C" int arvae_gru_seq_bwd(const arvae_gru_seq_t *seqs, int32_t nseq, int32_t steps, int32_t rows, int32_t hidden,
                                 arvae_stream_t stream) {
    return gru_seq_bwd_masked(seqs, nullptr, nseq, steps, rows, hidden, stream);
}

int arvae::gru_seq_bwd_masked(const arvae_gru_seq_t *seqs, const GruSeqMask *masks, int32_t nseq, int32_t steps, int32_t rows, int32_t hidden,
                              arvae_stream_t stream) {
    ARVAE_REQUIRE(seqs != nullptr && nseq >= 1 && nseq <= GRU_SEQ_MAX, "gru_seq_bwd: 1..%d sequences per launch", GRU_SEQ_MAX);
    ARVAE_REQUIRE(!any_mask(masks, nseq) || gru_seq_masks_supported(), "gru_seq_bwd: output dropout needs the fp16 two-term kernels");
    for (int i = 0; masks != nullptr && i < nseq; ++i)
        ARVAE_REQUIRE(masks[i].mask == nullptr || seqs[i].dh_all != nullptr, "gru_seq_bwd: a masked sequence needs the gradient of its output");
    ARVAE_REQUIRE(steps >= 1 && rows >= 1, "gru_seq_bwd: empty sequence");
    ARVAE_REQUIRE(arvae_gru_seq_supported(hidden), "gru_seq_bwd: hidden size %d is not built (32, 64, 128)", hidden);
    for (int i = 0; i < nseq; ++i)
        ARVAE_REQUIRE(seqs[i].w_hh && seqs[i].h_all && seqs[i].saved && seqs[i].dgi && seqs[i].dgh, "gru_seq_bwd: null pointer");
    GruSeqBatch b{};
    fill_batch(&b, seqs, nseq, hidden, masks);
    // (the recurrence addresses its arrays with 32-bit byte offsets: gru_rsrc)
    const int64_t span = (int64_t)steps * rows * 4;
    for (int i = 0; i < nseq; ++i) {
        const GruSeq &q = b.seq[i];
        const int64_t widest = std::max<int64_t>({q.dh_stride, q.h_stride, q.h0_stride, q.dgi_rstride, 4 * (int64_t)hidden});
        ARVAE_REQUIRE(span * widest < GRU_RANGE, "gru_seq_bwd: %d steps x %d rows of %lld floats do not fit 2 GB per array", steps, rows,
                      (long long)widest);
    }
    hipStream_t st = as_stream(stream);
    const dim3 grid((rows + 15) / 16, nseq);
    if (gru_fp32_mfma()) {
        if (hidden == 128) ARVAE_LAUNCH(gru_seq_bwd_kernel<128>, grid, dim3(512), 0, st, b, steps, rows);
        else if (hidden == 64) ARVAE_LAUNCH(gru_seq_bwd_kernel<64>, grid, dim3(256), 0, st, b, steps, rows);
        else ARVAE_LAUNCH(gru_seq_bwd_kernel<32>, grid, dim3(128), 0, st, b, steps, rows);
    } else if (gru_bf16_backward()) {
        const int rw = gru_rows_per_wg(rows, nseq);
        GRU_LAUNCH(gru_seq_bwd_x3_kernel, rw, b, steps, rows)
    } else {
        const int rw = gru_rows_per_wg(rows, nseq);
        GRU_LAUNCH(gru_seq_bwd_h2_kernel, rw, b, steps, rows)
    }
    return check_launch("gru_seq_bwd_kernel");
}

extern "C" int64_t arvae_tick_free_run_ws_floats(int32_t hidden) {
    // three matrices [3H][H] as three bf16 terms each (tick_weight_prep_kernel)
    return arvae_gru_seq_supported(hidden) ? (int64_t)3 * 3 * hidden * hidden * 3 / 2 : 0;
}

extern "C" int arvae_tick_free_run_supported(int32_t hidden, int32_t vocab) {
    return arvae_gru_seq_supported(hidden) && vocab >= 1 && vocab <= 64 && vocab <= 16 * (hidden / 16);
}

extern "C" int arvae_tick_free_run(const arvae_tick_weights_t *wts, const float *h0_l0, const float *h0_l1, int64_t h0_stride, const float *gib,
                                   const float *ptab, const uint8_t *mask, float keep_scale, int32_t batch, int32_t beats,
                                   int32_t ticks_per_beat, int32_t hidden, int32_t vocab, int64_t *tokens, float *ws,
                                   arvae_stream_t stream) {
    ARVAE_REQUIRE(wts && h0_l0 && h0_l1 && gib && ptab && tokens, "tick_free_run: null pointer");
    ARVAE_REQUIRE(wts->w_hh0 && wts->b_hh0 && wts->w_ih1 && wts->b_ih1 && wts->w_hh1 && wts->b_hh1 && wts->w_out && wts->b_out,
                  "tick_free_run: null weight pointer");
    ARVAE_REQUIRE(batch >= 1 && beats >= 1 && ticks_per_beat >= 1, "tick_free_run: empty problem");
    ARVAE_REQUIRE(arvae_gru_seq_supported(hidden), "tick_free_run: hidden size %d is not built (32, 64, 128)", hidden);
    ARVAE_REQUIRE(arvae_tick_free_run_supported(hidden, vocab), "tick_free_run: vocabulary of %d notes not supported at hidden size %d",
                  vocab, hidden);
    TickFreeRun p{};
    p.w_hh0 = wts->w_hh0; p.b_hh0 = wts->b_hh0; p.w_ih1 = wts->w_ih1; p.b_ih1 = wts->b_ih1;
    p.w_hh1 = wts->w_hh1; p.b_hh1 = wts->b_hh1; p.w_out = wts->w_out; p.b_out = wts->b_out;
    p.h0_l0 = h0_l0; p.h0_l1 = h0_l1; p.h0_stride = h0_stride != 0 ? h0_stride : hidden; p.gib = gib; p.ptab = ptab; p.mask = mask; p.keep_scale = keep_scale;
    p.batch = batch; p.beats = beats; p.tpb = ticks_per_beat; p.vocab = vocab; p.tokens = tokens;
    hipStream_t st = as_stream(stream);
    const dim3 grid((batch + 15) / 16);
    const bool m = mask != nullptr;
    if (gru_fp32_mfma() || ws == nullptr) {
        if (hidden == 128 && m) ARVAE_LAUNCH((tick_free_run_kernel<128, 2, true>), grid, dim3(512), 0, st, p);
        else if (hidden == 128) ARVAE_LAUNCH((tick_free_run_kernel<128, 2, false>), grid, dim3(512), 0, st, p);
        else if (hidden == 64 && m) ARVAE_LAUNCH((tick_free_run_kernel<64, 2, true>), grid, dim3(256), 0, st, p);
        else if (hidden == 64) ARVAE_LAUNCH((tick_free_run_kernel<64, 2, false>), grid, dim3(256), 0, st, p);
        else if (m) ARVAE_LAUNCH((tick_free_run_kernel<32, 2, true>), grid, dim3(128), 0, st, p);
        else ARVAE_LAUNCH((tick_free_run_kernel<32, 2, false>), grid, dim3(128), 0, st, p);
        return check_launch("tick_free_run_kernel");
    }
    ARVAE_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15) == 0, "tick_free_run: workspace must be 16-byte aligned");
    TickPrep tp{{wts->w_hh0, wts->w_ih1, wts->w_hh1}, reinterpret_cast<uint4 *>(ws)};
    const int items = 3 * (hidden / 32) * (hidden / 16) * 3 * 64;
    const uint4 *packed = reinterpret_cast<const uint4 *>(ws);
    if (!gru_bf16_forward()) {
        const int rw = gru_rows_per_wg(batch, 1);
        const dim3 gr((batch + rw - 1) / rw);
#define TICK_H2_RW(HH, MM, RWV)                                                                                                  \
        {                                                                                                                        \
            if ((RWV) == 4) ARVAE_LAUNCH((tick_free_run_h2_kernel<HH, MM, 4>), gr, dim3(4 * HH), 0, st, p, packed);               \
            else if ((RWV) == 8) ARVAE_LAUNCH((tick_free_run_h2_kernel<HH, MM, 8>), gr, dim3(4 * HH), 0, st, p, packed);          \
            else ARVAE_LAUNCH((tick_free_run_h2_kernel<HH, MM, 16>), gr, dim3(4 * HH), 0, st, p, packed);                         \
        }
#define TICK_H2(HH)                                                                                                              \
        {                                                                                                                        \
            float *wmax_ = ws + 9 * HH * HH;      /* (behind the two-term layout: the workspace is sized for three terms) */ \
            ARVAE_LAUNCH(tick_weight_amax_kernel<HH>, dim3(3), dim3(1024), 0, st, tp, wmax_);                                     \
            ARVAE_LAUNCH(tick_weight_prep_h2_kernel<HH>, dim3((items + 255) / 256), dim3(256), 0, st, tp, wmax_);                 \
            if (m) TICK_H2_RW(HH, true, rw)                                                                                      \
            else TICK_H2_RW(HH, false, rw)                                                                                       \
        }
        if (hidden == 128) TICK_H2(128)
        else if (hidden == 64) TICK_H2(64)
        else TICK_H2(32)
#undef TICK_H2
#undef TICK_H2_RW
        return check_launch("tick_free_run_h2_kernel");
    }
    if (hidden == 128) {
        ARVAE_LAUNCH(tick_weight_prep_kernel<128>, dim3((items + 255) / 256), dim3(256), 0, st, tp);
        if (m) ARVAE_LAUNCH((tick_free_run_x3_kernel<128, true>), grid, dim3(512), 0, st, p, packed);
        else ARVAE_LAUNCH((tick_free_run_x3_kernel<128, false>), grid, dim3(512), 0, st, p, packed);
    } else if (hidden == 64) {
        ARVAE_LAUNCH(tick_weight_prep_kernel<64>, dim3((items + 255) / 256), dim3(256), 0, st, tp);
        if (m) ARVAE_LAUNCH((tick_free_run_x3_kernel<64, true>), grid, dim3(256), 0, st, p, packed);
        else ARVAE_LAUNCH((tick_free_run_x3_kernel<64, false>), grid, dim3(256), 0, st, p, packed);
    } else {
        ARVAE_LAUNCH(tick_weight_prep_kernel<32>, dim3((items + 255) / 256), dim3(256), 0, st, tp);
        if (m) ARVAE_LAUNCH((tick_free_run_x3_kernel<32, true>), grid, dim3(128), 0, st, p, packed);
        else ARVAE_LAUNCH((tick_free_run_x3_kernel<32, false>), grid, dim3(128), 0, st, p, packed);
    }
    return check_launch("tick_free_run_x3_kernel");
}

#ifdef ARVAE_GRU_STAMPS
extern "C" int arvae_debug_tick_stamps(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(arvae::g_tick_stamps), sizeof(unsigned long long) * 9);
}
extern "C" int arvae_debug_gru_stamps(unsigned long long *out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(arvae::g_gru_stamps), sizeof(unsigned long long) * 8);
}
#endif
