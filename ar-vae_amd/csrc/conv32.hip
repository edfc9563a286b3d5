// Specialised gfx950 kernels for the dSprites stack's 32-channel links (kernel 4x4, stride 2, pad 1):
// the six Conv2d / ConvTranspose2d layers between the 32x32, 16x16, 8x8 and 4x4 feature maps, which
// carry 85 % of the training step's FLOPs.  Same math and C-ABI as the
// generic gather-GEMM (link_gemm.hip); arvae_link_down/up/wgrad dispatch here when the geometry fits.
//
// The kernels run the bf16 MFMA at fp32 accuracy: every fp32 operand is split into three bf16 terms when it enters LDS
// (weights: once per step, conv32_weight_prep) and every multiply-add is six partial products on v_mfma_f32_32x32x16_bf16
// with fp32 accumulation: the result is within one fp32 rounding of the fp32 MFMA's at 2.7x fewer MFMA cycles, and the bf16
// MFMA leaves the vector ALU free for the splitting.  (The first generation -- the same tilings on v_mfma_f32_32x32x2_f32,
// 45-50 us per 16x16-layer launch against 29-33 -- and a two-term experiment were removed in round 3; the K-split small-tile
// Down kernel of the 4x4 layers, down32s, is the one fp32-MFMA kernel left.)
//
// Design rules (measured, profiles/r1_down32_phase_stamps.txt, tools/stamp_conv32.py): on gfx950 the fp32 MFMA
// (64 cycles each, runs at the fp32 vector rate) does NOT co-execute with VALU work of another wave on the same
// SIMD, so SIMD time = MFMA cycles + VALU cycles; a CU moves store data at ~16 B/clk.  Hence:
//   * one 256-thread workgroup per CU (one wave per SIMD), persistent over tiles of 128 / 64 / 32 lo pixels
//     (full-width row blocks / whole images, so a tile is one contiguous span of the lo tensor);
//   * every operand read from LDS is one 16-byte read that feeds several MFMAs;
//   * weights live in registers for the whole kernel (Wgrad: none, pixels are the K axis);
//   * global traffic uses raw buffer loads / stores: the hardware bounds check zero-fills everything outside
//     the tensor (images past the end of the batch, rows before its start), per-thread offsets are computed
//     once per kernel, per tile a slot costs an add, a compare and a select;
//   * a tile's patch (+halo) is prefetched into registers while the previous tile's MFMAs run and is
//     committed to LDS between two barriers (pixel strides 36 floats / 20 dwords per bf16 plane: conflict-free
//     16-byte reads for stride-1/2 pixel walks);
//   * epilogues are 16-byte stores of a lane's consecutive channels, in the Up kernels spread between the next
//     tile's MFMAs with the waves staggered.
#include "common.h"
#include "conv32_common.h"
#include "reduce.h"
#include "midprep.h"
#include "prep32.h"
#include "regloss.h"

#include <type_traits>

namespace arvae {

// ------------------------------------------------------------------------------------------------
// Register-staged patch loader for a [n_img, SZ, SZ, 32] fp32 tensor.
//   STRIDE 2: hi patch of a lo tile: rows [2*r0-1, 2*r0-1+PR), cols [-1, PC-1), PR = 2*TR+2, PC = 2*TC+2
//   STRIDE 1: lo patch with a 1-pixel halo: rows [r0-1, r0-1+PR), cols [-1, PC-1), PR = TR+2, PC = TC+2
// Plain operands only (no fused act'(y) factor: the callers route such gradients to the generic kernel).
// Per thread and slot: the byte offset relative to the tile's first patch row (OOB for the column halo and
// padding slots) and the patch row (the row halo at image borders is a compare + select per tile).
// Slots are issued one at a time so that the callers can spread them between the MFMAs of the previous tile.
// ------------------------------------------------------------------------------------------------
template <int LO, int STRIDE, int PX = 128>
struct PatchLoader {
    using T = Tile<LO, PX>;
    static constexpr int SZ = STRIDE * LO;
    static constexpr int PR = STRIDE * T::TR + 2, PC = STRIDE * T::TC + 2;
    static constexpr int SLOTS = T::TI * PR * PC * 8;
    static constexpr int ITERS = (SLOTS + 255) / 256;
    static constexpr int PATCH_FLOATS = T::TI * PR * PC * PS;
    float4 r[ITERS];
    unsigned rel[ITERS];     // byte offset relative to the tile's first patch row (multiple of 16) | bit 0: first patch row,
                             // bit 1: last patch row -- the only rows that can fall outside the image (row halo)
    __amdgpu_buffer_rsrc_t rs_v;
    bool valid;
    unsigned badrows;        // per tile: bit 0 / 1 set when the first / last patch row is outside the image
    int base;

    __device__ __forceinline__ void init(const float *src, int n_img) {
        rs_v = make_rsrc(src, (int64_t)n_img * SZ * SZ * PIXB);
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int idx = threadIdx.x + it * 256;
            const int q = idx & 7, pix = idx >> 3;
            const int pc = pix % PC, pr = (pix / PC) % PR, im = pix / (PC * PR);
            const int gx = pc - 1;
            const bool ok = idx < SLOTS && (unsigned)gx < (unsigned)SZ;
            // image im's rows follow image 0's SZ rows later; patch row pr is tensor row gy0 + pr of its image
            rel[it] = ok ? (unsigned)(((im * SZ + pr) * SZ + gx) * PIXB + q * 16) | (pr == 0 ? 1u : 0u) | (pr == PR - 1 ? 2u : 0u)
                         : OOB;
        }
    }
    // next tile to fetch: first image img0, first lo row r0; !ok -> every slot reads zeros without touching memory
    __device__ __forceinline__ void set_tile(int img0, int r0, bool ok) {
        const int gy0 = STRIDE * r0 - 1;
        base = ((img0 * SZ + gy0) * SZ) * PIXB;                  // negative for the very first patch row of the tensor
        badrows = (gy0 < 0 ? 1u : 0u) | (gy0 + PR - 1 >= SZ ? 2u : 0u);
        valid = ok;
    }
    __device__ __forceinline__ void issue_slot(int it) {
        const bool row_ok = valid && (rel[it] & badrows) == 0;
        // OOB + base stays out of range (tensors are < 2^31 - 2^20 bytes, |negative base| < 2^20)
        const unsigned off = row_ok ? (rel[it] & ~3u) + (unsigned)base : OOB;
        r[it] = buf_load4(rs_v, off);
    }
    // the slots that belong to step `step` of `steps` evenly spaced issue points (compile-time after unrolling)
    template <int STEPS, int STEP> __device__ __forceinline__ void issue_step() {
        static_for<0, ITERS>([&](auto ic) __attribute__((always_inline)) {
            constexpr int it = decltype(ic)::value;
            if constexpr (it * STEPS / ITERS == STEP) issue_slot(it);
        });
    }
    __device__ __forceinline__ void issue_all() {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) issue_slot(it);
    }
    // Split commit for the bf16 MFMA kernels: every fp32 value x becomes hi = bf16(x) and mid = bf16(x - hi) (split_pair);
    // two planes of PLANE_DW dwords, a pixel is PSB dwords per plane (32 channels x 2 bytes + pad), channel pair
    // (2i, 2i+1) shares a dword, even channel in the low half
    static constexpr int PLANE_DW = T::TI * PR * PC * PSB;
    __device__ __forceinline__ void commit_split(unsigned *planes) const {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int idx = threadIdx.x + it * 256;
            if (idx < SLOTS) {
                const int q = idx & 7, pix = idx >> 3;
                uint2 hv, mv;
                split_pair(r[it].x, r[it].y, hv.x, mv.x);
                split_pair(r[it].z, r[it].w, hv.y, mv.y);
                *reinterpret_cast<uint2 *>(planes + pix * PSB + q * 2) = hv;
                *reinterpret_cast<uint2 *>(planes + PLANE_DW + pix * PSB + q * 2) = mv;
            }
        }
    }
    // three planes (hi, mid, lo): the exact split used by the fp32-accurate bf16 kernels; BIAS_SUM as in commit();
    // PITCH = dwords per pixel and plane
    template <bool BIAS_SUM = false, int PITCH = PSB>
    __device__ __forceinline__ void commit_split3(unsigned *planes, float4 *bsum = nullptr) const {
        constexpr int PLANE_P = T::TI * PR * PC * PITCH;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int idx = threadIdx.x + it * 256;
            // every lane "uses" the slot's registers, also those past SLOTS: a load whose only use sits under a branch is never
            // waited for on the path around it, and the compiler then drains the whole queue (vmcnt(0): every store of the
            // previous tile's epilogue as well) before it reuses the register
            float4 v = r[it];
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
            if (idx < SLOTS) {
                const int q = idx & 7, pix = idx >> 3;
                uint2 hv, mv, lv;
                split_pair3(v.x, v.y, hv.x, mv.x, lv.x);
                split_pair3(v.z, v.w, hv.y, mv.y, lv.y);
                *reinterpret_cast<uint2 *>(planes + pix * PITCH + q * 2) = hv;
                *reinterpret_cast<uint2 *>(planes + PLANE_P + pix * PITCH + q * 2) = mv;
                *reinterpret_cast<uint2 *>(planes + 2 * PLANE_P + pix * PITCH + q * 2) = lv;
                if (BIAS_SUM) {
                    const int pc = pix % PC, pr = (pix / PC) % PR;
                    if (pr >= 1 && pr <= PR - 2 && pc >= 1 && pc <= PC - 2) {
                        bsum->x += v.x; bsum->y += v.y; bsum->z += v.z; bsum->w += v.w;
                    }
                }
            }
        }
    }
    // packed three-term image (PSB3): one slot, all slots, or the slots of one of STEPS issue points followed by the
    // load of the same slots for the next tile (the registers are free again once their values are in LDS)
    static constexpr int BUF3_DW = T::TI * PR * PC * PSB3;
    __device__ __forceinline__ void commit_slot3p(unsigned *buf, int it) const {
        const int idx = threadIdx.x + it * 256;
        if (idx < SLOTS) {
            const int q = idx & 7, pix = idx >> 3;
            uint2 hv, mv, lv;
            split_pair3(r[it].x, r[it].y, hv.x, mv.x, lv.x);
            split_pair3(r[it].z, r[it].w, hv.y, mv.y, lv.y);
            unsigned *d = buf + pix * PSB3 + q * 2;
            *reinterpret_cast<uint2 *>(d) = hv;
            *reinterpret_cast<uint2 *>(d + 16) = mv;
            *reinterpret_cast<uint2 *>(d + 32) = lv;
        }
    }
    __device__ __forceinline__ void commit_all3p(unsigned *buf) const {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) commit_slot3p(buf, it);
    }
    template <int STEPS, int STEP> __device__ __forceinline__ void commit_issue_step3p(unsigned *buf) {
        static_for<0, ITERS>([&](auto ic) __attribute__((always_inline)) {
            constexpr int it = decltype(ic)::value;
            if constexpr (it * STEPS / ITERS == STEP) {
                commit_slot3p(buf, it);
                issue_slot(it);
            }
        });
    }
    // BIAS_SUM: also accumulate the pixels this tile owns (not the halo) per channel chunk q = threadIdx.x & 7
    template <bool BIAS_SUM>
    __device__ __forceinline__ void commit(float *patch, float4 &bsum) const {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int idx = threadIdx.x + it * 256;
            if (idx < SLOTS) {
                const int q = idx & 7, pix = idx >> 3;
                *reinterpret_cast<float4 *>(patch + pix * PS + q * 4) = r[it];
                if (BIAS_SUM) {
                    const int pc = pix % PC, pr = (pix / PC) % PR;
                    if (pr >= 1 && pr <= PR - 2 && pc >= 1 && pc <= PC - 2) {
                        bsum.x += r[it].x; bsum.y += r[it].y; bsum.z += r[it].z; bsum.w += r[it].w;
                    }
                }
            }
        }
    }
};

#ifdef ARVAE_STAMPS
// diagnostic build only (tools/stamp_conv32.py): per-workgroup phase timeline, 64 slots, s_memtime + wall clock
__device__ unsigned long long g_stamps[512 * 64 * 2];
#define STAMP(slot)                                                                      \
    do {                                                                                 \
        if (LO == 16 && threadIdx.x == 0 && blockIdx.x < 512 && (slot) < 64) {           \
            g_stamps[(blockIdx.x * 64 + (slot)) * 2] = __builtin_readcyclecounter();     \
            g_stamps[(blockIdx.x * 64 + (slot)) * 2 + 1] = wall_clock64();               \
        }                                                                                \
    } while (0)
#define STAMP_WAIT() __builtin_amdgcn_s_waitcnt(0)
// up32p_kernel: row blockIdx.x = its consumers (thread 0), row 256 + blockIdx.x = its producers (thread 256)
#define PSTAMP(role, slot)                                                               \
    do {                                                                                 \
        if (threadIdx.x == 256 * (role) && blockIdx.x < 256 && (slot) < 64) {            \
            g_stamps[((blockIdx.x + 256 * (role)) * 64 + (slot)) * 2] = __builtin_readcyclecounter();     \
            g_stamps[((blockIdx.x + 256 * (role)) * 64 + (slot)) * 2 + 1] = wall_clock64();               \
        }                                                                                \
    } while (0)
#else
#define STAMP(slot)
#define STAMP_WAIT()
#define PSTAMP(role, slot)
#endif

// The weight value is the MFMA's A operand (row = output channel = lane & 31) and the pixel value its B operand
// (column = pixel = lane & 31), so a lane ends up with 4 x 4 consecutive output channels of ONE pixel:
// accumulator register reg holds channel CH0(reg) + 4 * (lane >> 5) + (reg & 3), and the epilogue is four
// 16-byte stores per 32x32 tile (dword stores run at a quarter of that rate).
#define MFMA4(ACC, A, W)                                                                \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((W)[0], (A).x, ACC, 0, 0, 0);            \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((W)[1], (A).y, ACC, 0, 0, 0);            \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((W)[2], (A).z, ACC, 0, 0, 0);            \
    ACC = __builtin_amdgcn_mfma_f32_32x32x2f32((W)[3], (A).w, ACC, 0, 0, 0);

// one channel group g (4 channels) of one pixel: bias -> ReLU / gate -> one 16-byte store at off + g*32;
// returns the group's four sign bits (for EP_RELU's bits_out)
template <int MODE>
__device__ __forceinline__ unsigned store_group(const f32x16 &acc, int g, const float4 &b, const float4 &gv, unsigned gbits,
                                                __amdgpu_buffer_rsrc_t rs_out, unsigned off) {
    float v[4] = {acc[4 * g] + b.x, acc[4 * g + 1] + b.y, acc[4 * g + 2] + b.z, acc[4 * g + 3] + b.w};
    const float gf[4] = {gv.x, gv.y, gv.z, gv.w};
    unsigned bits = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (MODE == EP_RELU) {
            v[j] = fmaxf(v[j], 0.f);
            bits |= (v[j] > 0.f ? 1u : 0u) << (4 * g + j);
        }
        if (MODE == EP_GATE_F) v[j] = gf[j] > 0.f ? v[j] : 0.f;
        if (MODE == EP_GATE_B) v[j] = ((gbits >> (4 * g + j)) & 1u) ? v[j] : 0.f;
    }
    buf_store4(make_float4(v[0], v[1], v[2], v[3]), rs_out, off + g * 32);
    return bits;
}

// all four groups of one pixel, gate fetched here (the non-pipelined epilogues)
template <int MODE>
__device__ __forceinline__ void store_pixel(const f32x16 &acc, const float4 (&b4)[4], __amdgpu_buffer_rsrc_t rs_out,
                                            __amdgpu_buffer_rsrc_t rs_gate, __amdgpu_buffer_rsrc_t rs_bits, bool want_bits,
                                            unsigned off, int half) {
    float4 gv[4];
    unsigned gb = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) gv[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (MODE == EP_GATE_F) {
#pragma unroll
        for (int g = 0; g < 4; ++g) gv[g] = buf_load4(rs_gate, off + g * 32);
    }
    if (MODE == EP_GATE_B) gb = buf_load_u16(rs_bits, bits_off(off, half));
    unsigned bits = 0;
#pragma unroll
    for (int g = 0; g < 4; ++g) bits |= store_group<MODE>(acc, g, b4[g], gv[g], gb, rs_out, off);
    if (MODE == EP_RELU && want_bits) buf_store_u16(bits, rs_bits, bits_off(off, half));
}

__device__ __forceinline__ void load_bias4(const float *bias, int half, float4 (&b4)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
        b4[g] = bias != nullptr ? *reinterpret_cast<const float4 *>(bias + 8 * g + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// the 64 KB weight tensor wt[32][32][4][4] as 4096 coalesced 16-byte loads of the workgroup (16 per thread)
__device__ __forceinline__ void load_weights(const float *wt, float4 (&v)[16]) {
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(wt, 16 * C32 * C32 * 4);
#pragma unroll
    for (int it = 0; it < 16; ++it) v[it] = buf_load4(rs_w, (threadIdx.x + it * 256) * 16);
}
constexpr int WROW_DOWN = 16 * C32 + 4;          // LDS floats per clo row (bank-conflict-free 16-byte reads)
constexpr int WROW_UP = 17;                      // LDS floats per (clo, chi) row (conflict-free dword reads)
constexpr int WSTAGE_DOWN = C32 * WROW_DOWN;     // floats of LDS needed while staging
constexpr int WSTAGE_UP = C32 * C32 * WROW_UP;



// ================================================================================================
// Down on the bf16 MFMA at fp32 accuracy (three-term split, six partial products: see up32x_kernel).  Three terms of a
// full output column (256 weights) do not fit the register file, so K is split two ways here: a tile is 64 lo pixels,
// wave w takes pixel half w >> 1 and kernel rows ky = 2 (w & 1), 2 (w & 1) + 1 (128 weights x 3 terms = 192 registers);
// the odd wave hands its partial tile to the even one through LDS, which runs the epilogue.
// Per-layer prepared weights (conv32_weight_prep, once per training step): the three-term split of wt in the exact
// per-lane register order of down32x_kernel and up32x_kernel, 16 bytes per (slot, lane) with lanes contiguous, so
// that a kernel starts with 48 / 24 coalesced loads instead of staging and splitting the tensor itself.
//   DOWN part: [kh 2][slot 48 = (tap 8, c 2, term 3)][lane 64]      UP part: [class 4][slot 24 = (ty, tx, c, term)][lane 64]
// (PREP_DOWN_SLOTS, PREP_UP_SLOTS, PREP_*_UINT4, PREP_FLOATS: conv32_common.h, shared with conv32k.hip)
__device__ __forceinline__ bf16x8 lds_bf16x8(const unsigned *p) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const i32x4v *>(p));
}

__global__ __launch_bounds__(256) void conv32_weight_prep_kernel(PrepArgs p) { conv32_prep_block(p, blockIdx.x); }

// the step's two weight preps as one launch: workgroups [0, conv_blocks) split the 32-channel conv weights, the rest lay out the
// latent block's matrices (midprep.h)
__global__ __launch_bounds__(256) void prep_all_kernel(PrepArgs p, MidPrepArgs mid, int conv_blocks) {
    prep_all_block(p, mid, conv_blocks, blockIdx.x);
}

template <int LO, int MODE>
__global__ __launch_bounds__(256, 1) void down32x_kernel(const float *__restrict__ hi, const float *__restrict__ wt, Ep32 ep,
                                                         int n_img, int n_tiles) {
    constexpr int PX = 64;
    using PL = PatchLoader<LO, 2, PX>;
    constexpr int PC = PL::PC, PR = PL::PR, BUF = PL::BUF3_DW;
    // two packed three-term images (the next tile is split and written while this one is multiplied) | red[2][2][16][64]
    extern __shared__ __attribute__((aligned(16))) float lds[];
    unsigned *ldsw = reinterpret_cast<unsigned *>(lds);
    float *red = lds + 2 * BUF;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, rc = lane & 31;
    const int mtile = wave >> 1, kh = wave & 1;

    PL pl;                                                       // first tile's loads fly while the weights are staged
    pl.init(hi, n_img);
    int img0, r0;
    tile_origin<LO, PX>(blockIdx.x, img0, r0);
    pl.set_tile(img0, r0, blockIdx.x < n_tiles);
    pl.issue_all();

    // w3[tap][c][term], tap = kyl*4 + kx with ky = 2*kh + kyl: channels c*16 + half*8 + j of wt[clo = rc][.][ky][kx]
    bf16x8 w3[8][2][3];
    if (ep.wprep != nullptr) {                                   // split once per step by conv32_weight_prep: 48 coalesced loads
        const uint4 *src = ep.wprep + (kh * PREP_DOWN_SLOTS) * 64 + lane;
#pragma unroll
        for (int tap = 0; tap < 8; ++tap)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int t = 0; t < 3; ++t) w3[tap][cc][t] = __builtin_bit_cast(bf16x8, src[((tap * 2 + cc) * 3 + t) * 64]);
    } else {
        float4 v[16];
        load_weights(wt, v);
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int idx4 = threadIdx.x + it * 256;             // (clo, chi, tap/4) = (idx4 >> 7, (idx4 >> 2) & 31, idx4 & 3)
            *reinterpret_cast<float4 *>(lds + (idx4 >> 7) * WROW_DOWN + (idx4 & 127) * 4) = v[it];
        }
        __syncthreads();
        static_for<0, 4>([&](auto gc) __attribute__((always_inline)) {
            constexpr int c = decltype(gc)::value >> 1, kyl = decltype(gc)::value & 1;
            float x[4][8];                                       // [kx][j]
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 q = *reinterpret_cast<const float4 *>(lds + rc * WROW_DOWN + (c * 16 + half * 8 + j) * 16 +
                                                                     (2 * kh + kyl) * 4);
                x[0][j] = q.x; x[1][j] = q.y; x[2][j] = q.z; x[3][j] = q.w;
            }
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) split8x3(x[kx], w3[kyl * 4 + kx][c][0], w3[kyl * 4 + kx][c][1], w3[kyl * 4 + kx][c][2]);
        });
    }

    int img, r, c;
    tile_pixel<LO, PX>(mtile * 32 + rc, img, r, c);
    const int aoff = ((img * PR + 2 * r + 2 * kh) * PC + 2 * c) * PSB3 + half * 4;    // dwords; + (kyl*PC + kx)*PSB3 + c*8 + term*16
    float4 b4[4];
    load_bias4(ep.bias, half, b4);
    const int64_t out_bytes = (int64_t)n_img * LO * LO * PIXB;
    const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(ep.out, out_bytes);
    const __amdgpu_buffer_rsrc_t rs_gate = make_rsrc(MODE == EP_GATE_F ? ep.gate : ep.out, out_bytes);
    const bool want_bits = MODE == EP_RELU && ep.bits_out != nullptr;
    const __amdgpu_buffer_rsrc_t rs_bits =
        make_rsrc(MODE == EP_GATE_B ? (const void *)ep.gate_bits : want_bits ? (const void *)ep.bits_out : (const void *)ep.out,
                  (int64_t)n_img * LO * LO * 4);
    const unsigned out_lane = (unsigned)((mtile * 32 + rc) * PIXB + half * 16);

    // pipeline: registers hold tile t+1 (loaded during tile t-1); during tile t's MFMAs each slot is split, written to
    // the other LDS image and refilled with tile t+2.  One barrier per tile.
    __syncthreads();                                             // weight staging (if any) no longer needs the LDS
    pl.commit_all3p(ldsw);                                       // first tile -> image 0
    {
        int ni, nr;
        tile_origin<LO, PX>(blockIdx.x + gridDim.x, ni, nr);
        pl.set_tile(ni, nr, blockIdx.x + gridDim.x < n_tiles);
        pl.issue_all();
    }
    __syncthreads();
    int cur = 0;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, cur ^= 1) {
        tile_origin<LO, PX>(tile, img0, r0);
        {
            int ni, nr;
            tile_origin<LO, PX>(tile + 2 * gridDim.x, ni, nr);
            pl.set_tile(ni, nr, tile + 2 * gridDim.x < n_tiles);
        }
        const unsigned *img_cur = ldsw + cur * BUF;
        unsigned *img_nxt = ldsw + (cur ^ 1) * BUF;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        bf16x8 a[2][2][3];                                       // [tap parity][c][term]
#pragma unroll
        for (int cc = 0; cc < 2; ++cc)
#pragma unroll
            for (int t = 0; t < 3; ++t) a[0][cc][t] = lds_bf16x8(img_cur + aoff + cc * 8 + t * 16);
        static_for<0, 8>([&](auto tc) __attribute__((always_inline)) {
            constexpr int tap = decltype(tc)::value;
            if constexpr (tap + 1 < 8) {                         // next tap's operands are in flight during these 12 MFMAs
                constexpr int kyl = (tap + 1) >> 2, kx = (tap + 1) & 3;
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                    for (int t = 0; t < 3; ++t)
                        a[(tap + 1) & 1][cc][t] = lds_bf16x8(img_cur + aoff + (kyl * PC + kx) * PSB3 + cc * 8 + t * 16);
            }
            pl.template commit_issue_step3p<8, tap>(img_nxt);
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {                     // smallest partial products first
                MFMA_B(acc, w3[tap][cc][2], a[tap & 1][cc][0]);
                MFMA_B(acc, w3[tap][cc][0], a[tap & 1][cc][2]);
                MFMA_B(acc, w3[tap][cc][1], a[tap & 1][cc][1]);
                MFMA_B(acc, w3[tap][cc][1], a[tap & 1][cc][0]);
                MFMA_B(acc, w3[tap][cc][0], a[tap & 1][cc][1]);
                MFMA_B(acc, w3[tap][cc][0], a[tap & 1][cc][0]);
            }
        });
        float *rd = red + cur * (2 * 16 * 64);
        if (kh == 1) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) rd[(mtile * 16 + reg) * 64 + lane] = acc[reg];
        }
        __syncthreads();                                         // partial tiles and the next image are written; this image is free
        if (kh == 0) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) acc[reg] += rd[(mtile * 16 + reg) * 64 + lane];
            store_pixel<MODE>(acc, b4, rs_out, rs_gate, rs_bits, want_bits,
                              out_lane + (unsigned)(((img0 * LO + r0) * LO) * PIXB), half);
        }
    }
}

// ================================================================================================
// Down, small-problem variant (4x4 / 8x8 layers at batch 512: too few 128-pixel tiles to fill or pipeline the CUs):
// a tile is 32 lo pixels, wave w takes kernel row ky = w (K split 4 ways, 64 weights per lane straight from
// global memory), the four partial tiles meet in LDS and wave 0 runs the epilogue.
template <int LO, int MODE>
__device__ __forceinline__ void down32s_body(const float *__restrict__ hi, const float *__restrict__ wt, Ep32 ep, int n_img, int n_tiles, const int BID, const int NBLK) {
    constexpr int PX = 32;
    using PL = PatchLoader<LO, 2, PX>;
    constexpr int PC = PL::PC, PR = PL::PR;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // PL::PATCH_FLOATS | red[4][16][64]
    float *red = lds + PL::PATCH_FLOATS;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, rc = lane & 31;

    PL pl;
    pl.init(hi, n_img);
    int img0, r0;
    tile_origin<LO, PX>(BID, img0, r0);
    pl.set_tile(img0, r0, BID < n_tiles);
    pl.issue_all();

    // w[kx][chunk][t] = wt[clo = rc][chi = chunk*8 + half*4 + t][ky = wave][kx]: the four kx are one 16-byte load
    float w[4][4][4];
#pragma unroll
    for (int ch = 0; ch < 4; ++ch)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float4 q = *reinterpret_cast<const float4 *>(wt + ((rc * C32) + ch * 8 + half * 4 + t) * 16 + wave * 4);
            w[0][ch][t] = q.x; w[1][ch][t] = q.y; w[2][ch][t] = q.z; w[3][ch][t] = q.w;
        }
    int img, r, c;
    tile_pixel<LO, PX>(rc, img, r, c);
    const int aoff = ((img * PR + 2 * r + wave) * PC + 2 * c) * PS + half * 4;      // + kx*PS + chunk*8
    float4 b4[4];
    load_bias4(ep.bias, half, b4);
    const int64_t out_bytes = (int64_t)n_img * LO * LO * PIXB;
    const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(ep.out, out_bytes);
    const __amdgpu_buffer_rsrc_t rs_gate = make_rsrc(MODE == EP_GATE_F ? ep.gate : ep.out, out_bytes);
    const bool want_bits = MODE == EP_RELU && ep.bits_out != nullptr;
    const __amdgpu_buffer_rsrc_t rs_bits =
        make_rsrc(MODE == EP_GATE_B ? (const void *)ep.gate_bits : want_bits ? (const void *)ep.bits_out : (const void *)ep.out,
                  (int64_t)n_img * LO * LO * 4);
    const unsigned out_lane = (unsigned)(rc * PIXB + half * 16);
    float4 dummy;

    for (int tile = BID; tile < n_tiles; tile += NBLK) {
        tile_origin<LO, PX>(tile, img0, r0);
        __syncthreads();                                         // the previous tile's patch and partials have been read
        pl.template commit<false>(lds, dummy);
        __syncthreads();
        {
            int ni, nr;
            tile_origin<LO, PX>(tile + NBLK, ni, nr);
            pl.set_tile(ni, nr, tile + NBLK < n_tiles);
            pl.issue_all();
        }
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int kx = 0; kx < 4; ++kx)
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                const float4 a = *reinterpret_cast<const float4 *>(lds + aoff + kx * PS + ch * 8);
                MFMA4(acc, a, w[kx][ch])
            }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) red[(wave * 16 + reg) * 64 + lane] = acc[reg];
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                acc[reg] = (acc[reg] + red[(16 + reg) * 64 + lane]) + (red[(32 + reg) * 64 + lane] + red[(48 + reg) * 64 + lane]);
            store_pixel<MODE>(acc, b4, rs_out, rs_gate, rs_bits, want_bits,
                              out_lane + (unsigned)(((img0 * LO + r0) * LO) * PIXB), half);
        }
    }
}
template <int LO, int MODE>
__global__ __launch_bounds__(256, 2) void down32s_kernel(const float *__restrict__ hi, const float *__restrict__ wt, Ep32 ep, int n_img, int n_tiles) {
    down32s_body<LO, MODE>(hi, wt, ep, n_img, n_tiles, blockIdx.x, gridDim.x);
}


// ================================================================================================
// Up: hi[n,hy,hx,chi] = ep( sum over the 2x2 taps valid for (hy,hx)'s parity and clo of lo * wt )
// wave w = parity class (py, px) = (w>>1, w&1) of the 4 x 128 hi pixels of a tile (four 32-pixel MFMA tiles)
// ================================================================================================
// Epilogue: a CU moves store data at only ~16 bytes per clock and a wave is held while its 1 KB store instruction
// drains (64 clocks alone, 256 when all four waves store together), so the 64 KB a tile produces would idle the
// matrix pipes for ~4K clocks.  The stores of tile t are therefore issued between the MFMAs of tile t+1, one
// 16-byte store per wave and 16-MFMA step, each wave in its own quarter of the step: the drain then hides
// behind the wave's previous MFMA.  Gate values are fetched one tile ahead into registers.

// ================================================================================================
// Up on the bf16 MFMA at fp32 accuracy: every fp32 operand is split into THREE bf16 numbers (hi + mid + lo, exact to
// 2^-26) and a product is the six partial products of weight >= 2^-18, accumulated in fp32 smallest first -- the
// result differs from the fp32 MFMA's by less than one fp32 rounding of the sum, so the parity bars do not move --
// at 6 x 32 cycles per 16 channels instead of 8 x 64 (2.7x fewer MFMA cycles).  Same tiling, loader, staggered
// epilogue and gating as up32_kernel; 64 weights x 3 terms live in 96 registers.
#ifndef ARVAE_UP_ISSUE_STEPS
#define ARVAE_UP_ISSUE_STEPS 4
#endif
constexpr int UP_ISSUE_STEPS = ARVAE_UP_ISSUE_STEPS;
template <int LO, int MODE, int PX = 128>
__device__ __forceinline__ void up32x_body(const float *__restrict__ lo, const float *__restrict__ wt, Ep32 ep, int n_img, int n_tiles, const int BID, const int NBLK) {
    using PL = PatchLoader<LO, 1, PX>;
    constexpr int MT = PX / 32;
    constexpr int HI = 2 * LO, PC = PL::PC, PR = PL::PR, PLANE = PL::PLANE_DW;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // max(3 planes, WSTAGE_UP floats)
    unsigned *ldsw = reinterpret_cast<unsigned *>(lds);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, rc = lane & 31;
    const int py = wave >> 1, px = wave & 1;
    const int ky0 = 1 - py, kx0 = 1 - px;
    STAMP(0);

    PL pl;                                                       // first tile's loads fly while the weights are staged
    pl.init(lo, n_img);
    int img0, r0;
    // a workgroup owns a contiguous run of tiles (the row groups of the same images): the halo rows two tiles share come from
    // this XCD's L2 the second time
    const int per_wg = (n_tiles + (int)NBLK - 1) / (int)NBLK, t_first = BID * per_wg;
    const int t_end = min(n_tiles, t_first + per_wg);
    tile_origin<LO, PX>(t_first, img0, r0);
    pl.set_tile(img0, r0, t_first < t_end);
    pl.issue_all();

    // w3[ty][tx][c][term]: the 8 input channels c*16 + half*8 + j of wt[.][chi = rc][ky0 + 2ty][kx0 + 2tx], split in three
    bf16x8 w3[2][2][2][3];
    if (ep.wprep != nullptr) {                                   // split once per step by conv32_weight_prep: 24 coalesced loads
        const uint4 *src = ep.wprep + PREP_DOWN_UINT4 + (wave * PREP_UP_SLOTS) * 64 + lane;
#pragma unroll
        for (int ty = 0; ty < 2; ++ty)
#pragma unroll
            for (int tx = 0; tx < 2; ++tx)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                    for (int t = 0; t < 3; ++t)
                        w3[ty][tx][cc][t] = __builtin_bit_cast(bf16x8, src[(((ty * 2 + tx) * 2 + cc) * 3 + t) * 64]);
    } else {
        float4 v[16];
        load_weights(wt, v);
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int idx4 = threadIdx.x + it * 256;             // row (clo*32 + chi) = idx4 >> 2, taps 4*(idx4 & 3)..+3
            float *d = lds + (idx4 >> 2) * WROW_UP + (idx4 & 3) * 4;
            d[0] = v[it].x; d[1] = v[it].y; d[2] = v[it].z; d[3] = v[it].w;
        }
        __syncthreads();
#pragma unroll
        for (int ty = 0; ty < 2; ++ty)
#pragma unroll
            for (int tx = 0; tx < 2; ++tx)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    float x[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        x[j] = lds[((c * 16 + half * 8 + j) * C32 + rc) * WROW_UP + (ky0 + 2 * ty) * 4 + kx0 + 2 * tx];
                    split8x3(x, w3[ty][tx][c][0], w3[ty][tx][c][1], w3[ty][tx][c][2]);
                }
    }

    STAMP(1);
    // patch origin is lo (r0-1, -1); tap (ty,tx) of class (py,px) reads lo (r + py - ty, c + px - tx)
    int aoff[MT];                                                 // dwords into a plane
    unsigned orel[MT];                                            // output byte offset of this lane's pixel in M-tile mt
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int img, r, c;
        tile_pixel<LO, PX>(mt * 32 + rc, img, r, c);
        aoff[mt] = ((img * PR + r + 1 + py) * PC + c + 1 + px) * PSB + half * 4;
        orel[mt] = (unsigned)(((img * HI + 2 * r + py) * HI + 2 * c + px) * PIXB + half * 16);
    }
    float4 b4[4];
    load_bias4(ep.bias, half, b4);
    const int64_t out_bytes = (int64_t)n_img * HI * HI * PIXB;
    const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(ep.out, out_bytes);
    const __amdgpu_buffer_rsrc_t rs_gate = make_rsrc(MODE == EP_GATE_F ? ep.gate : ep.out, out_bytes);
    const bool want_bits = MODE == EP_RELU && ep.bits_out != nullptr;
    const __amdgpu_buffer_rsrc_t rs_bits =
        make_rsrc(MODE == EP_GATE_B ? (const void *)ep.gate_bits : want_bits ? (const void *)ep.bits_out : (const void *)ep.out,
                  (int64_t)n_img * HI * HI * 4);

    f32x16 prev[MT];                                              // previous tile's accumulators, stored during this tile
    float4 gq[4 * MT];
    unsigned gqb[MT] = {}, gqn[MT] = {}, pbits[MT] = {};
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int i = 0; i < 16; ++i) prev[mt][i] = 0.f;
#pragma unroll
    for (int i = 0; i < 4 * MT; ++i) gq[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    unsigned prev_base = OOB;
    STAMP(2);
    int stamp_t = 0;
    (void)stamp_t;

    // The commit of tile t + 1 closes the body of tile t (the first one is peeled): at the loop header the entry path (loads
    // youngest) and the back edge (stores youngest) would otherwise merge into a vmcnt count that waits for the stores too.
    __syncthreads();
    pl.commit_split3(ldsw);
    __syncthreads();
    for (int tile = t_first; tile < t_end; ++tile) {
        STAMP(3 + 6 * stamp_t);
        tile_origin<LO, PX>(tile, img0, r0);
        STAMP(4 + 6 * stamp_t);
        STAMP(5 + 6 * stamp_t);
        {
            int ni, nr;
            tile_origin<LO, PX>(tile + 1, ni, nr);
            pl.set_tile(ni, nr, tile + 1 < t_end);
        }
        STAMP(6 + 6 * stamp_t);
        const unsigned obase = (unsigned)(((img0 * HI + 2 * r0) * HI) * PIXB);
        f32x16 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
        // operands of step 0; every later step's 3 * MT reads are issued one or two at a time behind the MFMA triples of the
        // step before (two operand sets): a burst of LDS reads in front of a step costs the matrix pipe ~10 idle cycles per
        // read plus the LDS round trip (tools/probes/mfma_barrier.hip)
        bf16x8 a[2][MT][3];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < 3; ++t) a[0][mt][t] = lds_bf16x8(ldsw + t * PLANE + aoff[mt]);
        static_for<0, 8>([&](auto sc) __attribute__((always_inline)) {
            constexpr int step = decltype(sc)::value, ty = step >> 2, tx = (step >> 1) & 1, c = step & 1;
            constexpr int cur = step & 1, nxt = cur ^ 1;
            constexpr int nstep = step + 1, nty = nstep >> 2, ntx = (nstep >> 1) & 1, nc = nstep & 1;
            constexpr int ntoff = -(nty * PC + ntx) * PSB + nc * 8;
            // the next tile's loads all leave in the first half of this tile: the commit at the top of the next tile waits for
            // them (in-order vmcnt), and a load issued in the last step would expose its whole HBM round trip there
            if constexpr (step < UP_ISSUE_STEPS) pl.template issue_step<UP_ISSUE_STEPS, step>();
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, 2 * MT>([&](auto mc) __attribute__((always_inline)) {
                constexpr int sub = decltype(mc)::value, grp = sub / MT, mt = sub % MT;
                // epilogue slots: one per four groups of three MFMAs, at the same program point in every wave (a wave-index
                // branch around them made every vmcnt count behind it conservative)
                constexpr int slot = (step * 2 + grp) * MT + mt;
                if constexpr ((slot & 3) == 0) {
                    constexpr int k = slot >> 2, em = k >> 2, eg = k & 3;
                    const unsigned poff = prev_base + orel[em];
                    if (eg == 0) pbits[em] = 0;
                    pbits[em] |= store_group<MODE>(prev[em], eg, b4[eg], gq[k], gqb[em], rs_out, poff);
                    // (unconditional: a store under a run-time branch makes the compiler's vmcnt count at the next tile's
                    // commit conservative -- it then waits for every store of this tile; without bits_out the offset is out of range)
                    if (MODE == EP_RELU && eg == 3)
                        buf_store_u16(pbits[em], rs_bits, (prev_base == OOB || !want_bits) ? OOB : bits_off(poff, half));
                    if (MODE == EP_GATE_F) gq[k] = buf_load4(rs_gate, obase + orel[em] + eg * 32);
                    // this tile's sign bits are requested in its first slots: the copies into gqb at the bottom of the loop then
                    // wait for loads that are two steps old, not for the one issued in the last step (plus every store before it)
                    if constexpr (MODE == EP_GATE_B && k < MT) gqn[k] = buf_load_u16(rs_bits, bits_off(obase + orel[k], half));
                }
                // Three of the step's 6 * MT MFMAs, taken in ROUND-ROBIN order over the MT accumulators (product-major: every
                // accumulator still sees its six partial products smallest first, so the sums are bit-identical): back-to-back
                // MFMAs into the same accumulator cost ~48 cycles each instead of the 32-cycle issue rate (conv32r.hip), which
                // was the whole gap between this phase's 8800 cycles per tile and the 6144 its MFMAs need.
                static_for<3 * sub, 3 * sub + 3>([&](auto qc) __attribute__((always_inline)) {
                    constexpr int m = decltype(qc)::value, prod = m / MT, pm = m % MT;
                    constexpr int tw = prod == 0 ? 2 : prod == 1 ? 0 : prod == 2 ? 1 : prod == 3 ? 1 : 0;
                    constexpr int ta = prod == 0 ? 0 : prod == 1 ? 2 : prod == 2 ? 1 : prod == 3 ? 0 : prod == 4 ? 1 : 0;
                    MFMA_B(acc[pm], w3[ty][tx][c][tw], a[cur][pm][ta]);
                });
                (void)grp; (void)mt;
                if constexpr (nstep < 8) {                       // this sub-step's share of the next step's operand reads
                    constexpr int r_lo = sub * (3 * MT) / (2 * MT), r_hi = (sub + 1) * (3 * MT) / (2 * MT);
                    static_for<r_lo, r_hi>([&](auto rc_) __attribute__((always_inline)) {
                        constexpr int ri = decltype(rc_)::value, rmt = ri / 3, rt = ri % 3;
                        a[nxt][rmt][rt] = lds_bf16x8(ldsw + rt * PLANE + aoff[rmt] + ntoff);
                    });
                }
                (void)nxt;
            });
            __builtin_amdgcn_sched_barrier(0);
        });
        STAMP(7 + 6 * stamp_t);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) prev[mt] = acc[mt];
        if (MODE == EP_GATE_B) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) gqb[mt] = gqn[mt];
        }
        prev_base = obase;
        // (unconditional, also behind the last tile, where it stages zeros: with the commit on one side of a branch the
        // compiler sinks the loads of the whole tile down to it)
        __syncthreads();
        pl.commit_split3(ldsw);
        __syncthreads();
        STAMP(8 + 6 * stamp_t);
        ++stamp_t;
    }
    // the last tile's epilogue has nothing to hide behind
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const unsigned poff = prev_base + orel[mt];
        unsigned bits = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) bits |= store_group<MODE>(prev[mt], g, b4[g], gq[mt * 4 + g], gqb[mt], rs_out, poff);
        if (MODE == EP_RELU) buf_store_u16(bits, rs_bits, (prev_base == OOB || !want_bits) ? OOB : bits_off(poff, half));
    }
    STAMP_WAIT();
    STAMP(63);
}
template <int LO, int MODE, int PX = 128>
__global__ __launch_bounds__(256, 1) void up32x_kernel(const float *__restrict__ lo, const float *__restrict__ wt, Ep32 ep, int n_img, int n_tiles) {
    up32x_body<LO, MODE, PX>(lo, wt, ep, n_img, n_tiles, blockIdx.x, gridDim.x);
}

// ================================================================================================================================
// Up map of the 16x16 layers with PRODUCER / CONSUMER waves (round 3; the prepared weights must exist, i.e. the fused step).
// In up32x_kernel every wave loads, splits, multiplies and stores: per 128-pixel tile 1190 cycles of commit between two barriers
// and an MFMA phase of 8650-9050 cycles for 6144 of MFMA issue (the next tile's loads, the operand reads and the previous tile's
// 64 KB of stores are issued between its MFMAs; profiles/r2_phase_stamps.txt).  Here waves 0-3 (consumers, wave = parity class)
// keep the weights and issue MFMAs with the operand reads of the next step between them and nothing else; when a tile is done
// they park its four accumulator tiles in LDS (64 KB).  Waves 4-7 (producers) meanwhile run the PREVIOUS tile's epilogue from
// there (bias, ReLU / gate, 16-byte stores, sign bits), split the NEXT tile's patch (exact truncation, single-issue
// instructions) into the other LDS image and issue the loads of the tile after that.  Two barriers per tile.
template <int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void up32p_kernel(const float *__restrict__ lo, Ep32 ep,
                                                                                               int n_img, int n_tiles) {
    constexpr int LO = 16, PX = 128, HI = 2 * LO, MT = PX / 32;
    using PL = PatchLoader<LO, 1, PX>;
    constexpr int PR = PL::PR, PC = PL::PC, PLANE = PL::PLANE_DW, SLOTS = PL::SLOTS, ITERS = PL::ITERS;
    constexpr int BUF = 3 * PLANE;                               // dwords of one three-plane patch image
    extern __shared__ __attribute__((aligned(16))) float lds_f[];                 // [2][BUF] patch images | handoff
    unsigned *lds = reinterpret_cast<unsigned *>(lds_f);
    float4 *hand = reinterpret_cast<float4 *>(lds + 2 * BUF);    // [wave][mt][group][lane]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, rc = lane & 31;
    const int per_wg = (n_tiles + (int)gridDim.x - 1) / (int)gridDim.x, t_first = blockIdx.x * per_wg;
    const int t_end = min(n_tiles, t_first + per_wg);
    const int cls = wave & 3, py = cls >> 1, px = cls & 1;       // parity class of a consumer wave / of the producer that finishes it
    unsigned orel[MT];                                           // output byte offset of this lane's pixel in M-tile mt
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int img, r, c;
        tile_pixel<LO, PX>(mt * 32 + rc, img, r, c);
        orel[mt] = (unsigned)(((img * HI + 2 * r + py) * HI + 2 * c + px) * PIXB + half * 16);
    }

    if (wave >= 4) {
        // ============================================================================================ producers
        PSTAMP(1, 0);
        int pst = 0;
        (void)pst;
        const int pt = threadIdx.x - 256;
        const __amdgpu_buffer_rsrc_t rs_lo = make_rsrc(lo, (int64_t)n_img * LO * LO * PIXB);
        const int64_t out_bytes = (int64_t)n_img * HI * HI * PIXB;
        const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(ep.out, out_bytes);
        const bool want_bits = MODE == EP_RELU && ep.bits_out != nullptr;
        const __amdgpu_buffer_rsrc_t rs_bits = make_rsrc(want_bits ? (const void *)ep.bits_out : (const void *)ep.out, (int64_t)n_img * HI * HI * 4);
        float4 b4[4];
        load_bias4(ep.bias, half, b4);
        // patch slots of this thread (as PatchLoader, with the producers' thread index): byte offset relative to the tile's first
        // patch row | bit 0: first patch row, bit 1: last patch row (the only rows that can fall outside the image)
        unsigned rel[ITERS];
        float4 rv[ITERS];
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int idx = pt + it * 256;
            const int q = idx & 7, pix = idx >> 3;
            const int pc = pix % PC, pr = (pix / PC) % PR;
            const int gx = pc - 1;
            const bool ok = idx < SLOTS && (unsigned)gx < (unsigned)LO;
            rel[it] = ok ? (unsigned)((pr * LO + gx) * PIXB + q * 16) | (pr == 0 ? 1u : 0u) | (pr == PR - 1 ? 2u : 0u) : OOB;
        }
        auto issue_tile = [&](int tile) __attribute__((always_inline)) {
            int img0, r0;
            tile_origin<LO, PX>(tile, img0, r0);
            const int gy0 = r0 - 1;
            const int base = ((img0 * LO + gy0) * LO) * PIXB;    // negative for the very first patch row of the tensor
            const unsigned bad = (gy0 < 0 ? 1u : 0u) | (gy0 + PR - 1 >= LO ? 2u : 0u);
            const bool valid = tile < t_end;
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const bool row_ok = valid && (rel[it] & bad) == 0;
                rv[it] = buf_load4(rs_lo, row_ok ? (rel[it] & ~3u) + (unsigned)base : OOB);
            }
        };
        auto commit_tile = [&](unsigned *planes) __attribute__((always_inline)) {
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int idx = pt + it * 256;
                float4 v = rv[it];
                asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));      // every lane uses the slot's registers (exact vmcnt)
                if (idx < SLOTS) {
                    const int q = idx & 7, pix = idx >> 3;
                    uint2 hv, mv, lv;
                    trunc_pair3(v.x, v.y, hv.x, mv.x, lv.x);
                    trunc_pair3(v.z, v.w, hv.y, mv.y, lv.y);
                    *reinterpret_cast<uint2 *>(planes + pix * PSB + q * 2) = hv;
                    *reinterpret_cast<uint2 *>(planes + PLANE + pix * PSB + q * 2) = mv;
                    *reinterpret_cast<uint2 *>(planes + 2 * PLANE + pix * PSB + q * 2) = lv;
                }
            }
        };
        // epilogue of the tile whose accumulators sit in the handoff area: this thread = lane `lane` of consumer wave `cls`
        unsigned prev_base = OOB;
        auto epilogue = [&]() __attribute__((always_inline)) {
            const float4 no_gate = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x16 acc;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 v = hand[((cls * MT + mt) * 4 + g) * 64 + lane];
                    acc[4 * g] = v.x; acc[4 * g + 1] = v.y; acc[4 * g + 2] = v.z; acc[4 * g + 3] = v.w;
                }
                const unsigned poff = prev_base + orel[mt];
                unsigned bits = 0;
#pragma unroll
                for (int g = 0; g < 4; ++g) bits |= store_group<MODE>(acc, g, b4[g], no_gate, 0u, rs_out, poff);
                if (MODE == EP_RELU) buf_store_u16(bits, rs_bits, (prev_base == OOB || !want_bits) ? OOB : bits_off(poff, half));
            }
        };
        issue_tile(t_first);
        __syncthreads();                                         // (the consumers' prologue barrier)
        commit_tile(lds);
        issue_tile(t_first + 1);
        __syncthreads();                                         // first tile staged
        for (int tile = t_first; tile < t_end; ++tile) {
            const int cur = (tile - t_first) & 1;
            PSTAMP(1, 3 + 6 * pst);
            epilogue();                                          // tile - 1 (nothing the first time: prev_base is out of range)
            PSTAMP(1, 4 + 6 * pst);
            commit_tile(lds + (cur ^ 1) * BUF);                  // tile + 1
            PSTAMP(1, 5 + 6 * pst);
            int img0, r0;
            tile_origin<LO, PX>(tile, img0, r0);
            prev_base = (unsigned)(((img0 * HI + 2 * r0) * HI) * PIXB);
            issue_tile(tile + 2);
            PSTAMP(1, 6 + 6 * pst);
            __syncthreads();                                     // A: handoff area free, tile + 1 staged
            PSTAMP(1, 7 + 6 * pst);
            __syncthreads();                                     // B: this tile's accumulators are in the handoff area
            PSTAMP(1, 8 + 6 * pst);
            ++pst;
        }
        epilogue();                                              // the last tile
        PSTAMP(1, 63);
        return;
    }

    // ================================================================================================ consumers (wave = parity class)
    // w3[ty][tx][c][term]: the 8 input channels c*16 + half*8 + j of wt[.][chi = rc][ky0 + 2ty][kx0 + 2tx], split in three
    PSTAMP(0, 0);
    int cst = 0;
    (void)cst;
    bf16x8 w3[2][2][2][3];
    {
        const uint4 *src = ep.wprep + PREP_DOWN_UINT4 + (wave * PREP_UP_SLOTS) * 64 + lane;
#pragma unroll
        for (int ty = 0; ty < 2; ++ty)
#pragma unroll
            for (int tx = 0; tx < 2; ++tx)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                    for (int t = 0; t < 3; ++t)
                        w3[ty][tx][cc][t] = __builtin_bit_cast(bf16x8, src[(((ty * 2 + tx) * 2 + cc) * 3 + t) * 64]);
    }
    // patch origin is lo (r0-1, -1); tap (ty,tx) of class (py,px) reads lo (r + py - ty, c + px - tx)
    int aoff[MT];                                                // dwords into a plane
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int img, r, c;
        tile_pixel<LO, PX>(mt * 32 + rc, img, r, c);
        aoff[mt] = ((img * PR + r + 1 + py) * PC + c + 1 + px) * PSB + half * 4;
    }
    __syncthreads();
    __syncthreads();                                             // first tile staged
    PSTAMP(0, 1);
    for (int tile = t_first; tile < t_end; ++tile) {
        PSTAMP(0, 3 + 6 * cst);
        const unsigned *ldsw = lds + ((tile - t_first) & 1) * BUF;
        f32x16 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
        bf16x8 a[2][MT][3];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < 3; ++t) a[0][mt][t] = lds_bf16x8(ldsw + t * PLANE + aoff[mt]);
        static_for<0, 8>([&](auto sc) __attribute__((always_inline)) {
            constexpr int step = decltype(sc)::value, ty = step >> 2, tx = (step >> 1) & 1, c = step & 1;
            constexpr int cur = step & 1, nxt = cur ^ 1;
            constexpr int nstep = step + 1, nty = nstep >> 2, ntx = (nstep >> 1) & 1, nc = nstep & 1;
            constexpr int ntoff = -(nty * PC + ntx) * PSB + nc * 8;
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, 2 * MT>([&](auto mc) __attribute__((always_inline)) {
                constexpr int sub = decltype(mc)::value;
                // three of the step's 6 * MT MFMAs, round-robin over the MT accumulators (product-major: every accumulator
                // sees its six partial products smallest first), then this sub-step's share of the next step's operand reads
                static_for<3 * sub, 3 * sub + 3>([&](auto qc) __attribute__((always_inline)) {
                    constexpr int m = decltype(qc)::value, prod = m / MT, pm = m % MT;
                    constexpr int tw = prod == 0 ? 2 : prod == 1 ? 0 : prod == 2 ? 1 : prod == 3 ? 1 : 0;
                    constexpr int ta = prod == 0 ? 0 : prod == 1 ? 2 : prod == 2 ? 1 : prod == 3 ? 0 : prod == 4 ? 1 : 0;
                    MFMA_B(acc[pm], w3[ty][tx][c][tw], a[cur][pm][ta]);
                });
                if constexpr (nstep < 8) {
                    constexpr int r_lo = sub * (3 * MT) / (2 * MT), r_hi = (sub + 1) * (3 * MT) / (2 * MT);
                    static_for<r_lo, r_hi>([&](auto rc_) __attribute__((always_inline)) {
                        constexpr int ri = decltype(rc_)::value, rmt = ri / 3, rt = ri % 3;
                        a[nxt][rmt][rt] = lds_bf16x8(ldsw + rt * PLANE + aoff[rmt] + ntoff);
                    });
                }
                (void)nxt;
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        PSTAMP(0, 4 + 6 * cst);
        __syncthreads();                                         // A: the producers are done with the previous tile's accumulators
        PSTAMP(0, 5 + 6 * cst);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                hand[((wave * MT + mt) * 4 + g) * 64 + lane] = make_float4(acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]);
        PSTAMP(0, 6 + 6 * cst);
        __syncthreads();                                         // B
        PSTAMP(0, 7 + 6 * cst);
        ++cst;
    }
    PSTAMP(0, 63);
}

// The decoder's first convolution (4x4 -> 8x8: a few tiles per CU, 6.5 us) and the all-pairs attribute regularisation of the
// same forward pass (regloss.h: ~65 workgroups, 6.3 us, needs only z and the labels) in ONE grid: workgroups [0, grid_up) run
// the convolution, the rest the regulariser's (row block, dim) pairs, staging their columns in the launch's dynamic LDS.
template <int MODE>
__global__ __launch_bounds__(256, 1) void up32x_reg_kernel(const float *__restrict__ lo, const float *__restrict__ wt, Ep32 ep, int n_img,
                                                            int n_tiles, int grid_up, RegArgs reg, int reg_bx) {
    if ((int)blockIdx.x < grid_up) {
        up32x_body<4, MODE, 32>(lo, wt, ep, n_img, n_tiles, blockIdx.x, grid_up);
        return;
    }
    extern __shared__ __attribute__((aligned(16))) float reg_lds[];
    const int b = blockIdx.x - grid_up;
    reg_loss_block(reg, b % reg_bx, b / reg_bx, reg_lds, reg_lds + REG_CHUNK);
}


// ================================================================================================
// Wgrad: dwt[clo][chi][ky][kx] += sum_{n,ly,lx} lo[n,ly,lx,clo] * hi[n,2ly-1+ky,2lx-1+kx,chi]
// wave w = ky; acc[kx] = 32(chi) x 32(clo); pixels are the MFMA K axis (2 per instruction, 64 steps per tile).
// slab layout per workgroup: [ky][kx][clo][chi] (16384 floats) + 32 bias sums.
// BIAS: 0 none, 1 = sum of the lo operand per clo, 2 = sum of the hi operand per chi (interior pixels)
// ================================================================================================
constexpr int WG32_SLAB = SLAB_C32_FLOATS;


// ================================================================================================
// Wgrad on the bf16 MFMA at fp32 accuracy (three-term split, six partial products).  Pixels are the K axis, 16 per
// MFMA with 8 consecutive pixels per lane, but LDS holds [pixel][channel] images: the operands come in through
// ds_read_b64_tr_b16, the transposing read (4 pixels x 16 channels per 16-lane group, two reads per operand).
// Tile = 64 lo pixels (three planes of the hi patch fit LDS only at this size); wave = ky, 4 accumulator tiles (kx).
template <int LO, int BIAS>
__device__ __forceinline__ void wgrad32x_body(const float *__restrict__ lo, const float *__restrict__ hi, float *__restrict__ slab, int n_img,
                                              int n_tiles, const int BID, const int NBLK) {
    constexpr int PX = 64, KB = PX / 16;                         // pixels per tile, 16-pixel K blocks per tile
    using PL = PatchLoader<LO, 2, PX>;
    // plane pitches (dwords per pixel) chosen for the transposed reads: a block is 4 consecutive K pixels x 16 dwords and
    // the 64-bank LDS serves it without conflicts when the four rows land on the four 16-dword quarters -- consecutive lo
    // pixels are WG_PSB_L apart, the hi pixels under them 2 * WG_PSB_H (stride 2): 16 and 48 dwords.  (With the common
    // pitch of 20 the fourth row wrapped onto the first: 17 % of the kernel's wave cycles were LDS bank conflicts.)
    constexpr int WG_PSB_L = WGRAD_PSB_L, WG_PSB_H = WGRAD_PSB_H;
    constexpr int PC = PL::PC, PR = PL::PR, PLANE = PL::PLANE_DW / PSB * WG_PSB_H, LPLANE = PX * WG_PSB_L;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // hi patch: 3 planes | lo tile: 3 planes
    unsigned *ldsw = reinterpret_cast<unsigned *>(lds);
    unsigned *lo_w = ldsw + 3 * PLANE;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, rc = lane & 31;

    // transposed-read addresses (dwords into a plane): 16-lane group g16 reads channels 16 (g16 & 1) .. +15 of the pixels
    // 16 b + 8 (g16 >> 1) + 4 i + q; lane 4q + p of the group points at channels 4p .. 4p+3 of pixel row q
    int loff[KB][2], hoff[KB][2];
    {
        const int g16 = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
        for (int b = 0; b < KB; ++b)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int P = 16 * b + 8 * (g16 >> 1) + 4 * i + q;
                int img, r, c;
                tile_pixel<LO, PX>(P, img, r, c);
                loff[b][i] = P * WG_PSB_L + 8 * (g16 & 1) + 2 * pp;
                hoff[b][i] = ((img * PR + 2 * r + wave) * PC + 2 * c) * WG_PSB_H + 8 * (g16 & 1) + 2 * pp;      // + kx * WG_PSB_H
            }
    }

    f32x16 acc[4];
#pragma unroll
    for (int kx = 0; kx < 4; ++kx)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[kx][i] = 0.f;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);             // BIAS 1: lo sums, BIAS 2: hi sums, channels 4 (tid & 7) .. +3

    PL pl;
    pl.init(hi, n_img);
    const int64_t lo_bytes = (int64_t)n_img * LO * LO * PIXB;
    const __amdgpu_buffer_rsrc_t rs_lo = make_rsrc(lo, lo_bytes);
    float4 lr[2];                                                // lo tile: 64 contiguous pixels x 8 float4 = 2 slots per thread
    unsigned lo_base = 0;
    auto set_lo = [&](int i0, int rr0, bool ok) {
        lo_base = ok ? (unsigned)(((i0 * LO + rr0) * LO) * PIXB) + threadIdx.x * 16 : OOB;
    };
    int img0, r0;
    tile_origin<LO, PX>(BID, img0, r0);
    pl.set_tile(img0, r0, BID < n_tiles);
    set_lo(img0, r0, BID < n_tiles);
    pl.issue_all();
#pragma unroll
    for (int it = 0; it < 2; ++it) lr[it] = buf_load4(rs_lo, lo_base + it * 4096);

    for (int tile = BID; tile < n_tiles; tile += NBLK) {
        __syncthreads();
        pl.template commit_split3<BIAS == 2, WG_PSB_H>(ldsw, &bias4);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = threadIdx.x + it * 256, pix = idx >> 3, q = idx & 7;
            uint2 hv, mv, lv;
            split_pair3(lr[it].x, lr[it].y, hv.x, mv.x, lv.x);
            split_pair3(lr[it].z, lr[it].w, hv.y, mv.y, lv.y);
            *reinterpret_cast<uint2 *>(lo_w + pix * WG_PSB_L + q * 2) = hv;
            *reinterpret_cast<uint2 *>(lo_w + LPLANE + pix * WG_PSB_L + q * 2) = mv;
            *reinterpret_cast<uint2 *>(lo_w + 2 * LPLANE + pix * WG_PSB_L + q * 2) = lv;
            if (BIAS == 1) { bias4.x += lr[it].x; bias4.y += lr[it].y; bias4.z += lr[it].z; bias4.w += lr[it].w; }
        }
        __syncthreads();
        {
            int ni, nr;
            tile_origin<LO, PX>(tile + NBLK, ni, nr);
            pl.set_tile(ni, nr, tile + NBLK < n_tiles);
            set_lo(ni, nr, tile + NBLK < n_tiles);
        }
        static_for<0, KB>([&](auto bc) __attribute__((always_inline)) {
            constexpr int b = decltype(bc)::value;
            bf16x8 b3[3];                                        // lo values: B operand, column = clo
#pragma unroll
            for (int t = 0; t < 3; ++t) b3[t] = lds_tr_bf16x8(lo_w + t * LPLANE + loff[b][0], lo_w + t * LPLANE + loff[b][1]);
            pl.template issue_step<KB, b>();
            if constexpr (b < 2) lr[b] = buf_load4(rs_lo, lo_base + b * 4096);
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                bf16x8 a3[3];                                    // hi values at tap (ky = wave, kx): A operand, row = chi
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    a3[t] = lds_tr_bf16x8(ldsw + t * PLANE + hoff[b][0] + kx * WG_PSB_H, ldsw + t * PLANE + hoff[b][1] + kx * WG_PSB_H);
                MFMA_B(acc[kx], a3[2], b3[0]);                   // smallest partial products first
                MFMA_B(acc[kx], a3[0], b3[2]);
                MFMA_B(acc[kx], a3[1], b3[1]);
                MFMA_B(acc[kx], a3[1], b3[0]);
                MFMA_B(acc[kx], a3[0], b3[1]);
                MFMA_B(acc[kx], a3[0], b3[0]);
            }
        });
    }

    // partial results -> slab[blockIdx][ky][kx][clo = rc][chi = 8g + 4*half + j]: 16-byte stores
    float *out = slab + (int64_t)BID * WG32_SLAB;
#pragma unroll
    for (int kx = 0; kx < 4; ++kx)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4 *>(out + ((wave * 4 + kx) * C32 + rc) * C32 + 8 * g + 4 * half) =
                make_float4(acc[kx][4 * g], acc[kx][4 * g + 1], acc[kx][4 * g + 2], acc[kx][4 * g + 3]);
    if (BIAS != 0) {
        __syncthreads();
        // every thread summed channel chunk q = threadIdx.x & 7 (slot stride 256 keeps q fixed)
        *reinterpret_cast<float4 *>(lds + threadIdx.x * 4) = bias4;
        __syncthreads();
        if (threadIdx.x < C32) {
            const int q = threadIdx.x >> 2, e = threadIdx.x & 3;
            float tot = 0.f;
            for (int j = 0; j < 32; ++j) tot += lds[(j * 8 + q) * 4 + e];
            out[16 * C32 * C32 + threadIdx.x] = tot;
        }
    }
}
template <int LO, int BIAS>
__global__ __launch_bounds__(256, 1) void wgrad32x_kernel(const float *__restrict__ lo, const float *__restrict__ hi,
                                                          float *__restrict__ slab, int n_img, int n_tiles) {
    wgrad32x_body<LO, BIAS>(lo, hi, slab, n_img, n_tiles, blockIdx.x, gridDim.x);
}

// ---- the 4x4 layers: data gradient and weight gradient of a layer in ONE launch ---------------------------------------
// Both read the same incoming gradient and neither needs the other; alone each fills half the chip (128 workgroups) for
// 7-11 us, most of it launch ramp and one memory round trip.  Workgroups [0, grid_a) run the data-gradient body over two
// 32-pixel tiles each, the rest the weight-gradient body (one 64-pixel tile and one slab each): 256 workgroups, one per CU.
template <int MODE, int BIAS>
__global__ __launch_bounds__(256, 1) void pair4_down_kernel(const float *__restrict__ g_hi, const float *__restrict__ wt, Ep32 ep,
                                                           const float *__restrict__ w_lo, const float *__restrict__ w_hi,
                                                           float *__restrict__ slab, int n_img, int tiles_a, int tiles_b, int grid_a) {
    if ((int)blockIdx.x < grid_a) down32s_body<4, MODE>(g_hi, wt, ep, n_img, tiles_a, blockIdx.x, grid_a);
    else wgrad32x_body<4, BIAS>(w_lo, w_hi, slab, n_img, tiles_b, blockIdx.x - grid_a, gridDim.x - grid_a);
}
template <int MODE, int BIAS>
__global__ __launch_bounds__(256, 1) void pair4_up_kernel(const float *__restrict__ g_lo, const float *__restrict__ wt, Ep32 ep,
                                                         const float *__restrict__ w_lo, const float *__restrict__ w_hi,
                                                         float *__restrict__ slab, int n_img, int tiles_a, int tiles_b, int grid_a) {
    if ((int)blockIdx.x < grid_a) up32x_body<4, MODE, 32>(g_lo, wt, ep, n_img, tiles_a, blockIdx.x, grid_a);
    else wgrad32x_body<4, BIAS>(w_lo, w_hi, slab, n_img, tiles_b, blockIdx.x - grid_a, gridDim.x - grid_a);
}


// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <int LO, int PX = 128> static constexpr int tiles_for(int n) {
    using T = Tile<LO, PX>;
    return T::TI == 1 ? n * (LO / T::TR) : (n + T::TI - 1) / T::TI;
}
static int grid_for_tiles(int tiles) { return tiles < cu_count() ? tiles : cu_count(); }

bool conv32_fits(const arvae_link_t *l) {
    return l->chi == 32 && l->clo == 32 && l->kh == 4 && l->kw == 4 && l->stride == 2 && l->pad == 1 &&
           l->hh == l->hw && l->lh == l->lw && (l->lh == 16 || l->lh == 8 || l->lh == 4) && l->hi_perm_c == 0 &&
           l->lo_perm_c == 0 && (int64_t)l->n * l->hh * l->hw * PIXB < (1ll << 31) - (1ll << 20);
}

template <class K> static void allow_lds(K kernel, int bytes) {
    (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

template <int A, int B> struct MaxOf { static constexpr int value = A > B ? A : B; };

template <int LO, int MODE>
static void launch_down_small(const Operand &hi, const float *wt, const Ep32 &ep, int n, hipStream_t s) {
    constexpr int LDS = (PatchLoader<LO, 2, 32>::PATCH_FLOATS + 4 * 16 * 64) * 4;
    const int tiles = tiles_for<LO, 32>(n);
    const int grid = tiles < 2 * cu_count() ? tiles : 2 * cu_count();
    static bool attr = false;
    if (!attr) { allow_lds(down32s_kernel<LO, MODE>, LDS); attr = true; }
    ARVAE_LAUNCH((down32s_kernel<LO, MODE>), dim3(grid), dim3(256), LDS, s, hi.v, wt, ep, n, tiles);
}
template <int LO, int MODE>
static void launch_down_v(int, const Operand &hi, const float *wt, const Ep32 &ep, int n, int, hipStream_t s) {
    // 4x4 layers: the K-split 32-pixel kernel (two packed three-term images of eight 4x4 patches, 166 KB, do not fit the LDS)
    if (LO == 4) return launch_down_small<4, MODE>(hi, wt, ep, n, s);
    // three-term bf16 at fp32 accuracy, 64-pixel tiles.  (A two-term / four-product variant measured 28.5-31 us against 45 for
    // the fp32 MFMA at the 16x16 layers in round 1, but flipped ReLU units moved per-tensor gradients by up to 5e-3 relative
    // L2: removed in round 3 together with the fp32-MFMA generation of these kernels.)
    constexpr int LDSX = MaxOf<2 * PatchLoader<LO, 2, 64>::BUF3_DW + 2 * 2 * 16 * 64, WSTAGE_DOWN>::value * 4;
    const int tiles64 = tiles_for<LO, 64>(n);
    static bool attrx = false;
    if (!attrx) { allow_lds(down32x_kernel<LO, MODE>, LDSX); attrx = true; }
    ARVAE_LAUNCH((down32x_kernel<LO, MODE>), dim3(grid_for_tiles(tiles64)), dim3(256), LDSX, s, hi.v, wt, ep, n, tiles64);
}
template <int LO, int MODE, int PX>
static void launch_up_px(const Operand &lo, const float *wt, const Ep32 &ep, int n, hipStream_t s) {
    const int tiles = tiles_for<LO, PX>(n), grid = grid_for_tiles(tiles);
    if constexpr (LO == 16 && PX == 128) {                      // producer / consumer form (needs the step's prepared weights)
        static const bool pc = getenv("ARVAE_UP32_NO_PC") == nullptr;       // A/B: up32x_kernel
        if (pc && ep.wprep != nullptr && (MODE == EP_PLAIN || MODE == EP_RELU)) {
            constexpr int LDSP = (2 * 3 * PatchLoader<16, 1, 128>::PLANE_DW) * 4 + 4 * 4 * 4 * 64 * 16;
            static bool attrp = false;
            if (!attrp) { allow_lds(up32p_kernel<MODE>, LDSP); attrp = true; }
            ARVAE_LAUNCH((up32p_kernel<MODE>), dim3(grid), dim3(512), LDSP, s, lo.v, ep, n, tiles);
            return;
        }
    }
    constexpr int LDSX = MaxOf<3 * PatchLoader<LO, 1, PX>::PLANE_DW, WSTAGE_UP>::value * 4;
    static bool attrx = false;
    if (!attrx) { allow_lds(up32x_kernel<LO, MODE, PX>, LDSX); attrx = true; }
    ARVAE_LAUNCH((up32x_kernel<LO, MODE, PX>), dim3(grid), dim3(256), LDSX, s, lo.v, wt, ep, n, tiles);
}
// 128-pixel tiles, or 32-pixel tiles when the former give a CU at most one tile (nothing to pipeline, or idle CUs:
// the 8x8 and 4x4 layers at batch 512)
template <int LO, int MODE>
static void launch_up_v(int, const Operand &lo, const float *wt, const Ep32 &ep, int n, int, hipStream_t s) {
    static const bool small_ok = getenv("ARVAE_NO_SMALL_TILES") == nullptr;     // diagnostic switch
    if (LO == 4 && small_ok && 2 * tiles_for<LO, 128>(n) <= cu_count()) launch_up_px<4, MODE, 32>(lo, wt, ep, n, s);
    else if (LO == 8 && small_ok && tiles_for<LO, 128>(n) <= cu_count()) launch_up_px<8, MODE, 32>(lo, wt, ep, n, s);
    else launch_up_px<LO, MODE, 128>(lo, wt, ep, n, s);
}

static int ep_mode(const Ep32 &ep, int relu) {
    return ep.gate_bits != nullptr ? EP_GATE_B : ep.gate != nullptr ? EP_GATE_F : relu ? EP_RELU : EP_PLAIN;
}

// (conv32k.hip: the four-way reduction-split kernel of the 16x16 / 8x8 layers; needs the prepared weights)
bool conv32_down_ksplit_fits(const arvae_link_t *l, const Ep32 &ep);
void conv32_down_ksplit(const arvae_link_t *l, const float *hi, const Ep32 &ep, int mode, hipStream_t s);

template <int LO> static int launch_down(const arvae_link_t *l, const Operand &hi, const float *wt, const Ep32 &ep, int relu, hipStream_t s) {
    const int tiles = tiles_for<LO>(l->n), grid = grid_for_tiles(tiles);
    if (LO != 4 && conv32_down_ksplit_fits(l, ep)) {
        conv32_down_ksplit(l, hi.v, ep, ep_mode(ep, relu), s);
        return check_launch(LO == 16 ? "down32_kernel<16>" : "down32_kernel<8>");
    }
    switch (ep_mode(ep, relu)) {
        case EP_GATE_B: launch_down_v<LO, EP_GATE_B>(grid, hi, wt, ep, l->n, tiles, s); break;
        case EP_GATE_F: launch_down_v<LO, EP_GATE_F>(grid, hi, wt, ep, l->n, tiles, s); break;
        case EP_RELU: launch_down_v<LO, EP_RELU>(grid, hi, wt, ep, l->n, tiles, s); break;
        default: launch_down_v<LO, EP_PLAIN>(grid, hi, wt, ep, l->n, tiles, s); break;
    }
    return check_launch(LO == 16 ? "down32_kernel<16>" : LO == 8 ? "down32_kernel<8>" : "down32_kernel<4>");
}

// gate (float activation) or gate_bits (relu_bits16) select a gated epilogue; with relu, bits_out (may be null) receives
// the sign bits of the result
int conv32_down(const arvae_link_t *l, const Operand &hi, const float *wt, const float *bias, int relu, const float *gate,
                const uint16_t *gate_bits, uint16_t *bits_out, float *out, hipStream_t s, const float *wprep) {
    Ep32 ep{bias, gate, gate_bits, bits_out, out, reinterpret_cast<const uint4 *>(wprep)};
    switch (l->lh) {
        case 16: return launch_down<16>(l, hi, wt, ep, relu, s);
        case 8: return launch_down<8>(l, hi, wt, ep, relu, s);
        default: return launch_down<4>(l, hi, wt, ep, relu, s);
    }
}

template <int LO> static int launch_up(const arvae_link_t *l, const Operand &lo, const float *wt, const Ep32 &ep, int relu, hipStream_t s) {
    const int tiles = tiles_for<LO>(l->n), grid = grid_for_tiles(tiles);
    switch (ep_mode(ep, relu)) {
        case EP_GATE_B: launch_up_v<LO, EP_GATE_B>(grid, lo, wt, ep, l->n, tiles, s); break;
        case EP_GATE_F: launch_up_v<LO, EP_GATE_F>(grid, lo, wt, ep, l->n, tiles, s); break;
        case EP_RELU: launch_up_v<LO, EP_RELU>(grid, lo, wt, ep, l->n, tiles, s); break;
        default: launch_up_v<LO, EP_PLAIN>(grid, lo, wt, ep, l->n, tiles, s); break;
    }
    return check_launch(LO == 16 ? "up32_kernel<16>" : LO == 8 ? "up32_kernel<8>" : "up32_kernel<4>");
}

// conv32_up of a 4x4 -> 8x8 ReLU layer (forward pass, prepared weights, small tiles) with the regulariser's workgroups riding in
// the same grid (up32x_reg_kernel); false: not that case, launch the two separately
bool conv32_up_reg_fits(const arvae_link_t *l, const float *wprep) {
    static const bool off = getenv("ARVAE_NO_PAIR_REG") != nullptr || getenv("ARVAE_NO_SMALL_TILES") != nullptr;
    return !off && conv32_fits(l) && l->lh == 4 && wprep != nullptr && 2 * tiles_for<4, 128>(l->n) <= cu_count();
}
int conv32_up_reg(const arvae_link_t *l, const Operand &lo, const float *wt, const float *bias, uint16_t *bits_out, float *out,
                  const float *wprep, const RegArgs &reg, int r, hipStream_t s) {
    Ep32 ep{bias, nullptr, nullptr, bits_out, out, reinterpret_cast<const uint4 *>(wprep)};
    const int tiles = tiles_for<4, 32>(l->n), grid_up = grid_for_tiles(tiles);
    constexpr int LDSX = MaxOf<MaxOf<3 * PatchLoader<4, 1, 32>::PLANE_DW, WSTAGE_UP>::value, 2 * REG_CHUNK>::value * 4;
    static bool attr = false;
    if (!attr) { allow_lds(up32x_reg_kernel<EP_RELU>, LDSX); attr = true; }
    const int reg_bx = (int)((reg.n_rows + REG_ROWS_PER_BLOCK - 1) / REG_ROWS_PER_BLOCK);
    ARVAE_LAUNCH((up32x_reg_kernel<EP_RELU>), dim3(grid_up + reg_bx * r), dim3(256), LDSX, s, lo.v, wt, ep, l->n, tiles, grid_up, reg, reg_bx);
    return check_launch("up32_kernel<4>(+ reg_loss)");
}

int conv32_up(const arvae_link_t *l, const Operand &lo, const float *wt, const float *bias, int relu, const float *gate,
              const uint16_t *gate_bits, uint16_t *bits_out, float *out, hipStream_t s, const float *wprep) {
    Ep32 ep{bias, gate, gate_bits, bits_out, out, reinterpret_cast<const uint4 *>(wprep)};
    switch (l->lh) {
        case 16: return launch_up<16>(l, lo, wt, ep, relu, s);
        case 8: return launch_up<8>(l, lo, wt, ep, relu, s);
        default: return launch_up<4>(l, lo, wt, ep, relu, s);
    }
}

// floats of workspace per layer for conv32_weight_prep, and the batched launch (up to 8 layers)
int64_t conv32_prep_floats() { return PREP_FLOATS; }

int conv32_weight_prep(const float *const *wts, float *const *preps, int n_layers, hipStream_t s) {
    if (n_layers <= 0) return ARVAE_OK;
    ARVAE_REQUIRE(n_layers <= PREP_MAX_LAYERS, "conv32_weight_prep: at most %d layers per launch", PREP_MAX_LAYERS);
    PrepArgs p{};
    for (int i = 0; i < n_layers; ++i) {
        p.wt[i] = wts[i];
        p.out[i] = reinterpret_cast<uint4 *>(preps[i]);
    }
    ARVAE_LAUNCH(conv32_weight_prep_kernel, dim3(16 * n_layers), dim3(256), 0, s, p);
    return check_launch("conv32_weight_prep");
}

// the same together with the latent block's layout prep (mid_prep_args, midblock.hip): one launch
int conv32_weight_prep_with_mid(const float *const *wts, float *const *preps, int n_layers, const MidPrepArgs &mid, hipStream_t s) {
    ARVAE_REQUIRE(n_layers > 0 && n_layers <= PREP_MAX_LAYERS && mid.count > 0, "conv32_weight_prep_with_mid: nothing to prepare");
    PrepArgs p{};
    for (int i = 0; i < n_layers; ++i) {
        p.wt[i] = wts[i];
        p.out[i] = reinterpret_cast<uint4 *>(preps[i]);
    }
    const int conv_blocks = 16 * n_layers;
    ARVAE_LAUNCH(prep_all_kernel, dim3(conv_blocks + mid.blk_end[mid.count - 1]), dim3(256), 0, s, p, mid, conv_blocks);
    return check_launch("weight_prep(conv32 + latent block)");
}

// row-stream weight gradient with producer / consumer waves (conv32r.hip): the 16x16 and 8x8 layers
bool conv32_wgrad_stream_fits(const arvae_link_t *l);
int conv32_wgrad_stream_groups(const arvae_link_t *l);
int conv32_wgrad_stream(const arvae_link_t *l, const Operand &lo, const Operand &hi, float *slab, int bias_mode, hipStream_t s);

int conv32_wgrad_groups(const arvae_link_t *l) {
    if (conv32_wgrad_stream_fits(l)) return conv32_wgrad_stream_groups(l);
    // the patch-staged three-term kernel (4x4 layers; every size with ARVAE_WGRAD_NO_STREAM): 64-pixel tiles, one persistent
    // workgroup per CU, one 64 KB partial each
    const int tiles = l->lh == 16 ? tiles_for<16, 64>(l->n) : l->lh == 8 ? tiles_for<8, 64>(l->n) : tiles_for<4, 64>(l->n);
    return grid_for_tiles(tiles);
}

int64_t conv32_wgrad_ws_floats(const arvae_link_t *l) {
    return (int64_t)conv32_wgrad_groups(l) * WG32_SLAB;
}

template <int LO> static int launch_wgrad_x(const arvae_link_t *l, const Operand &lo, const Operand &hi, float *slab, int bias_mode,
                                            int grid, hipStream_t s) {
    constexpr int LDS = 3 * (PatchLoader<LO, 2, 64>::PLANE_DW / PSB * WGRAD_PSB_H + 64 * WGRAD_PSB_L) * 4;
    const int tiles = tiles_for<LO, 64>(l->n);
    static bool attr = false;
    if (!attr) {
        allow_lds(wgrad32x_kernel<LO, 0>, LDS);
        allow_lds(wgrad32x_kernel<LO, 1>, LDS);
        allow_lds(wgrad32x_kernel<LO, 2>, LDS);
        attr = true;
    }
    if (bias_mode == 1)
        ARVAE_LAUNCH((wgrad32x_kernel<LO, 1>), dim3(grid), dim3(256), LDS, s, lo.v, hi.v, slab, l->n, tiles);
    else if (bias_mode == 2)
        ARVAE_LAUNCH((wgrad32x_kernel<LO, 2>), dim3(grid), dim3(256), LDS, s, lo.v, hi.v, slab, l->n, tiles);
    else
        ARVAE_LAUNCH((wgrad32x_kernel<LO, 0>), dim3(grid), dim3(256), LDS, s, lo.v, hi.v, slab, l->n, tiles);
    return check_launch(LO == 16 ? "wgrad32_kernel<16>" : LO == 8 ? "wgrad32_kernel<8>" : "wgrad32_kernel<4>");
}

// per-workgroup partial sums into `slab`; the returned job describes the reduction that finishes the layer
// bias_mode: 0 none, 1 dbias[clo] += sum lo, 2 dbias[chi] += sum hi
int conv32_wgrad_partial(const arvae_link_t *l, const Operand &lo, const Operand &hi, float *dwt, float *dbias, int bias_mode,
                         float *slab, hipStream_t s, SlabJob *job) {
    const int grid = conv32_wgrad_groups(l);
    const int rc = conv32_wgrad_stream_fits(l) ? conv32_wgrad_stream(l, lo, hi, slab, bias_mode, s)
                   : l->lh == 16             ? launch_wgrad_x<16>(l, lo, hi, slab, bias_mode, grid, s)
                   : l->lh == 8              ? launch_wgrad_x<8>(l, lo, hi, slab, bias_mode, grid, s)
                                             : launch_wgrad_x<4>(l, lo, hi, slab, bias_mode, grid, s);
    *job = SlabJob{slab, dwt, bias_mode ? dbias : nullptr, grid, SLAB_C32, bias_mode};
    return rc;
}

// ---- 4x4 layers: gated data gradient + weight-gradient partials of one layer in one launch (pair4_*_kernel) ---------------
// up == true: the layer is a forward UP link (data gradient = DOWN map on g, weight gradient with g on the hi side, bias mode 2);
// up == false: a forward DOWN link (data gradient = UP map, g on the lo side, bias mode 1).  Only the combinations the image
// executor produces are instantiated; everything else (and the experiment switches) goes the two-launch way.
bool conv32_pair4_fits(const arvae_link_t *l, bool up, const float *gate, const uint16_t *gate_bits, const float *wprep, int bias_mode) {
    static const bool off = getenv("ARVAE_NO_PAIR4") != nullptr || getenv("ARVAE_NO_SMALL_TILES") != nullptr;
    if (off || l->lh != 4 || (gate == nullptr && gate_bits == nullptr) || bias_mode != (up ? 2 : 1)) return false;
    if (!up && wprep == nullptr) return false;                   // (the UP kernel would stage its weights through LDS)
    const int wg_tiles = tiles_for<4, 64>(l->n), dg_tiles = tiles_for<4, 32>(l->n);
    return wg_tiles >= 8 && wg_tiles + (dg_tiles + 1) / 2 <= cu_count() && conv32_wgrad_groups(l) == wg_tiles;
}

int conv32_pair4(const arvae_link_t *l, bool up, const float *g, const float *x_in, const float *wt, const float *gate,
                 const uint16_t *gate_bits, float *d_in, const float *wprep, float *dwt, float *dbias, float *slab, hipStream_t s,
                 SlabJob *job) {
    constexpr int LDS_W = 3 * (PatchLoader<4, 2, 64>::PLANE_DW / PSB * WGRAD_PSB_H + 64 * WGRAD_PSB_L) * 4;
    constexpr int LDS_D = (PatchLoader<4, 2, 32>::PATCH_FLOATS + 4 * 16 * 64) * 4;
    constexpr int LDS_U = MaxOf<3 * PatchLoader<4, 1, 32>::PLANE_DW, WSTAGE_UP>::value * 4;
    constexpr int LDS = MaxOf<LDS_W, MaxOf<LDS_D, LDS_U>::value>::value;
    const int wg_tiles = tiles_for<4, 64>(l->n), dg_tiles = tiles_for<4, 32>(l->n), grid_a = (dg_tiles + 1) / 2;
    Ep32 ep{nullptr, gate_bits ? nullptr : gate, gate_bits, nullptr, d_in, reinterpret_cast<const uint4 *>(wprep)};
    static bool attr = false;
    if (!attr) {
        allow_lds(pair4_down_kernel<EP_GATE_F, 2>, LDS);
        allow_lds(pair4_down_kernel<EP_GATE_B, 2>, LDS);
        allow_lds(pair4_up_kernel<EP_GATE_F, 1>, LDS);
        allow_lds(pair4_up_kernel<EP_GATE_B, 1>, LDS);
        attr = true;
    }
    const dim3 grid(grid_a + wg_tiles);
    if (up) {                                                    // DOWN map of g (hi side); weight gradient: lo = layer input, hi = g
        if (gate_bits) ARVAE_LAUNCH((pair4_down_kernel<EP_GATE_B, 2>), grid, dim3(256), LDS, s, g, wt, ep, x_in, g, slab, l->n, dg_tiles, wg_tiles, grid_a);
        else ARVAE_LAUNCH((pair4_down_kernel<EP_GATE_F, 2>), grid, dim3(256), LDS, s, g, wt, ep, x_in, g, slab, l->n, dg_tiles, wg_tiles, grid_a);
    } else {                                                     // UP map of g (lo side); weight gradient: lo = g, hi = layer input
        if (gate_bits) ARVAE_LAUNCH((pair4_up_kernel<EP_GATE_B, 1>), grid, dim3(256), LDS, s, g, wt, ep, g, x_in, slab, l->n, dg_tiles, wg_tiles, grid_a);
        else ARVAE_LAUNCH((pair4_up_kernel<EP_GATE_F, 1>), grid, dim3(256), LDS, s, g, wt, ep, g, x_in, slab, l->n, dg_tiles, wg_tiles, grid_a);
    }
    *job = SlabJob{slab, dwt, dbias, wg_tiles, SLAB_C32, up ? 2 : 1};
    return check_launch(up ? "pair4(down32 + wgrad32)" : "pair4(up32 + wgrad32)");
}

int conv32_wgrad(const arvae_link_t *l, const Operand &lo, const Operand &hi, float *dwt, float *dbias, int bias_mode,
                 float *slab, hipStream_t s) {
    SlabJob job;
    if (int rc = conv32_wgrad_partial(l, lo, hi, dwt, dbias, bias_mode, slab, s, &job)) return rc;
    return slab_reduce(job, s);
}

}  // namespace arvae

#ifdef ARVAE_STAMPS
extern "C" int arvae_debug_stamps(unsigned long long *out, int count) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(arvae::g_stamps), sizeof(unsigned long long) * count);
}
#endif
