// Specialised gfx950 kernels for the dSprites stack's 32-channel links (kernel 4x4, stride 2, pad 1):
// the six Conv2d / ConvTranspose2d layers between the 32x32, 16x16, 8x8 and 4x4 feature maps, which
// carry 85 % of the training step's FLOPs.  Same math and C-ABI as the
// generic gather-GEMM (link_gemm.hip); arvae_link_down/up/wgrad dispatch here when the geometry fits.
//
// The kernels run the fp16 MFMA on scaled two-term operands (conv32_common.h): every fp32 operand is scaled by its tensor's
// power-of-two scale and split into two fp16 terms when it enters LDS (weights: once per step, prep32.h) and every multiply-add
// is three partial products on v_mfma_f32_32x32x16_f16 with fp32 accumulation: the result is within one fp32 rounding of an
// fp32 FMA chain's at a fifth of the fp32 MFMA's cycles, and the fp16 MFMA leaves the vector ALU free for the splitting.
// (Generations removed on the way: the same tilings on v_mfma_f32_32x32x2_f32, 45-50 us per 16x16-layer launch; a two-term
// bf16 experiment, 16 bits: flipped ReLU units; three-term bf16 with six products, 26-33 us per 16x16-layer launch, round 3.)
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
#include <mutex>
#include "diag.h"
#include "common.h"
#include "conv32_common.h"
#include "reduce.h"
#include "midprep.h"
#include "prep32.h"
#include "regloss.h"
#include "down32p.h"
#include "wgrad32r.h"

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
    // Split commit: every fp32 value x becomes the fp16 pair (h, l) of s x (split_pair_h2); two planes of PLANE_DW dwords, a
    // pixel is PITCH dwords per plane (32 channels x 2 bytes + pad), channel pair (2i, 2i+1) shares a dword, even channel in the
    // low half.  BIAS_SUM: also accumulate the pixels this tile owns (not the halo) per channel chunk q = threadIdx.x & 7.
    static constexpr int PLANE_DW = T::TI * PR * PC * PSB;
    template <bool BIAS_SUM = false, int PITCH = PSB>
    __device__ __forceinline__ void commit_split2(unsigned *planes, float sc, float4 *bsum = nullptr) const {
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
                uint2 hv, lv;
                split_pair_h2(v.x, v.y, sc, hv.x, lv.x);
                split_pair_h2(v.z, v.w, sc, hv.y, lv.y);
                *reinterpret_cast<uint2 *>(planes + pix * PITCH + q * 2) = hv;
                *reinterpret_cast<uint2 *>(planes + PLANE_P + pix * PITCH + q * 2) = lv;
                if (BIAS_SUM) {
                    const int pc = pix % PC, pr = (pix / PC) % PR;
                    if (pr >= 1 && pr <= PR - 2 && pc >= 1 && pc <= PC - 2) {
                        bsum->x += v.x; bsum->y += v.y; bsum->z += v.z; bsum->w += v.w;
                    }
                }
            }
        }
    }
    // packed two-term image (PSB2)
    static constexpr int BUF2_DW = T::TI * PR * PC * PSB2;
    __device__ __forceinline__ void commit_all2p(unsigned *buf, float sc) const {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int idx = threadIdx.x + it * 256;
            if (idx < SLOTS) {
                const int q = idx & 7, pix = idx >> 3;
                uint2 hv, lv;
                split_pair_h2(r[it].x, r[it].y, sc, hv.x, lv.x);
                split_pair_h2(r[it].z, r[it].w, sc, hv.y, lv.y);
                unsigned *d = buf + pix * PSB2 + q * 2;
                *reinterpret_cast<uint2 *>(d) = hv;
                *reinterpret_cast<uint2 *>(d + 16) = lv;
            }
        }
    }
};

#ifdef ARVAE_STAMPS
// diagnostic build only (tools/stamp_conv32.py): per-workgroup phase timeline, 64 slots, s_memtime + wall clock
__device__ unsigned long long g_stamps[512 * 64 * 2];
#define STAMP(slot)                                                                      \
    do {                                                                                 \
        if (LO == 16 && threadIdx.x == 0 && BID < 512 && (slot) < 64) {                  \
            g_stamps[(BID * 64 + (slot)) * 2] = __builtin_readcyclecounter();            \
            g_stamps[(BID * 64 + (slot)) * 2 + 1] = wall_clock64();                      \
        }                                                                                \
    } while (0)
#define STAMP_WAIT() __builtin_amdgcn_s_waitcnt(0)
// up32p_kernel: row blockIdx.x = its consumers (thread 0), row 256 + blockIdx.x = its producers (thread 256)
#define PSTAMP(role, slot)                                                               \
    do {                                                                                 \
        if (threadIdx.x == 256 * (role) && BID < 256 && (slot) < 64) {                   \
            g_stamps[((BID + 256 * (role)) * 64 + (slot)) * 2] = __builtin_readcyclecounter();            \
            g_stamps[((BID + 256 * (role)) * 64 + (slot)) * 2 + 1] = wall_clock64();                      \
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

// one channel group g (4 channels) of one pixel: accumulator * inv (the operands' inverse scales) + bias -> ReLU / gate -> one
// 16-byte store at off + g*32; returns the group's four sign bits (for EP_RELU's bits_out); `amax` collects the magnitude of
// what is stored when `live` (a pipelined epilogue's first pass stores nothing: its offset is out of range)
template <int MODE>
__device__ __forceinline__ unsigned store_group(const f32x16 &acc, int g, float inv, const float4 &b, const float4 &gv, unsigned gbits,
                                                __amdgpu_buffer_rsrc_t rs_out, unsigned off, bool live, float &amax) {
    float v[4] = {fmaf(acc[4 * g], inv, b.x), fmaf(acc[4 * g + 1], inv, b.y), fmaf(acc[4 * g + 2], inv, b.z), fmaf(acc[4 * g + 3], inv, b.w)};
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
    const float4 o = make_float4(v[0], v[1], v[2], v[3]);
    amax = live ? fmaxf(amax, amax4(o)) : amax;
    buf_store4(o, rs_out, off + g * 32);
    return bits;
}

// all four groups of one pixel, gate fetched here (the non-pipelined epilogues)
template <int MODE>
__device__ __forceinline__ void store_pixel(const f32x16 &acc, float inv, const float4 (&b4)[4], __amdgpu_buffer_rsrc_t rs_out,
                                            __amdgpu_buffer_rsrc_t rs_gate, __amdgpu_buffer_rsrc_t rs_bits, bool want_bits,
                                            unsigned off, int half, bool live, float &amax) {
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
    for (int g = 0; g < 4; ++g) bits |= store_group<MODE>(acc, g, inv, b4[g], gv[g], gb, rs_out, off, live, amax);
    if (MODE == EP_RELU && want_bits) buf_store_u16(bits, rs_bits, bits_off(off, half));
}

__device__ __forceinline__ void load_bias4(const float *bias, int half, float4 (&b4)[4]) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
        b4[g] = bias != nullptr ? *reinterpret_cast<const float4 *>(bias + 8 * g + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
}

__device__ __forceinline__ f16x8 lds_f16x8(const unsigned *p) {
    return __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4v *>(p));
}

__global__ __launch_bounds__(256) void conv32_weight_prep_kernel(PrepArgs p) { conv32_prep_block(p, blockIdx.x); }

// the step's two weight preps as one launch: workgroups [0, conv_blocks) split the 32-channel conv weights, the rest lay out the
// latent block's matrices (midprep.h)
__global__ __launch_bounds__(256) void prep_all_kernel(PrepArgs p, MidPrepArgs mid, int conv_blocks) {
    prep_all_block(p, mid, conv_blocks, blockIdx.x);
}

// maximum magnitude of a tensor the caller brought along without one (the per-layer C-ABI entry points): AMAX array of `x`
__global__ __launch_bounds__(256) void amax_kernel(const float *__restrict__ x, int64_t count4, unsigned *__restrict__ out) {
    __shared__ float wm[4];
    float m = 0.f;
    // four loads in flight per lane (one at a time, a 25 MB tensor took 21 us: 1.2 TB/s)
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count4; i += 4 * stride) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t j = i + u * stride;
            v[u] = reinterpret_cast<const float4 *>(x)[j < count4 ? j : i];
        }
        m = fmaxf(fmaxf(m, amax4(v[0])), fmaxf(fmaxf(amax4(v[1]), amax4(v[2])), amax4(v[3])));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x < 64) amax_publish(out, blockIdx.x, gridDim.x, fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])));
}

// ================================================================================================
// Down, small-problem variant (the 4x4 layers at batch 512: too few 64-pixel tiles to fill or pipeline the CUs): a tile is 32
// lo pixels (two whole images), wave w takes kernel row ky = w (K split 4 ways, 16 prepared-weight loads per lane), the four
// partial tiles meet in LDS and wave 0 runs the epilogue.  (The 16x16 and 8x8 layers: down32p_kernel, conv32k.hip.)
template <int LO, int MODE>
__device__ __forceinline__ void down32s_body(const float *__restrict__ hi, Ep32 ep, int n_img, int n_tiles, const int BID, const int NBLK) {
    constexpr int PX = 32;
    using PL = PatchLoader<LO, 2, PX>;
    constexpr int PC = PL::PC, PR = PL::PR;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // packed two-term patch (BUF2_DW) | red[4][16][64]
    unsigned *ldsw = reinterpret_cast<unsigned *>(lds);
    float *red = lds + PL::BUF2_DW;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, rc = lane & 31;

    const AmaxLoad al = amax_issue(ep.amax_in);
    PL pl;
    pl.init(hi, n_img);
    int img0, r0;
    tile_origin<LO, PX>(BID, img0, r0);
    pl.set_tile(img0, r0, BID < n_tiles);
    pl.issue_all();

    // w2[kx][c][term]: input channels c*16 + half*8 + j of wt[clo = rc][.][ky = wave][kx]
    f16x8 w2[4][2][2];
    {
        const uint4 *src = ep.wprep + ((wave >> 1) * PREP_DOWN_SLOTS + (wave & 1) * 16) * 64 + lane;
#pragma unroll
        for (int kx = 0; kx < 4; ++kx)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int t = 0; t < 2; ++t) w2[kx][cc][t] = __builtin_bit_cast(f16x8, src[((kx * 2 + cc) * 2 + t) * 64]);
    }
    const Pow2 sc = amax_scale(al);
    const float inv = sc.inv * prep_inv_scale(ep.wprep);
    int img, r, c;
    tile_pixel<LO, PX>(rc, img, r, c);
    const int aoff = ((img * PR + 2 * r + wave) * PC + 2 * c) * PSB2 + half * 4;      // + kx*PSB2 + c*8 + term*16
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
    float amax_run = 0.f;

    for (int tile = BID; tile < n_tiles; tile += NBLK) {
        tile_origin<LO, PX>(tile, img0, r0);
        __syncthreads();                                         // the previous tile's patch and partials have been read
        pl.commit_all2p(ldsw, sc.s);
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
            for (int cc = 0; cc < 2; ++cc) {
                const f16x8 ah = lds_f16x8(ldsw + aoff + kx * PSB2 + cc * 8), al = lds_f16x8(ldsw + aoff + kx * PSB2 + cc * 8 + 16);
                MFMA_H(acc, w2[kx][cc][1], ah);                  // smallest partial products first
                MFMA_H(acc, w2[kx][cc][0], al);
                MFMA_H(acc, w2[kx][cc][0], ah);
            }
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) red[(wave * 16 + reg) * 64 + lane] = acc[reg];
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg)
                acc[reg] = (acc[reg] + red[(16 + reg) * 64 + lane]) + (red[(32 + reg) * 64 + lane] + red[(48 + reg) * 64 + lane]);
            store_pixel<MODE>(acc, inv, b4, rs_out, rs_gate, rs_bits, want_bits,
                              out_lane + (unsigned)(((img0 * LO + r0) * LO) * PIXB), half, true, amax_run);
        }
    }
    if (wave == 0) amax_publish(ep.amax_out, BID, NBLK, amax_run);
}
template <int LO, int MODE>
__global__ __launch_bounds__(256, 2) void down32s_kernel(const float *__restrict__ hi, Ep32 ep, int n_img, int n_tiles) {
    down32s_body<LO, MODE>(hi, ep, n_img, n_tiles, blockIdx.x, gridDim.x);
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
// Up on the fp16 MFMA with scaled two-term operands (conv32_common.h): 3 x 32 cycles per 16 channels.  64 weights x 2 terms
// live in 64 registers (prepared per step, prep32.h); staggered epilogue and gating as described above.
#ifndef ARVAE_UP_ISSUE_STEPS
#define ARVAE_UP_ISSUE_STEPS 4
#endif
constexpr int UP_ISSUE_STEPS = ARVAE_UP_ISSUE_STEPS;
template <int LO, int MODE, int PX = 128>
__device__ __forceinline__ void up32x_body(const float *__restrict__ lo, Ep32 ep, int n_img, int n_tiles, const int BID, const int NBLK) {
    using PL = PatchLoader<LO, 1, PX>;
    constexpr int MT = PX / 32;
    constexpr int HI = 2 * LO, PC = PL::PC, PR = PL::PR, PLANE = PL::PLANE_DW;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // 2 planes
    unsigned *ldsw = reinterpret_cast<unsigned *>(lds);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, rc = lane & 31;
    const int py = wave >> 1, px = wave & 1;
    STAMP(0);

    const AmaxLoad al = amax_issue(ep.amax_in);
    PL pl;                                                       // first tile's loads fly while the weights arrive
    pl.init(lo, n_img);
    int img0, r0;
    // a workgroup owns a contiguous run of tiles (the row groups of the same images): the halo rows two tiles share come from
    // this XCD's L2 the second time
    const int per_wg = (n_tiles + (int)NBLK - 1) / (int)NBLK, t_first = BID * per_wg;
    const int t_end = min(n_tiles, t_first + per_wg);
    tile_origin<LO, PX>(t_first, img0, r0);
    pl.set_tile(img0, r0, t_first < t_end);
    pl.issue_all();

    // w2[ty][tx][c][term]: the 8 input channels c*16 + half*8 + j of wt[.][chi = rc][1 - py + 2ty][1 - px + 2tx]
    f16x8 w2[2][2][2][2];
    {
        const uint4 *src = ep.wprep + PREP_DOWN_UINT4 + (wave * PREP_UP_SLOTS) * 64 + lane;
#pragma unroll
        for (int ty = 0; ty < 2; ++ty)
#pragma unroll
            for (int tx = 0; tx < 2; ++tx)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        w2[ty][tx][cc][t] = __builtin_bit_cast(f16x8, src[(((ty * 2 + tx) * 2 + cc) * 2 + t) * 64]);
    }
    const Pow2 sc = amax_scale(al);
    const float inv = sc.inv * prep_inv_scale(ep.wprep);

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
    float amax_run = 0.f;
    STAMP(2);
    int stamp_t = 0;
    (void)stamp_t;

    // The commit of tile t + 1 closes the body of tile t (the first one is peeled): at the loop header the entry path (loads
    // youngest) and the back edge (stores youngest) would otherwise merge into a vmcnt count that waits for the stores too.
    __syncthreads();
    pl.commit_split2(ldsw, sc.s);
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
        // operands of step 0; every later step's 2 * MT reads are issued two at a time behind the MFMA triples of the step
        // before (two operand sets): a burst of LDS reads in front of a step costs the matrix pipe ~10 idle cycles per read plus
        // the LDS round trip (tools/probes/mfma_barrier.hip)
        f16x8 a[2][MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < 2; ++t) a[0][mt][t] = lds_f16x8(ldsw + t * PLANE + aoff[mt]);
        static_for<0, 8>([&](auto sc_) __attribute__((always_inline)) {
            constexpr int step = decltype(sc_)::value, ty = step >> 2, tx = (step >> 1) & 1, c = step & 1;
            constexpr int cur = step & 1, nxt = cur ^ 1;
            constexpr int nstep = step + 1, nty = nstep >> 2, ntx = (nstep >> 1) & 1, nc = nstep & 1;
            constexpr int ntoff = -(nty * PC + ntx) * PSB + nc * 8;
            // the next tile's loads all leave in the first half of this tile: the commit at the top of the next tile waits for
            // them (in-order vmcnt), and a load issued in the last step would expose its whole HBM round trip there
            if constexpr (step < UP_ISSUE_STEPS) pl.template issue_step<UP_ISSUE_STEPS, step>();
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, MT>([&](auto mc) __attribute__((always_inline)) {
                constexpr int sub = decltype(mc)::value;
                // epilogue slots: one per two groups of three MFMAs, at the same program point in every wave (a wave-index
                // branch around them made every vmcnt count behind it conservative)
                constexpr int slot = step * MT + sub;
                if constexpr ((slot & 1) == 0) {
                    constexpr int k = slot >> 1, em = k >> 2, eg = k & 3;
                    const unsigned poff = prev_base + orel[em];
                    if (eg == 0) pbits[em] = 0;
                    pbits[em] |= store_group<MODE>(prev[em], eg, inv, b4[eg], gq[k], gqb[em], rs_out, poff, prev_base != OOB, amax_run);
                    // (unconditional: a store under a run-time branch makes the compiler's vmcnt count at the next tile's
                    // commit conservative -- it then waits for every store of this tile; without bits_out the offset is out of range)
                    if (MODE == EP_RELU && eg == 3)
                        buf_store_u16(pbits[em], rs_bits, (prev_base == OOB || !want_bits) ? OOB : bits_off(poff, half));
                    if (MODE == EP_GATE_F) gq[k] = buf_load4(rs_gate, obase + orel[em] + eg * 32);
                    // this tile's sign bits are requested in its first slots: the copies into gqb at the bottom of the loop then
                    // wait for loads that are two steps old, not for the one issued in the last step (plus every store before it)
                    if constexpr (MODE == EP_GATE_B && k < MT) gqn[k] = buf_load_u16(rs_bits, bits_off(obase + orel[k], half));
                }
                // Three of the step's 3 * MT MFMAs, taken in ROUND-ROBIN order over the MT accumulators (product-major: every
                // accumulator still sees its three partial products smallest first): back-to-back MFMAs into the same
                // accumulator cost ~48 cycles each instead of the 32-cycle issue rate (conv32r.hip).
                static_for<3 * sub, 3 * sub + 3>([&](auto qc) __attribute__((always_inline)) {
                    constexpr int m = decltype(qc)::value, prod = m / MT, pm = m % MT;
                    constexpr int tw = prod == 0 ? 1 : 0, ta = prod == 1 ? 1 : 0;      // (l, h), (h, l), (h, h)
                    MFMA_H(acc[pm], w2[ty][tx][c][tw], a[cur][pm][ta]);
                });
                if constexpr (nstep < 8) {                       // this sub-step's share of the next step's operand reads
                    static_for<2 * sub, 2 * sub + 2>([&](auto rc_) __attribute__((always_inline)) {
                        constexpr int ri = decltype(rc_)::value, rmt = ri / 2, rt = ri % 2;
                        a[nxt][rmt][rt] = lds_f16x8(ldsw + rt * PLANE + aoff[rmt] + ntoff);
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
        pl.commit_split2(ldsw, sc.s);
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
        for (int g = 0; g < 4; ++g) bits |= store_group<MODE>(prev[mt], g, inv, b4[g], gq[mt * 4 + g], gqb[mt], rs_out, poff, prev_base != OOB, amax_run);
        if (MODE == EP_RELU) buf_store_u16(bits, rs_bits, (prev_base == OOB || !want_bits) ? OOB : bits_off(poff, half));
    }
    amax_publish(ep.amax_out, BID * 4 + wave, NBLK * 4, amax_run);
    STAMP_WAIT();
    STAMP(63);
}
template <int LO, int MODE, int PX = 128>
__global__ __launch_bounds__(256, 1) void up32x_kernel(const float *__restrict__ lo, Ep32 ep, int n_img, int n_tiles) {
    up32x_body<LO, MODE, PX>(lo, ep, n_img, n_tiles, blockIdx.x, gridDim.x);
}

// ================================================================================================================================
// Up map of the 16x16 layers with COMPUTE and STORE waves.
// What bounds this map is its output: 64 KB per 128-pixel tile against ~16 bytes per clock of store path per CU = 2 us per tile,
// 1.5 us of MFMA issue.  In up32x_kernel every wave loads, splits, multiplies and stores, and the stores of a tile are
// issued between the MFMAs of the next.  Here
//   * waves 0-3 (compute, wave = parity class) keep the weights, request the next tile's patch at the top of a tile, multiply
//     (operand reads of the next step between the MFMAs and nothing else), then split the patch that has arrived into the
//     other LDS image and park the finished accumulator tiles in LDS (64 KB);
//   * waves 4-7 (store) do nothing but epilogues from there: scale, bias, ReLU / sign-bit gate, 16-byte stores, sign bits, the
//     output's maximum -- the store path is busy for all but the hand-over.
// Two barriers per tile.  (Round 3's first form had the patch loading and splitting on the store waves: they were busy 3.2 us
// per tile, the compute waves 2.4 -- and as a gated data gradient its loader waited for the far patch loads, which the compute
// waves now request a whole tile ahead.)  profiles/r3_phase_stamps.txt
// C1W (round 5): this map is the data gradient of the layer BEHIND a single-channel first layer (Conv2d(1, 32, 4, 2, 1),
// imagevae/dsprites_vae.py:12-13), and all that layer's backward pass wants from the 67 MB this map would write is its own weight
// gradient dwt1[c][ky][kx] = sum over images and hi pixels (Y, X) of out[Y][X][c] * img[2 Y - 1 + ky][2 X - 1 + kx] (+ the bias sums).
// The store waves then store NOTHING: each takes the gated results it holds (a 32-pixel x 32-channel block per M-tile) through a
// wave-private LDS image into the fp16 MFMA with pixels as the K axis -- two-term operands, three products, the block's scale the
// running maximum of what the wave has seen (a power of two: the accumulator is rescaled exactly when it grows) -- against the
// image taps of those pixels, and leaves one [32][16] + 32 partial per workgroup (reduce.h, SLAB_C1).
struct C1Wgrad {
    const float *img;            // [n_img][64][64]: the first layer's input; |img| < 64 (a fixed operand scale of 2^10)
    float *slab;                 // [workgroups of this body][SLAB_C1_FLOATS]
};
constexpr int C1W_VP = 16, C1W_XP = 16;                         // dwords per pixel of a wave's result / tap image (one fp16 term; taps 0-15, the constant 1 as "tap" 16, zeros)
constexpr int C1W_WAVE_DW = 2 * 32 * C1W_VP + 2 * 32 * C1W_XP;   // per store wave: two terms of each
constexpr float C1W_SX = 1024.f, C1W_SX_INV = 1.f / 1024.f;

template <int MODE, bool C1W = false>
__device__ __forceinline__ void up32p_body(const float *__restrict__ lo, Ep32 ep, int n_img, int n_tiles, const int BID, const int NBLK,
                                           C1Wgrad c1 = C1Wgrad{nullptr, nullptr}) {
    static_assert(MODE != EP_GATE_F, "up32p_kernel: float gates stay on up32x_kernel");
    static_assert(!C1W || MODE == EP_GATE_B, "the fused first-layer weight gradient takes the sign-bit gated data gradient");
    constexpr int LO = 16, PX = 128, HI = 2 * LO, MT = PX / 32;
    using PL = PatchLoader<LO, 1, PX>;
    constexpr int PR = PL::PR, PC = PL::PC, PLANE = PL::PLANE_DW, SLOTS = PL::SLOTS, ITERS = PL::ITERS;
    constexpr int BUF = 2 * PLANE;                               // dwords of one two-plane patch image
    extern __shared__ __attribute__((aligned(16))) float lds_f[];                 // [2][BUF] patch images | handoff
    unsigned *lds = reinterpret_cast<unsigned *>(lds_f);
    float4 *hand = reinterpret_cast<float4 *>(lds + 2 * BUF);    // [wave][mt][group][lane]
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, rc = lane & 31;
    const int per_wg = (n_tiles + NBLK - 1) / NBLK, t_first = BID * per_wg;
    const int t_end = min(n_tiles, t_first + per_wg);
    const int cls = wave & 3, py = cls >> 1, px = cls & 1;       // parity class of a compute wave / of the store wave that finishes it

    if (wave >= 4) {
        // ============================================================================================ store waves
        PSTAMP(1, 0);
        int pst = 0;
        (void)pst;
        const AmaxLoad al = amax_issue(ep.amax_in);
        unsigned orel[MT];                                       // output byte offset of this lane's pixel in M-tile mt
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            int img, r, c;
            tile_pixel<LO, PX>(mt * 32 + rc, img, r, c);
            orel[mt] = (unsigned)(((img * HI + 2 * r + py) * HI + 2 * c + px) * PIXB + half * 16);
        }
        const int64_t out_bytes = (int64_t)n_img * HI * HI * PIXB;
        const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(ep.out, out_bytes);
        const bool want_bits = MODE == EP_RELU && ep.bits_out != nullptr;
        const __amdgpu_buffer_rsrc_t rs_bits =
            make_rsrc(MODE == EP_GATE_B ? (const void *)ep.gate_bits : want_bits ? (const void *)ep.bits_out : (const void *)ep.out,
                      (int64_t)n_img * HI * HI * 4);
        float4 b4[4];
        load_bias4(ep.bias, half, b4);
        const float inv = amax_scale(al).inv * prep_inv_scale(ep.wprep);
        float amax_run = 0.f;
        auto tile_base = [&](int tile) -> unsigned {
            int img0, r0;
            tile_origin<LO, PX>(tile, img0, r0);
            return tile < t_end ? (unsigned)(((img0 * HI + 2 * r0) * HI) * PIXB) : OOB;
        };
        // sign bits that gate a tile's outputs: requested one tile ahead (before the stores of the tile in hand: memory returns
        // them first, and the wait in front of their use leaves those stores in flight)
        unsigned gb[MT] = {}, gbn[MT] = {};
        auto request_gates = [&](unsigned base) __attribute__((always_inline)) {
            if (MODE == EP_GATE_B) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) gbn[mt] = buf_load_u16(rs_bits, base == OOB ? OOB : bits_off(base + orel[mt], half));
            }
        };
        // epilogue of the tile whose accumulators sit in the handoff area: this thread = lane `lane` of compute wave `cls`
        auto epilogue = [&](unsigned base) __attribute__((always_inline)) {
            const float4 no_gate = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                f32x16 acc;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 v = hand[((cls * MT + mt) * 4 + g) * 64 + lane];
                    acc[4 * g] = v.x; acc[4 * g + 1] = v.y; acc[4 * g + 2] = v.z; acc[4 * g + 3] = v.w;
                }
                const unsigned poff = base + orel[mt];
                unsigned bits = 0;
#pragma unroll
                for (int g = 0; g < 4; ++g) bits |= store_group<MODE>(acc, g, inv, b4[g], no_gate, gb[mt], rs_out, poff, base != OOB, amax_run);
                if (MODE == EP_RELU) buf_store_u16(bits, rs_bits, (base == OOB || !want_bits) ? OOB : bits_off(poff, half));
            }
        };
        // ---- C1W: the first layer's weight gradient from the results this wave holds (see C1Wgrad) ----
        unsigned *c1_v = lds + 2 * BUF + 4 * MT * 4 * 64 * 4 + (wave - 4) * C1W_WAVE_DW;     // behind the handoff area: [term][pixel][C1W_VP]
        unsigned *c1_x = c1_v + 2 * 32 * C1W_VP;                                             //                          [term][pixel][C1W_XP]
        f32x16 c1_acc, c1_acc2;
#pragma unroll
        for (int i = 0; i < 16; ++i) { c1_acc[i] = 0.f; c1_acc2[i] = 0.f; }
        const __amdgpu_buffer_rsrc_t rs_img = make_rsrc(C1W ? c1.img : ep.out, C1W ? (int64_t)n_img * 4 * HI * HI * 4 : 0);
        // transposed-read addresses (wgrad32x_body): 16-lane group g16 reads pixels 16 b + 8 (g16 >> 1) + 4 i + q; lane 4 q + p of
        // the group points at channels 4 p .. 4 p + 3 (+ 16 (g16 & 1)) of pixel row q; the tap image's "channels" 16-31 are the constant 1
        // and zeros: accumulator column 16 collects the bias sums
        int c1_voff[2][2], c1_xoff[2][2];
        const int c1_sw = 2 * ((rc >> 2) & 7);                   // this lane's row swizzle (its pixel row is rc)
        {
            const int g16 = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int P = 16 * b + 8 * (g16 >> 1) + 4 * i + q;
                    // (rows are 16 dwords = a quarter of the banks apart, so the WRITES -- every lane its own pixel row -- would meet
                    // eight to a bank; the dword offset inside a row is XORed with 2 ((P >> 2) & 7): the four rows of a transposed
                    // read share P >> 2 and stay on their four quarters, the 64 row pieces of a write spread over all banks)
                    const int sw = 2 * ((P >> 2) & 7);
                    c1_voff[b][i] = P * C1W_VP + ((8 * (g16 & 1) + 2 * pp) ^ sw);
                    c1_xoff[b][i] = P * C1W_XP + ((8 * (g16 & 1) + 2 * pp) ^ sw);
                }
            if constexpr (C1W) {                                 // the constant half of the tap image: column 16 = 1 (x C1W_SX = 1024 = 0x6400), 17-31 = 0
                const int csw = 2 * ((rc >> 2) & 7);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int col = 8 + 4 * half + d;
                    c1_x[rc * C1W_XP + (col ^ csw)] = col == 8 ? 0x6400u : 0u;
                    c1_x[32 * C1W_XP + rc * C1W_XP + (col ^ csw)] = 0u;
                }
            }
        }
        // the image taps of this lane's hi pixel in M-tile mt: rows ky = 2 half, 2 half + 1, columns 2 X - 1 .. 2 X + 2, requested one
        // tile ahead.  Column -1 (X = 0) and column 64 (X = 31) are outside the image: the first is fetched from column 0 and
        // shifted, the second masked; rows outside read nothing.
        float4 c1_tap[MT][2], c1_tapn[MT][2];
        auto c1_request = [&](int tile) __attribute__((always_inline)) {
            if constexpr (C1W) {
                int img0, r0;
                tile_origin<LO, PX>(tile, img0, r0);
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    int im, r, c;
                    tile_pixel<LO, PX>(mt * 32 + rc, im, r, c);
                    const int Y = 2 * (r0 + r) + py, X = 2 * c + px;
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int iy = 2 * Y - 1 + 2 * half + e, ix = X == 0 ? 0 : 2 * X - 1;
                        const bool ok = tile < t_end && (unsigned)iy < (unsigned)(2 * HI);
                        c1_tapn[mt][e] = buf_load4(rs_img, ok ? (unsigned)((((img0 + im) * 2 * HI + iy) * 2 * HI + ix) * 4) : OOB);
                    }
                }
            }
        };
        // The results' scale is STATIC: an accumulator is a sum of 128 products of operands below 2^15 each, so acc * 2^-23 is below
        // 2^14 whatever the data (typical blocks sit ~2^10 lower: the second fp16 term still has its bits there), and the factor
        // 2^23 * inv goes into the final scaling.  The bias sums ride in the MFMA: the tap image's column 16 is the constant 1
        // (1024 / C1W_SX in its scale), so accumulator column 16 is the sum of the results.
        auto c1_tile = [&](unsigned base) __attribute__((always_inline)) {
            if constexpr (C1W) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    // results -> [pixel rc][channel] (even channel in the low half of a dword), taps -> [pixel rc][tap]
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const float4 a = hand[((cls * MT + mt) * 4 + g) * 64 + lane];
                        const unsigned bits = base != OOB ? gb[mt] >> (4 * g) : 0u;
                        const float v0 = (bits & 1u) ? a.x : 0.f, v1 = (bits & 2u) ? a.y : 0.f, v2 = (bits & 4u) ? a.z : 0.f, v3 = (bits & 8u) ? a.w : 0.f;
                        uint2 hv, lv;
                        split_pair_h2(v0, v1, 0x1p-23f, hv.x, lv.x);
                        split_pair_h2(v2, v3, 0x1p-23f, hv.y, lv.y);
                        const int vd = (4 * g + 2 * half) ^ c1_sw;                     // (an even XOR keeps the dword pair together)
                        *reinterpret_cast<uint2 *>(c1_v + rc * C1W_VP + vd) = hv;
                        *reinterpret_cast<uint2 *>(c1_v + 32 * C1W_VP + rc * C1W_VP + vd) = lv;
                    }
                    {
                        int im, r, c;
                        tile_pixel<LO, PX>(mt * 32 + rc, im, r, c);
                        const int X = 2 * c + px;
                        uint4 hx, lx;
                        unsigned *hp = &hx.x, *lp = &lx.x;
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            float4 t = c1_tap[mt][e];
                            if (X == 0) t = make_float4(0.f, t.x, t.y, t.z);
                            if (X == HI - 1) t.w = 0.f;
                            split_pair_h2(t.x, t.y, C1W_SX, hp[2 * e], lp[2 * e]);
                            split_pair_h2(t.z, t.w, C1W_SX, hp[2 * e + 1], lp[2 * e + 1]);
                        }
                        // (dword pairs: the XOR moves pairs, not quads)
                        *reinterpret_cast<uint2 *>(c1_x + rc * C1W_XP + ((4 * half) ^ c1_sw)) = make_uint2(hx.x, hx.y);
                        *reinterpret_cast<uint2 *>(c1_x + rc * C1W_XP + ((4 * half + 2) ^ c1_sw)) = make_uint2(hx.z, hx.w);
                        *reinterpret_cast<uint2 *>(c1_x + 32 * C1W_XP + rc * C1W_XP + ((4 * half) ^ c1_sw)) = make_uint2(lx.x, lx.y);
                        *reinterpret_cast<uint2 *>(c1_x + 32 * C1W_XP + rc * C1W_XP + ((4 * half + 2) ^ c1_sw)) = make_uint2(lx.z, lx.w);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (wave-private image: its own writes, in order)
                    f16x8 a2[2][2], b2[2][2];
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            a2[b][t] = lds_tr_f16x8(c1_v + t * 32 * C1W_VP + c1_voff[b][0], c1_v + t * 32 * C1W_VP + c1_voff[b][1]);
                            b2[b][t] = lds_tr_f16x8(c1_x + t * 32 * C1W_XP + c1_xoff[b][0], c1_x + t * 32 * C1W_XP + c1_xoff[b][1]);
                        }
                    // (two accumulators: an MFMA into the result of the previous one waits for it)
                    MFMA_H(c1_acc, a2[0][1], b2[0][0]);          // smallest partial products first
                    MFMA_H(c1_acc2, a2[1][1], b2[1][0]);
                    MFMA_H(c1_acc, a2[0][0], b2[0][1]);
                    MFMA_H(c1_acc2, a2[1][0], b2[1][1]);
                    MFMA_H(c1_acc, a2[0][0], b2[0][0]);
                    MFMA_H(c1_acc2, a2[1][0], b2[1][0]);
                }
            }
        };
        request_gates(tile_base(t_first));
        c1_request(t_first);
        __syncthreads();                                         // (the compute waves' prologue barriers)
        __syncthreads();
        for (int tile = t_first; tile < t_end; ++tile) {
            PSTAMP(1, 3 + 6 * pst);
            __syncthreads();                                     // A: (the handoff area is free)
            PSTAMP(1, 4 + 6 * pst);
            __syncthreads();                                     // B: this tile's accumulators are in the handoff area
            PSTAMP(1, 5 + 6 * pst);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) gb[mt] = gbn[mt];
            request_gates(tile_base(tile + 1));
            if constexpr (C1W) {
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) { c1_tap[mt][0] = c1_tapn[mt][0]; c1_tap[mt][1] = c1_tapn[mt][1]; }
                c1_request(tile + 1);
            }
            PSTAMP(1, 6 + 6 * pst);
            if constexpr (C1W) c1_tile(tile_base(tile));
            else epilogue(tile_base(tile));
            PSTAMP(1, 7 + 6 * pst);
            PSTAMP(1, 8 + 6 * pst);
            ++pst;
        }
        if constexpr (C1W) {
            // the four classes' partials meet in the handoff area (free: the compute waves have left their loop); wave 4 finishes.
            // accumulator register i of lane (tap = rc, half): channel 8 (i >> 2) + 4 half + (i & 3); columns (taps) 16-31 are repeats.
            __syncthreads();                                     // (pairs with the compute waves' closing barrier)
            float *fin = reinterpret_cast<float *>(hand);
            const float k = inv * 0x1p+23f * C1W_SX_INV;
#pragma unroll
            for (int i = 0; i < 16; ++i) fin[((wave - 4) * 16 + i) * 64 + lane] = (c1_acc[i] + c1_acc2[i]) * k;
            __syncthreads();
            if (wave == 4) {
                float *out = c1.slab + (int64_t)BID * SLAB_C1_FLOATS;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const float t = (fin[(0 * 16 + i) * 64 + lane] + fin[(1 * 16 + i) * 64 + lane]) + (fin[(2 * 16 + i) * 64 + lane] + fin[(3 * 16 + i) * 64 + lane]);
                    const int ch = 8 * (i >> 2) + 4 * half + (i & 3);
                    if (rc < 16) out[ch * 16 + rc] = t;
                    if (rc == 16) out[32 * 16 + ch] = t;         // (column 16: the results times the constant 1)
                }
                if (lane == 0) out[32 * 16 + 32] = 0.f;
            }
            return;
        }
        amax_publish(ep.amax_out, BID * 4 + cls, NBLK * 4, amax_run);
        PSTAMP(1, 63);
        return;
    }

    // ================================================================================================ compute waves (wave = parity class)
    PSTAMP(0, 0);
    int cst = 0;
    (void)cst;
    const AmaxLoad al = amax_issue(ep.amax_in);
    const __amdgpu_buffer_rsrc_t rs_lo = make_rsrc(lo, (int64_t)n_img * LO * LO * PIXB);
    // patch slots of this thread (as PatchLoader): byte offset relative to the tile's first patch row | bit 0: first patch row,
    // bit 1: last patch row (the only rows that can fall outside the image)
    unsigned rel[ITERS];
    float4 rv[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int idx = threadIdx.x + it * 256;
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
        const int base = ((img0 * LO + gy0) * LO) * PIXB;        // negative for the very first patch row of the tensor
        const unsigned bad = (gy0 < 0 ? 1u : 0u) | (gy0 + PR - 1 >= LO ? 2u : 0u);
        const bool valid = tile < t_end;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const bool row_ok = valid && (rel[it] & bad) == 0;
            rv[it] = buf_load4(rs_lo, row_ok ? (rel[it] & ~3u) + (unsigned)base : OOB);
        }
    };
    float sc_in = 1.f;
    auto commit_tile = [&](unsigned *planes) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int idx = threadIdx.x + it * 256;
            float4 v = rv[it];
            asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));          // every lane uses the slot's registers (exact vmcnt)
            if (idx < SLOTS) {
                const int q = idx & 7, pix = idx >> 3;
                uint2 hv, lv;
                split_pair_h2(v.x, v.y, sc_in, hv.x, lv.x);
                split_pair_h2(v.z, v.w, sc_in, hv.y, lv.y);
                *reinterpret_cast<uint2 *>(planes + pix * PSB + q * 2) = hv;
                *reinterpret_cast<uint2 *>(planes + PLANE + pix * PSB + q * 2) = lv;
            }
        }
    };
    issue_tile(t_first);
    // w2[ty][tx][c][term]: the 8 input channels c*16 + half*8 + j of wt[.][chi = rc][1 - py + 2ty][1 - px + 2tx]
    f16x8 w2[2][2][2][2];
    {
        const uint4 *src = ep.wprep + PREP_DOWN_UINT4 + (wave * PREP_UP_SLOTS) * 64 + lane;
#pragma unroll
        for (int ty = 0; ty < 2; ++ty)
#pragma unroll
            for (int tx = 0; tx < 2; ++tx)
#pragma unroll
                for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        w2[ty][tx][cc][t] = __builtin_bit_cast(f16x8, src[(((ty * 2 + tx) * 2 + cc) * 2 + t) * 64]);
    }
    // patch origin is lo (r0-1, -1); tap (ty,tx) of class (py,px) reads lo (r + py - ty, c + px - tx)
    int aoff[MT];                                                // dwords into a plane
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        int img, r, c;
        tile_pixel<LO, PX>(mt * 32 + rc, img, r, c);
        aoff[mt] = ((img * PR + r + 1 + py) * PC + c + 1 + px) * PSB + half * 4;
    }
    sc_in = amax_scale(al).s;
    __syncthreads();
    commit_tile(lds);
    __syncthreads();                                             // first tile staged
    PSTAMP(0, 1);
    for (int tile = t_first; tile < t_end; ++tile) {
        PSTAMP(0, 3 + 6 * cst);
        const int cur = (tile - t_first) & 1;
        const unsigned *ldsw = lds + cur * BUF;
        issue_tile(tile + 1);                                    // a whole tile ahead of its split
        f32x16 acc[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
        f16x8 a[2][MT][2];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < 2; ++t) a[0][mt][t] = lds_f16x8(ldsw + t * PLANE + aoff[mt]);
        static_for<0, 8>([&](auto sc_) __attribute__((always_inline)) {
            constexpr int step = decltype(sc_)::value, ty = step >> 2, tx = (step >> 1) & 1, c = step & 1;
            constexpr int cu = step & 1, nxt = cu ^ 1;
            constexpr int nstep = step + 1, nty = nstep >> 2, ntx = (nstep >> 1) & 1, nc = nstep & 1;
            constexpr int ntoff = -(nty * PC + ntx) * PSB + nc * 8;
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, MT>([&](auto mc) __attribute__((always_inline)) {
                constexpr int sub = decltype(mc)::value;
                // three of the step's 3 * MT MFMAs, round-robin over the MT accumulators (product-major: every accumulator
                // sees its three partial products smallest first), then this sub-step's share of the next step's operand reads
                static_for<3 * sub, 3 * sub + 3>([&](auto qc) __attribute__((always_inline)) {
                    constexpr int m = decltype(qc)::value, prod = m / MT, pm = m % MT;
                    constexpr int tw = prod == 0 ? 1 : 0, ta = prod == 1 ? 1 : 0;      // (l, h), (h, l), (h, h)
                    MFMA_H(acc[pm], w2[ty][tx][c][tw], a[cu][pm][ta]);
                });
                if constexpr (nstep < 8) {
                    static_for<2 * sub, 2 * sub + 2>([&](auto rc_) __attribute__((always_inline)) {
                        constexpr int ri = decltype(rc_)::value, rmt = ri / 2, rt = ri % 2;
                        a[nxt][rmt][rt] = lds_f16x8(ldsw + rt * PLANE + aoff[rmt] + ntoff);
                    });
                }
                (void)nxt;
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        PSTAMP(0, 4 + 6 * cst);
        commit_tile(lds + (cur ^ 1) * BUF);                      // tile + 1 (zeros behind the last one)
        PSTAMP(0, 5 + 6 * cst);
        __syncthreads();                                         // A: the store waves are done with the previous tile's accumulators
        PSTAMP(0, 6 + 6 * cst);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                hand[((wave * MT + mt) * 4 + g) * 64 + lane] = make_float4(acc[mt][4 * g], acc[mt][4 * g + 1], acc[mt][4 * g + 2], acc[mt][4 * g + 3]);
        PSTAMP(0, 7 + 6 * cst);
        __syncthreads();                                         // B: accumulators parked, tile + 1 staged
        PSTAMP(0, 8 + 6 * cst);
        ++cst;
    }
    PSTAMP(0, 63);
    if constexpr (C1W) {                                         // (the store waves' closing exchange)
        __syncthreads();
        __syncthreads();
    }
}
template <int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void up32p_kernel(const float *__restrict__ lo, Ep32 ep,
                                                                                               int n_img, int n_tiles) {
    up32p_body<MODE>(lo, ep, n_img, n_tiles, blockIdx.x, gridDim.x);
}

// The decoder's first convolution (4x4 -> 8x8: a few tiles per CU, 6.5 us) and the all-pairs attribute regularisation of the
// same forward pass (regloss.h: ~65 workgroups, 6.3 us, needs only z and the labels) in ONE grid: workgroups [0, grid_up) run
// the convolution, the rest the regulariser's (row block, dim) pairs, staging their columns in the launch's dynamic LDS.
// (LO = 8 since round 5: when the latent block computes the 4x4 -> 8x8 layer itself, the regulariser rides with the next one)
template <int LO, int MODE>
__global__ __launch_bounds__(256, 1) void up32x_reg_kernel(const float *__restrict__ lo, Ep32 ep, int n_img,
                                                            int n_tiles, int grid_up, RegArgs reg, int reg_bx) {
    if ((int)blockIdx.x < grid_up) {
        up32x_body<LO, MODE, 32>(lo, ep, n_img, n_tiles, blockIdx.x, grid_up);
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
// Wgrad on the fp16 MFMA with scaled two-term operands (three partial products).  Pixels are the K axis, 16 per
// MFMA with 8 consecutive pixels per lane, but LDS holds [pixel][channel] images: the operands come in through
// ds_read_b64_tr_b16, the transposing read (4 pixels x 16 channels per 16-lane group, two reads per operand).
// Tile = 64 lo pixels; wave = ky, 4 accumulator tiles (kx).  The 4x4 layers (the larger ones: wgrad32r_kernel, conv32r.hip).
template <int LO, int BIAS>
__device__ __forceinline__ void wgrad32x_body(const float *__restrict__ lo, const float *__restrict__ hi, float *__restrict__ slab, int n_img,
                                              int n_tiles, const unsigned *amax_lo, const unsigned *amax_hi, const int BID, const int NBLK) {
    constexpr int PX = 64, KB = PX / 16;                         // pixels per tile, 16-pixel K blocks per tile
    using PL = PatchLoader<LO, 2, PX>;
    // plane pitches (dwords per pixel) chosen for the transposed reads: a block is 4 consecutive K pixels x 16 dwords and
    // the 64-bank LDS serves it without conflicts when the four rows land on the four 16-dword quarters -- consecutive lo
    // pixels are WG_PSB_L apart, the hi pixels under them 2 * WG_PSB_H (stride 2): 16 and 48 dwords.  (With the common
    // pitch of 20 the fourth row wrapped onto the first: 17 % of the kernel's wave cycles were LDS bank conflicts.)
    constexpr int WG_PSB_L = WGRAD_PSB_L, WG_PSB_H = WGRAD_PSB_H;
    constexpr int PC = PL::PC, PR = PL::PR, PLANE = PL::PLANE_DW / PSB * WG_PSB_H, LPLANE = PX * WG_PSB_L;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // hi patch: 2 planes | lo tile: 2 planes
    unsigned *ldsw = reinterpret_cast<unsigned *>(lds);
    unsigned *lo_w = ldsw + 2 * PLANE;
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

    const AmaxLoad al_l = amax_issue(amax_lo), al_h = amax_issue(amax_hi);
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
    const Pow2 sc_lo = amax_scale(al_l), sc_hi = amax_scale(al_h);

    for (int tile = BID; tile < n_tiles; tile += NBLK) {
        __syncthreads();
        pl.template commit_split2<BIAS == 2, WG_PSB_H>(ldsw, sc_hi.s, &bias4);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = threadIdx.x + it * 256, pix = idx >> 3, q = idx & 7;
            uint2 hv, lv;
            split_pair_h2(lr[it].x, lr[it].y, sc_lo.s, hv.x, lv.x);
            split_pair_h2(lr[it].z, lr[it].w, sc_lo.s, hv.y, lv.y);
            *reinterpret_cast<uint2 *>(lo_w + pix * WG_PSB_L + q * 2) = hv;
            *reinterpret_cast<uint2 *>(lo_w + LPLANE + pix * WG_PSB_L + q * 2) = lv;
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
            f16x8 b2[2];                                         // lo values: B operand, column = clo
#pragma unroll
            for (int t = 0; t < 2; ++t) b2[t] = lds_tr_f16x8(lo_w + t * LPLANE + loff[b][0], lo_w + t * LPLANE + loff[b][1]);
            pl.template issue_step<KB, b>();
            if constexpr (b < 2) lr[b] = buf_load4(rs_lo, lo_base + b * 4096);
#pragma unroll
            for (int kx = 0; kx < 4; ++kx) {
                f16x8 a2[2];                                     // hi values at tap (ky = wave, kx): A operand, row = chi
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    a2[t] = lds_tr_f16x8(ldsw + t * PLANE + hoff[b][0] + kx * WG_PSB_H, ldsw + t * PLANE + hoff[b][1] + kx * WG_PSB_H);
                MFMA_H(acc[kx], a2[1], b2[0]);                   // smallest partial products first
                MFMA_H(acc[kx], a2[0], b2[1]);
                MFMA_H(acc[kx], a2[0], b2[0]);
            }
        });
    }

    // partial results (times the operands' inverse scales: exact) -> slab[blockIdx][ky][kx][clo = rc][chi = 8g + 4*half + j]
    float *out = slab + (int64_t)BID * WG32_SLAB;
    const float inv = sc_lo.inv * sc_hi.inv;
#pragma unroll
    for (int kx = 0; kx < 4; ++kx)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<float4 *>(out + ((wave * 4 + kx) * C32 + rc) * C32 + 8 * g + 4 * half) =
                make_float4(acc[kx][4 * g] * inv, acc[kx][4 * g + 1] * inv, acc[kx][4 * g + 2] * inv, acc[kx][4 * g + 3] * inv);
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
                                                          float *__restrict__ slab, int n_img, int n_tiles, const unsigned *amax_lo,
                                                          const unsigned *amax_hi) {
    wgrad32x_body<LO, BIAS>(lo, hi, slab, n_img, n_tiles, amax_lo, amax_hi, blockIdx.x, gridDim.x);
}

// ---- the 4x4 layers: data gradient and weight gradient of a layer in ONE launch ---------------------------------------
// Both read the same incoming gradient and neither needs the other; alone each fills half the chip (128 workgroups) for
// 7-11 us, most of it launch ramp and one memory round trip.  Workgroups [0, grid_a) run the data-gradient body over two
// 32-pixel tiles each, the rest the weight-gradient body (one 64-pixel tile and one slab each): 256 workgroups, one per CU.
template <int MODE, int BIAS>
__global__ __launch_bounds__(256, 1) void pair4_down_kernel(const float *__restrict__ g_hi, Ep32 ep,
                                                           const float *__restrict__ w_lo, const float *__restrict__ w_hi,
                                                           float *__restrict__ slab, int n_img, int tiles_a, int tiles_b, int grid_a,
                                                           const unsigned *amax_lo, const unsigned *amax_hi) {
    if ((int)blockIdx.x < grid_a) down32s_body<4, MODE>(g_hi, ep, n_img, tiles_a, blockIdx.x, grid_a);
    else wgrad32x_body<4, BIAS>(w_lo, w_hi, slab, n_img, tiles_b, amax_lo, amax_hi, blockIdx.x - grid_a, gridDim.x - grid_a);
}
template <int MODE, int BIAS>
__global__ __launch_bounds__(256, 1) void pair4_up_kernel(const float *__restrict__ g_lo, Ep32 ep,
                                                         const float *__restrict__ w_lo, const float *__restrict__ w_hi,
                                                         float *__restrict__ slab, int n_img, int tiles_a, int tiles_b, int grid_a,
                                                         const unsigned *amax_lo, const unsigned *amax_hi) {
    if ((int)blockIdx.x < grid_a) up32x_body<4, MODE, 32>(g_lo, ep, n_img, tiles_a, blockIdx.x, grid_a);
    else wgrad32x_body<4, BIAS>(w_lo, w_hi, slab, n_img, tiles_b, amax_lo, amax_hi, blockIdx.x - grid_a, gridDim.x - grid_a);
}


// ---- the 16x16 and 8x8 layers: the same horizontal pair (the two products of a layer's backward pass both read the incoming
// gradient and neither needs the other).  Alone each is a persistent one-workgroup-per-CU launch of 11-26 us of which 5-8 us
// are launch ramp, cold first loads and drain; side by side on disjoint CUs those overlap, and the weight gradient leaves
// half as many slabs for the closing reduction.
template <int LO, int MODE, int BIAS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void pair_down_wgrad_kernel(
    const float *__restrict__ g_hi, Ep32 ep, int n_img, int tiles_a, int grid_a, const float *__restrict__ w_lo, const float *__restrict__ w_hi,
    float *__restrict__ slab, int total_steps, int steps_per_wg, const unsigned *amax_lo, const unsigned *amax_hi) {
    if ((int)blockIdx.x < grid_a) down32p_body<LO, MODE>(g_hi, ep, n_img, tiles_a, blockIdx.x, grid_a);
    else wgrad32r_body<LO, BIAS>(w_lo, w_hi, slab, n_img, total_steps, steps_per_wg, amax_lo, amax_hi, blockIdx.x - grid_a);
}
template <int MODE, int BIAS, bool C1W = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void pair_up16_wgrad_kernel(
    const float *__restrict__ g_lo, Ep32 ep, int n_img, int tiles_a, int grid_a, const float *__restrict__ w_lo, const float *__restrict__ w_hi,
    float *__restrict__ slab, int total_steps, int steps_per_wg, const unsigned *amax_lo, const unsigned *amax_hi, C1Wgrad c1) {
    if ((int)blockIdx.x < grid_a) up32p_body<MODE, C1W>(g_lo, ep, n_img, tiles_a, blockIdx.x, grid_a, c1);
    else wgrad32r_body<16, BIAS>(w_lo, w_hi, slab, n_img, total_steps, steps_per_wg, amax_lo, amax_hi, blockIdx.x - grid_a);
}
// (the 8x8 Up body is a 256-thread workgroup: its launch-mates' other four waves leave at once)
template <int MODE, int BIAS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void pair_up8_wgrad_kernel(
    const float *__restrict__ g_lo, Ep32 ep, int n_img, int tiles_a, int grid_a, const float *__restrict__ w_lo, const float *__restrict__ w_hi,
    float *__restrict__ slab, int total_steps, int steps_per_wg, const unsigned *amax_lo, const unsigned *amax_hi) {
    if ((int)blockIdx.x < grid_a) {
        if (threadIdx.x >= 256) return;
        up32x_body<8, MODE, 32>(g_lo, ep, n_img, tiles_a, blockIdx.x, grid_a);
    } else wgrad32r_body<8, BIAS>(w_lo, w_hi, slab, n_img, total_steps, steps_per_wg, amax_lo, amax_hi, blockIdx.x - grid_a);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int cu_count() { return device_cu_count(); }

template <int LO, int PX = 128> static constexpr int tiles_for(int n) {
    using T = Tile<LO, PX>;
    return T::TI == 1 ? n * (LO / T::TR) : (n + T::TI - 1) / T::TI;
}
// persistent workgroups: one per CU, and never more writer units (four waves each) than an AMAX array has entries
static int grid_for_tiles(int tiles) {
    const int cap = cu_count() < AMAX_N / 4 ? cu_count() : AMAX_N / 4;
    return tiles < cap ? tiles : cap;
}

bool conv32_fits(const arvae_link_t *l) {
    return l->chi == 32 && l->clo == 32 && l->kh == 4 && l->kw == 4 && l->stride == 2 && l->pad == 1 &&
           l->hh == l->hw && l->lh == l->lw && (l->lh == 16 || l->lh == 8 || l->lh == 4) && l->hi_perm_c == 0 &&
           l->lo_perm_c == 0 && (int64_t)l->n * l->hh * l->hw * PIXB < (1ll << 31) - (1ll << 20);
}

template <class K> static void allow_lds(K kernel, int bytes) {
    (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

template <int A, int B> struct MaxOf { static constexpr int value = A > B ? A : B; };

constexpr int LDS_DOWN_S = (PatchLoader<4, 2, 32>::BUF2_DW + 4 * 16 * 64) * 4;
static int down_small_grid(int tiles) {                          // two workgroups per CU; one writer unit each (wave 0)
    int g = tiles < 2 * cu_count() ? tiles : 2 * cu_count();
    return g < AMAX_N ? g : AMAX_N;
}

template <int MODE> static void launch_down_small(const Operand &hi, const Ep32 &ep, int n, hipStream_t s) {
    const int tiles = tiles_for<4, 32>(n);
    static std::once_flag attr;
    std::call_once(attr, [&] { allow_lds(down32s_kernel<4, MODE>, LDS_DOWN_S); });
    ARVAE_LAUNCH((down32s_kernel<4, MODE>), dim3(down_small_grid(tiles)), dim3(256), LDS_DOWN_S, s, hi.v, ep, n, tiles);
}
template <int LO, int MODE, int PX>
static void launch_up_px(const Operand &lo, const Ep32 &ep, int n, hipStream_t s) {
    const int tiles = tiles_for<LO, PX>(n), grid = grid_for_tiles(tiles);
    if constexpr (LO == 16 && PX == 128) {                      // compute / store waves
        static const bool pc = diag_env("ARVAE_UP32_NO_PC") == nullptr;       // A/B: up32x_kernel
        if constexpr (MODE != EP_GATE_F) if (pc) {
            constexpr int LDSP = (2 * 2 * PatchLoader<16, 1, 128>::PLANE_DW) * 4 + 4 * 4 * 4 * 64 * 16;
            static std::once_flag attrp;
            std::call_once(attrp, [&] { allow_lds(up32p_kernel<MODE>, LDSP); });
            ARVAE_LAUNCH((up32p_kernel<MODE>), dim3(grid), dim3(512), LDSP, s, lo.v, ep, n, tiles);
            return;
        }
    }
    constexpr int LDSX = 2 * PatchLoader<LO, 1, PX>::PLANE_DW * 4;
    static std::once_flag attrx;
    std::call_once(attrx, [&] { allow_lds(up32x_kernel<LO, MODE, PX>, LDSX); });
    ARVAE_LAUNCH((up32x_kernel<LO, MODE, PX>), dim3(grid), dim3(256), LDSX, s, lo.v, ep, n, tiles);
}
// 128-pixel tiles, or 32-pixel tiles when the former give a CU at most one tile (nothing to pipeline, or idle CUs:
// the 8x8 and 4x4 layers at batch 512)
template <int LO, int MODE>
static void launch_up_v(const Operand &lo, const Ep32 &ep, int n, hipStream_t s) {
    static const bool small_ok = diag_env("ARVAE_NO_SMALL_TILES") == nullptr;     // diagnostic switch
    if (LO == 4 && small_ok && 2 * tiles_for<LO, 128>(n) <= cu_count()) launch_up_px<4, MODE, 32>(lo, ep, n, s);
    else if (LO == 8 && small_ok && tiles_for<LO, 128>(n) <= cu_count()) launch_up_px<8, MODE, 32>(lo, ep, n, s);
    else launch_up_px<LO, MODE, 128>(lo, ep, n, s);
}

static int ep_mode(const Ep32 &ep, int relu) {
    return ep.gate_bits != nullptr ? EP_GATE_B : ep.gate != nullptr ? EP_GATE_F : relu ? EP_RELU : EP_PLAIN;
}

// the four-way reduction-split producer / consumer kernel of the 16x16 / 8x8 layers (down32p.h)
template <int LO, int MODE> static void launch_down_p(const float *hi, const Ep32 &ep, int n, hipStream_t s) {
    constexpr int LDS = DownK<LO>::LDS_DW * 4;
    static std::once_flag attr;
    std::call_once(attr, [&] { allow_lds(down32p_kernel<LO, MODE>, LDS); });
    const int tiles = n * DownK<LO>::TILES_PER_IMG;
    ARVAE_LAUNCH((down32p_kernel<LO, MODE>), dim3(grid_for_tiles(tiles)), dim3(512), LDS, s, hi, ep, n, tiles);
}
template <int LO> static void launch_down_p_mode(const float *hi, const Ep32 &ep, int mode, int n, hipStream_t s) {
    switch (mode) {
        case EP_GATE_B: launch_down_p<LO, EP_GATE_B>(hi, ep, n, s); break;
        case EP_GATE_F: launch_down_p<LO, EP_GATE_F>(hi, ep, n, s); break;
        case EP_RELU: launch_down_p<LO, EP_RELU>(hi, ep, n, s); break;
        default: launch_down_p<LO, EP_PLAIN>(hi, ep, n, s); break;
    }
}
static void conv32_down_ksplit(const arvae_link_t *l, const float *hi, const Ep32 &ep, int mode, hipStream_t s) {
    if (l->lh == 16) launch_down_p_mode<16>(hi, ep, mode, l->n, s);
    else launch_down_p_mode<8>(hi, ep, mode, l->n, s);
}

// The operands of every entry point below: plain fp32 tensors that come with the AMAX array of their values (conv32_common.h;
// conv32_amax makes one for a tensor that has none), the layer's prepared weights (conv32_weight_prep), and -- when somebody
// will multiply the result on the matrix pipe -- the AMAX array the result's maxima go to.
// gate (float activation) or gate_bits (relu_bits16) select a gated epilogue; with relu, bits_out (may be null) receives
// the sign bits of the result
int conv32_down(const arvae_link_t *l, const Operand &hi, const float *bias, int relu, const float *gate, const uint16_t *gate_bits,
                uint16_t *bits_out, float *out, hipStream_t s, const float *wprep, const unsigned *amax_in, unsigned *amax_out) {
    ARVAE_REQUIRE(wprep != nullptr && amax_in != nullptr, "conv32_down: prepared weights and the input's maxima are needed");
    Ep32 ep{bias, gate, gate_bits, bits_out, out, reinterpret_cast<const uint4 *>(wprep), amax_in, amax_out};
    const int mode = ep_mode(ep, relu);
    if (l->lh != 4) {
        conv32_down_ksplit(l, hi.v, ep, mode, s);
        return check_launch(l->lh == 16 ? "down32_kernel<16>" : "down32_kernel<8>");
    }
    switch (mode) {
        case EP_GATE_B: launch_down_small<EP_GATE_B>(hi, ep, l->n, s); break;
        case EP_GATE_F: launch_down_small<EP_GATE_F>(hi, ep, l->n, s); break;
        case EP_RELU: launch_down_small<EP_RELU>(hi, ep, l->n, s); break;
        default: launch_down_small<EP_PLAIN>(hi, ep, l->n, s); break;
    }
    return check_launch("down32_kernel<4>");
}

// ---- two stacked ReLU DOWN layers of the forward pass (16x16 then 8x8 output) in ONE launch (round 6) ---------------------------
// Every tile of the lower layer needs ONE image of the upper layer's output, and down32p_body gives a workgroup a contiguous run
// of tiles: with the same grid for both layers and runs of whole images, workgroup w computes the 8x8 layer on exactly the images
// whose 16x16 layer it has just stored -- a dependency inside the workgroup (its stores drained, one barrier), not between
// launches.  The second body's input scale is the workgroup's own maximum (down32p.h CHAIN); the tensor-wide AMAX arrays of both
// outputs are published as always (the backward pass's weight gradients read them).  Saves the lower layer's launch: ramp, cold
// prologue and drain of a 12 us launch that multiplies for 3.
template <int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void chain_down_kernel(const float *__restrict__ hi, Ep32 ep_a, Ep32 ep_b,
                                                                                                    int n_img, int tiles_a, int tiles_b) {
    extern __shared__ __attribute__((aligned(16))) unsigned chain_lds[];
    constexpr int BODY_DW = MaxOf<DownK<16>::LDS_DW, DownK<8>::LDS_DW>::value;
    float *chain_max = reinterpret_cast<float *>(chain_lds + BODY_DW);
    down32p_body<16, MODE, 1>(hi, ep_a, n_img, tiles_a, blockIdx.x, gridDim.x, chain_max);
    __syncthreads();                                             // (a workgroup-scope release / acquire: this workgroup's stores are visible to its loads)
    down32p_body<8, MODE, 2>(ep_a.out, ep_b, n_img, tiles_b, blockIdx.x, gridDim.x, chain_max);
}
static int chain_grid(const arvae_link_t *a, const arvae_link_t *b) {       // 0: the runs of the two layers do not cover the same images
    if (a->n != b->n || a->n < 1) return 0;
    const int tiles_a = a->n * DownK<16>::TILES_PER_IMG, tiles_b = b->n * DownK<8>::TILES_PER_IMG;
    const int grid = grid_for_tiles(tiles_a);
    const int run_a = (tiles_a + grid - 1) / grid, run_b = (tiles_b + grid - 1) / grid;
    return (run_a * DownK<8>::TILES_PER_IMG == run_b * DownK<16>::TILES_PER_IMG) ? grid : 0;
}
bool conv32_down_chain_fits(const arvae_link_t *a, const arvae_link_t *b) {
    static const bool off = diag_env("ARVAE_NO_DOWN_CHAIN") != nullptr;     // A/B switch: two launches
    return !off && conv32_fits(a) && conv32_fits(b) && a->lh == 16 && b->lh == 8 && b->hh == a->lh && chain_grid(a, b) > 0;
}
// both layers: bias + ReLU, sign bits to bits_*; amax_in: of `hi`; amax_a / amax_b: where the outputs' maxima go
int conv32_down_chain(const arvae_link_t *a, const arvae_link_t *b, const float *hi, const unsigned *amax_in, const float *bias_a,
                      uint16_t *bits_a, float *out_a, const float *wprep_a, unsigned *amax_a, const float *bias_b, uint16_t *bits_b,
                      float *out_b, const float *wprep_b, unsigned *amax_b, hipStream_t s) {
    ARVAE_REQUIRE(conv32_down_chain_fits(a, b), "conv32_down_chain: these two layers do not chain");
    ARVAE_REQUIRE(hi && amax_in && out_a && out_b && wprep_a && wprep_b, "conv32_down_chain: null pointer");
    Ep32 ep_a{bias_a, nullptr, nullptr, bits_a, out_a, reinterpret_cast<const uint4 *>(wprep_a), amax_in, amax_a};
    Ep32 ep_b{bias_b, nullptr, nullptr, bits_b, out_b, reinterpret_cast<const uint4 *>(wprep_b), nullptr, amax_b};
    constexpr int LDS = (MaxOf<DownK<16>::LDS_DW, DownK<8>::LDS_DW>::value + 4) * 4;
    static std::once_flag attr;
    std::call_once(attr, [&] { allow_lds(chain_down_kernel<EP_RELU>, LDS); });
    ARVAE_LAUNCH((chain_down_kernel<EP_RELU>), dim3(chain_grid(a, b)), dim3(512), LDS, s, hi, ep_a, ep_b, a->n, a->n * DownK<16>::TILES_PER_IMG,
                 b->n * DownK<8>::TILES_PER_IMG);
    return check_launch("chain(down32<16> + down32<8>)");
}

template <int LO> static int launch_up(const arvae_link_t *l, const Operand &lo, const Ep32 &ep, int relu, hipStream_t s) {
    switch (ep_mode(ep, relu)) {
        case EP_GATE_B: launch_up_v<LO, EP_GATE_B>(lo, ep, l->n, s); break;
        case EP_GATE_F: launch_up_v<LO, EP_GATE_F>(lo, ep, l->n, s); break;
        case EP_RELU: launch_up_v<LO, EP_RELU>(lo, ep, l->n, s); break;
        default: launch_up_v<LO, EP_PLAIN>(lo, ep, l->n, s); break;
    }
    return check_launch(LO == 16 ? "up32_kernel<16>" : LO == 8 ? "up32_kernel<8>" : "up32_kernel<4>");
}

// conv32_up of a 4x4 -> 8x8 ReLU layer (forward pass, small tiles) with the regulariser's workgroups riding in
// the same grid (up32x_reg_kernel); false: not that case, launch the two separately
bool conv32_up_reg_fits(const arvae_link_t *l) {
    static const bool off = diag_env("ARVAE_NO_PAIR_REG") != nullptr || diag_env("ARVAE_NO_SMALL_TILES") != nullptr;
    return !off && conv32_fits(l) && ((l->lh == 4 && 2 * tiles_for<4, 128>(l->n) <= cu_count()) ||
                                      (l->lh == 8 && tiles_for<8, 128>(l->n) <= cu_count()));      // (the cases launch_up_v runs on 32-pixel tiles)
}
int conv32_up_reg(const arvae_link_t *l, const Operand &lo, const float *bias, uint16_t *bits_out, float *out, const float *wprep,
                  const unsigned *amax_in, unsigned *amax_out, const RegArgs &reg, int r, hipStream_t s) {
    ARVAE_REQUIRE(wprep != nullptr && amax_in != nullptr, "conv32_up_reg: prepared weights and the input's maxima are needed");
    Ep32 ep{bias, nullptr, nullptr, bits_out, out, reinterpret_cast<const uint4 *>(wprep), amax_in, amax_out};
    const int reg_bx = (int)((reg.n_rows + REG_ROWS_PER_BLOCK - 1) / REG_ROWS_PER_BLOCK);
    if (l->lh == 8) {
        const int tiles = tiles_for<8, 32>(l->n), grid_up = grid_for_tiles(tiles);
        constexpr int LDSX8 = MaxOf<2 * PatchLoader<8, 1, 32>::PLANE_DW, 2 * REG_CHUNK>::value * 4;
        static std::once_flag attr8;
        std::call_once(attr8, [&] { allow_lds(up32x_reg_kernel<8, EP_RELU>, LDSX8); });
        ARVAE_LAUNCH((up32x_reg_kernel<8, EP_RELU>), dim3(grid_up + reg_bx * r), dim3(256), LDSX8, s, lo.v, ep, l->n, tiles, grid_up, reg, reg_bx);
        return check_launch("up32_kernel<8>(+ reg_loss)");
    }
    const int tiles = tiles_for<4, 32>(l->n), grid_up = grid_for_tiles(tiles);
    constexpr int LDSX = MaxOf<2 * PatchLoader<4, 1, 32>::PLANE_DW, 2 * REG_CHUNK>::value * 4;
    static std::once_flag attr;
    std::call_once(attr, [&] { allow_lds(up32x_reg_kernel<4, EP_RELU>, LDSX); });
    ARVAE_LAUNCH((up32x_reg_kernel<4, EP_RELU>), dim3(grid_up + reg_bx * r), dim3(256), LDSX, s, lo.v, ep, l->n, tiles, grid_up, reg, reg_bx);
    return check_launch("up32_kernel<4>(+ reg_loss)");
}

int conv32_up(const arvae_link_t *l, const Operand &lo, const float *bias, int relu, const float *gate, const uint16_t *gate_bits,
              uint16_t *bits_out, float *out, hipStream_t s, const float *wprep, const unsigned *amax_in, unsigned *amax_out) {
    ARVAE_REQUIRE(wprep != nullptr && amax_in != nullptr, "conv32_up: prepared weights and the input's maxima are needed");
    Ep32 ep{bias, gate, gate_bits, bits_out, out, reinterpret_cast<const uint4 *>(wprep), amax_in, amax_out};
    switch (l->lh) {
        case 16: return launch_up<16>(l, lo, ep, relu, s);
        case 8: return launch_up<8>(l, lo, ep, relu, s);
        default: return launch_up<4>(l, lo, ep, relu, s);
    }
}

// floats of workspace per layer for conv32_weight_prep, and the batched launch (up to 8 layers)
int64_t conv32_prep_floats() { return PREP_FLOATS; }
// floats of scratch a caller WITHOUT prepared weights and maxima needs for one call: one layer's prep + two AMAX arrays
int64_t conv32_scratch_floats() { return (PREP_FLOATS + 3) / 4 * 4 + 2 * AMAX_N; }

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
    ARVAE_LAUNCH(prep_all_kernel, dim3(conv_blocks + mid_prep_blocks(mid)), dim3(256), 0, s, p, mid, conv_blocks);
    return check_launch("weight_prep(conv32 + latent block)");
}

// AMAX array of a plain tensor of `count` floats (a multiple of 4, 16-byte aligned)
int conv32_amax(const float *x, int64_t count, unsigned *out, hipStream_t s) {
    ARVAE_REQUIRE(x != nullptr && out != nullptr && count % 4 == 0, "conv32_amax: null pointer or a count that is not a multiple of 4");
    int64_t blocks = (count / 4 + 255) / 256;
    if (blocks > AMAX_N) blocks = AMAX_N;
    if (blocks < 1) blocks = 1;
    ARVAE_LAUNCH(amax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, count / 4, out);
    return check_launch("conv32_amax");
}

// row-stream weight gradient with producer / consumer waves (wgrad32r.h): the 16x16 and 8x8 layers
static bool conv32_wgrad_stream_fits(const arvae_link_t *l) {
    static const bool off = diag_env("ARVAE_WGRAD_NO_STREAM") != nullptr;     // diagnostic: the patch-staged wgrad32x_kernel instead
    return !off && (l->lh == 16 || l->lh == 8);
}
// steps of the whole launch, steps per workgroup and workgroups when `groups` workgroups (at most) share the stream
static void stream_geometry(const arvae_link_t *l, int groups, int &total, int &spw, int &grid) {
    total = l->n * (l->lh * l->lh / 32);
    spw = (total + groups - 1) / groups;
    static const int forced = diag_env("ARVAE_WGR_SPW") != nullptr ? atoi(diag_env("ARVAE_WGR_SPW")) : 0;   // diagnostic: steps per workgroup
    if (forced > 0) spw = forced;
    if (spw < 1) spw = 1;
    grid = (total + spw - 1) / spw;
}
static int conv32_wgrad_stream_groups(const arvae_link_t *l) {
    int total, spw, grid;
    stream_geometry(l, cu_count(), total, spw, grid);
    return grid;
}
template <int LO> static int launch_stream(const arvae_link_t *l, const float *lo, const float *hi, float *slab, int bias_mode,
                                           const unsigned *amax_lo, const unsigned *amax_hi, hipStream_t s) {
    constexpr int LDS = RowStream<LO>::LDS_DW * 4;
    int total, spw, grid;
    stream_geometry(l, cu_count(), total, spw, grid);
    static std::once_flag attr;
    std::call_once(attr, [&] {
        allow_lds(wgrad32r_kernel<LO, 0>, LDS);
        allow_lds(wgrad32r_kernel<LO, 1>, LDS);
        allow_lds(wgrad32r_kernel<LO, 2>, LDS);
    });
    if (bias_mode == 1) ARVAE_LAUNCH((wgrad32r_kernel<LO, 1>), dim3(grid), dim3(512), LDS, s, lo, hi, slab, l->n, total, spw, amax_lo, amax_hi);
    else if (bias_mode == 2) ARVAE_LAUNCH((wgrad32r_kernel<LO, 2>), dim3(grid), dim3(512), LDS, s, lo, hi, slab, l->n, total, spw, amax_lo, amax_hi);
    else ARVAE_LAUNCH((wgrad32r_kernel<LO, 0>), dim3(grid), dim3(512), LDS, s, lo, hi, slab, l->n, total, spw, amax_lo, amax_hi);
    return check_launch(LO == 16 ? "wgrad32_kernel<16>" : "wgrad32_kernel<8>");
}
static int conv32_wgrad_stream(const arvae_link_t *l, const Operand &lo, const Operand &hi, float *slab, int bias_mode, const unsigned *amax_lo,
                               const unsigned *amax_hi, hipStream_t s) {
    return l->lh == 16 ? launch_stream<16>(l, lo.v, hi.v, slab, bias_mode, amax_lo, amax_hi, s)
                       : launch_stream<8>(l, lo.v, hi.v, slab, bias_mode, amax_lo, amax_hi, s);
}

int conv32_wgrad_groups(const arvae_link_t *l) {
    if (conv32_wgrad_stream_fits(l)) return conv32_wgrad_stream_groups(l);
    // the patch-staged kernel (4x4 layers; every size with ARVAE_WGRAD_NO_STREAM): 64-pixel tiles, one persistent
    // workgroup per CU, one 64 KB partial each
    const int tiles = l->lh == 16 ? tiles_for<16, 64>(l->n) : l->lh == 8 ? tiles_for<8, 64>(l->n) : tiles_for<4, 64>(l->n);
    return grid_for_tiles(tiles);
}

int64_t conv32_wgrad_ws_floats(const arvae_link_t *l) {
    return (int64_t)conv32_wgrad_groups(l) * WG32_SLAB;
}

constexpr int LDS_WGRAD_X4 = 2 * (PatchLoader<4, 2, 64>::PLANE_DW / PSB * WGRAD_PSB_H + 64 * WGRAD_PSB_L) * 4;
template <int LO> static int launch_wgrad_x(const arvae_link_t *l, const Operand &lo, const Operand &hi, float *slab, int bias_mode,
                                            int grid, const unsigned *amax_lo, const unsigned *amax_hi, hipStream_t s) {
    constexpr int LDS = 2 * (PatchLoader<LO, 2, 64>::PLANE_DW / PSB * WGRAD_PSB_H + 64 * WGRAD_PSB_L) * 4;
    const int tiles = tiles_for<LO, 64>(l->n);
    static std::once_flag attr;
    std::call_once(attr, [&] {
        allow_lds(wgrad32x_kernel<LO, 0>, LDS);
        allow_lds(wgrad32x_kernel<LO, 1>, LDS);
        allow_lds(wgrad32x_kernel<LO, 2>, LDS);
    });
    if (bias_mode == 1)
        ARVAE_LAUNCH((wgrad32x_kernel<LO, 1>), dim3(grid), dim3(256), LDS, s, lo.v, hi.v, slab, l->n, tiles, amax_lo, amax_hi);
    else if (bias_mode == 2)
        ARVAE_LAUNCH((wgrad32x_kernel<LO, 2>), dim3(grid), dim3(256), LDS, s, lo.v, hi.v, slab, l->n, tiles, amax_lo, amax_hi);
    else
        ARVAE_LAUNCH((wgrad32x_kernel<LO, 0>), dim3(grid), dim3(256), LDS, s, lo.v, hi.v, slab, l->n, tiles, amax_lo, amax_hi);
    return check_launch(LO == 16 ? "wgrad32_kernel<16>" : LO == 8 ? "wgrad32_kernel<8>" : "wgrad32_kernel<4>");
}

// per-workgroup partial sums into `slab`; the returned job describes the reduction that finishes the layer
// bias_mode: 0 none, 1 dbias[clo] += sum lo, 2 dbias[chi] += sum hi
int conv32_wgrad_partial(const arvae_link_t *l, const Operand &lo, const Operand &hi, float *dwt, float *dbias, int bias_mode,
                         float *slab, hipStream_t s, SlabJob *job, const unsigned *amax_lo, const unsigned *amax_hi) {
    ARVAE_REQUIRE(amax_lo != nullptr && amax_hi != nullptr, "conv32_wgrad: the operands' maxima are needed");
    const int grid = conv32_wgrad_groups(l);
    const int rc = conv32_wgrad_stream_fits(l) ? conv32_wgrad_stream(l, lo, hi, slab, bias_mode, amax_lo, amax_hi, s)
                   : l->lh == 16             ? launch_wgrad_x<16>(l, lo, hi, slab, bias_mode, grid, amax_lo, amax_hi, s)
                   : l->lh == 8              ? launch_wgrad_x<8>(l, lo, hi, slab, bias_mode, grid, amax_lo, amax_hi, s)
                                             : launch_wgrad_x<4>(l, lo, hi, slab, bias_mode, grid, amax_lo, amax_hi, s);
    *job = SlabJob{slab, dwt, bias_mode ? dbias : nullptr, grid, SLAB_C32, bias_mode};
    return rc;
}

// ---- gated data gradient + weight-gradient partials of one layer in ONE launch (pair4_*_kernel, pair_*_wgrad_kernel) ------------
// up == true: the layer is a forward UP link (data gradient = DOWN map on g, weight gradient with g on the hi side, bias mode 2);
// up == false: a forward DOWN link (data gradient = UP map, g on the lo side, bias mode 1).  Only the combinations the image
// executor produces are instantiated; everything else (and the experiment switches) goes the two-launch way.
// ... when the Up half of the 16x16 pair stores nothing (C1Wgrad: the first layer's weight gradient in its store waves)
static int pair_split_c1_percent() {
    static const int forced = diag_env("ARVAE_PAIR_SPLIT_C1") != nullptr ? atoi(diag_env("ARVAE_PAIR_SPLIT_C1")) : 0;
    return (forced > 0 && forced < 100) ? forced : 62;        // (same-box sweep at B = 512: 38 / 44 / 50 / 56 / 62 / 68 % -> 55.1 / 50.8 / 43.6 / 43.8 / 39.1 / 43.3 us)
}
static int pair_split_percent(int lh, bool up) {              // share of the workgroups that runs the data gradient
    // ARVAE_PAIR_SPLIT16 / _SPLIT8: both pairs of that size; ..._SPLIT16U / 16D / 8U / 8D: the pair whose data gradient is the Up
    // (forward Down layer) / Down map
    auto env = [](const char *name) { const char *v = diag_env(name); return v != nullptr ? atoi(v) : 0; };
    static const int e16 = env("ARVAE_PAIR_SPLIT16"), e8 = env("ARVAE_PAIR_SPLIT8");
    static const int e16u = env("ARVAE_PAIR_SPLIT16U"), e16d = env("ARVAE_PAIR_SPLIT16D"), e8u = env("ARVAE_PAIR_SPLIT8U"), e8d = env("ARVAE_PAIR_SPLIT8D");
    const int one = lh == 16 ? (up ? e16d : e16u) : (up ? e8d : e8u), both = lh == 16 ? e16 : e8;
    const int forced = one > 0 ? one : both;
    if (forced > 0 && forced < 100) return forced;
    // 50 % everywhere: both halves then give workgroup w the SAME images (equal ranges of whole images at the benchmark batches)
    // and workgroup w of either half sits on XCD w % 8 (grid_a a multiple of 8): the tensor both halves read (the gradient g) is
    // fetched from HBM once and the later reader hits that XCD's L2.  Round 6, pair(down32<16> + wgrad32<16>) at B = 512, same
    // box: 48 % (121 + 133 workgroups, 17 tiles against 31 steps: the ranges and the XCDs drift apart) 177.7 MB and 39.1 us per
    // launch, 50 % (128 + 128, 16 tiles and 32 steps = the same four images) 112.7 MB and 36.9 us: one read of g (67 MB) gone
    // (tools/run_l2_align_ab.sh, profiles/r6_l2_align_ab.txt; the counters are calibrated in profiles/r6_fetch_calib.txt)
    return 50;
}
bool conv32_pair_fits(const arvae_link_t *l, bool up, const float *gate, const uint16_t *gate_bits, int bias_mode) {
    static const bool off4 = diag_env("ARVAE_NO_PAIR4") != nullptr || diag_env("ARVAE_NO_SMALL_TILES") != nullptr;
    static const bool off = diag_env("ARVAE_NO_PAIR32") != nullptr;
    if ((gate == nullptr && gate_bits == nullptr) || bias_mode != (up ? 2 : 1)) return false;
    if (l->lh == 4) {
        const int wg_tiles = tiles_for<4, 64>(l->n), dg_tiles = tiles_for<4, 32>(l->n);
        return !off4 && wg_tiles >= 8 && wg_tiles + (dg_tiles + 1) / 2 <= cu_count() && conv32_wgrad_groups(l) == wg_tiles;
    }
    // the larger layers: sign-bit gates only; both halves must keep several tiles / steps per workgroup; the 8x8 Up body is the
    // 32-pixel-tile kernel (what launch_up_v picks at this batch)
    if (off || gate_bits == nullptr || !conv32_wgrad_stream_fits(l)) return false;
    if (l->n * (l->lh * l->lh / 32) < 4 * cu_count()) return false;
    if (!up && l->lh == 8 && !(diag_env("ARVAE_NO_SMALL_TILES") == nullptr && tiles_for<8, 128>(l->n) <= cu_count())) return false;
    if (!up && l->lh == 16 && diag_env("ARVAE_UP32_NO_PC") != nullptr) return false;
    return true;
}

template <int LO> static int launch_pair_big(const arvae_link_t *l, bool up, const float *g, const float *x_in, const Ep32 &ep, float *slab,
                                             const unsigned *amax_g, const unsigned *amax_x, hipStream_t s, int *grid_b_out,
                                             const C1Wgrad *c1 = nullptr, int *grid_a_out = nullptr) {
    const int cus = cu_count() < AMAX_N / 4 ? cu_count() : AMAX_N / 4;
    int grid_a = cus * (c1 != nullptr ? pair_split_c1_percent() : pair_split_percent(LO, up)) / 100;
    if (grid_a < 1) grid_a = 1;
    // every body walks a contiguous run of ceil(tiles / grid_a) tiles: the workgroups past the last non-empty run would find nothing
    // to do, so the weight gradient gets them instead (round 6; the first-layer pair at B = 512: 158 workgroups asked for, 1024 tiles
    // in runs of 7 -> 147 in use, the weight gradient 109 instead of 98 workgroups = 38 instead of 42 steps each)
    const int tiles_a = up ? l->n * DownK<LO>::TILES_PER_IMG : (LO == 16 ? tiles_for<16, 128>(l->n) : tiles_for<8, 32>(l->n));
    {
        const int run = (tiles_a + grid_a - 1) / grid_a;
        grid_a = (tiles_a + run - 1) / run;
    }
    if (grid_a_out != nullptr) *grid_a_out = grid_a;             // (after the clamp: the caller sizes the slab reduction with it)
    // the first layer's slabs (one per workgroup of the Up half) live in that layer's weight-gradient workspace, which make_layout
    // sizes for wgrad_c1_groups() <= 256 workgroups (conv_c1.hip): a split or a CU cap that asks for more must not write past it
    ARVAE_REQUIRE(c1 == nullptr || grid_a <= 256, "pair_up16_wgrad: %d workgroups of the Up half exceed the first layer's 256 weight-gradient slabs", grid_a);
    int total, spw, grid_b;
    stream_geometry(l, cus - grid_a, total, spw, grid_b);
    *grid_b_out = grid_b;
    constexpr int LDS_W = RowStream<LO>::LDS_DW * 4;
    const dim3 grid(grid_a + grid_b);
    if (up) {                                                    // DOWN map of g (hi side); weight gradient: lo = layer input, hi = g
        constexpr int LDS = MaxOf<LDS_W, DownK<LO>::LDS_DW * 4>::value;
        static std::once_flag attr;
        std::call_once(attr, [&] { allow_lds(pair_down_wgrad_kernel<LO, EP_GATE_B, 2>, LDS); });
        ARVAE_LAUNCH((pair_down_wgrad_kernel<LO, EP_GATE_B, 2>), grid, dim3(512), LDS, s, g, ep, l->n, tiles_a, grid_a, x_in, g, slab, total, spw,
                     amax_x, amax_g);
        return check_launch(LO == 16 ? "pair(down32<16> + wgrad32<16>)" : "pair(down32<8> + wgrad32<8>)");
    }
    if constexpr (LO == 16) {                                    // UP map of g (lo side); weight gradient: lo = g, hi = layer input
        constexpr int LDS_U = (2 * 2 * PatchLoader<16, 1, 128>::PLANE_DW) * 4 + 4 * 4 * 4 * 64 * 16;
        constexpr int LDS = MaxOf<LDS_W, LDS_U>::value;
        static std::once_flag attr;
        if (c1 != nullptr) {                                     // + the first layer's weight gradient in the Up half's store waves
            constexpr int LDS_C = MaxOf<LDS_W, LDS_U + 4 * C1W_WAVE_DW * 4>::value;
            static std::once_flag attr_c;
            std::call_once(attr_c, [&] { allow_lds(pair_up16_wgrad_kernel<EP_GATE_B, 1, true>, LDS_C); });
            ARVAE_LAUNCH((pair_up16_wgrad_kernel<EP_GATE_B, 1, true>), grid, dim3(512), LDS_C, s, g, ep, l->n, tiles_a, grid_a, g, x_in,
                         slab, total, spw, amax_g, amax_x, *c1);
            return check_launch("pair(up32<16> + wgrad32<16> + wgrad_c1)");
        }
        std::call_once(attr, [&] { allow_lds(pair_up16_wgrad_kernel<EP_GATE_B, 1>, LDS); });
        ARVAE_LAUNCH((pair_up16_wgrad_kernel<EP_GATE_B, 1>), grid, dim3(512), LDS, s, g, ep, l->n, tiles_a, grid_a, g, x_in, slab,
                     total, spw, amax_g, amax_x, C1Wgrad{nullptr, nullptr});
        return check_launch("pair(up32<16> + wgrad32<16>)");
    } else {
        constexpr int LDS = MaxOf<LDS_W, 2 * PatchLoader<8, 1, 32>::PLANE_DW * 4>::value;
        static std::once_flag attr;
        std::call_once(attr, [&] { allow_lds(pair_up8_wgrad_kernel<EP_GATE_B, 1>, LDS); });
        ARVAE_LAUNCH((pair_up8_wgrad_kernel<EP_GATE_B, 1>), grid, dim3(512), LDS, s, g, ep, l->n, tiles_a, grid_a, g, x_in, slab, total,
                     spw, amax_g, amax_x);
        return check_launch("pair(up32<8> + wgrad32<8>)");
    }
}

// amax_g / amax_x: AMAX arrays of the incoming gradient and of the layer's input; amax_out: of d_in (may be null)
// c1_img / c1_slab / c1_job (all or none; the 16x16 layer behind a single-channel first layer, sign-bit gates): the first
// layer's weight gradient comes out of this launch too (up32p_body, C1Wgrad) and the data gradient is NOT stored (d_in unused)
bool conv32_pair_c1_fits(const arvae_link_t *l, bool up, const uint16_t *gate_bits) {
    static const bool off = diag_env("ARVAE_NO_PAIR_C1W") != nullptr;
    return !off && !up && l->lh == 16 && gate_bits != nullptr && diag_env("ARVAE_UP32_NO_PC") == nullptr;
}
int conv32_pair(const arvae_link_t *l, bool up, const float *g, const float *x_in, const float *gate, const uint16_t *gate_bits,
                float *d_in, const float *wprep, float *dwt, float *dbias, float *slab, hipStream_t s, SlabJob *job,
                const unsigned *amax_g, const unsigned *amax_x, unsigned *amax_out, const float *c1_img, float *c1_slab, SlabJob *c1_job) {
    ARVAE_REQUIRE(wprep != nullptr && amax_g != nullptr && amax_x != nullptr, "conv32_pair: prepared weights and the operands' maxima are needed");
    Ep32 ep{nullptr, gate_bits ? nullptr : gate, gate_bits, nullptr, d_in, reinterpret_cast<const uint4 *>(wprep), amax_g, amax_out};
    if (l->lh != 4) {
        int grid_b = 0, grid_a = 0;
        C1Wgrad c1{c1_img, c1_slab};
        const bool fuse = c1_img != nullptr && c1_slab != nullptr && c1_job != nullptr && conv32_pair_c1_fits(l, up, gate_bits);
        ARVAE_REQUIRE(fuse || c1_img == nullptr, "conv32_pair: this layer cannot carry the first layer's weight gradient");
        const int rc = l->lh == 16 ? launch_pair_big<16>(l, up, g, x_in, ep, slab, amax_g, amax_x, s, &grid_b, fuse ? &c1 : nullptr, &grid_a)
                                   : launch_pair_big<8>(l, up, g, x_in, ep, slab, amax_g, amax_x, s, &grid_b);
        *job = SlabJob{slab, dwt, dbias, grid_b, SLAB_C32, up ? 2 : 1};
        if (fuse) *c1_job = SlabJob{c1_slab, nullptr, nullptr, grid_a, SLAB_C1, 1};       // (dwt / dbias: the caller's)
        return rc;
    }
    constexpr int LDS_U = 2 * PatchLoader<4, 1, 32>::PLANE_DW * 4;
    constexpr int LDS = MaxOf<LDS_WGRAD_X4, MaxOf<LDS_DOWN_S, LDS_U>::value>::value;
    const int wg_tiles = tiles_for<4, 64>(l->n), dg_tiles = tiles_for<4, 32>(l->n), grid_a = (dg_tiles + 1) / 2;
    static std::once_flag attr;
    std::call_once(attr, [&] {
        allow_lds(pair4_down_kernel<EP_GATE_F, 2>, LDS);
        allow_lds(pair4_down_kernel<EP_GATE_B, 2>, LDS);
        allow_lds(pair4_up_kernel<EP_GATE_F, 1>, LDS);
        allow_lds(pair4_up_kernel<EP_GATE_B, 1>, LDS);
    });
    const dim3 grid(grid_a + wg_tiles);
    if (up) {                                                    // DOWN map of g (hi side); weight gradient: lo = layer input, hi = g
        if (gate_bits) ARVAE_LAUNCH((pair4_down_kernel<EP_GATE_B, 2>), grid, dim3(256), LDS, s, g, ep, x_in, g, slab, l->n, dg_tiles, wg_tiles, grid_a, amax_x, amax_g);
        else ARVAE_LAUNCH((pair4_down_kernel<EP_GATE_F, 2>), grid, dim3(256), LDS, s, g, ep, x_in, g, slab, l->n, dg_tiles, wg_tiles, grid_a, amax_x, amax_g);
    } else {                                                     // UP map of g (lo side); weight gradient: lo = g, hi = layer input
        if (gate_bits) ARVAE_LAUNCH((pair4_up_kernel<EP_GATE_B, 1>), grid, dim3(256), LDS, s, g, ep, g, x_in, slab, l->n, dg_tiles, wg_tiles, grid_a, amax_g, amax_x);
        else ARVAE_LAUNCH((pair4_up_kernel<EP_GATE_F, 1>), grid, dim3(256), LDS, s, g, ep, g, x_in, slab, l->n, dg_tiles, wg_tiles, grid_a, amax_g, amax_x);
    }
    *job = SlabJob{slab, dwt, dbias, wg_tiles, SLAB_C32, up ? 2 : 1};
    return check_launch(up ? "pair4(down32 + wgrad32)" : "pair4(up32 + wgrad32)");
}

int conv32_wgrad(const arvae_link_t *l, const Operand &lo, const Operand &hi, float *dwt, float *dbias, int bias_mode,
                 float *slab, hipStream_t s, const unsigned *amax_lo, const unsigned *amax_hi) {
    SlabJob job;
    if (int rc = conv32_wgrad_partial(l, lo, hi, dwt, dbias, bias_mode, slab, s, &job, amax_lo, amax_hi)) return rc;
    return slab_reduce(job, s);
}

}  // namespace arvae

#ifdef ARVAE_STAMPS
extern "C" int arvae_debug_stamps(unsigned long long *out, int count) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(arvae::g_stamps), sizeof(unsigned long long) * count);
}
#endif
#ifdef D32K_STAMPS
extern "C" int arvae_debug_d32k_stamps(unsigned long long *out, int count) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(arvae::g_d32k_stamps), sizeof(unsigned long long) * count);
}
#endif
#ifdef WGR_STAMPS
extern "C" int arvae_debug_wgr_stamps(unsigned long long *out, int count) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(arvae::g_wgr_stamps), sizeof(unsigned long long) * count);
}
#endif
