// Down map of the 32-channel k4/s2/p1 links (Conv2d forward, ConvTranspose2d data gradient) at 16x16 and 8x8 output size,
// FOUR-WAY REDUCTION-SPLIT form with PRODUCER / CONSUMER waves (gfx950).
//
//   lo[n,ly,lx,clo] = ep( sum_{ky,kx,chi} hi[n,2ly-1+ky,2lx-1+kx,chi] * wt[clo][chi][ky][kx] )
//
// down32x_kernel (conv32.hip) splits the reduction two ways with every wave loader and multiplier.  Here the reduction is split
// FOUR ways (consumer wave = kernel row): a wave's weights are 64 registers (resident for the whole launch, 16 coalesced loads
// from the prepared DOWN part, prep32.h) and it owns two accumulator tiles (its MFMAs alternate between them).
//   * a tile = 64 lo pixels (4 rows at 16x16, one image at 8x8) = two 32-pixel MFMA column tiles; weight = A operand
//     (row = output channel), pixels = B operand, 32x32x16 fp16 on scaled two-term operands, three partial products per
//     multiply-add, smallest first (conv32_common.h);
//   * consumer wave = kernel row ky (K split four ways): 8 reduction steps (kx, 16-channel chunk) of 6 MFMAs per tile.  The
//     four partial sums meet through a 24 KB LDS area; wave w finishes column tile w & 1, channel groups 2 (w >> 1),
//     2 (w >> 1) + 1 (16-byte stores, one byte of ReLU sign bits per lane);
//   * the input patch (2 TR + 2 rows x 2 LO columns, no column halo: a tap outside reads one shared zero pixel) is scaled and
//     split into two fp16 terms once and packed per pixel (PSB2), two LDS buffers: the producer waves write tile t+1's image
//     while tile t is multiplied and keep the loads of tiles t+2, t+3 in flight.
// Round 2's form of this kernel (every wave loader AND multiplier, 256 threads) is gone: see the comment at the kernel.
#pragma once
#include "common.h"
#include "conv32_common.h"

namespace arvae {

#ifdef D32K_STAMPS
// diagnostic build only (tools/stamp_d32p.py): phase timeline of the first 32 workgroups, 100 MHz wall clock
__device__ unsigned long long g_d32k_stamps[64 * 64];
// down32p_kernel<16, *>: rows 0..31 = the consumers of workgroups 0..31 (thread 0), rows 32..63 = their producers (thread 256)
#define DSTAMP(role, slot) do { if (LO == 16 && threadIdx.x == 256 * (role) && BID < 32 && (slot) < 64) g_d32k_stamps[(BID + 32 * (role)) * 64 + (slot)] = wall_clock64(); } while (0)
#else
#define DSTAMP(role, slot)
#endif

template <int LO> struct DownK {
    using T = Tile<LO, 64>;
    static constexpr int HW = 2 * LO;                          // hi pixels per row
    static constexpr int PR = 2 * T::TR + 2;                   // patch rows
    static constexpr int PIX = PR * HW;                        // staged pixels (320 / 288)
    static constexpr int BUF = (PIX + 1) * PSB2;               // dwords, + the zero pixel
    static constexpr int SLOTS = (PIX * 8 + 255) / 256;        // 16-byte loader slots per thread and tile (10 / 9)
    static constexpr int XCH = 4 * 3 * 2 * 64 * 4;             // exchange area, dwords: [owner][source][8 regs as 2 x float4][lane]
    static constexpr int LDS_DW = 2 * BUF + XCH;
    static constexpr int TILES_PER_IMG = LO / T::TR;
};

// ================================================================================================================================
// Producer / consumer waves (round 3).  In round 2's down32k_kernel every wave was loader AND multiplier: per reduction step a
// wave issued 12 MFMAs and ~90 vector instructions of loading, splitting and LDS writing between them, and one wave per SIMD hides
// at most ~5 per MFMA: the step took 1.5x its MFMA issue time (stamps: 2.3 us per tile for 1.54 us of MFMA; 29-33 us per 16x16
// launch).  Here, as in wgrad32r_kernel, the two jobs live in different waves of the same SIMD (same-box A/B, three boxes:
// 1.5 / 3.5 / 7.4 us per training step in favour of this form; stamps: tools/stamp_d32p.py, profiles/r3_phase_stamps.txt): waves 0-3 (consumers, wave = kernel row) keep the weights, read their
// operands from LDS and issue MFMAs -- 6 LDS reads per 12 MFMAs and nothing else in the reduction loop; waves 4-7 (producers)
// fetch tile t+1 .. t+3, scale and split (single-issue instructions: they co-issue beside the partner's MFMAs) and write
// tile t+1's LDS image while tile t is multiplied.  (Round 3 measured this with six bf16 products per multiply-add: 1.9-2.1 us
// per tile for 1.54 us of MFMA issue, the producers done after 1.4 us; with three fp16 products the MFMA issue is 0.77 us.)
// (a body: the launch may carry another kernel's workgroups beside these -- BID / NBLK: this workgroup's index and their number)
// CHAIN (round 6, chain_down_kernel in conv32.hip: two layers of the forward pass in one launch, workgroup w running the lower layer
// on exactly the images whose upper layer it has just computed): 1 = the first body of a chain -- its consumer waves leave the
// maxima of what they stored in chain_max[0..3] (LDS); 2 = the second body -- the input's scale comes from those four values
// (the WORKGROUP's own maximum: a scale need only cover the values this workgroup reads; the tensor-wide AMAX array is complete
// only when the whole launch is), ep.amax_in is not read.
template <int LO, int MODE, int CHAIN = 0>
__device__ __forceinline__ void down32p_body(const float *__restrict__ hi, Ep32 ep, int n_img, int n_tiles, const int BID, const int NBLK,
                                             float *chain_max = nullptr) {
    using K = DownK<LO>;
    constexpr int HW = K::HW, PIX = K::PIX, SLOTS = K::SLOTS, HI = 2 * LO;
    extern __shared__ __attribute__((aligned(16))) unsigned ldsd[];
    unsigned *xch = ldsd + 2 * K::BUF;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int per_wg = (n_tiles + NBLK - 1) / NBLK, t_first = BID * per_wg;
    const int t_end = min(n_tiles, t_first + per_wg);
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    // the input's scale: from its AMAX array, or (second body of a chain) from the workgroup's own four maxima
    auto chain_scale = [&]() __attribute__((always_inline)) -> Pow2 {
        const float m = fmaxf(fmaxf(chain_max[0], chain_max[1]), fmaxf(chain_max[2], chain_max[3]));
        return pow2_for((unsigned)__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, m)));
    };

    if (wave >= 4) {
        // ============================================================================================ producers
        DSTAMP(1, 0);
        int pst = 0;
        (void)pst;
        const int pt = threadIdx.x - 256;
        const __amdgpu_buffer_rsrc_t rs_hi = make_rsrc(hi, (int64_t)n_img * HI * HI * PIXB);
        int q = pt & 7, pix0 = pt >> 3;                          // slot s of this thread = staged pixel pix0 + 32 s, channels 4 q .. 4 q + 3
        float4 lv[2][SLOTS];
        float sc_in = 1.f;                                       // the input tensor's scale (set behind the first tiles' loads)
        AmaxLoad al{};
        if constexpr (CHAIN != 2) al = amax_issue(ep.amax_in);
        auto issue = [&](auto set_, int s, int tile) __attribute__((always_inline)) {
            constexpr int set = decltype(set_)::value;
            int img0, r0;
            tile_origin<LO, 64>(tile, img0, r0);
            const int pix = pix0 + 32 * s, pr = pix / HW, px = pix - pr * HW;
            const int gy = 2 * r0 - 1 + pr;
            const bool ok = tile < t_end && pix < PIX && (unsigned)gy < (unsigned)HI;
            lv[set][s] = buf_load4(rs_hi, ok ? (unsigned)((((img0 * HI + gy) * HI + px) * C32 + 4 * q) * 4) : OOB);
        };
        auto commit = [&](auto set_, int s, unsigned *buf) __attribute__((always_inline)) {
            constexpr int set = decltype(set_)::value;
            const int pix = pix0 + 32 * s;
            if (SLOTS * 32 > PIX && pix >= PIX) return;
            uint2 hv, lw;
            split_pair_h2(lv[set][s].x, lv[set][s].y, sc_in, hv.x, lw.x);
            split_pair_h2(lv[set][s].z, lv[set][s].w, sc_in, hv.y, lw.y);
            unsigned *d = buf + pix * PSB2 + q * 2;
            *reinterpret_cast<uint2 *>(d) = hv;
            *reinterpret_cast<uint2 *>(d + 16) = lw;
        };
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) issue(S0{}, s, t_first);
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) issue(S1{}, s, t_first + 1);
        if (pt < 2 * PSB2) ldsd[(pt / PSB2) * K::BUF + PIX * PSB2 + pt % PSB2] = 0u;   // the zero pixels
        if constexpr (CHAIN != 2) sc_in = amax_scale(al).s;
        else sc_in = chain_scale().s;
        __syncthreads();                                         // (the consumers' prologue barrier)
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) commit(S0{}, s, ldsd);    // first tile -> buffer 0
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) issue(S0{}, s, t_first + 2);
        __syncthreads();
        // per thread, for the loads issued inside the tile loop (hand-reduced): byte offset of slot 0 in a patch and the slots
        // (bit s) that sit in the patch's first / last row or past its end -- a slot then costs a test, an add and a select
        const unsigned rel0 = (unsigned)(pix0 * PIXB + q * 16);
        unsigned first_row = 0, last_row = 0, no_slot = 0;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int pix = pix0 + 32 * s, pr = pix / HW;
            if (pr == 0) first_row |= 1u << s;
            if (pr == PIX / HW - 1) last_row |= 1u << s;
            if (pix >= PIX) no_slot |= 1u << s;
        }
        // tile `tile` is being multiplied from buffer cur: register set SET (tile + 1) goes to the other buffer and is refilled
        // with tile + 3
        auto do_tile = [&](auto set_, int tile, int cur) __attribute__((always_inline)) {
            constexpr int set = decltype(set_)::value;
            asm volatile("" : "+v"(q), "+v"(pix0));              // keep the slot addresses out of loop-invariant hoisting (spills)
            DSTAMP(1, 4 + 5 * pst);
            unsigned *nb = ldsd + (cur ^ 1) * K::BUF;
            unsigned n_base, n_bad;
            {
                const int nt = tile + 3;
                int ni, nr;
                tile_origin<LO, 64>(nt, ni, nr);
                const int gy0 = 2 * nr - 1;
                n_base = nt < t_end ? (unsigned)(((ni * HI + gy0) * HI) * PIXB) + rel0 : OOB;
                n_bad = (gy0 < 0 ? first_row : 0u) | (gy0 + PIX / HW - 1 >= HI ? last_row : 0u) | no_slot;
            }
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                commit(set_, s, nb);
                lv[set][s] = buf_load4(rs_hi, (n_bad & (1u << s)) != 0 ? OOB : n_base + 4096u * s);
            }
            DSTAMP(1, 5 + 5 * pst);
            __syncthreads();                                     // exchange area free
            DSTAMP(1, 6 + 5 * pst);
            __syncthreads();                                     // every read of `cur` is done, `cur ^ 1` is staged
            DSTAMP(1, 7 + 5 * pst);
            ++pst;
        };
        for (int tile = t_first; tile < t_end; tile += 2) {
            do_tile(S1{}, tile, 0);
            do_tile(S0{}, tile + 1, 1);
        }
        return;
    }

    // ================================================================================================ consumers (wave = kernel row)
    DSTAMP(0, 0);
    int cst = 0;
    (void)cst;
    const int half = lane >> 5, rc = lane & 31;
    AmaxLoad al{};
    if constexpr (CHAIN != 2) al = amax_issue(ep.amax_in);
    int xoff[2][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        int img, r, c;
        tile_pixel<LO, 64>(mt * 32 + rc, img, r, c);
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
            const int col = 2 * c - 1 + kx;
            xoff[mt][kx] = ((unsigned)col < (unsigned)HW ? (2 * r + wave) * HW + col : PIX) * PSB2 + half * 4;
        }
    }
    const int om = wave & 1, og = wave >> 1;
    float4 b4[2];
#pragma unroll
    for (int e = 0; e < 2; ++e)
        b4[e] = ep.bias != nullptr ? *reinterpret_cast<const float4 *>(ep.bias + 8 * (2 * og + e) + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
    const int64_t out_bytes = (int64_t)n_img * LO * LO * PIXB;
    const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(ep.out, out_bytes);
    const __amdgpu_buffer_rsrc_t rs_gate = make_rsrc(MODE == EP_GATE_F ? ep.gate : ep.out, out_bytes);
    const bool want_bits = MODE == EP_RELU && ep.bits_out != nullptr;
    const __amdgpu_buffer_rsrc_t rs_bits =
        make_rsrc(MODE == EP_GATE_B ? (const void *)ep.gate_bits : want_bits ? (const void *)ep.bits_out : (const void *)ep.out,
                  (int64_t)n_img * LO * LO * 4);
    const unsigned out_lane = (unsigned)((om * 32 + rc) * PIXB + (2 * og) * 32 + half * 16);
    f16x8 w2[8][2];
    {
        const uint4 *wp = ep.wprep + ((wave >> 1) * PREP_DOWN_SLOTS + ((wave & 1) * 4) * 2 * 2) * 64 + lane;
#pragma unroll
        for (int st = 0; st < 8; ++st)
#pragma unroll
            for (int t = 0; t < 2; ++t) w2[st][t] = __builtin_bit_cast(f16x8, wp[(st * 2 + t) * 64]);
    }
    float inv;                                                   // accumulators -> fp32 results (exact)
    if constexpr (CHAIN != 2) inv = amax_scale(al).inv * prep_inv_scale(ep.wprep);
    else inv = chain_scale().inv * prep_inv_scale(ep.wprep);
    float amax_run = 0.f;                                        // maximum magnitude of what this wave stores
    __syncthreads();                                             // zero pixels written
    __syncthreads();                                             // first tile staged
    DSTAMP(0, 1);

    // The epilogue of tile t (sum of the four kernel rows' partial tiles, bias, ReLU / gate, stores) is DEFERRED into the
    // reduction loop of tile t + 1, one small piece behind an MFMA at a time: between two tiles the matrix pipe then waits for
    // the exchange alone (stamps, B = 512, 16x16 layer: k-loop 1.9 us, exchange 0.3 us, epilogue 0.4 us per tile before).
    float4 own[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};     // this wave's share of the previous tile
    float4 gq_prev[2] = {own[0], own[0]};
    unsigned obase_prev = OOB, gb_prev = 0;
    const float4 *xq_own = reinterpret_cast<const float4 *>(xch) + (wave * 3) * 2 * 64 + lane;
    unsigned bits_acc = 0;
    // the three other kernel rows' partial sums of this wave's share, requested at the top of a tile (an LDS read whose value
    // is used by the next instruction stalls the wave's MFMA stream for the whole round trip)
    float4 part[2][3];
    auto load_parts = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int k = 0; k < 3; ++k) part[e][k] = xq_own[(k * 2 + e) * 64];
    };
    // piece i of the deferred epilogue: 0-2 / 4-6: add the partial of source i (mod 4) to half e = i / 4; 3 / 7: finish + store
    // half e; 8: the sign bits
    auto epi_item = [&](auto ic) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i < 8) {
            constexpr int e = i / 4, k = i % 4;
            if constexpr (k < 3) {
                const float4 p = part[e][k];
                own[e].x += p.x; own[e].y += p.y; own[e].z += p.z; own[e].w += p.w;
            } else {
                float o4[4] = {fmaf(own[e].x, inv, b4[e].x), fmaf(own[e].y, inv, b4[e].y), fmaf(own[e].z, inv, b4[e].z), fmaf(own[e].w, inv, b4[e].w)};
                const float gf[4] = {gq_prev[e].x, gq_prev[e].y, gq_prev[e].z, gq_prev[e].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (MODE == EP_RELU) {
                        o4[j] = fmaxf(o4[j], 0.f);
                        bits_acc |= (o4[j] > 0.f ? 1u : 0u) << (4 * e + j);
                    }
                    if (MODE == EP_GATE_F) o4[j] = gf[j] > 0.f ? o4[j] : 0.f;
                    if (MODE == EP_GATE_B) o4[j] = ((gb_prev >> (8 * og + 4 * e + j)) & 1u) ? o4[j] : 0.f;
                }
                const float4 ov = make_float4(o4[0], o4[1], o4[2], o4[3]);
                amax_run = obase_prev != OOB ? fmaxf(amax_run, amax4(ov)) : amax_run;
                buf_store4(ov, rs_out, obase_prev + e * 32);
            }
        } else if constexpr (i == 8) {
            if (MODE == EP_RELU)        // unconditional (exact vmcnt counts in the loop); dropped through its offset without bits_out
                __builtin_amdgcn_raw_buffer_store_b8((unsigned char)bits_acc, rs_bits,
                                                     (int)((obase_prev == OOB || !want_bits) ? OOB : bits_off(obase_prev, half) + og), 0, 0);
            bits_acc = 0;
        }
    };
    auto do_tile = [&](int tile, int cur) __attribute__((always_inline)) {
        DSTAMP(0, 4 + 5 * cst);
        const unsigned *xb = ldsd + cur * K::BUF;
        // two accumulator sets per column tile: an MFMA that accumulates into the result of the one issued two slots earlier
        // waits for it (~44 cycles per MFMA measured with one set, 32 is the issue rate); the sets swap roles every step so
        // that four MFMAs always lie between two into the same registers
        f32x16 acc[2], accb[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) { acc[mt][i] = 0.f; accb[mt][i] = 0.f; }
        f16x8 x2[2][2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                x2[0][mt][t] = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4v *>(xb + xoff[mt][0] + t * 16));
        load_parts();
        int img0, r0;
        tile_origin<LO, 64>(tile, img0, r0);
        const unsigned obase = tile < t_end ? (unsigned)(((img0 * LO + r0) * LO) * PIXB) + out_lane : OOB;
        // gate values / sign bits of this tile's outputs: requested now, used one tile later
        float4 gq[2];
        unsigned gb = 0;
        if (MODE == EP_GATE_F) {
            gq[0] = buf_load4(rs_gate, obase);
            gq[1] = buf_load4(rs_gate, obase + 32);
        }
        if (MODE == EP_GATE_B) gb = buf_load_u16(rs_bits, bits_off(obase, half));
        // issue order pinned by hand (a scheduling barrier after every MFMA; left to the scheduler a step came out as a block of
        // MFMAs followed by a block of vector instructions): behind every MFMA at most one operand read of the next step or one
        // piece of the previous tile's epilogue (steps 1 .. 5: two pieces each, the sign bits last)
        static_for<0, 8>([&](auto kc) __attribute__((always_inline)) {
            constexpr int step = decltype(kc)::value, cu = step & 1, nx = cu ^ 1;
            constexpr int e_lo = step >= 1 ? 2 * (step - 1) : 0, e_hi = step >= 1 ? (2 * step < 9 ? 2 * step : 9) : 0;
            constexpr int n_epi = e_hi > e_lo ? e_hi - e_lo : 0, n_items = 4 + n_epi;
            auto item = [&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i < 4) {
                    if constexpr (step + 1 < 8) {
                        constexpr int nkx = (step + 1) >> 1, nc = (step + 1) & 1, mt = i / 2, t = i % 2;
                        x2[nx][mt][t] = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4v *>(xb + xoff[mt][nkx] + t * 16 + nc * 8));
                    }
                } else {
                    epi_item(std::integral_constant<int, e_lo + (i - 4)>{});
                }
            };
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, 6>([&](auto mc) __attribute__((always_inline)) {
                constexpr int m = decltype(mc)::value, prod = m >> 1, mt = m & 1;
                constexpr int tw = prod == 0 ? 1 : 0;            // (weight term, pixel term): (l, h), (h, l), (h, h)
                constexpr int tx = prod == 1 ? 1 : 0;
                if constexpr ((prod == 1) == ((step & 1) == 0)) MFMA_H(accb[mt], w2[step][tw], x2[cu][mt][tx]);
                else MFMA_H(acc[mt], w2[step][tw], x2[cu][mt][tx]);
                __builtin_amdgcn_sched_barrier(0);
                static_for<m * n_items / 6, (m + 1) * n_items / 6>(item);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][i] += accb[mt][i];
        DSTAMP(0, 5 + 5 * cst);
        __syncthreads();                                         // the previous tile's exchange has been read by everybody
        DSTAMP(0, 6 + 5 * cst);
        float4 *xq = reinterpret_cast<float4 *>(xch);
        // acc registers 4 (2 G + e) + j of column tile M: owner (M, G) = wave M + 2 G
        auto piece = [&](auto m_, auto g_, int e) __attribute__((always_inline)) -> float4 {
            constexpr int M = decltype(m_)::value, G = decltype(g_)::value;
            return e == 0 ? make_float4(acc[M][8 * G], acc[M][8 * G + 1], acc[M][8 * G + 2], acc[M][8 * G + 3])
                          : make_float4(acc[M][8 * G + 4], acc[M][8 * G + 5], acc[M][8 * G + 6], acc[M][8 * G + 7]);
        };
        static_for<0, 4>([&](auto oc) __attribute__((always_inline)) {      // owner ow receives this wave's piece as source (wave - ow - 1) & 3
            constexpr int ow = decltype(oc)::value;
            if (ow != wave) {
                const int src = (wave - ow - 1) & 3;
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    xq[((ow * 3 + src) * 2 + e) * 64 + lane] = piece(std::integral_constant<int, (ow & 1)>{}, std::integral_constant<int, (ow >> 1)>{}, e);
            }
        });
#pragma unroll
        for (int e = 0; e < 2; ++e)
            own[e] = wave == 0 ? piece(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, e)
                   : wave == 1 ? piece(std::integral_constant<int, 1>{}, std::integral_constant<int, 0>{}, e)
                   : wave == 2 ? piece(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, e)
                               : piece(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, e);
        obase_prev = obase;
        gb_prev = gb;
        if (MODE == EP_GATE_F) { gq_prev[0] = gq[0]; gq_prev[1] = gq[1]; }
        __syncthreads();                                         // partial sums written; every read of `cur` is done, `cur ^ 1` is staged
        DSTAMP(0, 7 + 5 * cst);
        DSTAMP(0, 8 + 5 * cst);
        ++cst;
    };
    for (int tile = t_first; tile < t_end; tile += 2) {
        do_tile(tile, 0);
        do_tile(tile + 1, 1);
    }
    load_parts();
    static_for<0, 9>(epi_item);                                  // the last tile's epilogue
    amax_publish(ep.amax_out, BID * 4 + wave, NBLK * 4, amax_run);
    if constexpr (CHAIN == 1) {
        const float m = wave_max(amax_run);
        if (lane == 0) chain_max[wave] = m;
    }
}
template <int LO, int MODE>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void down32p_kernel(const float *__restrict__ hi, Ep32 ep,
                                                                                                 int n_img, int n_tiles) {
    down32p_body<LO, MODE>(hi, ep, n_img, n_tiles, blockIdx.x, gridDim.x);
}

}  // namespace arvae
