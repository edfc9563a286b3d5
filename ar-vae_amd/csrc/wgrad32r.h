// Weight gradient of the 32-channel k4/s2/p1 links as a ROW STREAM with producer / consumer waves (gfx950).
//
//   dW[ky][kx][clo][chi] = sum over lo pixels (n, r, c) of  lo[n][r][c][clo] * hi[n][2r-1+ky][2c-1+kx][chi]
//
// wgrad32x_kernel (conv32.hip) stages a whole 64-pixel patch (+25 % halo rows), splits it and only then starts its MFMAs:
// commit and MFMA phases are serial and the MFMA pipe is busy a third of the time.  Here
// (measured at B = 512 with the three-term bf16 arithmetic of rounds 1-3, rocprofv3: 32.4 us against 35.3 us for the 16x16
// layers, 15.2 against 15.7 us for the 8x8 ones)
//   * the hi tensor is walked as ONE stream of rows (images are contiguous, so row G = 2*LO*n + y is a linear walk through
//     memory): a step is 32 lo pixels (16 KB of new hi rows + 4 KB of lo), the rows it shares with the previous step stay in
//     an LDS ring -- every hi value is fetched from HBM and split exactly once (no halo re-reads);
//   * waves 4-7 (producers) load the next steps' rows into registers (three steps in flight), split them into bf16 terms
//     and write the ring while waves 0-3 (consumers, wave = ky, four accumulator tiles = kx) run the previous step's
//     MFMAs: one barrier per step, the vector ALU work of the split co-issues with the bf16 MFMAs of the SIMD's other wave;
//   * zero padding: the column halo is two LDS pixels per ring row that are zeroed once; a row above / below an image is
//     whatever the stream holds there (the neighbouring image's row), so the consumers zero the lo values of the pixels
//     whose tap row falls outside their image instead (ky = 0 on an image's first lo row, ky = 3 on its last).
// Same numerics as wgrad32x_kernel: scaled two-term fp16 operands (conv32_common.h), three partial products, smallest first,
// fp32 accumulation, the inverse scales applied to the slab, per-workgroup slabs reduced in fixed order (reduce.hip).
#pragma once
#include "common.h"
#include "conv32_common.h"
#include "reduce.h"

namespace arvae {

template <int LO> struct RowStream {
    static constexpr int HW = 2 * LO;                    // hi pixels per row
    static constexpr int RB = HW * PIXB;                 // bytes per hi row
    static constexpr int HN = 16384 / RB;                // new hi rows per step (32 lo pixels = 32 / LO lo rows)
    static constexpr int TR = 32 / LO;                   // lo rows per step
    static constexpr int RING = 3 * HN;                  // ring rows: HN + 2 in use by the consumers, HN being written
    static constexpr int PCOLS = HW + 2;                 // LDS pixels per ring row (column halo left and right)
    static constexpr int HP = WGRAD_PSB_H, LP = WGRAD_PSB_L;   // dwords per pixel and plane (see wgrad32x_kernel)
    static constexpr int HPLANE = RING * PCOLS * HP;     // dwords per hi plane
    static constexpr int LPLANE = 32 * LP, LBUF = 2 * LPLANE;
    static constexpr int LDS_DW = 2 * HPLANE + 2 * LBUF;
    static constexpr int PRO = 2 * RB / 4096;            // prologue slots per producer thread (the two rows above a range)
    static constexpr int STEPS_PER_IMG = LO * LO / 32;
};

#ifdef WGR_STAMPS
// diagnostic build only (tools/stamp_wgr.py): [workgroup][role 0 consumer / 1 producer][slot][cycle counter, 100 MHz wall clock]
__device__ unsigned long long g_wgr_stamps[256 * 2 * 64 * 2];
#define WGR_STAMP(role, slot)                                                                             \
    do {                                                                                                  \
        if ((threadIdx.x & 255) == 0 && BID < 256 && (slot) < 64) {                                \
            g_wgr_stamps[((BID * 2 + (role)) * 64 + (slot)) * 2] = __builtin_readcyclecounter();   \
            g_wgr_stamps[((BID * 2 + (role)) * 64 + (slot)) * 2 + 1] = wall_clock64();             \
        }                                                                                                 \
    } while (0)
#else
#define WGR_STAMP(role, slot)
#endif

constexpr int WGR_DEPTH = 3;                             // steps of global loads in flight per producer thread

// BID: the workgroup's index in the grid
template <int LO, int BIAS>
__device__ __forceinline__ void wgrad32r_body(const float *__restrict__ lo, const float *__restrict__ hi, float *__restrict__ slab,
                                              int n_img, int total_steps, int steps_per_wg, const unsigned *amax_lo,
                                              const unsigned *amax_hi, const int BID) {
    using RS = RowStream<LO>;
    constexpr int HN = RS::HN, RING = RS::RING, PCOLS = RS::PCOLS, HP = RS::HP, LP = RS::LP, RB = RS::RB;
    constexpr int HPLANE = RS::HPLANE, LPLANE = RS::LPLANE, LBUF = RS::LBUF;
    extern __shared__ __attribute__((aligned(16))) unsigned ring[];     // hi: 2 planes of RING rows | lo: 2 buffers x 2 planes
    unsigned *lo_w = ring + 2 * HPLANE;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int k0 = BID * steps_per_wg;
    const int k1 = min(k0 + steps_per_wg, total_steps);
    float *out = slab + (int64_t)BID * SLAB_C32_FLOATS;

    if (wave >= 4) {
        // ================================================================ producers
        const int pt = threadIdx.x - 256, chunk = pt & 7;
        const __amdgpu_buffer_rsrc_t rs_hi = make_rsrc(hi, (int64_t)n_img * RS::HW * RB);
        const __amdgpu_buffer_rsrc_t rs_lo = make_rsrc(lo, (int64_t)n_img * LO * LO * PIXB);
        float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);         // BIAS 1: lo sums, BIAS 2: hi sums, channels 4 chunk .. +3
        // (the split: split_pair_h2, conv32_common.h -- single-issue instructions only)
        float sc_h = 1.f, sc_l = 1.f;                            // the operands' scales (set behind the first loads)
        const AmaxLoad al_h = amax_issue(amax_hi), al_l = amax_issue(amax_lo);
        auto put = [&](unsigned *dst, int plane, const float4 &v, float sc) __attribute__((always_inline)) {
            uint2 hv, lv;
            split_pair_h2(v.x, v.y, sc, hv.x, lv.x);
            split_pair_h2(v.z, v.w, sc, hv.y, lv.y);
            *reinterpret_cast<uint2 *>(dst) = hv;
            *reinterpret_cast<uint2 *>(dst + plane) = lv;
        };
        // ring slot of stream row HN*k + j, j in [-1, HN]
        auto slot_of = [&](int k, int j) { const int s = HN * (k % 3) + j; return s < 0 ? s + RING : (s >= RING ? s - RING : s); };
        // byte offset o inside a run of whole rows -> (row, pixel) of this thread's 16 bytes
        auto row_of = [&](int o) { return o / RB; };
        auto px_of = [&](int o) { return (o % RB) / PIXB; };

        // every load of the first steps is issued before anything else (the first barrier waits for a cold HBM round trip)
        float4 pv[RS::PRO];                                      // the two rows above this range's first step: HN*k0 - 1, HN*k0
        {
            const unsigned base = (unsigned)((HN * k0 - 1) * RB);       // wraps for k0 = 0: out of range = zeros
#pragma unroll
            for (int j = 0; j < RS::PRO; ++j) {
                const int o = j * 4096 + pt * 16;
                // row -1 of the tensor does not exist; row HN*k0 does: test the row, not the run
                const bool ok = k0 < k1 && (HN * k0 - 1 + row_of(o)) >= 0;
                pv[j] = buf_load4(rs_hi, ok ? base + (unsigned)o : OOB);
            }
        }
        float4 hv[WGR_DEPTH][4], lv[WGR_DEPTH];
        auto issue = [&](auto dc, int k) __attribute__((always_inline)) {
            constexpr int d = decltype(dc)::value;
            const bool ok = k < k1;
            const unsigned hb = (unsigned)((HN * k + 1) * RB) + pt * 16, lb = (unsigned)k * 4096u + pt * 16;
#pragma unroll
            for (int j = 0; j < 4; ++j) hv[d][j] = buf_load4(rs_hi, ok ? hb + j * 4096 : OOB);
            lv[d] = buf_load4(rs_lo, ok ? lb : OOB);
        };
        auto commit = [&](auto dc, int k) __attribute__((always_inline)) {
            constexpr int d = decltype(dc)::value;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = j * 4096 + pt * 16;
                put(ring + (slot_of(k, row_of(o) + 1) * PCOLS + px_of(o) + 1) * HP + chunk * 2, HPLANE, hv[d][j], sc_h);
                if (BIAS == 2) { bias4.x += hv[d][j].x; bias4.y += hv[d][j].y; bias4.z += hv[d][j].z; bias4.w += hv[d][j].w; }
            }
            put(lo_w + (k & 1) * LBUF + (pt >> 3) * LP + chunk * 2, LPLANE, lv[d], sc_l);
            if (BIAS == 1) { bias4.x += lv[d].x; bias4.y += lv[d].y; bias4.z += lv[d].z; bias4.w += lv[d].w; }
        };
        WGR_STAMP(1, 0);
        static_for<0, WGR_DEPTH>([&](auto dc) __attribute__((always_inline)) { issue(dc, k0 + decltype(dc)::value); });
        sc_h = amax_scale(al_h).s;
        sc_l = amax_scale(al_l).s;
        // column halo of every ring row: zero, once
        for (int e = pt; e < 2 * RING * 2 * 16; e += 256) {
            const int dw = e & 15, side = (e >> 4) & 1, row = (e >> 5) % RING, t = (e >> 5) / RING;
            ring[t * HPLANE + (row * PCOLS + side * (PCOLS - 1)) * HP + dw] = 0u;
        }
#pragma unroll
        for (int j = 0; j < RS::PRO; ++j) {
            const int o = j * 4096 + pt * 16;
            put(ring + (slot_of(k0, row_of(o) - 1) * PCOLS + px_of(o) + 1) * HP + chunk * 2, HPLANE, pv[j], sc_h);
            if (BIAS == 2 && k0 == 0) { bias4.x += pv[j].x; bias4.y += pv[j].y; bias4.z += pv[j].z; bias4.w += pv[j].w; }
        }
        commit(std::integral_constant<int, 0>{}, k0);
        WGR_STAMP(1, 1);
        __syncthreads();                                         // step k0 is in LDS
        // step k is being consumed: refill its register set with step k + DEPTH, commit step k + 1
#if defined(WGR_NO_PRODUCE)                                     /* ablation build: barriers only */
#define ARVAE_WGR_PRODUCE(D)                                                          \
        if (k + D >= k1) break;                                                       \
        __syncthreads();
#elif defined(WGR_NO_COMMIT)                                    /* ablation build: loads, no split / LDS writes */
#define ARVAE_WGR_PRODUCE(D)                                                          \
        if (k + D >= k1) break;                                                       \
        issue(std::integral_constant<int, D>{}, k + D + WGR_DEPTH);                   \
        asm volatile("" ::"v"(hv[(D + 1) % WGR_DEPTH][0]), "v"(hv[(D + 1) % WGR_DEPTH][3]), "v"(lv[(D + 1) % WGR_DEPTH])); \
        __syncthreads();
#else
#define ARVAE_WGR_PRODUCE(D)                                                          \
        if (k + D >= k1) break;                                                       \
        WGR_STAMP(1, 2 + 2 * (k + D - k0));                                           \
        issue(std::integral_constant<int, D>{}, k + D + WGR_DEPTH);                   \
        commit(std::integral_constant<int, (D + 1) % WGR_DEPTH>{}, k + D + 1);        \
        WGR_STAMP(1, 3 + 2 * (k + D - k0));                                           \
        __syncthreads();
#endif
        for (int k = k0;; k += WGR_DEPTH) {
            ARVAE_WGR_PRODUCE(0)
            ARVAE_WGR_PRODUCE(1)
            ARVAE_WGR_PRODUCE(2)
        }
#undef ARVAE_WGR_PRODUCE
        WGR_STAMP(1, 63);
        if (BIAS != 0) {
            // every producer thread summed channel chunk pt & 7; fold the 32 partial sums per chunk in fixed order
            float *red = reinterpret_cast<float *>(ring);        // the consumers are past their last LDS read (final barrier)
            *reinterpret_cast<float4 *>(red + pt * 4) = bias4;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            // producers only: a named barrier is not available, so the four producer waves meet through the full barrier below
        }
    } else {
        // ================================================================ consumers (wave = ky)
        const int lane = threadIdx.x & 63, half = lane >> 5, rc = lane & 31;
        const int g16 = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
        const AmaxLoad al_h = amax_issue(amax_hi), al_l = amax_issue(amax_lo);
        // per K block b (16 lo pixels) and read i: pixel P = 16 b + 8 half + 4 i + q of the step; lo row r = P / LO
        int loff[2][2], hcol[2][2], hrow[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            hrow[b] = 2 * ((16 * b + 8 * half) / LO) + wave;            // patch row (0 = stream row HN*k - 1) of this lane's pixels
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int P = 16 * b + 8 * half + 4 * i + q;
                loff[b][i] = P * LP + 8 * (g16 & 1) + 2 * pp;
                hcol[b][i] = (2 * (P % LO)) * HP + 8 * (g16 & 1) + 2 * pp;          // + kx * HP
            }
        }
        f32x16 acc[4];
#pragma unroll
        for (int kx = 0; kx < 4; ++kx)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[kx][i] = 0.f;
        // A step is two K blocks (16 lo pixels each) of 12 MFMAs: three partial products for each of the four tap columns kx.
        // Back-to-back MFMAs into the SAME accumulator cost ~48 cycles each on gfx950 (measured here: 45.6 per MFMA with six
        // dependent ones in a row, against the 32-cycle issue rate), so the products are issued round-robin over the four
        // kx accumulators (every accumulator still sees its three products in the same order, smallest first).  The transposed LDS reads of the next block are issued before the MFMAs of the current one (two operand
        // sets), and the step's barrier sits between the last block's reads and its MFMAs, so the first reads of the next
        // step fly during those MFMAs: the matrix pipe never waits for an LDS round trip.
        f16x8 a2[2][4][2], b2[2][2];
        const unsigned *lbuf = lo_w, *hrow0 = ring, *hrow1 = ring;
        auto set_step = [&](int k) __attribute__((always_inline)) {
            const int slot0 = (HN * (k % 3) + RING - 1) % RING;  // ring slot of stream row HN*k - 1
            lbuf = lo_w + (k & 1) * LBUF;
            int s0 = slot0 + hrow[0], s1 = slot0 + hrow[1];
            s0 = s0 >= RING ? s0 - RING : s0;
            s1 = s1 >= RING ? s1 - RING : s1;
            hrow0 = ring + s0 * (PCOLS * HP);
            hrow1 = ring + s1 * (PCOLS * HP);
        };
        auto read_block = [&](auto bc) __attribute__((always_inline)) {
            constexpr int b = decltype(bc)::value;
#ifdef WGR_NO_READS                                             // ablation build: MFMAs on whatever the registers hold
            return;
#endif
#pragma unroll
            for (int t = 0; t < 2; ++t) b2[b][t] = lds_tr_f16x8(lbuf + t * LPLANE + loff[b][0], lbuf + t * LPLANE + loff[b][1]);
            const unsigned *hb = b == 0 ? hrow0 : hrow1;
#pragma unroll
            for (int kx = 0; kx < 4; ++kx)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    a2[b][kx][t] = lds_tr_f16x8(hb + kx * HP + t * HPLANE + hcol[b][0], hb + kx * HP + t * HPLANE + hcol[b][1]);
        };
        // tap row outside the image for this lane's 8 pixels (one lo row or part of one): their lo values become zero
        auto mask_b = [&](auto bc, int k) __attribute__((always_inline)) {
            constexpr int b = decltype(bc)::value;
            if (wave == 0 || wave == 3) {
                const int rimg = (RS::TR * k + (16 * b + 8 * half) / LO) & (LO - 1);
                const bool outside = wave == 0 ? rimg == 0 : rimg == LO - 1;
                typedef int i32x4q __attribute__((ext_vector_type(4)));
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    i32x4q v = __builtin_bit_cast(i32x4q, b2[b][t]);
                    v.x = outside ? 0 : v.x; v.y = outside ? 0 : v.y; v.z = outside ? 0 : v.z; v.w = outside ? 0 : v.w;
                    b2[b][t] = __builtin_bit_cast(f16x8, v);
                }
            }
        };
        auto mfma_block = [&](auto bc) __attribute__((always_inline)) {
            constexpr int b = decltype(bc)::value;
#ifdef WGR_NO_MFMA                                              // ablation build: reads stay, products go
#pragma unroll
            for (int kx = 0; kx < 4; ++kx)
                asm volatile("" ::"v"(a2[b][kx][0]), "v"(a2[b][kx][1]), "v"(b2[b][0]), "v"(b2[b][1]));
#else
            // (a term, b term) of the three partial products, smallest first
#define ARVAE_WGR_PRODUCT(TA, TB)                                                     \
            _Pragma("unroll") for (int kx = 0; kx < 4; ++kx) MFMA_H(acc[kx], a2[b][kx][TA], b2[b][TB]);
            ARVAE_WGR_PRODUCT(1, 0)
            ARVAE_WGR_PRODUCT(0, 1)
            ARVAE_WGR_PRODUCT(0, 0)
#undef ARVAE_WGR_PRODUCT
#endif
        };
        const float inv = amax_scale(al_l).inv * amax_scale(al_h).inv;     // accumulators -> fp32 partial sums (exact)
        WGR_STAMP(0, 0);
        __syncthreads();                                         // step k0 is in LDS
        WGR_STAMP(0, 1);
        set_step(k0);
        read_block(std::integral_constant<int, 0>{});
        // issue order of one block: an MFMA, then two or one of the next block's 20 transposed reads = 40 ds_read_b64_tr_b16
        // (they ride in the MFMA's shadow; issued as a burst in front of the block each read costs the matrix pipe ~10 idle
        // cycles: tools/probes/mfma_barrier.hip)
#define ARVAE_WGR_INTERLEAVE                                                                                      \
        { _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) {                                                      \
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 4, 0); } \
          _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) {                                                      \
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x100, 2, 0); } }
        for (int k = k0; k < k1; ++k) {
            __builtin_amdgcn_sched_barrier(0);
            mask_b(std::integral_constant<int, 0>{}, k);
            read_block(std::integral_constant<int, 1>{});
            mfma_block(std::integral_constant<int, 0>{});
#if !defined(WGR_NO_MFMA) && !defined(WGR_NO_READS)
            ARVAE_WGR_INTERLEAVE
#endif
            __builtin_amdgcn_sched_barrier(0);
            WGR_STAMP(0, 2 + 2 * (k - k0));
            __syncthreads();                                     // every read of step k has landed; step k + 1 is in LDS
            WGR_STAMP(0, 3 + 2 * (k - k0));
            set_step(k + 1);
            mask_b(std::integral_constant<int, 1>{}, k);
            __builtin_amdgcn_sched_barrier(0);
            read_block(std::integral_constant<int, 0>{});        // (past the last step: harmless reads of stale rows)
            mfma_block(std::integral_constant<int, 1>{});
#if !defined(WGR_NO_MFMA) && !defined(WGR_NO_READS)
            ARVAE_WGR_INTERLEAVE
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
#undef ARVAE_WGR_INTERLEAVE
        WGR_STAMP(0, 62);
        // partial results (times the operands' inverse scales: exact) -> slab[blockIdx][ky][kx][clo = rc][chi = 8g + 4*half + j]
#pragma unroll
        for (int kx = 0; kx < 4; ++kx)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *reinterpret_cast<float4 *>(out + ((wave * 4 + kx) * C32 + rc) * C32 + 8 * g + 4 * half) =
                    make_float4(acc[kx][4 * g] * inv, acc[kx][4 * g + 1] * inv, acc[kx][4 * g + 2] * inv, acc[kx][4 * g + 3] * inv);
        WGR_STAMP(0, 63);
    }
    if (BIAS != 0) {
        __syncthreads();                                         // the producers' partial sums are in LDS
        if (threadIdx.x < C32) {
            const float *red = reinterpret_cast<const float *>(ring);
            const int qq = threadIdx.x >> 2, e = threadIdx.x & 3;
            float tot = 0.f;
            for (int j = 0; j < 32; ++j) tot += red[(j * 8 + qq) * 4 + e];
            out[16 * C32 * C32 + threadIdx.x] = tot;
        }
    }
}

template <int LO, int BIAS>
__global__ __launch_bounds__(512, 2) void wgrad32r_kernel(const float *__restrict__ lo, const float *__restrict__ hi,
                                                          float *__restrict__ slab, int n_img, int total_steps, int steps_per_wg,
                                                          const unsigned *amax_lo, const unsigned *amax_hi) {
    wgrad32r_body<LO, BIAS>(lo, hi, slab, n_img, total_steps, steps_per_wg, amax_lo, amax_hi, blockIdx.x);
}

}  // namespace arvae
