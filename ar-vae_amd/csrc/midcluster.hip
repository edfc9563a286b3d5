// The latent block of the dSprites-shaped conv VAE (imagevae/dsprites_vae.py:22-37 as executed by imagevae/mnist_vae.py:59-72:
// Linear 512 -> 256 -> 256 -> (mu | log_std) -> z -> 256 -> 256 -> 512, and its backward) on CLUSTERS of workgroups.
//
// midblock.hip gives every workgroup a few batch rows and lets it stream EVERY matrix (1.6 MB) through its own CU's L2 port:
// ~20 us per pass however the arithmetic is done.  Here a cluster of MC_S = 16 workgroups owns MC_R = 32 batch rows, and
// each member owns 1/16 of every wide layer's output columns: a workgroup reads ~100 KB of weights per pass (each value
// once, straight into the B operand registers of v_mfma_f32_16x16x4_f32 -- 16-byte loads in the cluster layout midprep.h
// writes), the products are exact-fp32 MFMAs, and what the members exchange between two layers is the 32 x 256 activation
// block (32 KB), through memory: every member stores its 32 x 16 slice write-through (sc1), arrives on the cluster's counter,
// polls it, and gathers the block with sc1 loads (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup
// visibility": the first row of the hand-off table; the exchanged tensors ARE the saved activations / pre-activation gradients
// the other pass and the weight-gradient launch need anyway).  The narrow middle of the chain (heads, z, decoder's first layer;
// their transposes in backward) is computed by every member for its cluster's rows, so a pass has three exchanges, not six.
// Forward: 3 exchanges; backward: 3.  256 workgroups at B = 512, one per CU.
//
// Membership is decided AT RUN TIME, by tickets (mc_ticket): a workgroup that has started takes the next free place of the
// next cluster (a few ticket heads, each serving every heads-th workgroup).  Members wait for each other, so what matters is
// that the workgroups that are RESIDENT form complete clusters: with places handed out in order of arrival every 16 arrivals
// on a head complete one, whichever 16 they are -- a grid that only partly fits the device (another process holds CUs, a larger
// batch) runs cluster after cluster instead of waiting for workgroups that cannot start.  (With places fixed by blockIdx, as through round 4, two processes sharing the device could each hold half of
// every cluster: a hang.)  What is left -- fewer than 16 places can ever be resident -- ends in mc_wait's bound: the poll gives
// up after MC_WAIT_TICKS (1 s) of its own running time, ORs a code into the caller's status word (arvae_image_vae_t.status) and the
// workgroup leaves; the host raises on that word where it synchronises anyway and stays on the row kernels (midblock.hip).
#include <mutex>

#include "diag.h"
#include "conv32_common.h"
#include "midcluster.h"

namespace arvae {
namespace {

constexpr int MC_T = 512;                 // 8 waves
constexpr int PA = MC_K0 + 4;             // LDS row pitches (floats): pitch % 64 == 4 keeps the 16-byte A-operand reads of 16 rows
constexpr int PB = MC_H + 4;              // on disjoint bank quartets
constexpr int PZ = 20, PO = 36;           // z rows (16 + 4); heads outputs / (d_mu | d_ls) rows (32 + 4)
constexpr int RED_FLOATS = 4 * MC_R * 20; // partial tiles: 4 k-quarters x 32 rows x (16 + 4), or 2 halves x 32 x (32 + 4)
constexpr int LDS_FLOATS = MC_R * PA + MC_R * PB + RED_FLOATS + MC_R * PZ + MC_R * PO;
constexpr int AUX_SC1 = 16;               // cache-policy bit of the raw buffer intrinsics: sc1 (agent scope: bypass L1 / write through)

#ifdef MIDC_STAMPS
__device__ unsigned long long g_midc_stamps[2 * 256 * 16];     // [pass][workgroup][slot]: wall clock (100 MHz) at phase boundaries
#define MC_STAMP(pass, slot) do { if (threadIdx.x == 0 && blockIdx.x < 256) g_midc_stamps[((pass) * 256 + blockIdx.x) * 16 + (slot)] = wall_clock64(); } while (0)
#else
#define MC_STAMP(pass, slot)
#endif

#ifdef ARVAE_DIAG
// diagnostic build: what the first workgroups that gave up saw (arvae_debug_midc_failures)
__device__ unsigned g_mc_fail_count;
__device__ unsigned g_mc_fail[64][8];
#define MC_FAIL_RECORD(code, target, seen, tk) do { const unsigned slot_ = atomicAdd(&g_mc_fail_count, 1u); if (slot_ < 64) { \
    unsigned xcc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_)); \
    g_mc_fail[slot_][0] = (code); g_mc_fail[slot_][1] = blockIdx.x; g_mc_fail[slot_][2] = xcc_; g_mc_fail[slot_][3] = (target); \
    g_mc_fail[slot_][4] = (seen); g_mc_fail[slot_][5] = (unsigned)wall_clock64(); g_mc_fail[slot_][6] = (tk); g_mc_fail[slot_][7] = 0; } } while (0)
#else
#define MC_FAIL_RECORD(code, target, seen, tk)
#endif

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

// KB: blocks of 16 along the reduce axis; CT: 16-column tiles of this workgroup's output columns
template <int KB_, int CT_>
struct Shape {
    static constexpr int KB = KB_, CT = CT_;
    static constexpr int T = 2 * CT;                        // 16 x 16 tiles of the 32-row block
    static constexpr int KS = T <= 8 ? 8 / T : 1;           // waves sharing a tile (the reduce axis split between them)
    static constexpr int TPW = T <= 8 ? 1 : T / 8;          // tiles per wave
    static constexpr int NKB = KB / KS;                     // k blocks per wave and tile
    static constexpr int NW = TPW * NKB;                    // 16-byte weight loads per lane
    static constexpr int NS = 16 * CT;                      // output columns of the slice
    static constexpr int P = NS + 4;                        // pitch of the partial tiles
    static constexpr int SLICE = CT * KB * 256;             // floats of one slice in cluster layout
    static_assert(KB % KS == 0 && NW <= 16 && (T > 8 || NKB % 2 == 0) && (T > 8 || KS * MC_R * P <= RED_FLOATS), "shape");
};
using ShE0 = Shape<MC_K0 / 16, 1>;        // 512 -> 16 of 256
using ShHH = Shape<MC_H / 16, 1>;         // 256 -> 16 of 256
using ShHD = Shape<MC_H / 16, 2>;         // 256 -> 32 (mu | log_std), every member
using ShD0 = Shape<1, 16>;                // 16 (z) -> 256, every member
using ShD2 = Shape<MC_H / 16, 2>;         // 256 -> 32 of 512
using ShZB = Shape<MC_H / 16, 1>;         // 256 -> 16 (d z), every member
using ShHB = Shape<2, 16>;                // 32 (d_mu | d_ls) -> 256, every member
constexpr int MC_TAP_SLAB = 32 * 32 + 32; // a member's weight-gradient partial of a folded layer: its tap's [clo][chi] block + its bias sums
using ShCD = Shape<32, 2>;                // a folded conv layer's DOWN map: 512 (16 taps x 32 channels) -> the 32 channels of this member's lo pixel
constexpr int UW = 9 * 32, PU = UW + 4;   // an UP window: the 3 x 3 lo pixels around this member's, 32 channels each; its LDS pitch

// this wave's weights of one product: issued early, consumed by mc_mma
template <class S, int N>
__device__ __forceinline__ void mc_load(const float *__restrict__ w, float4 (&wr)[N]) {
    static_assert(S::NW <= N, "weight registers");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (S::T <= 8) {
        const int tile = wave % S::T, ct = tile >> 1, kq = wave / S::T;
#pragma unroll
        for (int u = 0; u < S::NKB; ++u) wr[u] = ld4(w + ((int64_t)((ct * S::KB + kq * S::NKB + u) * 64 + lane)) * 4);
    } else {
#pragma unroll
        for (int t = 0; t < S::TPW; ++t) {
            const int ct = (wave * S::TPW + t) >> 1;
#pragma unroll
            for (int u = 0; u < S::KB; ++u) wr[t * S::KB + u] = ld4(w + ((int64_t)((ct * S::KB + u) * 64 + lane)) * 4);
        }
    }
}

// acc[t] = rows of `in` (LDS, pitch ld) x this wave's weights, for its tile(s) and its part of the reduce axis.
// A operand: lane (g = lane / 16, r = lane % 16) holds in[row r][16 b + 4 g + j] for step j of block b (one 16-byte LDS read);
// D: lane holds rows 4 g + j, column lane % 16.
template <class S, int N>
__device__ __forceinline__ void mc_mma(const float *in, int ld, const float4 (&wr)[N], f32x4v (&acc)[4]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15;
    if (S::T <= 8) {
        const int tile = wave % S::T, rt = tile & 1, kq = wave / S::T;
        const float *ap = in + (16 * rt + r) * ld + 16 * kq * S::NKB + 4 * g;
        // two interleaved chains (a 16x16x4 MFMA's result is 8 passes away): even / odd k blocks, summed at the end
        acc[0] = f32x4v{0.f, 0.f, 0.f, 0.f};
        f32x4v odd = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < S::NKB; u += 2) {
            const float4 a = ld4(ap + 16 * u), b = ld4(ap + 16 * (u + 1));
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wr[u].x, acc[0], 0, 0, 0);
            odd = __builtin_amdgcn_mfma_f32_16x16x4f32(b.x, wr[u + 1].x, odd, 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wr[u].y, acc[0], 0, 0, 0);
            odd = __builtin_amdgcn_mfma_f32_16x16x4f32(b.y, wr[u + 1].y, odd, 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wr[u].z, acc[0], 0, 0, 0);
            odd = __builtin_amdgcn_mfma_f32_16x16x4f32(b.z, wr[u + 1].z, odd, 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wr[u].w, acc[0], 0, 0, 0);
            odd = __builtin_amdgcn_mfma_f32_16x16x4f32(b.w, wr[u + 1].w, odd, 0, 0, 0);
        }
        acc[0] += odd;
    } else {
#pragma unroll
        for (int t = 0; t < S::TPW; ++t) {
            const int rt = (wave * S::TPW + t) & 1;
            const float *ap = in + (16 * rt + r) * ld + 4 * g;
            acc[t] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < S::KB; ++u) {
                const float4 a = ld4(ap + 16 * u);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wr[t * S::KB + u].x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wr[t * S::KB + u].y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wr[t * S::KB + u].z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wr[t * S::KB + u].w, acc[t], 0, 0, 0);
            }
        }
    }
}

// split products (T <= 8): this wave's partial tile -> red[kq][row][col]
template <class S>
__device__ __forceinline__ void mc_partials(const f32x4v &acc, float *red) {
    static_assert(S::T <= 8, "split products only");
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int tile = wave % S::T, ct = tile >> 1, rt = tile & 1, kq = wave / S::T;
    float *dst = red + (kq * MC_R + 16 * rt + 4 * g) * S::P + 16 * ct + c;
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[j * S::P] = acc[j];
}
// the k-parts of one output, summed in a fixed order
template <class S>
__device__ __forceinline__ float mc_sum(const float *red, int row, int col) {
    float v = red[row * S::P + col];
#pragma unroll
    for (int q = 1; q < S::KS; ++q) v += red[(q * MC_R + row) * S::P + col];
    return v;
}

// place b of the grid -> (cluster, member): b = h + heads * t
__device__ __forceinline__ void mc_place(const McArgs &p, int b, int &cl, int &m) {
    const int heads = p.heads, h = b % heads, t = b / heads;
    cl = h + heads * (t / MC_S);
    m = t % MC_S;
}

// This workgroup's place in the grid, handed out in order of arrival (see the top of the file); -1: none could be had (the heads
// are corrupt: status word set).  The grid's places are dealt from p.heads heads (4, 2 or 1: the largest that divides the number
// of clusters); a workgroup draws from head blockIdx % heads, so every head serves exactly grid / heads workgroups = whole
// clusters, and ticket t of head h is place h + heads * t: cluster h + heads * (t / 16), member t % 16.  More heads = fewer
// workgroups per atomic word (256 arrivals on ONE word take 2.8 us to serve, MI355X_MICROARCH.md "dequeue"; measured on the
// kernels: one head +1.3 us per pass over places by blockIdx, four heads +0.5); fewer heads = progress with fewer resident
// workgroups: a pass moves whenever one head has 16 resident drawers, i.e. with 15 * heads + 1 resident workgroups at the latest.
// EIGHT heads (one per XCD) is what round 5 measured too few for: two processes at B = 512 on one device were each held to 15
// workgroups per XCD for > 300 ms (the 16th member of every cluster not dispatched) in 1 of ~8 runs; with four heads a cluster
// draws on two XCDs.  The heads are zero at the start of a pass: the pass clears them itself once every cluster is placed
// (mc_placed / mc_clear_heads), the step's prep launch clears everything.
constexpr unsigned MC_E_FWD = ARVAE_STATUS_HANDOFF_FWD, MC_E_BWD = ARVAE_STATUS_HANDOFF_BWD, MC_E_TICKET = ARVAE_STATUS_HANDOFF_TICKET;     // arvae_hip.h: float-safe bit patterns
__device__ __forceinline__ int mc_ticket(const McArgs &p) {
    if (p.debug_static) return (int)blockIdx.x;          // diagnostic build, ARVAE_MIDC_STATIC: places by blockIdx (what tickets cost)
    __shared__ int place;
    if (threadIdx.x == 0) {
        const unsigned heads = (unsigned)p.heads, h = blockIdx.x % heads;
        const unsigned t = __hip_atomic_fetch_add(p.counters + MC_TICKET_BASE + 32 * h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int b = t < gridDim.x / heads ? (int)(h + heads * t) : -1;
        if (b < 0 && p.status != nullptr) __hip_atomic_fetch_or(p.status, MC_E_TICKET, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (b < 0) MC_FAIL_RECORD(MC_E_TICKET, 0u, t, h);
        place = b;
    }
    __syncthreads();
    return place;
}
// The heads are cleared for the next pass by the pass itself, without a wait of its own: behind a cluster's LAST hand-off (every
// member has arrived, so every member holds its place) member 0 counts its cluster as placed (mc_placed: a returning atomic whose
// result nobody needs for the ~3 us the last product takes), and at the end of the kernel the member that counted last clears the
// heads (mc_clear_heads).  A pass in which a cluster gave up leaves them as they are: the next pass of that step then finds no
// place (status word), the next step's prep launch clears everything.
__device__ __forceinline__ unsigned mc_placed(const McArgs &p, int m) {
    unsigned before = 0;
    if (m == 0 && threadIdx.x == 0)
        before = __hip_atomic_fetch_add(p.counters + MC_TICKET_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return before;
}
__device__ __forceinline__ void mc_clear_heads(const McArgs &p, int m, unsigned before) {
    if (m == 0 && threadIdx.x == 0 && before == (unsigned)p.clusters - 1u) {
#pragma unroll
        for (int i = 0; i < 10; ++i) __hip_atomic_store(p.counters + MC_TICKET_BASE + 32 * i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Hand-off, producer side then consumer side (every member is both).  mc_publish: all of this workgroup's sc1 stores have left
// (each storing wave drains its own queue), then ONE lane arrives on the cluster's counter.  mc_wait: that lane polls the counter
// until the phase is complete; the barrier releases the other waves, whose loads of the exchanged block are sc1 loads
// (mc_gather).  Whatever is requested BETWEEN the two (the next products' weights) is in flight during the poll instead of being
// waited for by the drain.  The counter only ever holds multiples of MC_S between phases (the prep launch zeroes it every step), so
// the phase's target follows from the value the arrival returned: any number of forward / backward passes may follow one prep.
__device__ __forceinline__ unsigned mc_publish(unsigned *ctr, bool arrive = true) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned target = 0;
    // (arrive == false: the diagnostic build's way to make a hand-off fail -- McArgs.debug_drop -- so that the bound is tested)
    if (threadIdx.x == 0) target = (__hip_atomic_fetch_add(ctr, arrive ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / MC_S + 1u) * MC_S;
    return target;
}
// The poll is BOUNDED: by the time this wave has spent polling (wall clock, 100 MHz; a gap of more than MC_GAP_TICKS between
// two polls is time the wave was not running -- a context switch under a shared device -- and does not count).  On expiry the
// status word takes `code` and every thread of the workgroup gets false: the caller returns.  The other members of the cluster
// run into the same bound within microseconds of this one (they wait for the same arrivals).
constexpr unsigned long long MC_WAIT_TICKS = 100ull * 1000 * 1000;     // 1 s of polling
constexpr unsigned long long MC_GAP_TICKS = 5000;                      // 50 us
__device__ __forceinline__ bool mc_wait(unsigned *ctr, unsigned target, unsigned *status, unsigned code, unsigned long long limit = MC_WAIT_TICKS) {
    __shared__ int arrived;
    if (threadIdx.x == 0) {
        bool ok = true;
        unsigned long long last = wall_clock64(), spent = 0;
        while ((int)(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            const unsigned long long now = wall_clock64(), dt = now - last;
            last = now;
            spent += dt < MC_GAP_TICKS ? dt : 0ull;
            if (spent > limit) { ok = false; break; }
        }
        if (!ok && status != nullptr) __hip_atomic_fetch_or(status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (!ok) MC_FAIL_RECORD(code, target, __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), (unsigned)(ctr - (unsigned *)nullptr));
        arrived = ok ? 1 : 0;
    }
    __syncthreads();
    return arrived != 0;
}

// the cluster's 32 x 256 block of a tensor every member has just stored a slice of -> LDS (pitch PB); rows past the batch: zeros
__device__ __forceinline__ void mc_gather(const float *src, int row0, int valid, float *dst) {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(src + (int64_t)row0 * MC_H, (int64_t)valid * MC_H * 4);
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = (int)threadIdx.x + MC_T * u;           // float4 index: row e / 64, quad e % 64 (rows past `valid`: out of
        v[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, e * 16, 0, AUX_SC1));   // range -> zeros)
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = (int)threadIdx.x + MC_T * u;
        *reinterpret_cast<float4 *>(dst + (e >> 6) * PB + 4 * (e & 63)) = v[u];
    }
    __syncthreads();
}

__device__ __forceinline__ void st_sc1(float v, float *base, int64_t index) {
    __hip_atomic_store(base + index, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // global_store_dword sc1
}

// Epilogue of a whole-width product (every member holds all 256 columns of its cluster's rows): value -> LDS block `dst` (pitch
// PB), this member's own 16 columns also to `own` in memory.  FWD: v = act(acc + bias); else v = acc * act'(y).  The activation
// is a template parameter: two waves share a SIMD and 16 values per lane go through here, so every instruction per value
// costs ~0.05 us of the pass.
template <class S, int ACT, bool FWD>
__device__ __forceinline__ void mc_wide_epilogue(const f32x4v (&acc)[4], const float (&bias)[4], const float (&y)[4][4], float *dst, float *own,
                                                 int m, int row0, int valid) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int t = 0; t < S::TPW; ++t) {
        const int tile = wave * S::TPW + t, ct = tile >> 1, rt = tile & 1, col = 16 * ct + c;
        float *d = dst + (16 * rt + 4 * g) * PB + col;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = FWD ? acc[t][j] + bias[t] : acc[t][j];
            if (FWD) v[j] = ACT == ARVAE_ACT_RELU ? fmaxf(x, 0.f) : (ACT == ARVAE_ACT_SELU ? act_fwd(x, ARVAE_ACT_SELU) : x);
            else v[j] = x * (ACT == ARVAE_ACT_NONE ? 1.f : act_bwd_from_out(y[t][j], ACT));
            v[j] = 16 * rt + 4 * g + j < valid ? v[j] : 0.f;
            d[j * PB] = v[j];
        }
        if (ct == m) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 16 * rt + 4 * g + j;
                if (r < valid) own[(int64_t)(row0 + r) * MC_H + col] = v[j];
            }
        }
    }
}
template <class S, bool FWD>
__device__ __forceinline__ void mc_wide_epilogue_act(int act, const f32x4v (&acc)[4], const float (&bias)[4], const float (&y)[4][4], float *dst,
                                                     float *own, int m, int row0, int valid) {
    if (act == ARVAE_ACT_RELU) mc_wide_epilogue<S, ARVAE_ACT_RELU, FWD>(acc, bias, y, dst, own, m, row0, valid);
    else if (act == ARVAE_ACT_SELU) mc_wide_epilogue<S, ARVAE_ACT_SELU, FWD>(acc, bias, y, dst, own, m, row0, valid);
    else mc_wide_epilogue<S, ARVAE_ACT_NONE, FWD>(acc, bias, y, dst, own, m, row0, valid);
}


// ================================================================================================ folded conv layers (round 5)
// Member m of a cluster owns lo pixel (i, j) = (m / 4, m % 4) of the 4 x 4 map for its cluster's 32 images.
// DOWN window of that pixel: the 4 x 4 hi pixels (2 i - 1 + ky, 2 j - 1 + kx) x 32 channels of each image, zero outside the 8 x 8
// map: 32 x 512 floats -> LDS rows of pitch PA.  `hi`: [batch][8][8][32].
__device__ __forceinline__ void mc_down_window_issue(const float *__restrict__ hi, int row0, int valid, int m, float4 (&xv)[8]) {
    const int i = m >> 2, j = m & 3;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = (int)threadIdx.x + MC_T * u, r = e >> 7, q = e & 127, tap = q >> 3, c4 = q & 7;
        const int y = 2 * i - 1 + (tap >> 2), x = 2 * j - 1 + (tap & 3);
        const bool ok = r < valid && (unsigned)y < 8u && (unsigned)x < 8u;
        xv[u] = ok ? ld4(hi + ((int64_t)(row0 + r) * 64 + y * 8 + x) * 32 + 4 * c4) : zero4();
    }
}
__device__ __forceinline__ void mc_down_window_commit(const float4 (&xv)[8], float *dst) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = (int)threadIdx.x + MC_T * u;
        *reinterpret_cast<float4 *>(dst + (e >> 7) * PA + 4 * (e & 127)) = xv[u];
    }
}
// the cluster's 32 x 512 block of a tensor every member has just stored 32 columns of -> LDS (pitch PA)
__device__ __forceinline__ void mc_gather_wide(const float *src, int row0, int valid, float *dst) {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(src + (int64_t)row0 * MC_K0, (int64_t)valid * MC_K0 * 4);
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = (int)threadIdx.x + MC_T * u;           // float4 index: row e / 128, quad e % 128
        v[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, e * 16, 0, AUX_SC1));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = (int)threadIdx.x + MC_T * u;
        *reinterpret_cast<float4 *>(dst + (e >> 7) * PA + 4 * (e & 127)) = v[u];
    }
    __syncthreads();
}
// UP window: the 3 x 3 lo pixels (i - 1 + wy, j - 1 + wx) x 32 channels of the cluster's rows from a 512-wide tensor whose
// slices the members have just stored (sc1 loads) -> LDS rows of pitch PU; zero outside the 4 x 4 map / past the batch
__device__ __forceinline__ void mc_up_window(const float *src, int row0, int valid, int m, float *dst) {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(src + (int64_t)row0 * MC_K0, (int64_t)valid * MC_K0 * 4);
    const int i = m >> 2, j = m & 3;
    float4 v[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const int e = (int)threadIdx.x + MC_T * u, r = e / 72, q = e - r * 72, w = q >> 3, c4 = q & 7;
        const int y = i - 1 + w / 3, x = j - 1 + w % 3;
        const bool ok = e < MC_R * 72 && (unsigned)y < 4u && (unsigned)x < 4u;
        v[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (r * MC_K0 + (y * 4 + x) * 32 + 4 * c4) * 4 : 0x7fffffff, 0, AUX_SC1));
    }
#pragma unroll
    for (int u = 0; u < 5; ++u) {
        const int e = (int)threadIdx.x + MC_T * u, r = e / 72, q = e - r * 72;
        if (e < MC_R * 72) *reinterpret_cast<float4 *>(dst + r * PU + 4 * q) = v[u];
    }
    __syncthreads();
}
// UP product: wave w = (class (a, b) = w >> 1, column tile ct = w & 1), both 16-row tiles; its 8 k blocks of the class's
// 128 x 32 matrix: [class][ct][b] = blocks 8 w .. 8 w + 7 of McConv.up
__device__ __forceinline__ void mc_up_load(const float *__restrict__ up, float4 (&wr)[8]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int u = 0; u < 8; ++u) wr[u] = ld4(up + ((int64_t)((wave * 8 + u) * 64 + lane)) * 4);
}
__device__ __forceinline__ void mc_up_mma(const float *win, const float4 (&wr)[8], f32x4v (&acc)[2]) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, r = lane & 15;
    const int a = wave >> 2, b = (wave >> 1) & 1;
    acc[0] = f32x4v{0.f, 0.f, 0.f, 0.f};
    acc[1] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int t = u >> 1, ty = t >> 1, tx = t & 1;
        const int wy = ty == 0 ? 1 : (a ? 2 : 0), wx = tx == 0 ? 1 : (b ? 2 : 0);
        const float *ap = win + r * PU + (wy * 3 + wx) * 32 + 16 * (u & 1) + 4 * g;
        const float4 a0 = ld4(ap), a1 = ld4(ap + 16 * PU);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, wr[u].x, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, wr[u].x, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, wr[u].y, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, wr[u].y, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, wr[u].z, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, wr[u].z, acc[1], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, wr[u].w, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, wr[u].w, acc[1], 0, 0, 0);
    }
}
// Weight gradient of a folded layer, partitioned by TAPS over the cluster: member m = tap (ky, kx) = (m / 4, m % 4) computes
//     part[clo][chi] = sum over the cluster's rows n and the 16 lo pixels (i, j) of lo[n][(i, j)][clo] * hi[n][2 i - 1 + ky][2 j - 1 + kx][chi]
// -- a 32 x 32 block with K = 512 -- and leaves 4 KB (a workgroup that took its own lo pixel's share of every tap instead, as the
// first build of this fold did, left 64 KB: 2 us of stores per layer and sixteen times the slab traffic).  Neither operand is
// reused beyond two MFMAs, so both come straight from L2 (one dword per lane and MFMA pair); wave w takes lo pixels 2 w, 2 w + 1,
// the eight partial blocks meet in `red` (8192 floats).  lo512: [batch][512] (SC1: freshly handed-off slices), hi: [batch][8][8][32].
template <bool SC1>
__device__ __forceinline__ void mc_wgrad_tap(const float *lo512, const float *__restrict__ hi, int row0, int valid, int m, float *red,
                                             float *__restrict__ out) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, g = lane >> 4, c = lane & 15;
    const int ky = m >> 2, kx = m & 3;
    const __amdgpu_buffer_rsrc_t rs_lo = make_rsrc(lo512 + (int64_t)row0 * MC_K0, (int64_t)valid * MC_K0 * 4);
    const __amdgpu_buffer_rsrc_t rs_hi = make_rsrc(hi + (int64_t)row0 * 2048, (int64_t)valid * 2048 * 4);
    f32x4v acc[2][2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
        const int px = 2 * wave + pi, y = 2 * (px >> 2) - 1 + ky, x = 2 * (px & 3) - 1 + kx;
        if ((unsigned)y >= 8u || (unsigned)x >= 8u) continue;    // (wave-uniform: the tap falls outside the 8 x 8 map)
        float av[8][2], bv[8][2];
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            const int n = 4 * st + g;                            // rows past `valid`: out of range -> zeros
            const int ao = (n * MC_K0 + px * 32 + c) * 4, bo = (n * 2048 + (y * 8 + x) * 32 + c) * 4;
            av[st][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_lo, ao, 0, SC1 ? AUX_SC1 : 0));
            av[st][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_lo, ao + 64, 0, SC1 ? AUX_SC1 : 0));
            bv[st][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_hi, bo, 0, 0));
            bv[st][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_hi, bo + 64, 0, 0));
        }
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st][0], bv[st][0], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st][0], bv[st][1], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st][1], bv[st][0], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[st][1], bv[st][1], acc[1][1], 0, 0, 0);
        }
    }
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
            *reinterpret_cast<float4 *>(red + ((wave * 4 + mt * 2 + nt) * 64 + lane) * 4) =
                make_float4(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]);
    __syncthreads();
    // D: lane (g, c) of tile (mt, nt) holds rows (clo) 16 mt + 4 g + j, column (chi) 16 nt + c; thread t finishes elements 2 t, 2 t + 1
    // of [tile][lane][j], the eight waves' partials in wave order
    {
        const int e = 2 * (int)threadIdx.x, tile = e >> 8, ln = (e >> 2) & 63, j0 = e & 3;
        float2 t = make_float2(0.f, 0.f);
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const float2 v = *reinterpret_cast<const float2 *>(red + (w * 4 + tile) * 256 + (e & 255));
            t.x += v.x; t.y += v.y;
        }
        const int clo = 16 * (tile >> 1) + 4 * (ln >> 4) + j0, chi = 16 * (tile & 1) + (ln & 15);
        out[clo * 32 + chi] = t.x;
        out[(clo + 1) * 32 + chi] = t.y;
    }
}

// ================================================================================================ forward
template <bool FOLD>
__device__ __forceinline__ void midc_forward_body(const McArgs &p, int place) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *bufA = lds, *bufB = bufA + MC_R * PA, *red = bufB + MC_R * PB, *zbuf = red + RED_FLOATS, *outs = zbuf + MC_R * PZ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & 15;
    int cl, m;
    mc_place(p, place, cl, m);
    const int row0 = cl * MC_R, valid = min(MC_R, p.batch - row0), zd = p.zdim;
    unsigned *ctr = p.counters + cl * 32;
    const int frow = tid >> 4, fc = tid & 15;                 // the output a thread finalises in a 16-column slice
    MC_STAMP(0, 0);
    float4 wa[8], wb[8];
    f32x4v acc[4];
    // Requests in the order their data is needed (memory returns loads in order): the conv features of the cluster's rows
    // (64 KB, eight 16-byte loads per thread) -- FOLD: the window of the conv layer that makes them, and its matrix --, enc0's
    // weights, then everything the later phases would otherwise wait for
    float4 xv[MC_R * (MC_K0 / 4) / MC_T];
    float4 wc[16];
    float2 b_cv = make_float2(0.f, 0.f);
    if constexpr (FOLD) {
        mc_down_window_issue(p.hi_e, row0, valid, m, xv);
        mc_load<ShCD>(p.cv_e.down, wc);
        if (p.cv_e.bias != nullptr) b_cv = *reinterpret_cast<const float2 *>(p.cv_e.bias + 2 * fc);
    } else {
#pragma unroll
        for (int u = 0; u < MC_R * (MC_K0 / 4) / MC_T; ++u) {
            const int i = tid + MC_T * u, r = i / (MC_K0 / 4), q = i % (MC_K0 / 4);
            xv[u] = r < valid ? ld4(p.x0 + (int64_t)(row0 + r) * MC_K0 + 4 * q) : zero4();
        }
    }
    mc_load<ShE0>(p.e0f.w + (int64_t)m * ShE0::SLICE, wa);
    mc_load<ShHH>(p.e1f.w + (int64_t)m * ShHH::SLICE, wb);
#pragma unroll
    for (int u = 0; u < MC_R * (MC_K0 / 4) / MC_T; ++u) {
        const int i = tid + MC_T * u, r = i / (MC_K0 / 4), q = i % (MC_K0 / 4);
        *reinterpret_cast<float4 *>(bufA + r * PA + 4 * q) = xv[u];
    }
    const float b_e0 = p.e0f.bias != nullptr ? p.e0f.bias[16 * m + fc] : 0.f;
    const float b_e1 = p.e1f.bias != nullptr ? p.e1f.bias[16 * m + fc] : 0.f;
    const float b_d1 = p.d1f.bias != nullptr ? p.d1f.bias[16 * m + fc] : 0.f;
    const float b_h0 = 2 * fc < 2 * zd ? p.hdf.bias[2 * fc] : 0.f, b_h1 = 2 * fc + 1 < 2 * zd ? p.hdf.bias[2 * fc + 1] : 0.f;
    float b_d0[ShD0::TPW];
#pragma unroll
    for (int t = 0; t < ShD0::TPW; ++t) b_d0[t] = p.d0f.bias != nullptr ? p.d0f.bias[16 * ((wave * ShD0::TPW + t) >> 1) + c] : 0.f;
    // this thread's element of the reparameterisation (threads below 32 x zdim): its noise is drawn / requested now
    const int zr = tid / zd, zj = tid - zr * zd;
    const bool z_mine = tid < MC_R * zd, z_on = z_mine && zr < valid;
    const int64_t zidx = (int64_t)(z_on ? row0 + zr : 0) * zd + zj;
    float z_eps = 0.f;
    if (z_mine) z_eps = p.eps_out != nullptr ? rng_normal(p.rng, (uint64_t)zidx) : p.eps[zidx];
    __syncthreads();
    MC_STAMP(0, 1);
    if constexpr (FOLD) {
        // ---- the conv layer in front of the block: this member's lo pixel of the cluster's rows = its 32 columns of x0 (ReLU is what
        //      dsprites_vae.py:19-20 puts there; the launcher folds no other activation), handed to the cluster like any slice
        mc_mma<ShCD>(bufA, PA, wc, acc);
        mc_partials<ShCD>(acc[0], red);
        __syncthreads();
        {
            const int o = 2 * fc;
            const float v0 = fmaxf(mc_sum<ShCD>(red, frow, o) + b_cv.x, 0.f), v1 = fmaxf(mc_sum<ShCD>(red, frow, o + 1) + b_cv.y, 0.f);
            if (frow < valid) {
                st_sc1(v0, p.x0_out, (int64_t)(row0 + frow) * MC_K0 + 32 * m + o);
                st_sc1(v1, p.x0_out, (int64_t)(row0 + frow) * MC_K0 + 32 * m + o + 1);
            }
        }
        const unsigned tc = mc_publish(ctr);
        if (!mc_wait(ctr, tc, p.status, MC_E_FWD, p.wait_ticks)) return;
        mc_gather_wide(p.x0_out, row0, valid, bufA);
    }
    // ---- enc0: 512 -> this member's 16 of 256
    mc_mma<ShE0>(bufA, PA, wa, acc);
    mc_partials<ShE0>(acc[0], red);
    __syncthreads();
    {
        const float v = act_fwd_sel(mc_sum<ShE0>(red, frow, fc) + b_e0, p.act_e0);
        if (frow < valid) st_sc1(v, p.y_e0, (int64_t)(row0 + frow) * MC_H + 16 * m + fc);
    }
    MC_STAMP(0, 2);
    const unsigned t0 = mc_publish(ctr, !(p.debug_drop != 0 && cl == 0 && m == 0));
    mc_load<ShHD>(p.hdf.w, wa);
    if (!mc_wait(ctr, t0, p.status, MC_E_FWD, p.wait_ticks)) return;
    MC_STAMP(0, 3);
    mc_gather(p.y_e0, row0, valid, bufB);
    MC_STAMP(0, 4);
    // ---- enc1: 256 -> 16 of 256
    mc_mma<ShHH>(bufB, PB, wb, acc);
    mc_partials<ShHH>(acc[0], red);
    __syncthreads();
    {
        const float v = act_fwd_sel(mc_sum<ShHH>(red, frow, fc) + b_e1, p.act_e1);
        if (frow < valid) st_sc1(v, p.y_e1, (int64_t)(row0 + frow) * MC_H + 16 * m + fc);
    }
    MC_STAMP(0, 5);
    const unsigned t1 = mc_publish(ctr);
    mc_load<ShD0>(p.d0f.w, wb);
    if (!mc_wait(ctr, t1, p.status, MC_E_FWD, p.wait_ticks)) return;
    MC_STAMP(0, 6);
    mc_gather(p.y_e1, row0, valid, bufA);                     // (bufA with pitch PB from here on)
    MC_STAMP(0, 7);
    // ---- heads: 256 -> (mu | log_std), every member for its cluster's rows
    mc_mma<ShHD>(bufA, PB, wa, acc);
    mc_partials<ShHD>(acc[0], red);
    if (tid < MC_R * 16) zbuf[frow * PZ + fc] = 0.f;          // z rows, padded to 16 columns
    __syncthreads();
    {
        const int o = 2 * fc;
        outs[frow * PO + o] = mc_sum<ShHD>(red, frow, o) + b_h0;
        outs[frow * PO + o + 1] = mc_sum<ShHD>(red, frow, o + 1) + b_h1;
    }
    mc_load<ShHH>(p.d1f.w + (int64_t)m * ShHH::SLICE, wa);
    __syncthreads();
    if (z_mine) {
        const int r = zr, j = zj;
        const bool on = z_on;
        const int64_t idx = zidx;
        const float mv = outs[r * PO + j], lv = outs[r * PO + zd + j], sv = expf(lv);
        const float e = z_eps;
        const float zv = fmaf(e, sv, mv);
        zbuf[r * PZ + j] = on ? zv : 0.f;
        if (on && (r & (MC_S - 1)) == m) {                    // every member holds the same values: member r % 16 stores row r
            p.mu[idx] = mv; p.log_std[idx] = lv; p.sigma[idx] = sv; p.z[idx] = zv;
            if (p.eps_out != nullptr) p.eps_out[idx] = e;
        }
    }
    __syncthreads();
    MC_STAMP(0, 8);
    // ---- dec0: z (16) -> 256, every member; its own 16 columns go to memory (saved activation)
    mc_mma<ShD0>(zbuf, PZ, wb, acc);
    MC_STAMP(0, 14);
    {
        const float no_y[4][4] = {};
        mc_wide_epilogue_act<ShD0, true>(p.act_d0, acc, b_d0, no_y, bufB, p.y_d0, m, row0, valid);
    }
    MC_STAMP(0, 15);
    mc_load<ShD2>(p.d2f.w + (int64_t)m * ShD2::SLICE, wb);
    __syncthreads();
    MC_STAMP(0, 9);
    // ---- dec1: 256 -> 16 of 256
    mc_mma<ShHH>(bufB, PB, wa, acc);
    mc_partials<ShHH>(acc[0], red);
    __syncthreads();
    {
        const float v = act_fwd_sel(mc_sum<ShHH>(red, frow, fc) + b_d1, p.act_d1);
        if (frow < valid) st_sc1(v, p.y_d1, (int64_t)(row0 + frow) * MC_H + 16 * m + fc);
    }
    const unsigned t_last = mc_publish(ctr);
    const float2 b_d2 = p.d2f.bias != nullptr ? *reinterpret_cast<const float2 *>(p.d2f.bias + 32 * m + 2 * fc) : make_float2(0.f, 0.f);
    if (!mc_wait(ctr, t_last, p.status, MC_E_FWD, p.wait_ticks)) return;
    const unsigned placed_early = FOLD ? 0u : mc_placed(p, m);     // (FOLD: the pass's LAST hand-off comes later)
    MC_STAMP(0, 10);
    mc_gather(p.y_d1, row0, valid, bufA);
    MC_STAMP(0, 11);
    // ---- dec2: 256 -> 32 of 512: the first deconv's input
    mc_mma<ShD2>(bufA, PB, wb, acc);
    mc_partials<ShD2>(acc[0], red);
    __syncthreads();
    MC_STAMP(0, 12);
    float am = 0.f;
    {
        const int o = 2 * fc;
        float2 v;
        v.x = act_fwd_sel(mc_sum<ShD2>(red, frow, o) + b_d2.x, p.act_d2);
        v.y = act_fwd_sel(mc_sum<ShD2>(red, frow, o + 1) + b_d2.y, p.act_d2);
        if (frow < valid) {
            if constexpr (FOLD) {                             // a hand-off follows: write through
                st_sc1(v.x, p.y_d2, (int64_t)(row0 + frow) * MC_K0 + 32 * m + o);
                st_sc1(v.y, p.y_d2, (int64_t)(row0 + frow) * MC_K0 + 32 * m + o + 1);
            } else {
                *reinterpret_cast<float2 *>(p.y_d2 + (int64_t)(row0 + frow) * MC_K0 + 32 * m + o) = v;
                am = fmaxf(fabsf(v.x), fabsf(v.y));
            }
        }
    }
    unsigned placed_before = 0;
    if constexpr (FOLD) {
        // ---- the transposed conv layer behind the block (4x4 -> 8x8, ReLU): the 2 x 2 hi pixels over this member's lo pixel for the
        //      cluster's rows, from the 3 x 3 lo pixels around it -- other members' columns of y_d2: the pass's last hand-off
        const unsigned tu = mc_publish(ctr);
        float4 wu[8];
        mc_up_load(p.cv_d.up, wu);
        const float b_up = p.cv_d.bias != nullptr ? p.cv_d.bias[16 * (wave & 1) + c] : 0.f;
        if (!mc_wait(ctr, tu, p.status, MC_E_FWD, p.wait_ticks)) return;
        placed_before = mc_placed(p, m);
        mc_up_window(p.y_d2, row0, valid, m, bufA);
        f32x4v au[2];
        mc_up_mma(bufA, wu, au);
        const int g = lane >> 4, a = wave >> 2, b = (wave >> 1) & 1, ct = wave & 1;
        const int pix = (2 * (m >> 2) + a) * 8 + 2 * (m & 3) + b;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 16 * rt + 4 * g + j;
                const float v = fmaxf(au[rt][j] + b_up, 0.f);
                const bool on = r < valid;
                // sign bits (relu_bits16, common.h): bit 4 (ch >> 3) + (ch & 3) of the pixel's half (ch >> 2) & 1; this wave holds
                // channels 16 ct + c of (row, pixel): one byte of each half, bits 8 ct .. 8 ct + 7
                const unsigned long long bal = __ballot(on && v > 0.f);
                const unsigned my = (unsigned)(bal >> (16 * g)) & 0xffffu;
                if (on) {
                    const int64_t at = ((int64_t)(row0 + r) * 64 + pix);
                    p.hi_d[at * 32 + 16 * ct + c] = v;
                    am = fmaxf(am, v);
                    if (c < 2) {
                        const unsigned sh = 4 * c, byte = ((my >> sh) & 0xfu) | (((my >> (8 + sh)) & 0xfu) << 4);
                        p.hi_d_bits[at * 4 + 2 * c + ct] = (unsigned char)byte;
                    }
                }
            }
    }
    if (unsigned *amax_to = FOLD ? p.hi_d_amax : p.amax_out) {   // one AMAX writer unit per workgroup
        float *slot = red + RED_FLOATS;                       // the z rows' LDS: read by nobody at this point
        am = wave_max(am);
        if (lane == 0) slot[wave] = am;
        __syncthreads();
        if (tid < 64) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < MC_T / 64; ++w) t = fmaxf(t, slot[w]);
            amax_publish(amax_to, blockIdx.x, gridDim.x, t);
        }
    }
    mc_clear_heads(p, m, FOLD ? placed_before : placed_early);
    MC_STAMP(0, 13);
}

// ================================================================================================ backward
__global__ __launch_bounds__(MC_T) void midc_forward_kernel(McArgs p) {
    const int place = mc_ticket(p);
    if (place < 0) return;
    if (p.fold) midc_forward_body<true>(p, place);
    else midc_forward_body<false>(p, place);
}

// bias sums of a folded layer (this member's share: its own lo pixel / its own 2 x 2 hi pixels; slab floats 1024 .. 1055):
// two-stage column sums over the 32 rows (partials in `red`, finished by
// the first 32 threads BEHIND the caller's next barrier): mc_wgrad_bias_part, barrier, mc_wgrad_bias_finish
//   UPSIDE (the transposed conv: bias per hi channel): the window's interior taps (ky, kx in {1, 2}) = this member's own 2 x 2 hi pixels
//   else (the conv: bias per lo channel): the lo-side operand's columns
template <bool UPSIDE>
__device__ __forceinline__ void mc_wgrad_bias_part(const float *sop, const float *win, float *red) {
    const int ch = threadIdx.x & 31, rg = threadIdx.x >> 5;
    float t = 0.f;
#pragma unroll
    for (int r = 2 * rg; r < 2 * rg + 2; ++r) {
        if (UPSIDE) t += (win[r * PA + 5 * 32 + ch] + win[r * PA + 6 * 32 + ch]) + (win[r * PA + 9 * 32 + ch] + win[r * PA + 10 * 32 + ch]);
        else t += sop[r * PO + ch];
    }
    red[rg * 32 + ch] = t;
}
__device__ __forceinline__ void mc_wgrad_bias_finish(const float *red, float *__restrict__ slab) {
    if (threadIdx.x < 32) {
        float t = 0.f;
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) t += red[rg * 32 + threadIdx.x];
        slab[32 * 32 + threadIdx.x] = t;
    }
}

template <bool FOLD>
__device__ __forceinline__ void midc_backward_body(const McArgs &p, int place) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *bufA = lds, *bufB = bufA + MC_R * PA, *red = bufB + MC_R * PB, *dml = red + RED_FLOATS + MC_R * PZ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    int cl, m;
    mc_place(p, place, cl, m);
    const int row0 = cl * MC_R, valid = min(MC_R, p.batch - row0), zd = p.zdim;
    unsigned *ctr = p.counters + cl * 32;
    const int frow = tid >> 4, fc = tid & 15;
    const bool fon = frow < valid;
    const int64_t fidx = (int64_t)(row0 + (fon ? frow : 0)) * MC_H + 16 * m + fc;     // this thread's element of a 256-wide tensor
    MC_STAMP(1, 0);
    float4 wa[8], wb[8];
    f32x4v acc[4];
    // requests in the order of need: the arriving gradient (and, where it still has to be taken through the activation, the saved
    // output), the first product's weights, then what the later phases would otherwise wait for
    constexpr int NG = MC_R * (MC_K0 / 4) / MC_T;
    float4 gv[NG], yv[NG];
    float4 wc[16];
    float2 ys = make_float2(0.f, 0.f);
    float *slab_d = FOLD ? p.slab_d + (int64_t)(cl * MC_S + m) * MC_TAP_SLAB : nullptr;       // (by place: a fixed summation order)
    if constexpr (FOLD) {
        // the gradient arrives at the transposed conv layer's (pre-activation) output: this member's lo pixel needs its 4 x 4 window
        mc_down_window_issue(p.g_hi_d, row0, valid, m, gv);
        mc_load<ShCD>(p.cv_d.down, wc);
        if (fon) ys = *reinterpret_cast<const float2 *>(p.y_d2 + (int64_t)(row0 + frow) * MC_K0 + 32 * m + 2 * fc);
    } else {
#pragma unroll
        for (int u = 0; u < NG; ++u) {
            const int i = tid + MC_T * u, r = i / (MC_K0 / 4), q = i % (MC_K0 / 4);
            const int64_t at = (int64_t)(row0 + (r < valid ? r : 0)) * MC_K0 + 4 * q;
            gv[u] = r < valid ? ld4(p.g_out + at) : zero4();
            yv[u] = (r < valid && !p.g_is_pre) ? ld4(p.y_d2 + at) : zero4();
        }
    }
    mc_load<ShE0>(p.d2b.w + (int64_t)m * ShE0::SLICE, wa);
    mc_load<ShHH>(p.d1b.w + (int64_t)m * ShHH::SLICE, wb);
    const float y_d1 = p.y_d1[fidx], y_d0 = p.y_d0[fidx], y_e0 = p.y_e0[fidx];
    // the latent arithmetic's operands (threads with fc < zdim: element (frow, fc) of the cluster's rows)
    const int64_t li = (int64_t)(row0 + (fon ? frow : 0)) * zd + (fc < zd ? fc : 0);
    const float l_gl = p.g_loss[0], l_kl = p.kl[0], l_cap = p.cap != nullptr ? p.cap[0] : 0.f;
    const float l_reg = p.dz_reg != nullptr ? p.dz_reg[li] : 0.f, l_ext = p.dz_extra != nullptr ? p.dz_extra[li] : 0.f;
    const float l_s = p.sigma[li], l_mu = p.mu[li], l_e = p.eps[li];
    if constexpr (FOLD) {
        mc_down_window_commit(gv, bufA);
        dml[frow * PO + 2 * fc] = ys.x;                       // the lo-side operand of the layer's weight gradient: its saved input,
        dml[frow * PO + 2 * fc + 1] = ys.y;                   // this member's 32 columns of y_d2 (zero rows past the batch)
        __syncthreads();
        MC_STAMP(1, 1);
        // ---- data gradient of the transposed conv layer (a DOWN map) -> this member's 32 columns of the gradient at y_d2, times
        //      act'(y_d2): a slice like any other, handed to the cluster
        mc_mma<ShCD>(bufA, PA, wc, acc);
        mc_partials<ShCD>(acc[0], red);
        __syncthreads();
        {
            const int o = 2 * fc;
            const float v0 = mc_sum<ShCD>(red, frow, o) * act_bwd_from_out_sel(ys.x, p.act_d2);
            const float v1 = mc_sum<ShCD>(red, frow, o + 1) * act_bwd_from_out_sel(ys.y, p.act_d2);
            if (fon) {
                st_sc1(v0, p.g_d2, (int64_t)(row0 + frow) * MC_K0 + 32 * m + o);
                st_sc1(v1, p.g_d2, (int64_t)(row0 + frow) * MC_K0 + 32 * m + o + 1);
            }
        }
        const unsigned tg = mc_publish(ctr);
        // its weight gradient's tap m over the cluster's rows while the others arrive (both operands are old: the saved input and
        // the arriving gradient)
        mc_wgrad_bias_part<true>(dml, bufA, red);
        mc_wgrad_tap<false>(p.y_d2, p.g_hi_d, row0, valid, m, bufB, slab_d);
        if (!mc_wait(ctr, tg, p.status, MC_E_BWD, p.wait_ticks)) return;
        mc_wgrad_bias_finish(red, slab_d);
        mc_gather_wide(p.g_d2, row0, valid, bufA);
    } else {
#pragma unroll
        for (int u = 0; u < NG; ++u) {          // -> bufA as a pre-activation gradient
            const int i = tid + MC_T * u, r = i / (MC_K0 / 4), q = i % (MC_K0 / 4);
            float4 g4 = gv[u];
            if (r < valid && !p.g_is_pre) {
                const float4 y = yv[u];
                g4 = make_float4(g4.x * act_bwd_from_out_sel(y.x, p.act_d2), g4.y * act_bwd_from_out_sel(y.y, p.act_d2),
                                 g4.z * act_bwd_from_out_sel(y.z, p.act_d2), g4.w * act_bwd_from_out_sel(y.w, p.act_d2));
                if ((q >> 3) == m) *reinterpret_cast<float4 *>(p.g_d2 + (int64_t)(row0 + r) * MC_K0 + 4 * q) = g4;   // this member's 32 columns
            }
            *reinterpret_cast<float4 *>(bufA + r * PA + 4 * q) = g4;
        }
        __syncthreads();
        MC_STAMP(1, 1);
    }
    // ---- through dec2: 512 -> 16 of 256, times act'(dec1's output)
    mc_mma<ShE0>(bufA, PA, wa, acc);
    mc_partials<ShE0>(acc[0], red);
    __syncthreads();
    {
        const float v = mc_sum<ShE0>(red, frow, fc) * act_bwd_from_out_sel(y_d1, p.act_d1);
        if (fon) st_sc1(v, p.g_d1, fidx);
    }
    MC_STAMP(1, 2);
    const unsigned t2 = mc_publish(ctr);
    mc_load<ShZB>(p.d0b.w, wa);
    if (!mc_wait(ctr, t2, p.status, MC_E_BWD, p.wait_ticks)) return;
    MC_STAMP(1, 3);
    mc_gather(p.g_d1, row0, valid, bufB);
    MC_STAMP(1, 4);
    // ---- through dec1: 256 -> 16 of 256, times act'(dec0's output)
    mc_mma<ShHH>(bufB, PB, wb, acc);
    mc_partials<ShHH>(acc[0], red);
    __syncthreads();
    {
        const float v = mc_sum<ShHH>(red, frow, fc) * act_bwd_from_out_sel(y_d0, p.act_d0);
        if (fon) st_sc1(v, p.g_d0, fidx);
    }
    MC_STAMP(1, 5);
    const unsigned t3 = mc_publish(ctr);
    mc_load<ShHB>(p.hdb.w, wb);
    if (!mc_wait(ctr, t3, p.status, MC_E_BWD, p.wait_ticks)) return;
    MC_STAMP(1, 6);
    mc_gather(p.g_d0, row0, valid, bufA);                     // (bufA with pitch PB from here on)
    MC_STAMP(1, 7);
    // ---- through dec0: 256 -> d z (16), every member; then d(mu, log_std): decoder path + regulariser + KL
    //      (the formulas of heads_latent_bwd_kernel, heads.hip)
    mc_mma<ShZB>(bufA, PB, wa, acc);
    mc_partials<ShZB>(acc[0], red);
    dml[frow * PO + fc] = 0.f;
    dml[frow * PO + 16 + fc] = 0.f;
    // saved output of enc1 at this wave's tiles of the NEXT product (16 loads in flight during the latent arithmetic)
    float y_e1[ShHB::TPW][4];
#pragma unroll
    for (int t = 0; t < ShHB::TPW; ++t) {
        const int tile = wave * ShHB::TPW + t, ct = tile >> 1, rt = tile & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 16 * rt + 4 * g + j;
            y_e1[t][j] = r < valid ? p.y_e1[(int64_t)(row0 + r) * MC_H + 16 * ct + c] : 0.f;
        }
    }
    mc_load<ShHH>(p.e1b.w + (int64_t)m * ShHH::SLICE, wa);
    __syncthreads();
    if (fc < zd) {
        const int j = fc;
        float gz = mc_sum<ShZB>(red, frow, j);
        const int64_t i = li;
        const float gl = l_gl;
        const float diff = l_kl - l_cap;
        const float kk = gl * p.beta * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) * p.inv_batch;
        if (p.dz_reg != nullptr) gz += gl * p.reg_scale * l_reg;
        if (p.dz_extra != nullptr) gz += l_ext;
        const float s = l_s, mu = l_mu, e = l_e;
        const float a = gz + kk * mu, b = (gz * e + kk * (s - 1.f / s)) * s;
        if (fon) {
            dml[frow * PO + j] = a;
            dml[frow * PO + zd + j] = b;
            if ((frow & (MC_S - 1)) == m) { p.d_mu[i] = a; p.d_ls[i] = b; }
        }
    }
    __syncthreads();
    MC_STAMP(1, 8);
    // ---- through the heads: (d_mu | d_ls) (32) -> 256, every member, times act'(enc1's output); own 16 columns to memory
    mc_mma<ShHB>(dml, PO, wb, acc);
    {
        const float no_bias[4] = {};
        mc_wide_epilogue_act<ShHB, false>(p.act_e1, acc, no_bias, y_e1, bufB, p.g_e1, m, row0, valid);
    }
    mc_load<ShD2>(p.e0b.w + (int64_t)m * ShD2::SLICE, wb);
    __syncthreads();
    MC_STAMP(1, 9);
    // ---- through enc1: 256 -> 16 of 256, times act'(enc0's output)
    mc_mma<ShHH>(bufB, PB, wa, acc);
    mc_partials<ShHH>(acc[0], red);
    __syncthreads();
    {
        const float v = mc_sum<ShHH>(red, frow, fc) * act_bwd_from_out_sel(y_e0, p.act_e0);
        if (fon) st_sc1(v, p.g_e0, fidx);
    }
    const unsigned t_last = mc_publish(ctr);
    const int64_t xat = (int64_t)(row0 + (fon ? frow : 0)) * MC_K0 + 32 * m + 2 * fc;
    const float2 gate = p.gate0 != nullptr ? *reinterpret_cast<const float2 *>(p.gate0 + xat) : make_float2(1.f, 1.f);
    if (!mc_wait(ctr, t_last, p.status, MC_E_BWD, p.wait_ticks)) return;
    unsigned placed_before = FOLD ? 0u : mc_placed(p, m);       // (FOLD: the pass's LAST hand-off comes later)
    MC_STAMP(1, 10);
    mc_gather(p.g_e0, row0, valid, bufA);
    MC_STAMP(1, 11);
    // ---- through enc0: 256 -> 32 of 512, gated by the last conv layer's ReLU: the gradient that layer's backward reads
    mc_mma<ShD2>(bufA, PB, wb, acc);
    mc_partials<ShD2>(acc[0], red);
    __syncthreads();
    MC_STAMP(1, 12);
    float am = 0.f;
    {
        const int o = 2 * fc;
        float2 v;
        v.x = mc_sum<ShD2>(red, frow, o);
        v.y = mc_sum<ShD2>(red, frow, o + 1);
        if (p.gate0 != nullptr) { v.x = gate.x > 0.f ? v.x : 0.f; v.y = gate.y > 0.f ? v.y : 0.f; }
        if constexpr (FOLD) {
            if (fon) {                                            // a hand-off follows: write through
                st_sc1(v.x, p.d_x0, xat);
                st_sc1(v.y, p.d_x0, xat + 1);
            }
            dml[frow * PO + o] = fon ? v.x : 0.f;                 // the lo-side operand of the conv layer's weight gradient
            dml[frow * PO + o + 1] = fon ? v.y : 0.f;
        } else if (fon) {
            *reinterpret_cast<float2 *>(p.d_x0 + xat) = v;
            am = fmaxf(fabsf(v.x), fabsf(v.y));
        }
    }
    if constexpr (FOLD) {
        // ---- the conv layer in front of the block, backwards: its weight-gradient partial (this member's slice of the gradient x
        //      the window of its saved input) while the slices are handed over, then its data gradient (an UP map: the 2 x 2 hi
        //      pixels over this member's lo pixel from the 3 x 3 lo pixels around it), gated by the saved input's sign
        float *slab_e = p.slab_e + (int64_t)(cl * MC_S + m) * MC_TAP_SLAB;
        const unsigned tx = mc_publish(ctr);                      // (its barrier: the operand of the bias sums is staged)
        float4 wu[8];
        mc_up_load(p.cv_e.up, wu);
        const int a = wave >> 2, b = (wave >> 1) & 1, ct = wave & 1;
        const int pix = (2 * (m >> 2) + a) * 8 + 2 * (m & 3) + b;
        float gt[2][4];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 16 * rt + 4 * g + j;
                gt[rt][j] = r < valid ? p.hi_e[((int64_t)(row0 + r) * 64 + pix) * 32 + 16 * ct + c] : 0.f;
            }
        mc_wgrad_bias_part<false>(dml, bufA, red);
        if (!mc_wait(ctr, tx, p.status, MC_E_BWD, p.wait_ticks)) return;
        placed_before = mc_placed(p, m);
        mc_wgrad_bias_finish(red, slab_e);
        mc_up_window(p.d_x0, row0, valid, m, bufA);
        f32x4v au[2];
        mc_up_mma(bufA, wu, au);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int r = 16 * rt + 4 * g + j;
                const float v = gt[rt][j] > 0.f ? au[rt][j] : 0.f;
                if (r < valid) {
                    p.d_hi_e[((int64_t)(row0 + r) * 64 + pix) * 32 + 16 * ct + c] = v;
                    am = fmaxf(am, fabsf(v));
                }
            }
        // its weight gradient's tap m: the gradient slices every member has handed over x the layer's saved input
        mc_wgrad_tap<true>(p.d_x0, p.hi_e, row0, valid, m, bufB, slab_e);
    }
    if (unsigned *amax_to = FOLD ? p.d_hi_e_amax : p.amax_out) {
        float *slot = red + RED_FLOATS;                       // the z rows' LDS: read by nobody at this point
        am = wave_max(am);
        if (lane == 0) slot[wave] = am;
        __syncthreads();
        if (tid < 64) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < MC_T / 64; ++w) t = fmaxf(t, slot[w]);
            amax_publish(amax_to, blockIdx.x, gridDim.x, t);
        }
    }
    mc_clear_heads(p, m, placed_before);
    MC_STAMP(1, 13);
}

__global__ __launch_bounds__(MC_T) void midc_backward_kernel(McArgs p) {
    const int place = mc_ticket(p);
    if (place < 0) return;
    if (p.fold) midc_backward_body<true>(p, place);
    else midc_backward_body<false>(p, place);
}

std::once_flag g_lds_once;
void allow_lds() {
    std::call_once(g_lds_once, [] {
        (void)hipFuncSetAttribute((const void *)midc_forward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FLOATS * 4);
        (void)hipFuncSetAttribute((const void *)midc_backward_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FLOATS * 4);
    });
}

}  // namespace

int64_t midc_counter_words(int) { return MC_COUNTER_WORDS; }
unsigned long long midc_wait_ticks() { return MC_WAIT_TICKS; }

// workgroups of the clustered kernels the device can hold at once: the occupancy the runtime reports for the heavier of the two
// (512 threads, LDS_FLOATS of dynamic LDS) x the CU count.  The launcher takes the clustered kernels when the whole grid fits --
// a question of speed since the tickets (a grid that does not fit still completes, cluster after cluster).
int midc_resident_capacity() {
    static const int cap = [] {
        allow_lds();
        int fwd = 0, bwd = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&fwd, (const void *)midc_forward_kernel, MC_T, LDS_FLOATS * sizeof(float)) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&bwd, (const void *)midc_backward_kernel, MC_T, LDS_FLOATS * sizeof(float)) != hipSuccess) {
            (void)hipGetLastError();
            return 0;
        }
        return (fwd < bwd ? fwd : bwd) * device_cu_count();
    }();
    return cap;
}

int midc_forward(const McArgs &a, hipStream_t s) {
    allow_lds();
    ARVAE_LAUNCH(midc_forward_kernel, dim3(a.clusters * MC_S), dim3(MC_T), LDS_FLOATS * sizeof(float), s, a);
    return check_launch(a.fold ? "midc_forward_kernel(+ conv4, deconv1)" : "midc_forward_kernel");
}

int midc_backward(const McArgs &a, hipStream_t s) {
    allow_lds();
    ARVAE_LAUNCH(midc_backward_kernel, dim3(a.clusters * MC_S), dim3(MC_T), LDS_FLOATS * sizeof(float), s, a);
    return check_launch(a.fold ? "midc_backward_kernel(+ conv4, deconv1)" : "midc_backward_kernel");
}

}  // namespace arvae

#ifdef ARVAE_DIAG
extern "C" int arvae_debug_midc_failures(unsigned *out /* [1 + 64 * 8] */) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(arvae::g_mc_fail_count), sizeof(unsigned)) != hipSuccess) return -1;
    return (int)hipMemcpyFromSymbol(out + 1, HIP_SYMBOL(arvae::g_mc_fail), sizeof(unsigned) * 64 * 8);
}
#endif

#ifdef MIDC_STAMPS
extern "C" int arvae_debug_midc_stamps(unsigned long long *out, int count) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(arvae::g_midc_stamps), sizeof(unsigned long long) * count);
}
#endif
