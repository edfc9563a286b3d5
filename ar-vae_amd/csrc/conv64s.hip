// Stride-1 4x4 convolutions between 64-channel layers (the Morpho-MNIST 64 <-> 64 layers, imagevae/mnist_vae.py:16-47),
// ROW-STAGED: the source rows an output row group meets are fetched from HBM, scaled, turned into two fp16 terms
// (conv32_common.h: the arithmetic of the 32-channel kernels, three partial products per multiply-add) and written to
// LDS exactly ONCE, and all 16 taps x 64 channels of the reduction are served from there.  conv_rows_x3_kernel (conv64.hip)
// re-gathers its 64-pixel A tile from L2 for each of its 32 reduction chunks (~4 GB per launch at B = 1024, 16x re-read) and
// runs load -> split -> LDS -> MFMA serially between two barriers per chunk (MFMA pipe ~30 % busy).
//
//   out[n][oy][ox][q] = ep( bias[q] + sum_{ky,kx,c} W(q, c, ky, kx) * src[n][oy + sgn ky + off][ox + sgn kx + off][c] )
//     sgn = +1: Conv2d forward / ConvTranspose2d data gradient      sgn = -1: ConvTranspose2d forward / Conv2d data gradient
//
// One 256-thread workgroup per CU, persistent over tiles; a tile = G output rows of one image (G * ow <= 32 MT pixels,
// MT = 3 or 4 MFMA column tiles) and needs G + 3 source rows: <= 176 pixels x (2 x 64 fp16 + pad) = 50 KB, two buffers.
//   * MFMA orientation as in conv32.hip: the WEIGHTS are the A operand (row = output channel), the pixels the B operand, so a
//     lane ends up with consecutive channels of one pixel (16-byte stores).  32x32x16 fp16, three partial products per
//     multiply-add, smallest first (fp32-accurate: conv32_common.h; six bf16 products through round 3: 357-378 us per
//     64 -> 64 launch at two thirds of the MFMA issue rate).
//   * wave = kernel row (K split four ways); a wave holds MT x 2 accumulator tiles (all 64 output channels of the group's
//     pixels for its four taps) -- each pixel operand read from LDS feeds 6 MFMAs, each weight operand MT x 3.  The four partial sums meet through the tile's own (now free) LDS buffer, each wave finishing a quarter
//     of every accumulator tile and storing it.
//   * weights: split once per launch by conv64s_weight_prep_kernel into the exact per-lane operand order,
//     [ky][kx][16-channel chunk][column tile][term][lane] x 16 bytes (262 KB, L2 resident) + the inverse weight scale; a wave
//     streams its 4 KB per reduction step straight into registers, two steps ahead.
//   * the source tensor's maxima (AMAX array, conv32_common.h) come with it or are taken by operand_amax_kernel first; the
//     result's are published by the epilogue.
//   * pipeline as down32x_kernel: registers hold tile t+1 (loaded during tile t-1); during tile t's MFMAs each loader slot
//     is split, written to the other buffer and refilled with tile t+2.  The activation derivative / keep-mask of a
//     gradient operand is applied at that point -- once per value instead of once per tap.
//   * padding: source columns outside [0, sw) are not stored; a lane whose tap falls there reads one shared zero pixel
//     (a select on the LDS address).  Source rows outside the image are staged as zeros.
#include <mutex>
#include "diag.h"
#include "common.h"
#include "conv32_common.h"

#ifdef C64S_STAMPS
// diagnostic build only (tools/stamp_c64s.py): phase timeline of the first 64 workgroups, 100 MHz wall clock
namespace arvae { __device__ unsigned long long g_c64s_stamps[64 * 64]; }
#define CSTAMP(slot) do { if (threadIdx.x == 0 && blockIdx.x < 64 && (slot) < 64) ::arvae::g_c64s_stamps[blockIdx.x * 64 + (slot)] = wall_clock64(); } while (0)
#else
#define CSTAMP(slot)
#endif

namespace arvae {

constexpr int S_PITCH = 72;                      // dwords per staged pixel: terms at +0, +32, 8 pad (conflict-free 16-byte reads for
                                                 // lanes walking consecutive pixels; a buffer also holds the 48 KB exchange)
constexpr int S_PIX = 176;                       // staged source pixels per tile
constexpr int S_BUF = (S_PIX + 1) * S_PITCH;     // + the zero pixel
constexpr int S_SLOTS = S_PIX * 16 / 256;        // 16-byte loader slots per thread (11)
constexpr int C64S_PREP_MAX = 2 * 2 * ARVAE_MAX_LAYERS;   // jobs of one batched weight prep: every layer of a model, both orientations
constexpr int S_WSTEP2 = 4 * 64;                 // uint4 per (ky, kx, channel chunk) with two row tiles: [row tile][term 2][lane 64]
constexpr int S_PREP_UINT4 = 16 * 4 * S_WSTEP2;  // 262 144 bytes (half of it for narrow outputs, one row tile); then one uint4 whose
                                                 // first dword is the inverse weight scale
static_assert((S_PIX + 1) * S_PITCH * 4 >= 4 * 3 * 4 * 64 * 16, "a staging buffer also serves as the exchange area of the four kernel rows");

struct ConvStage {
    Operand src;                 // [n][sh][sw][64]
    int n, sh, sw, oh, ow, q;    // q output channels: 64 (two 32-row MFMA tiles) or <= 32 (one, rows >= q are zero weights)
    int sgn, dmin;               // source coordinate = output coordinate + dmin + j, j = tap index if sgn > 0 else 3 - tap index
    int rows, groups;            // output rows per tile, tiles per image
    const uint4 *wprep;
    const float *bias;
    const uint8_t *mask;
    int act;
    float *out;                  // [n][oh][ow][q]
    GateOp gate;                 // data-gradient launches: result *= act'(gate.y) * 2 gate.mask at the output location
    const unsigned *amax_in;     // AMAX array of the source operand AS MULTIPLIED (derivative / keep-mask applied)
    unsigned *amax_out;          // AMAX array of `out`, or null
};

// wt = nn.Conv2d / nn.ConvTranspose2d weights [a][b][ky][kx]; (q, c) = (a, b) for the Conv2d-forward direction, (b, a) for
// the transposed one.  One thread = one (ky, kx, chunk, column tile, lane) = 8 reduction channels of one output channel.
// Every workgroup first takes the maximum magnitude of the whole tensor itself (q_count x 64 x 16 weights from L2).
__device__ __forceinline__ void conv64s_prep_block(const float *__restrict__ wt, uint4 *__restrict__ out, int transposed, int q_count,
                                                   int nt_count, const int block) {
    __shared__ float wmax[4];
    float m = 0.f;
    const int n4 = q_count * 64 * 4;                             // float4s of the tensor; eight loads in flight per thread
    for (int i4 = threadIdx.x; i4 < n4; i4 += 8 * 256) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4 *>(wt + 4 * min(i4 + u * 256, n4 - 1));
#pragma unroll
        for (int u = 0; u < 8; ++u) m = fmaxf(m, amax4(v[u]));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = m;
    __syncthreads();
    const Pow2 sc = pow2_for(__builtin_bit_cast(unsigned, fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
    const int i = block * 256 + threadIdx.x;                   // ((tap * 4 + c16) * nt_count + nt) * 64 + lane
    if (i == 0) out[16 * 4 * nt_count * 2 * 64] = make_uint4(__builtin_bit_cast(unsigned, sc.inv), 0u, 0u, 0u);
    const int lane = i & 63, rest = i >> 6, nt = rest % nt_count, c16 = (rest / nt_count) & 3, tap = rest / (4 * nt_count);
    // MFMA row i of kernel row ky's operand = output channel 8 ((i / 8 + wave) & 3) + i % 8 of the column tile, wave = the wave
    // that runs this kernel row (conv64s_kernel: ky for the Conv2d-forward orientation, 3 - ky for the transposed one): every wave
    // then finds ITS share of the outputs (channels 8 wave ..) in accumulator registers 0-3 and wave + o's in 4 o .. 4 o + 3 --
    // compile-time register numbers (indexed with the wave number they cost three instructions per register read)
    const int wave = transposed ? 3 - (tap >> 2) : (tap >> 2), row = lane & 31;
    const int q = nt * 32 + 8 * (((row >> 3) + wave) & 3) + (row & 7), c0 = c16 * 16 + 8 * (lane >> 5);
    const int qc = q < q_count ? q : 0;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = transposed ? wt[((c0 + j) * q_count + qc) * 16 + tap] : wt[(qc * 64 + c0 + j) * 16 + tap];
        x[j] = q < q_count ? v : 0.f;
    }
    uint4 h, l;
    split_pair_h2(x[0], x[1], sc.s, h.x, l.x);
    split_pair_h2(x[2], x[3], sc.s, h.y, l.y);
    split_pair_h2(x[4], x[5], sc.s, h.z, l.z);
    split_pair_h2(x[6], x[7], sc.s, h.w, l.w);
    uint4 *d = out + ((tap * 4 + c16) * nt_count + nt) * 2 * 64 + lane;
    d[0] = h; d[64] = l;
}
__global__ __launch_bounds__(256) void conv64s_weight_prep_kernel(const float *__restrict__ wt, uint4 *__restrict__ out, int transposed,
                                                                   int q_count, int nt_count) {
    conv64s_prep_block(wt, out, transposed, q_count, nt_count, blockIdx.x);
}
// every wide layer of a model in both orientations as ONE launch at the start of a training step (plan.hip): blockIdx.y = job.
// (Six per-launch preps of 5-9 us each sat in front of the Morpho-MNIST step's convolutions with the chip idle around them.)
struct C64sPrepJobs {
    int count;
    const float *wt[C64S_PREP_MAX];
    uint4 *out[C64S_PREP_MAX];
    int transposed[C64S_PREP_MAX], q[C64S_PREP_MAX];
};
__global__ __launch_bounds__(256) void conv64s_weight_prep_batch_kernel(C64sPrepJobs j) {
    const int k = blockIdx.y, nt = j.q[k] > 32 ? 2 : 1;
    if ((int)blockIdx.x >= 16 * 4 * nt * 64 / 256) return;
    conv64s_prep_block(j.wt[k], j.out[k], j.transposed[k], j.q[k], nt, blockIdx.x);
}

// AMAX array of an operand as a kernel multiplies it: value x activation derivative x keep-mask (Operand::at4); for the caller
// whose source tensor comes without one
__global__ __launch_bounds__(256) void operand_amax_kernel(Operand x, int64_t count4, unsigned *__restrict__ out) {
    __shared__ float wm[4];
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count4; i += (int64_t)gridDim.x * 256) m = fmaxf(m, amax4(x.at4(4 * i)));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x < 64) amax_publish(out, blockIdx.x, gridDim.x, fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])));
}

// The epilogue runs with the matrix pipe idle and ONE wave per SIMD: it is bound by its own vector instruction count; activation and
// gate in common.h's coefficient form (ActCoef / GateCoef: no select on the activation kind, no branch on the sign).
// MODE = Operand::mode() of the source: 0 plain, 1 activation derivative from the saved output, 2 also the keep-mask
template <int MT, int NT, int MODE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv64s_kernel(ConvStage g) {
    CSTAMP(0);
    constexpr int S_WSTEP = NT * 2 * 64;
#ifndef C64S_EP_STEPS
#define C64S_EP_STEPS 3
#endif
    // the epilogue operands (keep-mask bytes / gate values: scattered requests from HBM) are requested in the LAST reduction steps,
    // behind the tile's last weight request (step 13): memory operations return in order, and a weight request queued behind
    // one of these waits out an HBM round trip instead of an L2 hit
#ifndef C64S_W_AHEAD
#define C64S_W_AHEAD (NT == 1 ? 3 : 2)
#endif
    constexpr int W_AHEAD = C64S_W_AHEAD;        // reduction steps between a weight operand's request and its MFMAs
    constexpr int EP_STEPS = C64S_EP_STEPS, EP_FIRST = 16 - EP_STEPS, EP_PER = (MT * NT + EP_STEPS - 1) / EP_STEPS;
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];          // 2 x S_BUF
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, rc = lane & 31;
    const int n_tiles = g.n * g.groups;
    // a workgroup owns a contiguous run of tiles: consecutive row groups of an image share three source rows, which then come
    // from this XCD's L2 the second time
    const int per_wg = (n_tiles + (int)gridDim.x - 1) / (int)gridDim.x, t_first = blockIdx.x * per_wg;
    const int t_end = min(n_tiles, t_first + per_wg);
    const int src_rows = g.rows + 3;
    const AmaxLoad al = amax_issue(g.amax_in);
    float sc_in = 1.f;                                           // the source's scale (set behind the first tile's loads)

    // ---- loader: slot s of this thread = (staged pixel, channels 4 q4 .. + 3) -------------------------------------------
    float4 lv[S_SLOTS], ly[S_SLOTS];
    unsigned lm[S_SLOTS];
    // slot s: staged pixel pix0 + 16 s.  A tile's source rows are ONE contiguous run of pixels in memory, so the slot's address is
    // the tile's first source row (a scalar offset, + 4096 s) + this thread's (pix0, q4) (one per-lane offset for all slots), and a
    // staged pixel is inside the image when its number lies in [lo_pix, lo_pix + span): raw buffer loads, the per-lane offset
    // sent beyond the range for the others (they read as zero).  No division, no 64-bit address per slot, no load under a lane
    // test -- the generic-pointer form cost ~20 vector instructions per slot (two quarter-rate multiplies among them) and an
    // exec-mask branch, eleven times per tile.  The resource starts pad_b bytes BEFORE the tensor: a first tile's scalar offset
    // (source rows above the image) stays non-negative, and nothing is read there.
    int q4 = threadIdx.x & 15, pix0 = threadIdx.x >> 4, wlane = lane;
    const int pad_b = (g.dmin < 0 ? -g.dmin : 0) * g.sw * 256;
    const int64_t src_bytes = (int64_t)g.n * g.sh * g.sw * 256 + pad_b;
    const __amdgpu_buffer_rsrc_t rs_v = make_rsrc(reinterpret_cast<const char *>(g.src.v) - pad_b, src_bytes);
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(MODE >= 1 ? reinterpret_cast<const char *>(g.src.y) - pad_b : reinterpret_cast<const char *>(g.src.v), MODE >= 1 ? src_bytes : 0);
    const __amdgpu_buffer_rsrc_t rs_m = make_rsrc(MODE == 2 ? reinterpret_cast<const char *>(g.src.mask) - pad_b / 4 : reinterpret_cast<const char *>(g.src.v), MODE == 2 ? src_bytes / 4 : 0);
    constexpr unsigned OOBV = 0xfffffff0u;
    struct TileSrc { int soff, lo, span; };                      // scalar byte offset of the tile's first source row; valid staged pixels
    auto tile_src = [&](int tile) __attribute__((always_inline)) {
        const int img = tile / g.groups, oy0 = (tile - img * g.groups) * g.rows, sy0 = oy0 + g.dmin;
        const int lo = max(0, -sy0) * g.sw, hi = tile < t_end ? min(src_rows, g.sh - sy0) * g.sw : 0;
        return TileSrc{((img * g.sh + sy0) * g.sw) * 256 + pad_b, lo, hi > lo ? hi - lo : 0};
    };
    auto slot_offset = [&](int s, const TileSrc &ts) __attribute__((always_inline)) {
        return (unsigned)(pix0 + 16 * s - ts.lo) < (unsigned)ts.span ? (unsigned)(pix0 * 256 + q4 * 16) : OOBV;
    };
    auto slot_load = [&](int s, unsigned off, const TileSrc &ts) __attribute__((always_inline)) {
        const f32x4v v = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs_v, (int)off, ts.soff + 4096 * s, 0));
        lv[s] = make_float4(v.x, v.y, v.z, v.w);
        if (MODE >= 1) {
            const f32x4v y = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(rs_y, (int)off, ts.soff + 4096 * s, 0));
            ly[s] = make_float4(y.x, y.y, y.z, y.w);
        }
        if (MODE == 2) lm[s] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs_m, (int)(off == OOBV ? OOBV : off >> 2), (ts.soff + 4096 * s) >> 2, 0);
    };
    auto issue = [&](int s, int tile) __attribute__((always_inline)) {
        const TileSrc ts = tile_src(tile);
        slot_load(s, slot_offset(s, ts), ts);
    };
    auto commit = [&](int s, unsigned *buf) __attribute__((always_inline)) {
        float4 v = lv[s];
        if (MODE >= 1) {
            float4 y = ly[s];
            if (MODE == 2) {
                const unsigned m = lm[s];
                v.x *= 2.f * (float)(m & 255u); v.y *= 2.f * (float)((m >> 8) & 255u);
                v.z *= 2.f * (float)((m >> 16) & 255u); v.w *= 2.f * (float)(m >> 24);
                y.x *= 0.5f; y.y *= 0.5f; y.z *= 0.5f; y.w *= 0.5f;
            }
            v.x *= act_bwd_from_out_sel(y.x, g.src.act); v.y *= act_bwd_from_out_sel(y.y, g.src.act);
            v.z *= act_bwd_from_out_sel(y.z, g.src.act); v.w *= act_bwd_from_out_sel(y.w, g.src.act);
        }
        uint2 h, l;
        split_pair_h2(v.x, v.y, sc_in, h.x, l.x);
        split_pair_h2(v.z, v.w, sc_in, h.y, l.y);
        unsigned *d = buf + (pix0 + 16 * s) * S_PITCH + q4 * 2;
        *reinterpret_cast<uint2 *>(d) = h;
        *reinterpret_cast<uint2 *>(d + 32) = l;
    };

#pragma unroll
    for (int s = 0; s < S_SLOTS; ++s) issue(s, t_first);
    // zero pixel of both buffers, once
    if (threadIdx.x < 2 * S_PITCH) lds[(threadIdx.x / S_PITCH) * S_BUF + S_PIX * S_PITCH + threadIdx.x % S_PITCH] = 0u;

    // ---- consumer geometry: wave = kernel row index jy (source row oy + dmin + jy); pixel of (mt, lane) ------------------
    // xoff[mt][jx]: dword offset of this lane's pixel at tap column jx in a buffer (the zero pixel where that column is
    // outside the source), + half * 4
    int xoff[MT][4];
    int opix[MT];                                // output offset of the lane's pixel inside the tile, or -1
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        const int P = mt * 32 + rc, r = P / g.ow, c = P - r * g.ow;
        opix[mt] = (P < g.rows * g.ow) ? P : -1;
#pragma unroll
        for (int jx = 0; jx < 4; ++jx) {
            const int sx = c + g.dmin + jx;
            const bool ok = r < g.rows && (unsigned)sx < (unsigned)g.sw;
            xoff[mt][jx] = (ok ? ((r + wave) * g.sw + sx) : S_PIX) * S_PITCH + half * 4;
        }
    }
    const int ky = g.sgn > 0 ? wave : 3 - wave;
    // this wave's share of the outputs after the exchange: channels nt * 32 + 8 wave + 4 half + j
    float4 b4[NT];
    bool ch_ok[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        ch_ok[nt] = nt * 32 + 8 * wave + 4 * half < g.q;
        b4[nt] = (g.bias != nullptr && ch_ok[nt]) ? *reinterpret_cast<const float4 *>(g.bias + nt * 32 + 8 * wave + 4 * half)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // first tile -> buffer 0
    {
        const Pow2 sc = amax_scale(al);
        sc_in = sc.s;
    }
    const float inv = amax_scale(al).inv * __builtin_bit_cast(float, g.wprep[16 * 4 * NT * 2 * 64].x);     // accumulators -> fp32 results
    float amax_run = 0.f;
    __syncthreads();
#pragma unroll
    for (int s = 0; s < S_SLOTS; ++s) commit(s, lds);
    __syncthreads();
    // weight operands: a ring of three register sets, two reduction steps ahead of the MFMAs (the split weights live in L2: one
    // step is not enough to cover that round trip under load)
    f16x8 w2[W_AHEAD + 1][NT][2];
    // (one resource, the lane's 16 bytes as the per-lane offset, the (tap, chunk) block as the scalar offset, the row tile / term as
    // the instruction's immediate: no vector instruction per weight request)
    const __amdgpu_buffer_rsrc_t rs_w = make_rsrc(g.wprep, (int64_t)(16 * 4 * S_WSTEP + 1) * 16);
    auto load_w = [&](auto rc_, int kx, int c16) __attribute__((always_inline)) {
        constexpr int r = decltype(rc_)::value;
        const int so = ((ky * 4 + kx) * 4 + c16) * S_WSTEP * 16;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                w2[r][nt][t] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wlane * 16 + (nt * 2 + t) * 1024, so, 0));
    };
    const int kx_first = g.sgn > 0 ? 0 : 3;
    // epilogue operands and the result as buffer resources (conv64s_fits bounds the tensor at 2 GB): absent operands get an empty
    // range (loads return zero) and lanes without an output an out-of-range offset, so that no load or store sits under a branch
    const int64_t out_elems = (int64_t)g.n * g.oh * g.ow * g.q;
    const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(g.out, out_elems * 4);
    const __amdgpu_buffer_rsrc_t rs_gy = make_rsrc(g.gate.y != nullptr ? (const void *)g.gate.y : (const void *)g.out, g.gate.y != nullptr ? out_elems * 4 : 0);
    const uint8_t *km_ptr = g.gate.y != nullptr ? g.gate.mask : g.mask;
    const __amdgpu_buffer_rsrc_t rs_km = make_rsrc(km_ptr != nullptr ? (const void *)km_ptr : (const void *)g.out, km_ptr != nullptr ? out_elems : 0);
    const bool km_ones = g.gate.y != nullptr && g.gate.mask == nullptr;
    const ActCoef ac = act_coef(g.act);
    const GateCoef gc = gate_coef(g.gate.act, g.gate.mask != nullptr);

    static_for<0, W_AHEAD>([&](auto r_) __attribute__((always_inline)) { load_w(r_, kx_first, decltype(r_)::value); });
    // as many (dropped) stores behind the first weight requests as an epilogue leaves behind the later ones: the loop header then
    // sees the same queue on its entry edge and on its back edge, and the first MFMAs of a tile wait for their weights only
    // (vmcnt(19) ...) instead of draining the epilogue's stores as well (vmcnt(0))
#pragma unroll
    for (int i = 0; i < MT * NT; ++i) buf_store4(make_float4(0.f, 0.f, 0.f, 0.f), rs_out, OOB);

    int cur = 0;
    int cst = 0;
    (void)cst;
    CSTAMP(1);
    for (int tile = t_first; tile < t_end; ++tile, cur ^= 1) {
        CSTAMP(2 + 6 * cst);
        asm volatile("" : "+v"(q4), "+v"(pix0), "+v"(wlane));
        const unsigned *xb = lds + cur * S_BUF;
        unsigned *nb = lds + (cur ^ 1) * S_BUF;
        f32x16 acc[MT][NT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][nt][i] = 0.f;
        // keep-mask bytes / gate values of this wave's share of the outputs: requested in reduction step 11 (a lane's 16 bytes
        // sit 256 bytes from its neighbour's: slow, scattered requests with a full memory round trip), used after the exchange
        const int img = tile / g.groups, oy0 = (tile - img * g.groups) * g.rows;
        unsigned km[MT][NT];
        float4 gy[MT][NT];
        // (buffer loads through possibly EMPTY resources and no branch: a memory operation under a run-time condition makes the
        // compiler's vmcnt counts conservative everywhere after it -- see the stores below)
        auto fetch_epilogue = [&](auto kc_) __attribute__((always_inline)) {
            constexpr int mt = decltype(kc_)::value / NT, nt = decltype(kc_)::value % NT;
                {
                    const int P = opix[mt];
#ifdef C64S_ABL_NOEPLOAD
                    const bool ok = false;
#else
                    const bool ok = P >= 0 && oy0 + P / g.ow < g.oh && ch_ok[nt];
#endif
                    const unsigned o = ok ? (unsigned)((((int64_t)img * g.oh + oy0) * g.ow + P) * g.q + nt * 32 + 8 * wave + 4 * half) : 0u;
                    gy[mt][nt] = buf_load4(rs_gy, o * 4u);
                    const unsigned m = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs_km, (int)o, 0, 0);
                    km[mt][nt] = km_ones ? 0x01010101u : m;
                }
        };
        f16x8 x2[2][MT][2];
        // pixel operands of reduction step 0 (jx = 0, chunk 0)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                x2[0][mt][t] = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4v *>(xb + xoff[mt][0] + t * 32));
        // Issue order pinned by hand (one scheduling barrier after every MFMA; see conv32k.hip and DESIGN.md section 4, item 15c):
        // the step's other work -- the next step's 3 MT pixel operand reads, the weight loads of the step after that, and the
        // loader's pieces (address, loads; derivative / mask, split in two halves per value pair, LDS writes) -- sits between the
        // MFMAs in order.  Left to the scheduler a step was a block of MFMAs followed by a block of vector instructions.
        uint2 c_h, c_l;                                          // a slot's values between its pieces
        float4 c_v;
        unsigned c_off = OOBV;
        const TileSrc next_src = tile_src(tile + 1);
        static_for<0, 16>([&](auto kc) __attribute__((always_inline)) {
            constexpr int step = decltype(kc)::value, cu = step & 1, nx = cu ^ 1, wr = step % (W_AHEAD + 1);
            // loader slots of this step: requested in step s * 12 / 11, split + written four steps (~3 us) later
            constexpr int is_lo = (step * S_SLOTS + 11) / 12, is_hi = step < 12 ? ((step + 1) * S_SLOTS + 11) / 12 : is_lo;
            constexpr int cs = step - 4;
            constexpr int cm_lo = cs >= 0 ? (cs * S_SLOTS + 11) / 12 : 0, cm_hi = (cs >= 0 && cs < 12) ? ((cs + 1) * S_SLOTS + 11) / 12 : cm_lo;
            constexpr int n_read = step + 1 < 16 ? 2 * MT : 0, n_w = step + W_AHEAD < 16 ? 2 * NT : 0;
            constexpr int n_issue = 2 * (is_hi - is_lo), n_commit = 4 * (cm_hi - cm_lo), ep_lo = step >= EP_FIRST ? (step - EP_FIRST) * EP_PER : 0;
            constexpr int ep_hi = step >= EP_FIRST ? (ep_lo + EP_PER < MT * NT ? ep_lo + EP_PER : MT * NT) : 0, n_ep = ep_hi > ep_lo ? ep_hi - ep_lo : 0;
            constexpr int n_items = n_read + n_w + n_issue + n_commit + n_ep, n_mfma = 3 * MT * NT;
            auto item = [&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                if constexpr (i < n_read) {
                    constexpr int njx = (step + 1) >> 2, nc16 = (step + 1) & 3, mt = i / 2, t = i % 2;
                    x2[nx][mt][t] = __builtin_bit_cast(f16x8, *reinterpret_cast<const i32x4v *>(xb + xoff[mt][njx] + t * 32 + nc16 * 8));
                } else if constexpr (i < n_read + n_w) {
                    constexpr int k = i - n_read, nt = k / 2, t = k % 2, njx = (step + W_AHEAD) >> 2, nc16 = (step + W_AHEAD) & 3;
                    const int nkx = g.sgn > 0 ? njx : 3 - njx;
                    w2[(step + W_AHEAD) % (W_AHEAD + 1)][nt][t] = __builtin_bit_cast(
                        f16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_w, wlane * 16 + (nt * 2 + t) * 1024, ((ky * 4 + nkx) * 4 + nc16) * S_WSTEP * 16, 0));
                } else if constexpr (i < n_read + n_w + n_issue) {
                    constexpr int k = i - n_read - n_w, s = is_lo + k / 2, piece = k % 2;
                    if constexpr (piece == 0) c_off = slot_offset(s, next_src);     // slot s of the next tile
                    else slot_load(s, c_off, next_src);
                } else if constexpr (i < n_read + n_w + n_issue + n_commit) {
                    constexpr int k = i - n_read - n_w - n_issue, s = cm_lo + k / 4, piece = k % 4;
                    if constexpr (piece == 0) {                  // the value with a gradient operand's derivative / keep-mask
                        c_v = lv[s];
                        if (MODE >= 1) {
                            float4 y = ly[s];
                            if (MODE == 2) {
                                const unsigned m = lm[s];
                                c_v.x *= 2.f * (float)(m & 255u); c_v.y *= 2.f * (float)((m >> 8) & 255u);
                                c_v.z *= 2.f * (float)((m >> 16) & 255u); c_v.w *= 2.f * (float)(m >> 24);
                                y.x *= 0.5f; y.y *= 0.5f; y.z *= 0.5f; y.w *= 0.5f;
                            }
                            c_v.x *= act_bwd_from_out_sel(y.x, g.src.act); c_v.y *= act_bwd_from_out_sel(y.y, g.src.act);
                            c_v.z *= act_bwd_from_out_sel(y.z, g.src.act); c_v.w *= act_bwd_from_out_sel(y.w, g.src.act);
                        }
                    }
                    if constexpr (piece == 1) split_pair_h2(c_v.x, c_v.y, sc_in, c_h.x, c_l.x);
                    if constexpr (piece == 2) split_pair_h2(c_v.z, c_v.w, sc_in, c_h.y, c_l.y);
                    if constexpr (piece == 3) {
                        unsigned *d = nb + (pix0 + 16 * s) * S_PITCH + q4 * 2;
                        *reinterpret_cast<uint2 *>(d) = c_h;
                        *reinterpret_cast<uint2 *>(d + 32) = c_l;
                    }
                } else if constexpr (i < n_items) {
                    fetch_epilogue(std::integral_constant<int, n_ep ? ep_lo + (i - (n_items - n_ep)) : 0>{});
                }
            };
            __builtin_amdgcn_sched_barrier(0);
            // (weight term, pixel term) of the three partial products (l, h), (h, l), (h, h), round-robin over the accumulators
            static_for<0, n_mfma>([&](auto mc) __attribute__((always_inline)) {
                constexpr int m = decltype(mc)::value, prod = m / (MT * NT), mt = (m % (MT * NT)) / NT, nt = m % NT;
                constexpr int tw = prod == 0 ? 1 : 0, tx = prod == 1 ? 1 : 0;
                MFMA_H(acc[mt][nt], w2[wr][nt][tw], x2[cu][mt][tx]);
                __builtin_amdgcn_sched_barrier(0);
                static_for<m * n_items / n_mfma, (m + 1) * n_items / n_mfma>(item);
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        CSTAMP(3 + 6 * cst);
        // the next tile's first steps: the exchange hides them
        static_for<0, W_AHEAD>([&](auto r_) __attribute__((always_inline)) { load_w(r_, kx_first, decltype(r_)::value); });
        __syncthreads();                                         // every read of this buffer is done; the next tile is staged
        CSTAMP(4 + 6 * cst);

        // ---- the four kernel rows' partial sums meet in this tile's buffer: wave w finishes channels 8 w + 4 half + j of every
        //      tile -- ITS accumulator registers 0-3 (the weight prep rotates each kernel row's output channels) -- and hands
        //      registers 4 o .. 4 o + 3 to wave w + o; one column tile at a time
        float4 *xch = reinterpret_cast<float4 *>(lds + cur * S_BUF);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int o = 1; o < 4; ++o) {                        // to owner gw = (wave + o) & 3, as its source number 3 - o
                const int gw = (wave + o) & 3;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt)
                    xch[((gw * 3 + (3 - o)) * MT + mt) * 64 + lane] =
                        make_float4(acc[mt][nt][4 * o], acc[mt][nt][4 * o + 1], acc[mt][nt][4 * o + 2], acc[mt][nt][4 * o + 3]);
            }
            __syncthreads();
            CSTAMP(5 + 6 * cst + 2 * nt);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                float4 v = make_float4(acc[mt][nt][0], acc[mt][nt][1], acc[mt][nt][2], acc[mt][nt][3]);
#pragma unroll
                for (int sidx = 0; sidx < 3; ++sidx) {           // fixed order: the waves wave + 1, wave + 2, wave + 3 (mod 4)
                    const float4 p = xch[((wave * 3 + sidx) * MT + mt) * 64 + lane];
                    v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
                }
                const int P = opix[mt];
                const int r = P / g.ow;
                {
                    const bool ok = P >= 0 && oy0 + r < g.oh && ch_ok[nt];
                    const int64_t o = (((int64_t)img * g.oh + oy0) * g.ow + P) * g.q + nt * 32 + 8 * wave + 4 * half;
                    v.x = act_fwd_coef(fmaf(v.x, inv, b4[nt].x), ac); v.y = act_fwd_coef(fmaf(v.y, inv, b4[nt].y), ac);
                    v.z = act_fwd_coef(fmaf(v.z, inv, b4[nt].z), ac); v.w = act_fwd_coef(fmaf(v.w, inv, b4[nt].w), ac);
                    if (g.gate.y != nullptr) {
                        const unsigned m = km[mt][nt];
                        v.x *= gate_deriv(gy[mt][nt].x, gc) * (float)(m & 255u);
                        v.y *= gate_deriv(gy[mt][nt].y, gc) * (float)((m >> 8) & 255u);
                        v.z *= gate_deriv(gy[mt][nt].z, gc) * (float)((m >> 16) & 255u);
                        v.w *= gate_deriv(gy[mt][nt].w, gc) * (float)(m >> 24);
                    } else if (g.mask != nullptr) {
                        const unsigned m = km[mt][nt];
                        v.x *= 2.f * (float)(m & 255u); v.y *= 2.f * (float)((m >> 8) & 255u);
                        v.z *= 2.f * (float)((m >> 16) & 255u); v.w *= 2.f * (float)(m >> 24);
                    }
                    // unconditional: under the lane test the stores sat behind a branch, and the next tile's first weight operands
                    // (requested before the exchange) were then waited for with vmcnt(0) -- every store of this epilogue included
                    amax_run = ok ? fmaxf(amax_run, amax4(v)) : amax_run;
#ifdef C64S_ABL_NOSTORE
                    buf_store4(v, rs_out, (ok && v.x == 1234.5f) ? (unsigned)o * 4u : OOB);
#else
                    buf_store4(v, rs_out, ok ? (unsigned)o * 4u : OOB);
#endif
                }
            }
            __syncthreads();
            CSTAMP(6 + 6 * cst + 2 * nt);
        }
        ++cst;
    }
    amax_publish(g.amax_out, blockIdx.x * 4 + wave, gridDim.x * 4, amax_run);
}

// ---- host side ---------------------------------------------------------------------------------------------------------
// AMAX array of an operand as multiplied (count floats, a multiple of 4)
int conv64_operand_amax(const Operand &x, int64_t count, unsigned *out, hipStream_t s) {
    int64_t blocks = (count / 4 + 255) / 256;
    if (blocks > AMAX_N) blocks = AMAX_N;
    if (blocks < 1) blocks = 1;
    ARVAE_LAUNCH(operand_amax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, count / 4, out);
    return check_launch("operand_amax");
}

static int cu_count_s() { return device_cu_count(); }

// rows per tile and column tiles for an output width / source width, or false
static bool stage_geometry(int ow, int sw, int &rows, int &mt) {
    for (int m = 4; m >= 3; --m) {
        const int r = 32 * m / ow;
        if (r >= 1 && (r + 3) * sw <= S_PIX) { rows = r; mt = m; return true; }
    }
    return false;
}

int64_t conv64s_ws_floats() { return (S_PREP_UINT4 + 1) * 4 + AMAX_N; }      // split weights + their inverse scale | the source's AMAX array

// 64 source channels, 64 or 4..32 (a multiple of 4) output channels, 4x4 taps, stride 1, channels-last without permutation,
// a row group that fits the staging buffers
bool conv64s_fits(const arvae_link_t *l, bool up) {
    static const bool off = diag_env("ARVAE_CONV64_NO_STAGE") != nullptr;      // diagnostic: the gathering kernel instead
    int rows, mt;
    const int ow = up ? l->hw : l->lw, sw = up ? l->lw : l->hw, cs = up ? l->clo : l->chi, q = up ? l->chi : l->clo;
    const int oh = up ? l->hh : l->lh;
    return !off && l->stride == 1 && l->kh == 4 && l->kw == 4 && cs == 64 && (q == 64 || (q <= 32 && q >= 4 && (q & 3) == 0)) &&
           l->hi_perm_c == 0 && l->lo_perm_c == 0 && stage_geometry(ow, sw, rows, mt) &&
           (int64_t)l->n * oh * ow * q * 4 < ((int64_t)1 << 31) - 65536 &&    // the result and the source are addressed through
           (int64_t)l->n * (up ? l->lh : l->hh) * sw * 64 * 4 < ((int64_t)1 << 31) - (1 << 20);   // buffer resources (32-bit byte offsets)
}

template <int MT, int NT> static void launch_stage(const ConvStage &g, int grid, hipStream_t s) {
    constexpr int LDS = 2 * S_BUF * 4;
    static std::once_flag attr;
    std::call_once(attr, [&] {
        (void)hipFuncSetAttribute((const void *)conv64s_kernel<MT, NT, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void *)conv64s_kernel<MT, NT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void *)conv64s_kernel<MT, NT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    });
    const int mode = (g.src.y == nullptr || (g.src.act == ARVAE_ACT_NONE && g.src.mask == nullptr)) ? 0 : g.src.mode();
    if (mode == 0) ARVAE_LAUNCH((conv64s_kernel<MT, NT, 0>), dim3(grid), dim3(256), LDS, s, g);
    else if (mode == 1) ARVAE_LAUNCH((conv64s_kernel<MT, NT, 1>), dim3(grid), dim3(256), LDS, s, g);
    else ARVAE_LAUNCH((conv64s_kernel<MT, NT, 2>), dim3(grid), dim3(256), LDS, s, g);
}

// src [n][sh][sw][64] -> out [n][oh][ow][q]; source coordinate = output coordinate + sgn * k + off
int conv64s_prep_batch(const float *const *wts, float *const *outs, const int *transposed, const int *q, int count, hipStream_t s) {
    if (count <= 0) return ARVAE_OK;
    ARVAE_REQUIRE(count <= C64S_PREP_MAX, "conv64s_prep_batch: at most %d jobs", C64S_PREP_MAX);
    C64sPrepJobs j{};
    j.count = count;
    for (int k = 0; k < count; ++k) {
        j.wt[k] = wts[k]; j.out[k] = reinterpret_cast<uint4 *>(outs[k]); j.transposed[k] = transposed[k]; j.q[k] = q[k];
    }
    ARVAE_LAUNCH(conv64s_weight_prep_batch_kernel, dim3(16 * 4 * 2 * 64 / 256, count), dim3(256), 0, s, j);
    return check_launch("conv64s_weight_prep_batch_kernel");
}

// prepped: ws already holds the split weights (conv64s_prep_batch on this stream, this step)
int conv64s_run(const Operand &src, int n, int sh, int sw, int oh, int ow, int q, int sgn, int off, const float *wt, bool transposed,
                const float *bias, int act, const uint8_t *mask, float *out, float *ws, hipStream_t s, const char *what, const GateOp *gate,
                const unsigned *amax_in, unsigned *amax_out, bool prepped) {
    if (ws == nullptr || (reinterpret_cast<uintptr_t>(ws) & 15) != 0)
        return fail(ARVAE_E_INVALID, "%s: needs arvae_link_ws_floats() floats of 16-byte aligned workspace for the split weights", what);
    // (the weight prep rotates a kernel row's output channels by the number of the wave that runs it: sgn > 0 <-> wave = ky)
    ARVAE_REQUIRE((sgn > 0) == !transposed, "%s: the row-staged kernel runs Conv2d-forward weights with sgn > 0 and transposed ones with sgn < 0", what);
    ConvStage g{};
    int mt = 0;
    if (!stage_geometry(ow, sw, g.rows, mt)) return fail(ARVAE_E_INVALID, "%s: no staging geometry", what);
    g.src = src; g.n = n; g.sh = sh; g.sw = sw; g.oh = oh; g.ow = ow; g.q = q;
    g.sgn = sgn; g.dmin = sgn > 0 ? off : off - 3;
    g.groups = (oh + g.rows - 1) / g.rows;
    g.wprep = reinterpret_cast<const uint4 *>(ws);
    g.bias = bias; g.mask = mask; g.act = act; g.out = out;
    if (gate != nullptr) g.gate = *gate;
    const int nt = q > 32 ? 2 : 1;
    if (!prepped)
        ARVAE_LAUNCH(conv64s_weight_prep_kernel, dim3(16 * 4 * nt * 64 / 256), dim3(256), 0, s, wt, reinterpret_cast<uint4 *>(ws),
                     transposed ? 1 : 0, q, nt);
    const bool plain = src.y == nullptr || (src.act == ARVAE_ACT_NONE && src.mask == nullptr);
    if (amax_in == nullptr || !plain) {                         // no maxima with the tensor, or not of what is multiplied
        unsigned *am = reinterpret_cast<unsigned *>(ws + (S_PREP_UINT4 + 1) * 4);
        if (int rc = conv64_operand_amax(src, (int64_t)n * sh * sw * 64, am, s)) return rc;
        amax_in = am;
    }
    g.amax_in = amax_in;
    g.amax_out = amax_out;
    const int tiles = n * g.groups, cus = cu_count_s() < AMAX_N / 4 ? cu_count_s() : AMAX_N / 4;
    const int grid = tiles < cus ? tiles : cus;
    if (mt == 4 && nt == 2) launch_stage<4, 2>(g, grid, s);
    else if (mt == 3 && nt == 2) launch_stage<3, 2>(g, grid, s);
    else if (mt == 4) launch_stage<4, 1>(g, grid, s);
    else launch_stage<3, 1>(g, grid, s);
    return check_launch(what);
}

}  // namespace arvae

#ifdef C64S_STAMPS
extern "C" int arvae_debug_c64s_stamps(unsigned long long *out, int count) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(arvae::g_c64s_stamps), sizeof(unsigned long long) * count);
}
#endif
