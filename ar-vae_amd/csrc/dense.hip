// nn.Linear forward / data-gradient / weight-gradient for batch-sized GEMMs (M = batch 64..4096,
// features 10..2888) on the fp32 MFMA.  These are latency-bound, cache-resident problems: a
// workgroup owns ONE 32x32 output tile, its 16 wavefronts split the reduction axis (so a wave needs only
// one or two rounds of loads: the kernels are latency-, not throughput-bound) and sum through LDS, and
// operands go straight from L2 to registers (no LDS staging): a lane reads 4 consecutive
// reduction elements (16 B) when that axis is contiguous in memory, or 4 row-strided dwords that are
// coalesced across the wave when the OTHER axis is contiguous.  The reduction order is permuted
// (chunk of 8 = [half 0: 4][half 1: 4]) to fit the 32x32x2 MFMA's k = 2s + half lane layout.
//
// Channel permutations (hi_perm / lo_perm of arvae_link_t: the NCHW flatten between conv and dense
// stacks over channels-last activations) are index remaps on the feature axis.
#include <mutex>
#include "diag.h"
#include "common.h"
#include "conv32_common.h"
#include "dense.h"
#include "reduce.h"
#include "x3tile.h"
#include "wgrad_c1s.h"

namespace arvae {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void mfma4(f32x16 &acc, const float (&a)[4], const float (&b)[4]) {
#pragma unroll
    for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[t], acc, 0, 0, 0);
}

constexpr int NW = 16;                       // wavefronts per workgroup = reduction split
constexpr int DENSE_THREADS = 64 * NW;

// sum the NWT waves' 32x32 partial tiles through LDS; wave w returns accumulator registers w * RPW .. + RPW - 1 of the total
template <int NWT>
__device__ __forceinline__ void reduce_waves(float *red, const f32x16 &acc, int wave, int lane, float (&out)[16 / NWT]) {
    constexpr int RPW = 16 / NWT;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < RPW; ++e) {
        float s = 0.f;
#pragma unroll
        for (int ws = 0; ws < NWT; ++ws) s += red[(ws * 16 + wave * RPW + e) * 64 + lane];
        out[e] = s;
    }
}
// a tile is split over 16 waves when the reduction is long (one or two rounds of loads per wave: the kernels are latency-bound)
// and over 4 when it is short and the tiles are many (K <= 512 with >= 512 tiles: the 16-way LDS reduction of 16 accumulator
// registers per wave then costs more than the 8 MFMAs each wave contributes -- Morpho-MNIST's 256 <-> 2888 layers)
static bool dense_short_k(int k, int tiles) {
    static const bool off = diag_env("ARVAE_DENSE_NW16") != nullptr;
    // (and whenever the reduction is at most 160 long: 16 waves would contribute one or two MFMAs each to a tile and then spend
    // longer reducing it -- the tick RNN's vocab + 1 + 4B rows x 138 inputs: 24 us on sixteen waves)
    return !off && k <= 512 && (tiles >= 512 || k <= 160);
}

// ---- forward: Y[m][out_perm(n)] = act( sum_k X[m][km] * W[n][feat(km)] + b[n] ),  km = memory column ----------
template <int NW>
__global__ __launch_bounds__(64 * NW) void dense_fwd_kernel(DenseArgs p) {
    __shared__ float red[NW * 16 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, rc = lane & 31;
    const int m = blockIdx.x * 32 + rc, n = blockIdx.y * 32 + rc;
    const bool mok = m < p.batch, nok = n < p.n_out;
    const bool vec = (p.n_in & 3) == 0;
    const float *xrow = p.a.v + (int64_t)(mok ? m : 0) * p.n_in;
    const float *wrow = p.w + (int64_t)(nok ? n : 0) * p.n_in;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int chunks = (p.n_in + 7) / 8;

    // FU reduction chunks per wave and round, every load of a round issued before the first MFMA (unconditional loads from
    // clamped addresses, zeroed by selects): a wave's share of the reduction is 1-4 chunks, so a layer is one memory round
    // trip instead of one per chunk
    constexpr int FU = 4;
    // Channel-permuted input (the NCHW flatten in front of the first Linear layer): memory column p * C + c <-> feature
    // c * HW + p, so a run that is contiguous in X is a stride-HW gather in W and vice versa.  A lane takes a 4 x 4 block
    // (c0..c0+3) x (p0..p0+3) instead: four 16-byte loads of X (one per p), four of W (one per c), paired transposed in
    // registers -- 8 loads per 16 reduction elements instead of 4 + 16 (13.9 -> 7 us for the 512 -> 256 layer at B = 512).
    const bool perm4 = vec && p.in_perm.c_count > 0 && (p.in_perm.c_count & 3) == 0 && (p.in_perm.hw & 3) == 0;
    if (perm4) {
        const int cb_n = p.in_perm.c_count >> 2, blocks = cb_n * (p.in_perm.hw >> 2);
        for (int q = wave; 2 * q < blocks; q += NW) {
            const int blk = 2 * q + half;
            const bool ok = blk < blocks;
            const int bc = ok ? blk : 0, c0 = (bc % cb_n) * 4, p0 = (bc / cb_n) * 4;
            float4 xa[4], wb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                xa[i] = *reinterpret_cast<const float4 *>(xrow + (p0 + i) * p.in_perm.c_count + c0);
                wb[i] = *reinterpret_cast<const float4 *>(wrow + (c0 + i) * p.in_perm.hw + p0);
            }
            const bool am = ok && mok, bm = ok && nok;
#pragma unroll
            for (int i = 0; i < 4; ++i) {                        // position p0 + i, channels c0 .. c0 + 3
                const float a[4] = {am ? xa[i].x : 0.f, am ? xa[i].y : 0.f, am ? xa[i].z : 0.f, am ? xa[i].w : 0.f};
                const float w0 = i == 0 ? wb[0].x : i == 1 ? wb[0].y : i == 2 ? wb[0].z : wb[0].w;
                const float w1 = i == 0 ? wb[1].x : i == 1 ? wb[1].y : i == 2 ? wb[1].z : wb[1].w;
                const float w2 = i == 0 ? wb[2].x : i == 1 ? wb[2].y : i == 2 ? wb[2].z : wb[2].w;
                const float w3 = i == 0 ? wb[3].x : i == 1 ? wb[3].y : i == 2 ? wb[3].z : wb[3].w;
                const float b[4] = {bm ? w0 : 0.f, bm ? w1 : 0.f, bm ? w2 : 0.f, bm ? w3 : 0.f};
                mfma4(acc, a, b);
            }
        }
    }
    for (int q0 = wave; q0 < (perm4 ? 0 : chunks); q0 += FU * NW) {
        float a[FU][4], b[FU][4];
        if (vec && p.in_perm.c_count == 0) {
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                const int k0 = (q0 + u * NW) * 8 + half * 4;
                const bool ok = k0 < p.n_in;                     // n_in is a multiple of 4 here: the whole quad is inside
                const float4 av = *reinterpret_cast<const float4 *>(xrow + (ok ? k0 : 0));
                const float4 bv = *reinterpret_cast<const float4 *>(wrow + (ok ? k0 : 0));
                a[u][0] = ok ? av.x : 0.f; a[u][1] = ok ? av.y : 0.f; a[u][2] = ok ? av.z : 0.f; a[u][3] = ok ? av.w : 0.f;
                b[u][0] = bv.x; b[u][1] = bv.y; b[u][2] = bv.z; b[u][3] = bv.w;
            }
        } else if (vec) {
            // channel-permuted input that the 4 x 4 block path cannot take (Morpho-MNIST: 8 channels x 361 positions): the reduction
            // runs in FEATURE order, so W is one 16-byte load per lane and the four X values of a lane sit 4 c_count bytes apart in
            // one cache line of its row -- the other way round (X contiguous, W gathered at a stride of hw floats) every W load
            // touched its own line (93 us for the 2888 -> 256 layer at B = 1024)
            const unsigned inv_hw = (1u << 24) / (unsigned)p.in_perm.hw + 1u;          // f / hw = (f * inv_hw) >> 24 for f * hw < 2^24
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                const int f0 = (q0 + u * NW) * 8 + half * 4;
                const bool ok = f0 < p.n_in;
                const float4 bv = *reinterpret_cast<const float4 *>(wrow + (ok ? f0 : 0));
                b[u][0] = bv.x; b[u][1] = bv.y; b[u][2] = bv.z; b[u][3] = bv.w;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const unsigned f = ok ? (unsigned)(f0 + t) : 0u, c = (f * inv_hw) >> 24, pp = f - c * (unsigned)p.in_perm.hw;
                    const float av = xrow[pp * (unsigned)p.in_perm.c_count + c];
                    a[u][t] = ok ? av : 0.f;
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                const int k0 = (q0 + u * NW) * 8 + half * 4;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const bool ok = k0 + t < p.n_in;
                    const float av = xrow[ok ? k0 + t : 0];
                    a[u][t] = ok ? av : 0.f;
                    b[u][t] = wrow[p.in_perm.to_feat(ok ? k0 + t : 0)];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < FU; ++u) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                a[u][t] = mok ? a[u][t] : 0.f;
                b[u][t] = nok ? b[u][t] : 0.f;
            }
            mfma4(acc, a[u], b[u]);
        }
    }
    float v[16 / NW];
    reduce_waves<NW>(red, acc, wave, lane, v);
    if (nok) {
        const float bias = p.bias != nullptr ? p.bias[n] : 0.f;
        const int col = p.out_perm.to_mem(n);
#pragma unroll
        for (int e = 0; e < 16 / NW; ++e) {
            const int reg = wave * (16 / NW) + e;                                        // accumulator register
            const int row = blockIdx.x * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
            if (row < p.batch) p.out[(int64_t)row * p.n_out + col] = act_fwd(v[e] + bias, p.act);
        }
    }
}

// ---- dgrad: dX[m][in_mem] = sum_{nm} G[m][nm] * W[feat_out(nm)][feat_in(in_mem)]  ---------------------------
// reduction loop: FU chunks per wave and round, loads first (see dense_fwd_kernel); MODE = Operand::mode() of G
template <int MODE, int NW>
__device__ __forceinline__ void dense_dgrad_loop(const DenseArgs &p, int64_t grow, int kf, bool mok, bool kok, bool vec, int wave,
                                                 int half, f32x16 &acc) {
    constexpr int FU = 4;
    const int chunks = (p.n_out + 7) / 8;
    for (int q0 = wave; q0 < chunks; q0 += FU * NW) {
        float a[FU][4], ay[FU][4], am[FU][4], b[FU][4];
        bool okt[FU][4];
        if (vec && MODE <= 1) {
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                const int n0 = (q0 + u * NW) * 8 + half * 4;
                const bool ok = n0 < p.n_out;                    // n_out is a multiple of 4 here
                const int nc = ok ? n0 : 0;
                const float4 av = *reinterpret_cast<const float4 *>(p.a.v + grow + nc);
                a[u][0] = av.x; a[u][1] = av.y; a[u][2] = av.z; a[u][3] = av.w;
                if (MODE == 1) {
                    const float4 yv = *reinterpret_cast<const float4 *>(p.a.y + grow + nc);
                    ay[u][0] = yv.x; ay[u][1] = yv.y; ay[u][2] = yv.z; ay[u][3] = yv.w;
                }
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    okt[u][t] = ok;
                    am[u][t] = 1.f;
                    if (MODE == 0) ay[u][t] = 0.f;
                    b[u][t] = p.w[(int64_t)p.out_perm.to_feat(nc + t) * p.n_in + kf];
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < FU; ++u) {
                const int n0 = (q0 + u * NW) * 8 + half * 4;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    okt[u][t] = n0 + t < p.n_out;
                    const int nc = okt[u][t] ? n0 + t : 0;
                    p.a.template fetch<MODE>(grow + nc, a[u][t], ay[u][t], am[u][t]);
                    b[u][t] = p.w[(int64_t)p.out_perm.to_feat(nc) * p.n_in + kf];
                }
            }
        }
#pragma unroll
        for (int u = 0; u < FU; ++u) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float g = p.a.template apply<MODE>(a[u][t], ay[u][t], am[u][t]);
                a[u][t] = (okt[u][t] && mok) ? g : 0.f;
                b[u][t] = (okt[u][t] && kok) ? b[u][t] : 0.f;
            }
            mfma4(acc, a[u], b[u]);
        }
    }
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void dense_dgrad_kernel(DenseArgs p) {
    __shared__ float red[NW * 16 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, rc = lane & 31;
    // A tile owns 32 consecutive input FEATURES (columns of W, read coalesced in the reduction loop); with a channel
    // permutation their dX columns are scattered, which costs one strided store per lane instead of a strided weight
    // gather per reduction step.
    const int m = blockIdx.x * 32 + rc, kfl = blockIdx.y * 32 + rc;
    const bool mok = m < p.batch, kok = kfl < p.n_in;
    const int kf = kok ? kfl : 0;
    const int km = p.in_perm.to_mem(kf);                                 // memory column of dX
    const bool vec = (p.n_out & 3) == 0;
    const int64_t grow = (int64_t)(mok ? m : 0) * p.n_out;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    if (p.a.mode() == 0) dense_dgrad_loop<0, NW>(p, grow, kf, mok, kok, vec, wave, half, acc);
    else if (p.a.mode() == 1) dense_dgrad_loop<1, NW>(p, grow, kf, mok, kok, vec, wave, half, acc);
    else dense_dgrad_loop<2, NW>(p, grow, kf, mok, kok, vec, wave, half, acc);
    float v[16 / NW];
    reduce_waves<NW>(red, acc, wave, lane, v);
    if (kok) {
#pragma unroll
        for (int e = 0; e < 16 / NW; ++e) {
            const int reg = wave * (16 / NW) + e;
            const int row = blockIdx.x * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
            if (row < p.batch) {
                const int64_t idx = (int64_t)row * p.n_in + km;
                p.out[idx] = (p.gate == nullptr || p.gate[idx] > 0.f) ? v[e] : 0.f;
            }
        }
    }
}

// ---- long-batch variants (M >= DENSE_SPLIT_MIN_ROWS rows: the MeasureVAE's whole-sequence GEMMs, 24 ticks x batch) --
// Thousands of rows make these throughput problems: C[P][Q] = sum_r A(p,r) B(q,r) with both operand tiles staged
// through LDS in reduction-major order [r][p] by coalesced 16-byte global loads (next chunk prefetched into registers
// while the current one is multiplied), 64 x 64 outputs per workgroup = 2 x 2 waves on the 32x32x2 MFMA (small tiles: these
// are 0.3-1.2 GFLOP problems, and two or three workgroups per CU hide each other's load latency).  An operand is either
// "rows x K" (reduction index contiguous in memory: X, W and G of the forward / data-gradient products) or "K x rows"
// (output index contiguous: W of the data gradient, G and X of the weight gradient); the weight gradient additionally
// splits the row axis over blockIdx.z into workspace slices that dense_split_reduce_kernel adds up in slice order.
constexpr int RG_TP = 64, RG_TQ = 64;
constexpr int RG_PA = RG_TP + 4, RG_PB = RG_TQ + 4;
enum { RG_EP_FWD = 0, RG_EP_SLICE = 1 };

struct RowsGemm {
    const float *a, *b;
    int64_t lda, ldb;
    int P, Q, Rn;               // output rows (A side), output columns (B side), reduction length
    int rslice;                 // reduction indices per blockIdx.z
    const float *bias;          // FWD: per output column
    int act;
    float *out;                 // FWD: [P][ldo]; SLICE: ws, slice z at out + z*slice_floats, bias sums after the P*ldo tile
    int64_t ldo, slice_floats;
    int want_bias;              // SLICE: also write sum_r A(p, r)
};

// NV float4 (or 4 scalar) loads per thread for a TP x RG_R tile.  Every load is unconditional (clamped address, value
// selected afterwards) so that the whole batch is in flight at once; VEC = 16-byte loads (leading dimension, extents
// and base all multiples of 4 floats), otherwise dword loads.
template <int LAY, int TP, bool VEC>
struct TileLoader {
    static constexpr int NV = TP * RG_R / 4 / 256;
    float4 v[NV];
    __device__ __forceinline__ void load(const float *base, int64_t ld, int p0, int pmax, int r0, int rmax) {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = threadIdx.x + 256 * i;
            // (row, first column) of this thread's 4 consecutive elements in memory; lim = end of the contiguous axis
            int row, col, row_max, col_max;
            if (LAY == RG_ROWSK) { row = p0 + idx / (RG_R / 4); col = r0 + 4 * (idx % (RG_R / 4)); row_max = pmax; col_max = rmax; }
            else { row = r0 + idx / (TP / 4); col = p0 + 4 * (idx % (TP / 4)); row_max = rmax; col_max = pmax; }
            const bool rok = row < row_max;
            if (VEC) {
                // raw buffer load: the chunk's position along the reduction axis is the SCALAR offset (r0 elements or r0 rows), the
                // thread's place in the tile a per-lane offset that does not change from chunk to chunk (sent beyond the range
                // for an element outside the matrix: reads as zero) -- the generic-pointer form cost a 64-bit address, two selects
                // and a load under a lane test per 16 bytes (rows_gemm_fits_32 bounds the matrices at 2 GB)
                const bool ok = rok && col < col_max;
                const int rel_row = LAY == RG_ROWSK ? row : row - r0, rel_col = LAY == RG_ROWSK ? col - r0 : col;
                const unsigned off = ok ? (unsigned)((rel_row * (int)ld + rel_col) * 4) : 0xfffffff0u;
                const int so = LAY == RG_ROWSK ? r0 * 4 : r0 * (int)ld * 4;
                const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 0x7fffffff, 0x00020000);
                typedef float f32x4t_ __attribute__((ext_vector_type(4)));
                const f32x4t_ t = __builtin_bit_cast(f32x4t_, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, so, 0));
                v[i] = float4{t.x, t.y, t.z, t.w};
            } else {
                const float *src = base + (int64_t)(rok ? row : 0) * ld;
                float e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = rok && col + j < col_max;
                    const float t = src[ok ? col + j : 0];
                    e[j] = ok ? t : 0.f;
                }
                v[i] = float4{e[0], e[1], e[2], e[3]};
            }
        }
    }
    // LDS image [RG_R][pitch], reduction-major
    __device__ __forceinline__ void commit(float *lds, int pitch) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int idx = threadIdx.x + 256 * i;
            if (LAY == RG_ROWSK) {
                const int p = idx / (RG_R / 4), r = 4 * (idx % (RG_R / 4));
                lds[(r + 0) * pitch + p] = v[i].x; lds[(r + 1) * pitch + p] = v[i].y;
                lds[(r + 2) * pitch + p] = v[i].z; lds[(r + 3) * pitch + p] = v[i].w;
            } else {
                const int r = idx / (TP / 4), p = 4 * (idx % (TP / 4));
                *reinterpret_cast<float4 *>(lds + r * pitch + p) = v[i];
            }
        }
    }
    // three bf16 planes for the split-bf16 MFMA (x3tile.h)
    typedef X3Plane<LAY, TP> Plane;
    static constexpr int PLANE = Plane::PLANE;
    __device__ __forceinline__ void commit3(unsigned short *lds) const {
#pragma unroll
        for (int i = 0; i < NV; ++i) Plane::commit(lds, threadIdx.x + 256 * i, v[i]);
    }
    __device__ static __forceinline__ int lane_base(int w) { return Plane::lane_base(w); }
    __device__ static __forceinline__ rg_bf16x8 operand(const unsigned short *lds, int base, int t, int s) {
        return Plane::operand(lds, base, t, s);
    }
};

template <int LA, int LB, int EP, bool VA, bool VB>
__global__ __launch_bounds__(256) void rows_gemm_kernel(RowsGemm g) {
    __shared__ float As[RG_R * RG_PA];
    __shared__ float Bs[RG_R * RG_PB];
    const int lane = threadIdx.x & 63, half = lane >> 5, rc = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p0 = blockIdx.x * RG_TP, q0 = blockIdx.y * RG_TQ;
    const int rbeg = blockIdx.z * g.rslice, rend = min(g.Rn, rbeg + g.rslice);
    TileLoader<LA, RG_TP, VA> la;
    TileLoader<LB, RG_TQ, VB> lb;
    const int wp = wave & 1, wq = wave >> 1;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float bsum = 0.f;
    la.load(g.a, g.lda, p0, g.P, rbeg, rend);
    lb.load(g.b, g.ldb, q0, g.Q, rbeg, rend);
    for (int r0 = rbeg; r0 < rend; r0 += RG_R) {
        __syncthreads();
        la.commit(As, RG_PA);
        lb.commit(Bs, RG_PB);
        __syncthreads();
        if (r0 + RG_R < rend) {
            la.load(g.a, g.lda, p0, g.P, r0 + RG_R, rend);
            lb.load(g.b, g.ldb, q0, g.Q, r0 + RG_R, rend);
        }
#pragma unroll
        for (int s = 0; s < RG_R / 2; ++s) {
            const float a = As[(2 * s + half) * RG_PA + 32 * wp + rc];
            const float b = Bs[(2 * s + half) * RG_PB + 32 * wq + rc];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
        if (EP == RG_EP_SLICE && g.want_bias && blockIdx.y == 0 && threadIdx.x < RG_TP) {
#pragma unroll 8
            for (int r = 0; r < RG_R; ++r) bsum += As[r * RG_PA + threadIdx.x];
        }
    }
    float *out = g.out + (EP == RG_EP_SLICE ? blockIdx.z * g.slice_floats : 0);
    const int q = q0 + 32 * wq + rc;
    const float bias = (EP == RG_EP_FWD && g.bias != nullptr && q < g.Q) ? g.bias[q] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int p = p0 + 32 * wp + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = acc[r];
        if (EP == RG_EP_FWD) v = act_fwd(v + bias, g.act);
        if (p < g.P && q < g.Q) out[(int64_t)p * g.ldo + q] = v;
    }
    if (EP == RG_EP_SLICE && g.want_bias && blockIdx.y == 0 && threadIdx.x < RG_TP && p0 + (int)threadIdx.x < g.P)
        out[(int64_t)g.P * g.ldo + p0 + threadIdx.x] = bsum;
}

// The same product on the bf16 MFMA at fp32 accuracy: operands split into three bf16 terms when they are committed to
// LDS ([p][r] planes, r contiguous), six partial products per multiply-add on v_mfma_f32_32x32x16_bf16, smallest first
// (2.7x fewer MFMA cycles than the fp32 32x32x2; ARVAE_ROWS_GEMM_FP32=1 selects the kernel above).
#ifdef RG_STAMPS
// diagnostic build only (tools/stamp_rg.py): phase timeline of the first 512 workgroups of a rows-GEMM launch, 100 MHz wall clock
__device__ unsigned long long g_rg_stamps[512 * 32];
#define RGSTAMP(slot) do { if (threadIdx.x == 0 && (slot) < 32) { const int wg_ = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x; if (wg_ < 512) g_rg_stamps[wg_ * 32 + (slot)] = wall_clock64(); } } while (0)
#define RGWAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define RGSTAMP(slot)
#define RGWAIT()
#endif
template <int LA, int LB, int EP, bool VA, bool VB>
__device__ __forceinline__ void rows_gemm_x3_body(const RowsGemm &g, const int bx, const int by, const int bz) {
    RGSTAMP(0);
    typedef TileLoader<LA, RG_TP, VA> LoadA;
    typedef TileLoader<LB, RG_TQ, VB> LoadB;
    __shared__ __attribute__((aligned(16))) unsigned short As[3 * LoadA::PLANE];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[3 * LoadB::PLANE];
    const int lane = threadIdx.x & 63, half = lane >> 5, rc = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p0 = bx * RG_TP, q0 = by * RG_TQ;
    const int rbeg = bz * g.rslice, rend = min(g.Rn, rbeg + g.rslice);
    LoadA la;
    LoadB lb;
    const int wp = wave & 1, wq = wave >> 1;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float bsum = 0.f;
    la.load(g.a, g.lda, p0, g.P, rbeg, rend);
    lb.load(g.b, g.ldb, q0, g.Q, rbeg, rend);
    const int abase = LoadA::lane_base(wp), bbase = LoadB::lane_base(wq);
    RGSTAMP(1);
    int rgc = 0;
    (void)rgc;
    for (int r0 = rbeg; r0 < rend; r0 += RG_R) {
        __syncthreads();
        RGSTAMP(2 + 5 * rgc);
        RGWAIT();
        RGSTAMP(3 + 5 * rgc);
        la.commit3(As);
        lb.commit3(Bs);
        RGSTAMP(4 + 5 * rgc);
        __syncthreads();
        RGSTAMP(5 + 5 * rgc);
        if (r0 + RG_R < rend) {
            la.load(g.a, g.lda, p0, g.P, r0 + RG_R, rend);
            lb.load(g.b, g.ldb, q0, g.Q, r0 + RG_R, rend);
        }
#pragma unroll
        for (int s = 0; s < RG_R / 16; ++s) {
            const rg_bf16x8 ah = LoadA::operand(As, abase, 0, s), am = LoadA::operand(As, abase, 1, s), al = LoadA::operand(As, abase, 2, s);
            const rg_bf16x8 bh = LoadB::operand(Bs, bbase, 0, s), bm = LoadB::operand(Bs, bbase, 1, s), bl = LoadB::operand(Bs, bbase, 2, s);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        }
        RGSTAMP(6 + 5 * rgc);
        ++rgc;
        if (EP == RG_EP_SLICE && g.want_bias && by == 0 && threadIdx.x < RG_TP) {
            // sum over the chunk's reduction indices of A(p = threadIdx.x, r), terms re-added exactly
#pragma unroll 8
            for (int r = 0; r < RG_R; ++r) {
                const unsigned short *e = As + (LA == RG_ROWSK ? threadIdx.x * RG_XP + r : r * RG_TRP + threadIdx.x);
                bsum += (__builtin_bit_cast(float, (unsigned)e[0] << 16) + __builtin_bit_cast(float, (unsigned)e[LoadA::PLANE] << 16)) +
                        __builtin_bit_cast(float, (unsigned)e[2 * LoadA::PLANE] << 16);
            }
        }
    }
    RGSTAMP(30);
    float *out = g.out + (EP == RG_EP_SLICE ? bz * g.slice_floats : 0);
    const int q = q0 + 32 * wq + rc;
    const float bias = (EP == RG_EP_FWD && g.bias != nullptr && q < g.Q) ? g.bias[q] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int p = p0 + 32 * wp + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = acc[r];
        if (EP == RG_EP_FWD) v = act_fwd(v + bias, g.act);
        if (p < g.P && q < g.Q) out[(int64_t)p * g.ldo + q] = v;
    }
    if (EP == RG_EP_SLICE && g.want_bias && by == 0 && threadIdx.x < RG_TP && p0 + (int)threadIdx.x < g.P)
        out[(int64_t)g.P * g.ldo + p0 + threadIdx.x] = bsum;
    RGWAIT();
    RGSTAMP(31);
}

template <int LA, int LB, int EP, bool VA, bool VB>
__global__ __launch_bounds__(256) void rows_gemm_x3_kernel(RowsGemm g) {
    rows_gemm_x3_body<LA, LB, EP, VA, VB>(g, blockIdx.x, blockIdx.y, blockIdx.z);
}

// ================================================================================================================================
// Wide Linear layers on the matrix pipe (round 6; Morpho-MNIST's Linear(2888, 256) / Linear(256, 2888), imagevae/mnist_vae.py:27-38,
// BASELINE.json configs[2] "MFMA on fc z-projections").  The latent block's row kernels (midblock.hip) stream every matrix through
// every workgroup on fp32 FMA chains: 80 + 85 us per step for these two layers' 6 GFLOP.  Here a layer's forward product and its
// data gradient are tile GEMMs  C[M][N] = sum_k A[m][k] B(k, n)  on the three-term bf16 MFMA (x3tile.h: fp32-accurate, no operand
// scales to carry), 64 x 64 tiles, 2 x 2 waves:
//   * A is always "rows x K" (activations / gradients, [batch][features]); B is the layer's ONE per-step copy W'[n_mem][k_mem]
//     (mid_prep: the NCHW-flatten permutation folded in) read as "rows x K" for the forward product and as "K x rows" -- through the
//     transposing LDS read -- for the data gradient: no second layout;
//   * both operands go through DOUBLE-BUFFERED LDS planes, three 32-deep chunks of global loads in flight per thread (register
//     sets), ONE barrier per chunk; two workgroups per CU hide each other's split / commit work behind their MFMAs;
//   * a long reduction (K = 2888 against 16 x 4 tiles) is split over blockIdx.z into workspace slices (EP_PARTIAL) that the
//     CONSUMER sums in slice order -- the latent block's kernels do it in their prologue with the bias, the activation (forward)
//     or its derivative (backward): no reduction launch; a short one (K = 256, 16 x 46 tiles) finishes in place (EP_FULL: bias,
//     activation or ReLU gate, the tensor's AMAX entry for the conv kernel that reads it next).
enum { WG_EP_PARTIAL = 0, WG_EP_FULL = 1 };
constexpr int WG_DEPTH = 3;
#ifndef ARVAE_WG_SINGLE
#define ARVAE_WG_SINGLE 1
#endif
constexpr int WG_NBUF = ARVAE_WG_SINGLE ? 1 : 2;     // LDS buffers of the 64-tile kernels (128-tile: always 2)
// Operand sources (T = tile extent along the operand's output axis, 64 or 128).  WideF32: an fp32 matrix, split into the three bf16
// terms on its way to LDS (TileLoader: ~11 vector instructions per value pair).  WidePlanes: the three bf16 planes already in
// memory -- the weights (mid_prep writes them once per step) and the small activations that dozens of tiles re-read (the latent
// block's kernels write them beside the fp32 tensors): 16 bytes per plane and load straight into the LDS image, no vector ALU work.
template <int LAY, int T_>
struct WideF32 {
    static constexpr int T = T_;
    typedef X3Plane<LAY, T> Plane;
    typedef TileLoader<LAY, T, true> Loader;
    struct Regs { Loader t; };
    const float *base;
    int64_t ld;
    int p0, pmax;
    __device__ __forceinline__ void init(const void *b, int64_t ld_, int64_t, int p0_, int pmax_) { base = static_cast<const float *>(b); ld = ld_; p0 = p0_; pmax = pmax_; }
    __device__ __forceinline__ void load(Regs &r, int r0, int rend) const { r.t.load(base, ld, p0, pmax, r0, rend); }
    __device__ __forceinline__ void commit(const Regs &r, unsigned short *lds) const { r.t.commit3(lds); }
    __device__ __forceinline__ void keep(const Regs &r) const {
#pragma unroll
        for (int i = 0; i < Loader::NV; ++i) asm volatile("" ::"v"(r.t.v[i].x), "v"(r.t.v[i].w));
    }
};
template <int LAY, int T_>
struct WidePlanes {
    static constexpr int T = T_;
    typedef X3Plane<LAY, T> Plane;
    static constexpr int NV = T / 64;                            // 16-byte units per plane and thread: T x 32 bf16 = 256 x NV x 8
    struct Regs { uint4 v[NV][3]; };
    __amdgpu_buffer_rsrc_t rs;
    unsigned voff[NV];
    int lds_off[NV], rloc[NV], ld2, ps2;
    // Planes in the TILED layout [3][K / 32][rows][32] of bf16 (x3tile.h x3_tiled_index), `pstride` elements apart, K zero-padded to
    // whole chunks: the 32 x (rows) block of one chunk is contiguous, so a wave's load is whole cache lines whichever axis is the
    // reduction -- "rows x K" (reduce along K: chunk c = block c, 64 bytes per row, rows adjacent) or "K x rows" (reduce along the
    // rows: 32 consecutive rows of one K block are 2 KB).  (Row-major planes made every request half a line: 13 TB/s of L2 -> LDS,
    // tools/probes/wide_gemm.py.)  `rows_` = the row count of the planes.
    __device__ __forceinline__ void init(const void *b, int64_t rows_, int64_t pstride, int p0_, int pmax_) {
        rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(b), 0, 0x7fffffff, 0x00020000);
        ld2 = (int)rows_ * 64;                                   // bytes of one K block of all rows
        ps2 = (int)(pstride * 2);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int u = threadIdx.x + 256 * i;
            bool ok;
            unsigned off;
            if (LAY == RG_ROWSK) {                               // unit = (row u / 4, reduction indices 8 (u % 4) ..): chunk = K block
                const int row = u >> 2, kg = u & 3;
                rloc[i] = 0;
                ok = p0_ + row < pmax_;
                off = (unsigned)(((p0_ + row) * 32 + 8 * kg) * 2);
                lds_off[i] = row * RG_XP + 8 * kg;
            } else {                                             // unit = (reduction row u / (T / 8), outputs 8 (u % (T / 8)) ..)
                const int r = u / (T / 8), qg = u % (T / 8);
                const int q = p0_ + 8 * qg;
                rloc[i] = r;
                ok = q < pmax_;
                off = (unsigned)((q >> 5) * ld2 + (r * 32 + (q & 31)) * 2);
                lds_off[i] = r * Plane::TRP + 8 * qg;
            }
            voff[i] = ok ? off : 0xfffffff0u;
        }
    }
    // (a chunk at or past `rend` reads zeros: the loop multiplies whole rounds of three chunks without a branch; "K x rows": this
    // unit's reduction row must exist too -- activations' planes are not padded along the batch)
    __device__ __forceinline__ void load(Regs &r, int r0, int rend) const {
        const int so = r0 < rend ? (LAY == RG_ROWSK ? (r0 >> 5) * ld2 : r0 * 64) : 0;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const bool in = LAY == RG_ROWSK ? r0 < rend : r0 + rloc[i] < rend;
            const unsigned vo = in ? voff[i] : 0xfffffff0u;
#pragma unroll
            for (int t = 0; t < 3; ++t)
                r.v[i][t] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(vo == 0xfffffff0u ? vo : vo + (unsigned)(t * ps2)), so, 0));
        }
    }
    __device__ __forceinline__ void commit(const Regs &r, unsigned short *lds) const {
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int t = 0; t < 3; ++t) *reinterpret_cast<uint4 *>(lds + t * Plane::PLANE + lds_off[i]) = r.v[i][t];
    }
    __device__ __forceinline__ void keep(const Regs &r) const {
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int t = 0; t < 3; ++t) asm volatile("" ::"v"(r.v[i][t].x), "v"(r.v[i][t].w));
    }
};
template <class SA, class SB>
struct WideLds {
    static constexpr int BUF = 3 * SA::Plane::PLANE + 3 * SB::Plane::PLANE;      // unsigned shorts per buffer
    static constexpr int NBUF = SA::T == 128 ? 2 : WG_NBUF;
    static constexpr size_t BYTES = (size_t)NBUF * BUF * sizeof(unsigned short) + 2048;    // + the bias-sum / AMAX scratch
};
// acc[a][b] = sum over the reduction range [rbeg, rend) of A(p, r) B(q, r) for this wave's NT x NT MFMA tiles of the workgroup's
// (64 NT) x (64 NT) tile: wave (wp, wq) owns rows 32 (NT wp + a) .., columns 32 (NT wq + b) ...
// BSUM (A "K x rows"): bsum = this thread's share of sum_r A(p0 + threadIdx.x % T, r) (r = its group's rows of every chunk; the
// caller adds the 256 / T groups), the three terms re-added exactly.
template <class SA, class SB, bool BSUM, int NT>
__device__ __forceinline__ void wide_mainloop(const void *a, int64_t lda, int64_t a_pstride, int P, const void *b, int64_t ldb, int64_t b_pstride,
                                              int Q, int p0, int q0, int rbeg, int rend, unsigned short *wlds, f32x16 (&acc)[NT][NT], float &bsum,
                                              const int dbg = 0) {
    typedef typename SA::Plane PA;
    typedef typename SB::Plane PB;
    constexpr int BUF = WideLds<SA, SB>::BUF, NBUF = WideLds<SA, SB>::NBUF, T = 64 * NT;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nchunks = (rend - rbeg + RG_R - 1) / RG_R;
    const int wp = wave & 1, wq = wave >> 1;
#pragma unroll
    for (int x = 0; x < NT; ++x)
#pragma unroll
        for (int y = 0; y < NT; ++y)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[x][y][i] = 0.f;
    SA sa;
    SB sb;
    sa.init(a, lda, a_pstride, p0, P);
    sb.init(b, ldb, b_pstride, q0, Q);
    typename SA::Regs ra[WG_DEPTH];
    typename SB::Regs rb[WG_DEPTH];
    // chunk c -> register set c % 3.  Chunks at or past the end of the slice load zeros (lane offsets beyond the buffer range), so
    // the loop runs whole rounds of three chunks with NO branch inside: every s_waitcnt vmcnt in it is exact.  (With a `break`
    // per step the loop header merged three exits and the first commit of every round drained the whole memory queue -- a full
    // round trip per three chunks.)
    auto issue = [&](auto dc, int c) __attribute__((always_inline)) {
        constexpr int d = decltype(dc)::value;
        const int r0 = rbeg + RG_R * c;
        sa.load(ra[d], r0 < rend ? r0 : rend, rend);
        sb.load(rb[d], r0 < rend ? r0 : rend, rend);
    };
    auto commit = [&](auto dc, int c) __attribute__((always_inline)) {
        constexpr int d = decltype(dc)::value;
        unsigned short *buf = wlds + (NBUF == 2 ? (c & 1) * BUF : 0);
        if (dbg & 2) { sa.keep(ra[d]); sb.keep(rb[d]); return; }
        sa.commit(ra[d], buf);
        sb.commit(rb[d], buf + 3 * PA::PLANE);
    };
    int abase[NT], bbase[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) { abase[i] = PA::lane_base(NT * wp + i); bbase[i] = PB::lane_base(NT * wq + i); }
    // One MFMA tile per wave (NT = 1): two accumulators -- the three small products and the three large ones of a k-step go to
    // different registers (an MFMA into the result of the previous one waits for it), summed once at the end.  Four tiles
    // (NT = 2): product-major over the tiles, consecutive MFMAs never meet.  All operand reads of a k-step are requested before
    // its first MFMA.
    f32x16 acc2;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc2[i] = 0.f;
    auto multiply = [&](int c) __attribute__((always_inline)) {
        const unsigned short *As = wlds + (NBUF == 2 ? (c & 1) * BUF : 0), *Bs = As + 3 * PA::PLANE;
        if (dbg & 1) return;
#pragma unroll
        for (int s = 0; s < RG_R / 16; ++s) {
            rg_bf16x8 a3[NT][3], b3[NT][3];
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int t = 0; t < 3; ++t) { a3[i][t] = PA::operand(As, abase[i], t, s); b3[i][t] = PB::operand(Bs, bbase[i], t, s); }
            if (NT == 1) {
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0][2], b3[0][0], acc2, 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0][1], b3[0][0], acc[0][0], 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0][0], b3[0][2], acc2, 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0][0], b3[0][1], acc[0][0], 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0][1], b3[0][1], acc2, 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[0][0], b3[0][0], acc[0][0], 0, 0, 0);
            } else {
                // (a term, b term) of the six products, smallest first: (l, h) (h, l) (m, m) (m, h) (h, m) (h, h)
#pragma unroll
                for (int prod = 0; prod < 6; ++prod)
#pragma unroll
                    for (int x = 0; x < NT; ++x)
#pragma unroll
                        for (int y = 0; y < NT; ++y) {
                            const int ta = prod == 0 ? 2 : ((prod == 2 || prod == 3) ? 1 : 0);
                            const int tb = prod == 1 ? 2 : ((prod == 2 || prod == 4) ? 1 : 0);
                            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3[x][ta], b3[y][tb], acc[x][y], 0, 0, 0);
                        }
            }
        }
        if (BSUM) {                                              // 256 / T thread groups share a chunk's 32 reduction rows
            constexpr int G = 256 / T, RPG = RG_R / G;
            const int pp = threadIdx.x % T, r_lo = (threadIdx.x / T) * RPG;
#pragma unroll
            for (int r = 0; r < RPG; ++r) {
                const unsigned short *e = As + (r_lo + r) * PA::TRP + pp;
                bsum += (__builtin_bit_cast(float, (unsigned)e[0] << 16) + __builtin_bit_cast(float, (unsigned)e[PA::PLANE] << 16)) +
                        __builtin_bit_cast(float, (unsigned)e[2 * PA::PLANE] << 16);
            }
        }
    };
    using D0 = std::integral_constant<int, 0>;
    using D1 = std::integral_constant<int, 1>;
    using D2 = std::integral_constant<int, 2>;
    issue(D0{}, 0);
    issue(D1{}, 1);
    issue(D2{}, 2);
    commit(D0{}, 0);
    issue(D0{}, 3);
    __syncthreads();
    // chunk c is multiplied from buffer c & 1; meanwhile chunk c + 1 (register set (c + 1) % 3) goes to the other buffer and its set
    // is refilled with chunk c + 4.  (One LDS buffer: a second barrier between the multiply and the commit; the workgroup is then
    // 32 KB and four or five share a CU -- each other's loads, splits, MFMAs and stores overlap across workgroups instead.)
#define ARVAE_WG_STEP(D, DN)                                     \
        multiply(c + D);                                         \
        if (NBUF == 1) __syncthreads();                          \
        commit(DN{}, c + D + 1);                                 \
        issue(DN{}, c + D + 4);                                  \
        __syncthreads();
    for (int c = 0; c < nchunks; c += WG_DEPTH) {
        ARVAE_WG_STEP(0, D1)
        ARVAE_WG_STEP(1, D2)
        ARVAE_WG_STEP(2, D0)
    }
#undef ARVAE_WG_STEP
    if (NT == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[0][0][i] += acc2[i];
    }
}

template <class SA, class SB, int EP, int NT>
__global__ __launch_bounds__(256, NT == 1 ? (WG_NBUF == 1 ? 3 : 2) : 1) void wide_gemm_x3_kernel(WideGemm g) {
    constexpr int BUF = WideLds<SA, SB>::BUF, NBUF = WideLds<SA, SB>::NBUF, T = 64 * NT;
    extern __shared__ __attribute__((aligned(16))) unsigned short wlds[];
    const int lane = threadIdx.x & 63, half = lane >> 5, rc = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p0 = blockIdx.x * T, q0 = blockIdx.y * T;
    const int rbeg = blockIdx.z * g.kslice, rend = min(g.K, rbeg + g.kslice);
    const int wp = wave & 1, wq = wave >> 1;
    f32x16 acc[NT][NT];
    float unused = 0.f;
    wide_mainloop<SA, SB, false, NT>(g.a, g.lda, g.a_pstride, g.M, g.b, g.ldb, g.b_pstride, g.N, p0, q0, rbeg, rend, wlds, acc, unused, g.dbg);
    if (g.dbg & 4) { if (acc[0][0][0] == 123.456f) g.out[0] = acc[0][0][1]; return; }
    float amax = 0.f;
#pragma unroll
    for (int y = 0; y < NT; ++y) {
        const int q = q0 + 32 * (NT * wq + y) + rc;
        const float bias = (EP == WG_EP_FULL && g.bias != nullptr && q < g.N) ? g.bias[q] : 0.f;
#pragma unroll
        for (int x = 0; x < NT; ++x) {
            if (EP == WG_EP_PARTIAL) {
                float *out = g.out + (int64_t)blockIdx.z * g.slice_floats;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = p0 + 32 * (NT * wp + x) + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (p < g.M && q < g.N) out[(int64_t)p * g.ldo + q] = acc[x][y][r];
                }
            } else {
                float gatev[16];
                if (g.gate != nullptr) {                         // (all sixteen requested before the first is used)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int p = p0 + 32 * (NT * wp + x) + (r & 3) + 8 * (r >> 2) + 4 * half;
                        const bool ok = p < g.M && q < g.N;
                        gatev[r] = g.gate[(int64_t)(ok ? p : 0) * g.ldo + (ok ? q : 0)];
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = p0 + 32 * (NT * wp + x) + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const bool ok = p < g.M && q < g.N;
                    float v = act_fwd(acc[x][y][r] + bias, g.act);
                    if (g.gate != nullptr) v = gatev[r] > 0.f ? v : 0.f;
                    if (ok) { g.out[(int64_t)p * g.ldo + q] = v; amax = fmaxf(amax, fabsf(v)); }
                }
            }
        }
    }
    if (EP == WG_EP_FULL && g.amax_out != nullptr) {             // one AMAX writer unit per workgroup (conv32_common.h)
        float *red = reinterpret_cast<float *>(wlds + NBUF * BUF);
        amax = wave_max(amax);
        __syncthreads();
        if (lane == 0) red[wave] = amax;
        __syncthreads();
        const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (wave == 0) {
            const int unit = (blockIdx.y * gridDim.x + blockIdx.x), units = gridDim.x * gridDim.y;
            amax_publish(g.amax_out, unit, units, m);
        }
    }
}

// The wide layers' WEIGHT gradients on the same main loop:  dW'[p][q] = sum over the batch of A(p, m) B(q, m), both operands
// "K x rows" (activations / gradients are [batch][features]: the batch is the reduction axis), one of them arriving as planes (the
// small one: the latent block's kernels wrote it pre-split), the other fp32.  A tile adds into dW at the permuted place of its
// rows / columns (the gradient arena holds the reference's [n][k] layout in feature order) and, for its first column tile, the
// bias gradient = A's row sums.  Replaces the two largest jobs of the grouped 32 x 32-tile launch (dense_wgrad_batch_kernel:
// ~1450 of its tiles, ~50 of its 58 us at B = 1024).
template <class SA, class SB>
__device__ __forceinline__ void wide_wgrad_tile(const WideWgradJob &j, int tile, unsigned short *wlds) {
    constexpr int BUF = WideLds<SA, SB>::BUF, NBUF = WideLds<SA, SB>::NBUF;
    const int tp = (j.P + 63) / 64;
    const int bx = tile % tp, by = tile / tp;
    const int lane = threadIdx.x & 63, half = lane >> 5, rc = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wp = wave & 1, wq = wave >> 1;
    const int p0 = bx * 64, q0 = by * 64;
    const int q = q0 + 32 * wq + rc;
    const bool qok = q < j.Q;
    const int qf = j.q_perm.to_feat(qok ? q : 0);
    // this lane's 16 elements of dW: requested now, added to after the reduction
    float *outp[16];
    float old[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int p = p0 + 32 * wp + (r & 3) + 8 * (r >> 2) + 4 * half;
        const bool ok = qok && p < j.P;
        outp[r] = j.dw + (ok ? (int64_t)j.p_perm.to_feat(p) * j.ldw + qf : 0);
        old[r] = *outp[r];
    }
    f32x16 acc[1][1];
    float bsum = 0.f;
    if (by == 0) wide_mainloop<SA, SB, true, 1>(j.a, j.lda, j.a_pstride, j.P, j.b, j.ldb, j.b_pstride, j.Q, p0, q0, 0, j.R, wlds, acc, bsum);
    else wide_mainloop<SA, SB, false, 1>(j.a, j.lda, j.a_pstride, j.P, j.b, j.ldb, j.b_pstride, j.Q, p0, q0, 0, j.R, wlds, acc, bsum);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int p = p0 + 32 * wp + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (qok && p < j.P) *outp[r] = old[r] + acc[0][0][r];
    }
    if (by == 0 && j.dbias != nullptr) {                         // the four thread groups' shares of the row sums, in order
        float *red = reinterpret_cast<float *>(wlds + NBUF * BUF);
        red[threadIdx.x] = bsum;
        __syncthreads();
        if (threadIdx.x < 64 && p0 + (int)threadIdx.x < j.P) {
            float *db = j.dbias + j.p_perm.to_feat(p0 + threadIdx.x);
            *db += (red[threadIdx.x] + red[threadIdx.x + 64]) + (red[threadIdx.x + 128] + red[threadIdx.x + 192]);
        }
    }
}
__global__ __launch_bounds__(256, WG_NBUF == 1 ? 3 : 2) void wide_wgrad_x3_kernel(WideWgradBatch b) {
    extern __shared__ __attribute__((aligned(16))) unsigned short wlds[];
    typedef WideF32<RG_KROWS, 64> F;
    typedef WidePlanes<RG_KROWS, 64> PL;
    const int jn = (b.count > 1 && (int)blockIdx.x >= b.wg_end[0]) ? 1 : 0;
    const int tile = blockIdx.x - (jn ? b.wg_end[0] : 0);
    if (jn == 0) {
        if (b.job[0].a_planes) wide_wgrad_tile<PL, F>(b.job[0], tile, wlds);
        else wide_wgrad_tile<F, PL>(b.job[0], tile, wlds);
    } else {
        if (b.job[1].a_planes) wide_wgrad_tile<PL, F>(b.job[1], tile, wlds);
        else wide_wgrad_tile<F, PL>(b.job[1], tile, wlds);
    }
}
bool wide_wgrad_fits(const WideWgradJob &j) {
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    // (the planes: tiled layout, ld = their row count = the batch, reduced along the rows)
    const bool planes_ok = j.a_planes ? (j.lda >= j.R && (j.P & 7) == 0 && (j.a_pstride & 7) == 0 && (j.ldb & 3) == 0 && (j.Q & 3) == 0)
                                      : (j.ldb >= j.R && (j.Q & 7) == 0 && (j.b_pstride & 7) == 0 && (j.lda & 3) == 0 && (j.P & 3) == 0);
    return j.P > 0 && j.Q > 0 && j.R > 0 && (j.a_planes != 0) != (j.b_planes != 0) && planes_ok && al(j.a) && al(j.b) &&
           (int64_t)j.R * (j.a_planes ? j.ldb : j.lda) * 4 < ((int64_t)1 << 31) && (3 * (j.a_planes ? j.a_pstride : j.b_pstride)) * 2 < ((int64_t)1 << 31);
}
int wide_wgrad(const WideWgradJob *jobs, int count, hipStream_t s) {
    ARVAE_REQUIRE(count >= 1 && count <= 2, "wide_wgrad: one or two jobs");
    WideWgradBatch b{};
    b.count = count;
    int total = 0;
    for (int i = 0; i < count; ++i) {
        ARVAE_REQUIRE(wide_wgrad_fits(jobs[i]), "wide_wgrad: operand shapes / alignment");
        b.job[i] = jobs[i];
        total += ((jobs[i].P + 63) / 64) * ((jobs[i].Q + 63) / 64);
        b.wg_end[i] = total;
    }
    typedef WideLds<WideF32<RG_KROWS, 64>, WidePlanes<RG_KROWS, 64>> Lds;
    constexpr size_t lds_bytes = Lds::BYTES;
    static std::once_flag once;
    std::call_once(once, [] { (void)hipFuncSetAttribute((const void *)wide_wgrad_x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Lds::BYTES); });
    ARVAE_LAUNCH(wide_wgrad_x3_kernel, dim3(total), dim3(256), lds_bytes, s, b);
    return check_launch("wide_wgrad");
}

// tools/probes/wide_gemm.py (diagnostic build): one launch of the tile GEMM on caller buffers (contents are the caller's business: timing only)
#ifdef ARVAE_DIAG
extern "C" int arvae_debug_wide_gemm(int32_t a_planes, int32_t b_krows, int32_t partial, int32_t M, int32_t N, int32_t K, const void *a, const void *b,
                                     float *out, int32_t slices, int32_t dbg, arvae_stream_t stream) {
    const int kpad = (K + RG_R - 1) / RG_R * RG_R, npad = (N + 31) / 32 * 32;
    WideGemm g{};
    g.a = a; g.lda = a_planes ? M : K; g.a_pstride = (int64_t)M * kpad; g.a_planes = a_planes;
    g.b = b; g.b_planes = 1; g.b_krows = b_krows;
    g.ldb = b_krows ? kpad : npad; g.b_pstride = (int64_t)npad * kpad;      // (planes: ld = their row count; K x rows: the rows are the reduction)
    g.M = M; g.N = N; g.K = K; g.out = out; g.ldo = N; g.slice_floats = (int64_t)M * N; g.act = ARVAE_ACT_NONE; g.dbg = dbg;
    return wide_gemm(g, slices, partial != 0, as_stream(stream));
}
#endif

// The tile extent a product runs with.  64 x 64, four or five 32 KB workgroups per CU.  The 128 x 128 form (half the operand
// traffic per multiply-add, one 120 KB workgroup per CU) is kept in the diagnostic build for A/B runs: with every MFMA, LDS write
// and result store switched off it moves its bytes in 9.3 us against 12.3 (tools/probes/wide_gemm.py), but one wave per SIMD then
// runs loads, commits, MFMAs and stores one after the other: 31-34 us per product against 20-21.
static int wide_tile(int M, int N) {
#ifdef ARVAE_DIAG
    static const bool big = diag_env("ARVAE_WIDE_TILE128") != nullptr;      // diagnostic build only: the 128 x 128 form
    return (big && M > 64 && N > 64) ? 128 : 64;
#else
    (void)M; (void)N;
    return 64;
#endif
}
// K slices that fill the chip: tiles x slices ~ one (128) or two (64) workgroups per CU, a slice at least four chunks long
int wide_gemm_slices(int M, int N, int K) {
    const int T = wide_tile(M, N);
    const int tiles = ((M + T - 1) / T) * ((N + T - 1) / T), chunks = (K + RG_R - 1) / RG_R;
    int s = ((T == 128 ? 1 : 2) * device_cu_count() + tiles - 1) / tiles;
    if (s > chunks / 4) s = chunks / 4;
    if (s > WIDE_MAX_SLICES) s = WIDE_MAX_SLICES;
    return s < 1 ? 1 : s;
}
bool wide_gemm_fits(const WideGemm &g, int slices) {
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const int64_t a_bytes = g.a_planes ? 3 * g.a_pstride * 2 : (int64_t)g.M * g.lda * 4;
    const int64_t b_bytes = 3 * g.b_pstride * 2;
    const int T = wide_tile(g.M, g.N);
    const int tiles = ((g.M + T - 1) / T) * ((g.N + T - 1) / T);
    // fp32 A: 16-byte rows.  Plane operands (tiled layout, ld = their row count, K padded to whole chunks by their writer): "rows x K"
    // needs at least the output rows; "K x rows" reduces along the rows (padded with zero rows for weights) and emits 8 columns per load
    const bool a_ok = g.a_planes ? (g.lda >= g.M && (g.K % RG_R) == 0 && (g.a_pstride & 7) == 0) : ((g.lda & 3) == 0 && (g.K & 3) == 0);
    const bool b_ok = (g.b_pstride & 7) == 0 && (g.b_krows ? (g.N & 7) == 0 && g.ldb >= g.K : g.ldb >= g.N);
    return g.M > 0 && g.N > 0 && g.K > 0 && a_ok && b_ok && g.b_planes && (g.N & 3) == 0 && al(g.a) && al(g.b) &&
           a_bytes < ((int64_t)1 << 31) && b_bytes < ((int64_t)1 << 31) && slices >= 1 && slices <= WIDE_MAX_SLICES &&
           (g.amax_out == nullptr || tiles <= AMAX_N);
}
template <class SA, class SB, int EP, int NT>
static void launch_wide(const WideGemm &g, dim3 grid, hipStream_t s) {
    static std::once_flag once;
    std::call_once(once, [] {
        (void)hipFuncSetAttribute((const void *)wide_gemm_x3_kernel<SA, SB, EP, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WideLds<SA, SB>::BYTES);
    });
    constexpr size_t lds_bytes = WideLds<SA, SB>::BYTES;
    ARVAE_LAUNCH((wide_gemm_x3_kernel<SA, SB, EP, NT>), grid, dim3(256), lds_bytes, s, g);
}
template <int NT>
static void dispatch_wide(const WideGemm &g, bool partial, dim3 grid, hipStream_t s) {
    constexpr int T = 64 * NT;
    typedef WideF32<RG_ROWSK, T> AF;
    typedef WidePlanes<RG_ROWSK, T> AP;
    typedef WidePlanes<RG_ROWSK, T> BR;
    typedef WidePlanes<RG_KROWS, T> BK;
    const int sel = (g.a_planes ? 4 : 0) + (g.b_krows ? 2 : 0) + (partial ? 0 : 1);
    switch (sel) {
        case 0: launch_wide<AF, BR, WG_EP_PARTIAL, NT>(g, grid, s); break;
        case 1: launch_wide<AF, BR, WG_EP_FULL, NT>(g, grid, s); break;
        case 2: launch_wide<AF, BK, WG_EP_PARTIAL, NT>(g, grid, s); break;
        case 3: launch_wide<AF, BK, WG_EP_FULL, NT>(g, grid, s); break;
        case 4: launch_wide<AP, BR, WG_EP_PARTIAL, NT>(g, grid, s); break;
        case 5: launch_wide<AP, BR, WG_EP_FULL, NT>(g, grid, s); break;
        case 6: launch_wide<AP, BK, WG_EP_PARTIAL, NT>(g, grid, s); break;
        default: launch_wide<AP, BK, WG_EP_FULL, NT>(g, grid, s); break;
    }
}
// slices > 1 (or partial): g.out = workspace of `slices` x slice_floats, the consumer sums them; else the finished product
int wide_gemm(WideGemm g, int slices, bool partial, hipStream_t s) {
    ARVAE_REQUIRE(wide_gemm_fits(g, slices), "wide_gemm: operand shapes / alignment");
    ARVAE_REQUIRE(partial || slices == 1, "wide_gemm: a split reduction leaves partial sums");
    const int chunks = (g.K + RG_R - 1) / RG_R;
    g.kslice = ((chunks + slices - 1) / slices) * RG_R;
    const int T = wide_tile(g.M, g.N);
    const dim3 grid((g.M + T - 1) / T, (g.N + T - 1) / T, slices);
#ifdef ARVAE_DIAG
    if (T == 128) dispatch_wide<2>(g, partial, grid, s);
    else
#endif
        dispatch_wide<1>(g, partial, grid, s);
    return check_launch(partial ? "wide_gemm(partial)" : "wide_gemm(full)");
}

// The weight-gradient product on 128 x 128 output tiles (2 x 2 waves of 2 x 2 MFMA tiles each): four times the multiply-adds per
// staged operand byte and per barrier of the 64 x 64 body above.  Stamped, a 64 x 64 workgroup spent 2.45 us per 32-row chunk
// (0.8 in its two barriers, 0.3 waiting for loads, 0.45 splitting) on 0.16 us of MFMA issue, and the batched launch moved 405 MB
// through L2 for 113 MB of operands.  "K x rows" operands, slice epilogue, 16-byte loads only.
__device__ __forceinline__ void rows_wgrad128_body(const RowsGemm &g, const int bx, const int by, const int bz) {
    typedef TileLoader<RG_KROWS, 128, true> Load;
    __shared__ __attribute__((aligned(16))) unsigned short As[3 * Load::PLANE];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[3 * Load::PLANE];
    const int lane = threadIdx.x & 63, half = lane >> 5, rc = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p0 = bx * 128, q0 = by * 128;
    const int rbeg = bz * g.rslice, rend = min(g.Rn, rbeg + g.rslice);
    Load la, lb;
    const int wp = wave & 1, wq = wave >> 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    float bsum = 0.f;
    la.load(g.a, g.lda, p0, g.P, rbeg, rend);
    lb.load(g.b, g.ldb, q0, g.Q, rbeg, rend);
    int abase[2], bbase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { abase[i] = Load::lane_base(2 * wp + i); bbase[i] = Load::lane_base(2 * wq + i); }
    for (int r0 = rbeg; r0 < rend; r0 += RG_R) {
        __syncthreads();
        la.commit3(As);
        lb.commit3(Bs);
        __syncthreads();
        if (r0 + RG_R < rend) {
            la.load(g.a, g.lda, p0, g.P, r0 + RG_R, rend);
            lb.load(g.b, g.ldb, q0, g.Q, r0 + RG_R, rend);
        }
#pragma unroll
        for (int s = 0; s < RG_R / 16; ++s) {
            rg_bf16x8 ah[2], am[2], al[2], bh[2], bm[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = Load::operand(As, abase[i], 0, s); am[i] = Load::operand(As, abase[i], 1, s); al[i] = Load::operand(As, abase[i], 2, s);
                bh[i] = Load::operand(Bs, bbase[i], 0, s); bm[i] = Load::operand(Bs, bbase[i], 1, s); bl[i] = Load::operand(Bs, bbase[i], 2, s);
            }
            // product-major over the four accumulators: consecutive MFMAs never wait for each other's result
#pragma unroll
            for (int prod = 0; prod < 6; ++prod)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const rg_bf16x8 x = prod == 0 ? al[a] : ((prod == 2 || prod == 3) ? am[a] : ah[a]);
                        const rg_bf16x8 y = prod == 1 ? bl[b] : ((prod == 2 || prod == 4) ? bm[b] : bh[b]);
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a][b], 0, 0, 0);
                    }
        }
        if (g.want_bias && by == 0 && threadIdx.x < 128) {
#pragma unroll 8
            for (int r = 0; r < RG_R; ++r) {
                const unsigned short *e = As + r * Load::Plane::TRP + threadIdx.x;
                bsum += (__builtin_bit_cast(float, (unsigned)e[0] << 16) + __builtin_bit_cast(float, (unsigned)e[Load::PLANE] << 16)) +
                        __builtin_bit_cast(float, (unsigned)e[2 * Load::PLANE] << 16);
            }
        }
    }
    float *out = g.out + bz * g.slice_floats;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int q = q0 + 64 * wq + 32 * b + rc;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = p0 + 64 * wp + 32 * a + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (p < g.P && q < g.Q) out[(int64_t)p * g.ldo + q] = acc[a][b][r];
            }
        }
    if (g.want_bias && by == 0 && threadIdx.x < 128 && p0 + (int)threadIdx.x < g.P) out[(int64_t)g.P * g.ldo + p0 + threadIdx.x] = bsum;
}

// Several long-batch weight gradients (RG_KROWS x RG_KROWS operands, slice epilogue) in ONE launch: the whole-sequence layers
// of a backward pass each leave a 0.6-2.4 GFLOP product that nothing on the pass's critical path waits for, and alone each
// is a 15-35 us launch that fills a fraction of the chip; together (workgroup -> (job, tile, slice) through the running
// workgroup count) they are one full-chip launch and one reduction.
constexpr int RG_BATCH_MAX = 12;
struct RowsGemmBatch {
    int count;
    int wg_end[RG_BATCH_MAX];
    RowsGemm job[RG_BATCH_MAX];
};
__global__ __launch_bounds__(256) void rows_wgrad_batch_kernel(RowsGemmBatch b) {
    // consecutive workgroup ids go round the 8 XCDs; the tiles of one row slice read the same operand rows, so logical
    // workgroup w runs as physical id (w % per_xcd) * 8 + w / per_xcd: a slice's tiles share an XCD's L2 (grid: a multiple of 8)
    const int per_xcd = gridDim.x >> 3;
    const int wg = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (wg >= b.wg_end[b.count - 1]) return;
    int j = 0;
    while (j + 1 < b.count && wg >= b.wg_end[j]) ++j;
    const int local = wg - (j > 0 ? b.wg_end[j - 1] : 0);
    const RowsGemm &g = b.job[j];
    const int tp = (g.P + 127) / 128, tq = (g.Q + 127) / 128;
    rows_wgrad128_body(g, local % tp, (local / tp) % tq, local / (tp * tq));
}

// ---- wgrad: dW[n][feat_in(km)] += sum_m G[m][mem_out(n)] * X[m][km] ;  db[n] += sum_m G[m][mem_out(n)] ------
// KU chunks are fetched per wave before the first MFMA, with unconditional loads (clamped row, zeroed by a select): the
// operands were written a whole pass ago and come from HBM with a row pitch of 1-2 KB, so a tile lives on memory
// latency (a single tile takes ~16 us cold, 9 us when its input was just read: measured per job).

#ifdef DW_STAMPS
// diagnostic build only (tools/stamp_dw.py): phase timeline of the first 512 weight-gradient tiles, 100 MHz wall clock
__device__ unsigned long long g_dw_stamps[512 * 8];
#define DW_STAMP(slot) do { if (threadIdx.x == 0 && blockIdx.x < 512) g_dw_stamps[blockIdx.x * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define DW_STAMP(slot)
#endif

template <int MODE, int NWT>
__device__ __forceinline__ void dense_wgrad_loop(const DenseArgs &p, int ncol, int kcol, bool nok, bool kok, int wave, int half,
                                                 f32x16 &acc, float &bsum) {
    constexpr int KU = 4;                                        // 8 did not help (measured): the tile is not load-count bound
    const int chunks = (p.batch + 7) / 8;
    for (int q0 = wave; q0 < chunks; q0 += KU * NWT) {
        float a[KU][4], ay[KU][4], am[KU][4], b[KU][4];
        // loads only (Operand::fetch): with Operand::at() here every element was its own memory round trip
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int m0 = (q0 + u * NWT) * 8 + half * 4;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int mc = m0 + t < p.batch ? m0 + t : 0;
                p.a.template fetch<MODE>((int64_t)mc * p.n_out + ncol, a[u][t], ay[u][t], am[u][t]);
                b[u][t] = p.x[(int64_t)mc * p.n_in + kcol];
            }
        }
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int m0 = (q0 + u * NWT) * 8 + half * 4;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bool ok = m0 + t < p.batch;
                const float g = p.a.template apply<MODE>(a[u][t], ay[u][t], am[u][t]);
                a[u][t] = (ok && nok) ? g : 0.f;
                b[u][t] = (ok && kok) ? b[u][t] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < KU; ++u) {
#pragma unroll
            for (int t = 0; t < 4; ++t) bsum += a[u][t];
            mfma4(acc, a[u], b[u]);
        }
#ifdef DW_STAMPS
        if (q0 == wave) { __builtin_amdgcn_s_waitcnt(0); DW_STAMP(2); }
#endif
    }
}

// NWT waves split the batch (K) axis of one 32x32 tile of dW; wave w finishes accumulator registers w*16/NWT ...
template <int NWT>
__device__ __forceinline__ void dense_wgrad_tile(const DenseArgs &p, int bx, int by, float *red) {
    constexpr int RPW = 16 / NWT;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, rc = lane & 31;
    // a tile owns 32 consecutive MEMORY columns of G (coalesced loads in the reduction loop); with a channel permutation
    // (out_perm) its rows of dW are scattered instead, which only moves the 16 stores per lane at the end: the permuted
    // layer's tile took 19 us with feature-ordered lanes (64 cache lines per load instruction) against 8 us for the others
    const int nm = bx * 32 + rc, km = by * 32 + rc;
    const bool nok = nm < p.n_out, kok = km < p.n_in;
    const int ncol = nok ? nm : 0, kcol = kok ? km : 0;
    const int n = p.out_perm.to_feat(ncol);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float bsum = 0.f;
    DW_STAMP(0);
    // this lane's dW elements: read early, added to after the reduction
    float *outp[RPW];
    float old[RPW];
    bool out_ok[RPW];
#pragma unroll
    for (int e = 0; e < RPW; ++e) {
        const int reg = wave * RPW + e;
        const int out_mem = bx * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;      // memory column of G = lane rc of this row
        out_ok[e] = kok && out_mem < p.n_out;
        outp[e] = p.out + (out_ok[e] ? (int64_t)p.out_perm.to_feat(out_mem) * p.n_in + p.in_perm.to_feat(km) : 0);
        old[e] = p.store ? 0.f : *outp[e];
    }
    if (p.a.mode() == 0) dense_wgrad_loop<0, NWT>(p, ncol, kcol, nok, kok, wave, half, acc, bsum);
    else if (p.a.mode() == 1) dense_wgrad_loop<1, NWT>(p, ncol, kcol, nok, kok, wave, half, acc, bsum);
    else dense_wgrad_loop<2, NWT>(p, ncol, kcol, nok, kok, wave, half, acc, bsum);
    DW_STAMP(3);
    __syncthreads();
    DW_STAMP(4);
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    DW_STAMP(5);
#pragma unroll
    for (int e = 0; e < RPW; ++e) {
        float v = 0.f;
#pragma unroll
        for (int ws = 0; ws < NWT; ++ws) v += red[(ws * 16 + wave * RPW + e) * 64 + lane];
        if (out_ok[e]) *outp[e] = old[e] + v;
    }
    if (p.dbias != nullptr && by == 0) {
        __syncthreads();
        red[wave * 64 + lane] = bsum;
        __syncthreads();
        if (wave == 0 && half == 0 && nok) {
            float tot = 0.f;
#pragma unroll
            for (int ws = 0; ws < NWT; ++ws) tot += red[ws * 64 + rc] + red[ws * 64 + rc + 32];
            p.dbias[n] = (p.store ? 0.f : p.dbias[n]) + tot;
        }
    }
#ifdef DW_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
    DW_STAMP(6);
#endif
}

__global__ __launch_bounds__(DENSE_THREADS) void dense_wgrad_kernel(DenseArgs p) {
    __shared__ float red[NW * 16 * 64];
    dense_wgrad_tile<NW>(p, blockIdx.x, blockIdx.y, red);
}

// Adds the row slices written by the long-batch weight gradient (rows_gemm_kernel, RG_EP_SLICE) to dW / db in slice order.
__global__ __launch_bounds__(256) void dense_split_reduce_kernel(const float *__restrict__ ws, int64_t slice_floats, int slices,
                                                                  int64_t w_floats, int n_out, float *__restrict__ dw,
                                                                  float *__restrict__ dbias) {
    // 4 lanes per output element, lane q sums slices q, q+4, ... (8 loads in flight), then a fixed-order lane sum
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 2;
    const int q = threadIdx.x & 3;
    const int64_t total = w_floats + (dbias != nullptr ? n_out : 0);
    const int64_t ic = i < total ? i : 0;
    float s = 0.f;
    for (int z0 = q; z0 < slices; z0 += 32) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int z = z0 + 4 * u;
            const float v = ws[(int64_t)(z < slices ? z : 0) * slice_floats + ic];
            t[u] = z < slices ? v : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += t[u];
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (q != 0 || i >= total) return;
    if (i < w_floats) dw[i] += s;
    else dbias[i - w_floats] += s;
}

// the same for every job of a rows_wgrad_batch_kernel launch
struct SplitReduceJob {
    const float *ws;
    int64_t slice_floats, w_floats;
    int slices, n_out;
    float *dw, *dbias;
};
struct SplitReduceBatch {
    int count;
    int wg_end[RG_BATCH_MAX];
    SplitReduceJob job[RG_BATCH_MAX];
};
__global__ __launch_bounds__(256) void dense_split_reduce_batch_kernel(SplitReduceBatch b) {
    int j = 0;
    while (j + 1 < b.count && (int)blockIdx.x >= b.wg_end[j]) ++j;
    const int local = blockIdx.x - (j > 0 ? b.wg_end[j - 1] : 0);
    const SplitReduceJob &r = b.job[j];
    const int64_t i = ((int64_t)local * 256 + threadIdx.x) >> 2;
    const int q = threadIdx.x & 3;
    const int64_t total = r.w_floats + (r.dbias != nullptr ? r.n_out : 0);
    const int64_t ic = i < total ? i : 0;
    float s = 0.f;
    for (int z0 = q; z0 < r.slices; z0 += 32) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int z = z0 + 4 * u;
            const float v = r.ws[(int64_t)(z < r.slices ? z : 0) * r.slice_floats + ic];
            t[u] = z < r.slices ? v : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += t[u];
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (q != 0 || i >= total) return;
    if (i < r.w_floats) r.dw[i] += s;
    else r.dbias[i - r.w_floats] += s;
}

// Weight gradients of several Linear layers in ONE launch: they are independent once every layer's output gradient
// exists, and each alone is a launch-latency-bound 8..128-tile problem.  Workgroup -> (job, tile) through the
// running tile count.  8 waves per tile here (32 KB of LDS): the ~400 tiles of the dSprites stack are then all
// resident at once instead of in two rounds of 1024-thread workgroups.
constexpr int NWB = 8;
__global__ __launch_bounds__(64 * NWB, 2) void dense_wgrad_batch_kernel(DenseWgradBatch b) {
    __shared__ float red[NWB * 16 * 64];
    int j = 0;
    while (j + 1 < b.count && (int)blockIdx.x >= b.tile_end[j]) ++j;
    const int tile = blockIdx.x - (j > 0 ? b.tile_end[j - 1] : 0);
    const DenseArgs &p = b.job[j];
    const int tx = (p.n_out + 31) / 32;
    dense_wgrad_tile<NWB>(p, tile % tx, tile / tx, red);
}

// The grouped Linear weight gradients (latency-bound in L2, ~400 tiles of 14 us) beside the single-channel first layer's weight
// gradient (wgrad_c1s.h: 256 workgroups streaming 75 MB from HBM, 17 us): the two close the backward pass and need nothing
// from each other; both are 8-wave workgroups of ~100 registers, so a CU holds one of each.  Workgroups [0, grid_c1) run the
// streaming body (dispatched first: it is the longer one), the rest one Linear tile each.
__global__ __launch_bounds__(64 * NWB, 2) void dense_wgrad_c1_kernel(DenseWgradBatch b, Operand c1_lo, Operand c1_img, float *__restrict__ c1_slab,
                                                                      int c1_rows, int grid_c1) {
    __shared__ __attribute__((aligned(16))) float raw[WGS_WAVES * LO1 * WS1 + WGS_WAVES * 4 * IMS];
    static_assert(WGS_WAVES == NWB && WGS_WAVES * LO1 * WS1 >= NWB * 16 * 64, "dense_wgrad_c1_kernel: one workgroup shape, one LDS block");
    if ((int)blockIdx.x < grid_c1) {
        wgrad_c1s_body(c1_lo, c1_img, c1_slab, c1_rows, blockIdx.x, grid_c1, raw, raw + WGS_WAVES * LO1 * WS1);
        return;
    }
    const int bid = blockIdx.x - grid_c1;
    int j = 0;
    while (j + 1 < b.count && bid >= b.tile_end[j]) ++j;
    const int tile = bid - (j > 0 ? b.tile_end[j - 1] : 0);
    const DenseArgs &p = b.job[j];
    const int tx = (p.n_out + 31) / 32;
    dense_wgrad_tile<NWB>(p, tile % tx, tile / tx, raw);
}

// The two closing launches of the image VAE's backward pass as ONE grid (round 6): the grouped Linear weight gradients (~400 tiles,
// each two dependent round trips to L2: latency) and the fixed-order sum of the conv layers' weight-gradient slabs (~55 MB
// streamed from HBM, ~1500 workgroups).  They write disjoint parts of the gradient arena.  Tiles first: they are dispatched one or
// two to a CU and leave the memory system idle, the reducers fill the CUs' other slots (both are 512-thread workgroups under 128
// registers) and stream beside them.  (Round 2 tried this with the roles interleaved and with the reducers first: 40.8 us and
// the sum of the two, against 31.2 back to back -- the tile kernel then ran one 1024-thread workgroup per CU.)
__global__ __launch_bounds__(64 * NWB, 2) void dense_wgrad_slab_kernel(DenseWgradBatch b, SlabReduceBatch r, int n_tiles) {
    __shared__ __attribute__((aligned(16))) float raw[NWB * 16 * 64];
    static_assert(64 * RED_Z == 64 * NWB && RED_Z * RED_OUT <= NWB * 16 * 64, "dense_wgrad_slab_kernel: one workgroup shape, one LDS block");
    if ((int)blockIdx.x < n_tiles) {
        int j = 0;
        while (j + 1 < b.count && (int)blockIdx.x >= b.tile_end[j]) ++j;
        const int tile = blockIdx.x - (j > 0 ? b.tile_end[j - 1] : 0);
        const DenseArgs &p = b.job[j];
        const int tx = (p.n_out + 31) / 32;
        dense_wgrad_tile<NWB>(p, tile % tx, tile / tx, raw);
        return;
    }
    const int bid = blockIdx.x - n_tiles;
    float (*red)[RED_OUT] = reinterpret_cast<float (*)[RED_OUT]>(raw);
    int j = 0, start = 0;
#pragma unroll
    for (int q = 0; q + 1 < SLAB_BATCH_MAX; ++q)
        if (q + 1 < r.count && bid >= r.block_end[q]) { j = q + 1; start = r.block_end[q]; }
    switch (j) {                         // constant indices into the by-value argument block
#define ARVAE_JOB(J) case J: slab_reduce_block(r.job[J], bid - start, red); break;
        ARVAE_JOB(0) ARVAE_JOB(1) ARVAE_JOB(2) ARVAE_JOB(3) ARVAE_JOB(4) ARVAE_JOB(5) ARVAE_JOB(6) ARVAE_JOB(7)
#undef ARVAE_JOB
        default: break;
    }
}

// both queues in one launch when both hold work; either alone otherwise (ARVAE_NO_PAIR_CLOSE in the diagnostic build: back to back)
int dense_wgrad_slab_flush(DenseWgradBatch *b, SlabReduceBatch *r, hipStream_t s) {
    static const bool off = diag_env("ARVAE_NO_PAIR_CLOSE") != nullptr || diag_env("ARVAE_DENSE_BATCH_SPLIT") != nullptr;
    if (off || b->count == 0 || r->count == 0) {
        if (int rc = slab_reduce_flush(r, s)) return rc;
        return dense_wgrad_flush(b, s);
    }
    const int n_tiles = b->tile_end[b->count - 1];
    ARVAE_LAUNCH(dense_wgrad_slab_kernel, dim3(n_tiles + r->block_end[r->count - 1]), dim3(64 * NWB), 0, s, *b, *r, n_tiles);
    b->count = 0;
    r->count = 0;
    return check_launch("pair(dense_wgrad_batch + slab_reduce_batch)");
}

bool dense_fits(const arvae_link_t *l) {
    return l->hh == 1 && l->hw == 1 && l->lh == 1 && l->lw == 1 && l->kh == 1 && l->kw == 1;
}

template <int LA, int LB, int EP>
static void launch_rows_gemm(const RowsGemm &g, int slices, hipStream_t s) {
    // 16-byte loads need the contiguous axis (the reduction for "rows x K", the output index for "K x rows"), the
    // leading dimension and the base address in multiples of 4 floats
    auto vec_ok = [&](const float *base, int64_t ld, int lay, int out_extent) {
        const int contiguous = lay == RG_ROWSK ? g.Rn : out_extent;
        const bool slice_ok = lay != RG_ROWSK || (g.rslice & 3) == 0;
        // (and the matrix within 2 GB: the 16-byte form addresses it with 32-bit byte offsets)
        const int64_t rows = lay == RG_ROWSK ? out_extent : g.Rn;
        return (ld & 3) == 0 && (contiguous & 3) == 0 && slice_ok && (reinterpret_cast<uintptr_t>(base) & 15) == 0 &&
               rows * ld * 4 < ((int64_t)1 << 31);
    };
    const bool va = vec_ok(g.a, g.lda, LA, g.P), vb = vec_ok(g.b, g.ldb, LB, g.Q);
    const dim3 grid((g.P + RG_TP - 1) / RG_TP, (g.Q + RG_TQ - 1) / RG_TQ, slices);
    static const bool fp32_mfma = diag_env("ARVAE_ROWS_GEMM_FP32") != nullptr;
    if (fp32_mfma) {
        if (va && vb) ARVAE_LAUNCH((rows_gemm_kernel<LA, LB, EP, true, true>), grid, dim3(256), 0, s, g);
        else if (va) ARVAE_LAUNCH((rows_gemm_kernel<LA, LB, EP, true, false>), grid, dim3(256), 0, s, g);
        else if (vb) ARVAE_LAUNCH((rows_gemm_kernel<LA, LB, EP, false, true>), grid, dim3(256), 0, s, g);
        else ARVAE_LAUNCH((rows_gemm_kernel<LA, LB, EP, false, false>), grid, dim3(256), 0, s, g);
        return;
    }
    if (va && vb) ARVAE_LAUNCH((rows_gemm_x3_kernel<LA, LB, EP, true, true>), grid, dim3(256), 0, s, g);
    else if (va) ARVAE_LAUNCH((rows_gemm_x3_kernel<LA, LB, EP, true, false>), grid, dim3(256), 0, s, g);
    else if (vb) ARVAE_LAUNCH((rows_gemm_x3_kernel<LA, LB, EP, false, true>), grid, dim3(256), 0, s, g);
    else ARVAE_LAUNCH((rows_gemm_x3_kernel<LA, LB, EP, false, false>), grid, dim3(256), 0, s, g);
}

// the operand needs no per-element work (no activation derivative, no keep-mask)
static bool plain_operand(const Operand &g) { return g.y == nullptr || (g.act == ARVAE_ACT_NONE && g.mask == nullptr); }

static bool dense_long_batch(const DenseArgs &p) {
    static const bool off = diag_env("ARVAE_DENSE_NO_ROWS") != nullptr;       // A/B switch
    return !off && p.batch >= DENSE_SPLIT_MIN_ROWS && p.in_perm.c_count == 0 && p.out_perm.c_count == 0;
}

static DenseArgs dense_args(const arvae_link_t *l) {
    DenseArgs p{};
    p.batch = l->n;
    p.n_in = l->chi;
    p.n_out = l->clo;
    p.in_perm = Perm{l->hi_perm_c, l->hi_perm_hw};
    p.out_perm = Perm{l->lo_perm_c, l->lo_perm_hw};
    return p;
}

int dense_fwd(const arvae_link_t *l, const float *x, const float *w, const float *bias, int act, float *y, hipStream_t s) {
    DenseArgs p = dense_args(l);
    p.a = Operand{x, nullptr, nullptr, ARVAE_ACT_NONE};
    p.w = w; p.bias = bias; p.act = act; p.out = y;
    if (dense_long_batch(p)) {
        RowsGemm g{};
        g.a = x; g.lda = p.n_in; g.b = w; g.ldb = p.n_in; g.P = p.batch; g.Q = p.n_out; g.Rn = p.n_in; g.rslice = p.n_in;
        g.bias = bias; g.act = act; g.out = y; g.ldo = p.n_out;
        launch_rows_gemm<RG_ROWSK, RG_ROWSK, RG_EP_FWD>(g, 1, s);
        return check_launch("rows_gemm_kernel<fwd>");
    }
    const dim3 grid((p.batch + 31) / 32, (p.n_out + 31) / 32);
    if (dense_short_k(p.n_in, (int)(grid.x * grid.y))) ARVAE_LAUNCH(dense_fwd_kernel<4>, grid, dim3(256), 0, s, p);
    else ARVAE_LAUNCH(dense_fwd_kernel<16>, grid, dim3(1024), 0, s, p);
    return check_launch("dense_fwd_kernel");
}

int dense_dgrad(const arvae_link_t *l, const Operand &g, const float *w, const float *gate, float *dx, hipStream_t s) {
    DenseArgs p = dense_args(l);
    p.a = g; p.w = w; p.out = dx; p.gate = gate;
    if (dense_long_batch(p) && gate == nullptr && plain_operand(g)) {
        RowsGemm r{};
        r.a = g.v; r.lda = p.n_out; r.b = w; r.ldb = p.n_in; r.P = p.batch; r.Q = p.n_in; r.Rn = p.n_out; r.rslice = p.n_out;
        r.out = dx; r.ldo = p.n_in;
        launch_rows_gemm<RG_ROWSK, RG_KROWS, RG_EP_FWD>(r, 1, s);
        return check_launch("rows_gemm_kernel<dgrad>");
    }
    const dim3 grid((p.batch + 31) / 32, (p.n_in + 31) / 32);
    if (dense_short_k(p.n_out, (int)(grid.x * grid.y))) ARVAE_LAUNCH(dense_dgrad_kernel<4>, grid, dim3(256), 0, s, p);
    else ARVAE_LAUNCH(dense_dgrad_kernel<16>, grid, dim3(1024), 0, s, p);
    return check_launch("dense_dgrad_kernel");
}

static int wgrad_slices(int rows) { return (rows + DENSE_SPLIT_ROWS - 1) / DENSE_SPLIT_ROWS; }

int64_t dense_wgrad_ws_floats(const arvae_link_t *l) {
    if (l->n < DENSE_SPLIT_MIN_ROWS) return 0;
    return (int64_t)wgrad_slices(l->n) * ((int64_t)l->clo * l->chi + l->clo);
}

int dense_wgrad(const arvae_link_t *l, const Operand &g, const float *x, float *dw, float *dbias, float *ws, hipStream_t s) {
    DenseArgs p = dense_args(l);
    p.a = g; p.x = x; p.out = dw; p.dbias = dbias;
    if (ws != nullptr && dense_long_batch(p) && plain_operand(g)) {
        const int slices = wgrad_slices(p.batch);
        const int64_t w_floats = (int64_t)p.n_out * p.n_in, slice_floats = w_floats + p.n_out;
        RowsGemm r{};
        r.a = g.v; r.lda = p.n_out; r.b = x; r.ldb = p.n_in; r.P = p.n_out; r.Q = p.n_in; r.Rn = p.batch; r.rslice = DENSE_SPLIT_ROWS;
        r.out = ws; r.ldo = p.n_in; r.slice_floats = slice_floats; r.want_bias = dbias != nullptr;
        launch_rows_gemm<RG_KROWS, RG_KROWS, RG_EP_SLICE>(r, slices, s);
        const int64_t total = w_floats + (dbias != nullptr ? p.n_out : 0);
        ARVAE_LAUNCH(dense_split_reduce_kernel, dim3((unsigned)((total * 4 + 255) / 256)), dim3(256), 0, s, ws, slice_floats,
                           slices, w_floats, p.n_out, dw, dbias);
        return check_launch("rows_gemm_kernel<wgrad>");
    }
    ARVAE_LAUNCH(dense_wgrad_kernel, dim3((p.n_out + 31) / 32, (p.n_in + 31) / 32), dim3(DENSE_THREADS), 0, s, p);
    return check_launch("dense_wgrad_kernel");
}

// long-batch weight gradients queued for one launch (plan_measure.hip).  The queue owns a workspace [ws, ws + ws_cap): every job
// takes its slices from it; a job that does not qualify (short batch, folded operand, unaligned operands, queue or workspace
// full) is refused and the caller launches it by itself.
struct LongWgradQueue {
    RowsGemmBatch gemm;
    SplitReduceBatch red;
    float *ws;
    int64_t ws_cap, ws_used;
};
LongWgradQueue *dense_wgrad_long_new() { return new LongWgradQueue(); }
void dense_wgrad_long_delete(LongWgradQueue *q) { delete q; }
void dense_wgrad_long_begin(LongWgradQueue *q, float *ws, int64_t ws_cap) {
    q->gemm.count = q->red.count = 0;
    q->ws = ws;
    q->ws_cap = ws_cap;
    q->ws_used = 0;
}
// Rows per slice.  Workspace is reserved for the finest slicing (256 rows); the launch itself slices so that all its workgroups are
// resident at once -- two per CU -- as nearly as the jobs' tile counts allow (dense_wgrad_long_flush): with one workgroup more than
// the chip holds the launch takes two rounds (measured: 528 workgroups 90 us, 396 workgroups 72 us for the same products).
constexpr int LONG_RSLICE_MIN = 256;
static int long_rslice(int) { return LONG_RSLICE_MIN; }
int64_t dense_wgrad_long_ws_floats(const arvae_link_t *l) {
    if (l->n < DENSE_SPLIT_MIN_ROWS) return 0;
    const int rs = long_rslice(l->n);
    return (int64_t)((l->n + rs - 1) / rs) * (((int64_t)l->clo * l->chi + l->clo + 3) / 4 * 4);
}
bool dense_wgrad_long_defer(LongWgradQueue *q, const arvae_link_t *l, const Operand &g, const float *x, float *dw, float *dbias) {
    static const bool off = diag_env("ARVAE_NO_LONG_BATCH") != nullptr;       // A/B switch
    DenseArgs p = dense_args(l);
    p.a = g;
    if (off || q == nullptr || q->ws == nullptr || q->gemm.count >= RG_BATCH_MAX || !dense_long_batch(p) || !plain_operand(g)) return false;
    auto aligned = [](const float *ptr, int ld, int extent) { return (ld & 3) == 0 && (extent & 3) == 0 && (reinterpret_cast<uintptr_t>(ptr) & 15) == 0; };
    if (!aligned(g.v, p.n_out, p.n_out) || !aligned(x, p.n_in, p.n_in)) return false;
    // (the tile loaders address a matrix with 32-bit byte offsets)
    if ((int64_t)p.batch * p.n_out * 4 >= ((int64_t)1 << 31) || (int64_t)p.batch * p.n_in * 4 >= ((int64_t)1 << 31)) return false;
    const int rs = long_rslice(p.batch), slices = (p.batch + rs - 1) / rs;
    const int64_t w_floats = (int64_t)p.n_out * p.n_in, slice_floats = (w_floats + p.n_out + 3) / 4 * 4;
    if (q->ws_used + slices * slice_floats > q->ws_cap) return false;
    float *ws = q->ws + q->ws_used;
    q->ws_used += slices * slice_floats;
    const int k = q->gemm.count;
    RowsGemm &r = q->gemm.job[k];
    r = RowsGemm{};
    r.a = g.v; r.lda = p.n_out; r.b = x; r.ldb = p.n_in; r.P = p.n_out; r.Q = p.n_in; r.Rn = p.batch; r.rslice = rs;
    r.out = ws; r.ldo = p.n_in; r.slice_floats = slice_floats; r.want_bias = dbias != nullptr;
    const int wgs = ((p.n_out + 127) / 128) * ((p.n_in + 127) / 128) * slices;
    q->gemm.wg_end[k] = (k > 0 ? q->gemm.wg_end[k - 1] : 0) + wgs;
    q->gemm.count = k + 1;
    const int64_t total = w_floats + (dbias != nullptr ? p.n_out : 0);
    q->red.job[k] = SplitReduceJob{ws, slice_floats, w_floats, slices, p.n_out, dw, dbias};
    q->red.wg_end[k] = (k > 0 ? q->red.wg_end[k - 1] : 0) + (int)((total * 4 + 255) / 256);
    q->red.count = k + 1;
    return true;
}
int dense_wgrad_long_flush(LongWgradQueue *q, hipStream_t s) {
    if (q == nullptr || q->gemm.count == 0) return ARVAE_OK;
    {   // one round of workgroups: slices per job = resident slots / tiles of all jobs (never finer than the workspace allows)
        int tiles = 0;
        for (int k = 0; k < q->gemm.count; ++k) tiles += ((q->gemm.job[k].P + 127) / 128) * ((q->gemm.job[k].Q + 127) / 128);
        const int slots = 2 * device_cu_count();
        int target = slots / (tiles > 0 ? tiles : 1);
        if (target < 1) target = 1;
        int wg = 0;
        for (int k = 0; k < q->gemm.count; ++k) {
            RowsGemm &r = q->gemm.job[k];
            int rs = ((r.Rn + target - 1) / target + RG_R - 1) / RG_R * RG_R;
            if (rs < LONG_RSLICE_MIN) rs = LONG_RSLICE_MIN;
            const int slices = (r.Rn + rs - 1) / rs;
            r.rslice = rs;
            q->red.job[k].slices = slices;
            wg += ((r.P + 127) / 128) * ((r.Q + 127) / 128) * slices;
            q->gemm.wg_end[k] = wg;
        }
    }
    ARVAE_LAUNCH(rows_wgrad_batch_kernel, dim3((q->gemm.wg_end[q->gemm.count - 1] + 7) / 8 * 8), dim3(256), 0, s, q->gemm);
    if (int rc = check_launch("rows_wgrad_batch_kernel")) return rc;
    ARVAE_LAUNCH(dense_split_reduce_batch_kernel, dim3(q->red.wg_end[q->red.count - 1]), dim3(256), 0, s, q->red);
    q->gemm.count = q->red.count = 0;
    q->ws_used = 0;
    return check_launch("dense_split_reduce_batch_kernel");
}

// deferred weight gradients (plan.hip): add a job / launch all of them
bool dense_wgrad_defer(DenseWgradBatch *b, const arvae_link_t *l, const Operand &g, const float *x, float *dw, float *dbias) {
    if (b->count >= DENSE_BATCH_MAX) return false;
    DenseArgs p = dense_args(l);
    p.a = g; p.x = x; p.out = dw; p.dbias = dbias;
    const int tiles = ((p.n_out + 31) / 32) * ((p.n_in + 31) / 32);
    b->tile_end[b->count] = (b->count > 0 ? b->tile_end[b->count - 1] : 0) + tiles;
    b->job[b->count++] = p;
    return true;
}

int dense_wgrad_flush(DenseWgradBatch *b, hipStream_t s) {
    if (b->count == 0) return ARVAE_OK;
    static const bool split = diag_env("ARVAE_DENSE_BATCH_SPLIT") != nullptr;     // diagnostic: one launch per job
    if (split) {
        for (int j = 0; j < b->count; ++j) {
            DenseWgradBatch one{};
            one.count = 1;
            one.job[0] = b->job[j];
            one.tile_end[0] = b->tile_end[j] - (j > 0 ? b->tile_end[j - 1] : 0);
            ARVAE_LAUNCH(dense_wgrad_batch_kernel, dim3(one.tile_end[0]), dim3(64 * NWB), 0, s, one);
        }
        b->count = 0;
        return check_launch("dense_wgrad_batch_kernel");
    }
    ARVAE_LAUNCH(dense_wgrad_batch_kernel, dim3(b->tile_end[b->count - 1]), dim3(64 * NWB), 0, s, *b);
    b->count = 0;
    return check_launch("dense_wgrad_batch_kernel");
}

// the queued Linear weight gradients and the single-channel layer's weight-gradient partials (conv_c1.hip) in one launch
int wgrad_c1_groups(const arvae_link_t *l);
bool dense_wgrad_c1_fits(const DenseWgradBatch *b) {
    static const bool off = diag_env("ARVAE_NO_PAIR_TAIL") != nullptr || diag_env("ARVAE_DENSE_BATCH_SPLIT") != nullptr;
    return !off && b != nullptr && b->count > 0;
}
int dense_wgrad_flush_with_c1(DenseWgradBatch *b, const arvae_link_t *l, const Operand &lo, const Operand &img, float *dwt, float *dbias,
                              int bias_mode, float *slab, hipStream_t s, SlabJob *job) {
    const int grid_c1 = wgrad_c1_groups(l);
    ARVAE_LAUNCH(dense_wgrad_c1_kernel, dim3(grid_c1 + b->tile_end[b->count - 1]), dim3(64 * NWB), 0, s, *b, lo, img, slab, l->n * LO1, grid_c1);
    b->count = 0;
    *job = SlabJob{slab, dwt, dbias, grid_c1, SLAB_C1, bias_mode};
    return check_launch("pair(wgrad_c1 + dense_wgrad_batch)");
}

}  // namespace arvae

#ifdef DW_STAMPS
extern "C" int arvae_debug_dw_stamps(unsigned long long *out, int count) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(arvae::g_dw_stamps), sizeof(unsigned long long) * count);
}
#endif

// Several Linear weight gradients in one launch for callers outside the whole-model executor (the MeasureVAE's autograd
// graph queues its batch-sized ones and flushes them at the end of the backward pass).
extern "C" int arvae_dense_wgrad_batch(const arvae_dense_wgrad_job_t *jobs, int32_t njobs, arvae_stream_t stream) {
    using namespace arvae;
    ARVAE_REQUIRE(jobs != nullptr && njobs >= 1, "dense_wgrad_batch: no jobs");
    hipStream_t st = as_stream(stream);
    DenseWgradBatch b{};
    for (int j = 0; j < njobs; ++j) {
        const arvae_dense_wgrad_job_t &q = jobs[j];
        ARVAE_REQUIRE(q.g.v && q.x && q.dw && q.rows > 0 && q.n_in > 0 && q.n_out > 0, "dense_wgrad_batch: bad job %d", j);
        arvae_link_t l{};
        l.n = q.rows; l.hh = l.hw = l.lh = l.lw = l.kh = l.kw = 1; l.stride = 1; l.chi = q.n_in; l.clo = q.n_out;
        if (!dense_wgrad_defer(&b, &l, make_operand(&q.g), q.x, q.dw, q.dbias)) {
            if (int rc = dense_wgrad_flush(&b, st)) return rc;
            dense_wgrad_defer(&b, &l, make_operand(&q.g), q.x, q.dw, q.dbias);
        }
    }
    return dense_wgrad_flush(&b, st);
}

#ifdef RG_STAMPS
extern "C" int arvae_debug_rg_stamps(unsigned long long *out, int count) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(arvae::g_rg_stamps), sizeof(unsigned long long) * count);
}
#endif
