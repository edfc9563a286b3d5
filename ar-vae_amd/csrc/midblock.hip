// The latent block of the conv VAEs as ONE launch per pass (imagevae/mnist_vae.py:59-72, dsprites_vae.py:12-46):
//   forward : conv features -> enc_lin (Linear + act)* -> enc_mean / enc_log_std -> z = mu + eps * exp(log_std)
//             -> dec_lin (Linear + act)* -> first deconv's input
//   backward: the same chain in reverse: d(dec_lin) -> d z (+ KL and regulariser gradients) -> d(mu, log_std) -> heads ->
//             d(enc_lin) -> gradient for the last conv layer, leaving every layer's pre-activation gradient in memory for
//             the grouped weight-gradient launch (dense.hip).
// On by default (see mid_fusable; ARVAE_MIDBLOCK=0 for the per-layer launches).
// Per layer these are a few MFLOP per batch: as separate launches (5 + 5 Linear kernels, 2 head kernels) they cost
// ~6 us each of pure latency, 12 dependent launches per step.  A batch row never talks to another here, so a workgroup
// takes R rows through the WHOLE chain with its activations in LDS; what it streams is the weights (L2-resident,
// ~1.6 MB for dSprites), read as coalesced 16-byte loads because a prep launch (mid_prep) lays every matrix out
// reduce-major for the direction that uses it ([k][n] for the forward product, [n][k] for the backward one), with the
// NCHW-flatten permutations of the first / last layer folded in.  Exact fp32 FMA chains (no MFMA: at 4 rows a matrix tile
// would be three quarters padding and the weight stream, not the arithmetic, sets the pace).
#include <mutex>
#include "diag.h"
#include "common.h"
#include "dense.h"
#include "rng.h"
#include "midprep.h"
#include "midcluster.h"
#include "conv32_common.h"

namespace arvae {

int conv32_amax(const float *x, int64_t count, unsigned *out, hipStream_t s);      // conv32.hip


constexpr int MID_T = 512;           // threads per workgroup
constexpr int MID_WIDE_MIN = 1024;   // a first / last layer at least this wide on its outer side goes to the tile GEMMs (dense.hip)
constexpr int MID_MAX_W = 3072;      // widest layer (Morpho-MNIST: 2888)
constexpr int mid_red(int r) { return r * 4 * MID_T; }   // floats of cross-slice reduction scratch (slices * n <= 4 * MID_T)

struct MidLayer {
    const float *mf;     // [k][n] forward matrix   (memory order on both axes)
    const float *mb;     // [n][k] backward matrix
    const float *bias;   // [n] or null
    float *y;            // saved output [batch][n]
    float *gpre;         // gradient w.r.t. the pre-activation [batch][n] (backward writes it for the weight gradient)
    int k, n, act;
    int kb;              // row length of mb: k rounded up to a multiple of 4 (zero-padded: the decoder's first layer reads z)
    // round 6: the saved output / the pre-activation gradient ALSO as their three bf16 terms, planes [3][batch][n] (x3tile.h), for the
    // tile GEMMs that read them as pre-split operands (dense.hip wide_gemm_x3_kernel), or null
    unsigned short *y_planes, *g_planes;
    // ... and a wide layer's own prepared matrix for those GEMMs: planes [3][n_pad][kb_pad]
    const unsigned short *w_planes;
    int n_pad, kb_pad;
};

struct MidArgs {
    MidLayer enc[MID_MAX_LAYERS], dec[MID_MAX_LAYERS];
    int ne, nd, batch, zdim, h, ld;                  // h = width of the encoder's hidden vector, ld = LDS row pitch
    const float *x0;                                 // conv features [batch][enc[0].k]
    const float *w_mu, *b_mu, *w_ls, *b_ls;          // heads, reference layout [zdim][h]
    const float *hf, *hb, *hbias;                    // heads prepped: [h][2 zdim] (mu columns, then log_std), [2 zdim][h], [2 zdim]
    float *mu, *log_std, *sigma, *z;
    const float *eps;                                // forward: input unless eps_out; backward: input
    float *eps_out;                                  // forward: draw eps (rng) and write it here
    RngStream rng;
    // backward only
    const float *g_out;                              // gradient arriving at dec[nd-1]: w.r.t. its pre-activation (g_is_pre) or output
    int g_is_pre;
    const float *gate0;                              // saved ReLU output of the producer of x0, or null: d x0 *= (gate0 > 0)
    float *d_x0;                                     // gradient handed to the last conv layer [batch][enc[0].k]
    const float *dz_reg, *dz_extra, *g_loss, *kl, *cap;
    float beta, inv_batch, reg_scale;
    float *d_mu, *d_ls;
    // the matrices this pass streams, except the first one: requested once per XCD at the top of the kernel (mid_warm)
    unsigned *amax_out;                              // AMAX array (conv32_common.h) of the pass's last output -- dec[nd-1].y forward,
                                                     // d_x0 backward -- for the 32-channel conv kernel that reads it next, or null
    const float *warm_ptr[2 * MID_MAX_LAYERS + 1];
    int warm_lines[2 * MID_MAX_LAYERS + 1];          // 128-byte lines
    int n_warm;
    // Wide layers the tile GEMMs multiply (dense.hip wide_gemm_x3_kernel; round 6): the block's first encoder layer (skip_enc0)
    // and / or its last decoder layer (skip_dec_last) are NOT run here.  What this kernel does for them is the consumer's half
    // of a split reduction: forward, enc[0]'s output = act(bias + sum over `wide_slices` partial products, in slice order),
    // saved like every layer's; backward, the gradient at dec[nd - 2]'s output = the slices' sum, times act'(saved output).
    int skip_enc0, skip_dec_last, wide_slices;
    const float *wide_partial;                       // [wide_slices][batch][width]
    int64_t wide_slice_floats;
};

// The prep launch wrote the matrices from other XCDs, so a layer's first weight loads miss this XCD's L2 and every layer of the
// chain pays a round trip to memory (>= 3.3 us per layer however small).  Workgroups are dispatched round-robin over the 8 XCDs,
// so workgroup b shares its L2 with the workgroups b' = b (mod 8): each of them touches its 1/n-th of the lines the later layers
// will stream -- ONE dword per 128-byte line and lane, MID_WARM loads per thread, nobody waits for them (their registers are
// consumed by an empty asm at the very end of the kernel) -- and by the time the second layer starts its matrix is in L2.
constexpr int MID_WARM = 2;
__device__ __forceinline__ void mid_warm(const MidArgs &p, float (&w)[MID_WARM]) {
    const int xw = blockIdx.x >> 3, nxw = ((int)gridDim.x + 7) >> 3;
    int total = 0;
    for (int j = 0; j < p.n_warm; ++j) total += p.warm_lines[j];
    const int share = (total + nxw - 1) / nxw;
#pragma unroll
    for (int k = 0; k < MID_WARM; ++k) {
        const int mine = k * MID_T + (int)threadIdx.x;
        int f = xw * share + mine;
        const bool ok = p.n_warm > 0 && mine < share && f < total;
        if (!ok) f = 0;
        const float *ptr = p.warm_ptr[0];
        for (int j = 0; j + 1 < p.n_warm && f >= p.warm_lines[j]; ++j) { f -= p.warm_lines[j]; ptr = p.warm_ptr[j + 1]; }
        w[k] = p.n_warm > 0 ? ptr[(int64_t)f * 32] : 0.f;
    }
}
__device__ __forceinline__ void mid_warm_done(float (&w)[MID_WARM]) {
#pragma unroll
    for (int k = 0; k < MID_WARM; ++k) asm volatile("" ::"v"(w[k]));
}

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void fma4(float4 &a, float x, const float4 &w) {
    a.x = fmaf(x, w.x, a.x); a.y = fmaf(x, w.y, a.y); a.z = fmaf(x, w.z, a.z); a.w = fmaf(x, w.w, a.w);
}

// rows of `in` (LDS, pitch ld) times M [ni][no] (global, no % 4 == 0) for columns 4q .. 4q+3, reduction range [i_lo, i_hi).
// The weight stream is what this block waits for (an L2 round trip is ~1000 cycles, a 4-row step of 64 FMAs ~200), so the
// reduction walks blocks of 8 rows of M with the NEXT block's eight 16-byte loads issued before the current block's FMAs:
// two register sets, 16 loads in flight per thread at the block boundary (16-row blocks spill at 256 registers).
template <int R>
__device__ __forceinline__ void mid_step4(const float *in, int ld, int i, const float4 (&w)[4], float4 (&acc)[R]) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float4 x = ld4(in + r * ld + i);
        fma4(acc[r], x.x, w[0]); fma4(acc[r], x.y, w[1]); fma4(acc[r], x.z, w[2]); fma4(acc[r], x.w, w[3]);
    }
}
template <int R>
__device__ __forceinline__ void mid_dot(const float *in, int ld, const float *__restrict__ M, int no, int q, int i_lo, int i_hi,
                                        float4 (&acc)[R]) {
    const float *mp = M + 4 * q;
    int i = i_lo;
    if (i + 8 <= i_hi) {
        float4 wa[2][4], wb[2][4];
        auto load8 = [&](float4 (&w)[2][4], int at) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < 4; ++t) w[u][t] = ld4(mp + (int64_t)(at + 4 * u + t) * no);
        };
        auto fma8 = [&](const float4 (&w)[2][4], int at) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < 2; ++u) mid_step4<R>(in, ld, at + 4 * u, w[u], acc);
        };
        load8(wa, i);
        for (;;) {
            if (i + 16 > i_hi) { fma8(wa, i); i += 8; break; }
            load8(wb, i + 8);
            fma8(wa, i);
            i += 8;
            if (i + 16 > i_hi) { fma8(wb, i); i += 8; break; }
            load8(wa, i + 8);
            fma8(wb, i);
            i += 8;
        }
    }
    for (; i + 4 <= i_hi; i += 4) {
        float4 w[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) w[t] = ld4(mp + (int64_t)(i + t) * no);
        mid_step4<R>(in, ld, i, w, acc);
    }
    for (; i < i_hi; ++i) {
        const float4 w = ld4(mp + (int64_t)i * no);
#pragma unroll
        for (int r = 0; r < R; ++r) fma4(acc[r], in[r * ld + i], w);
    }
}

// out[r][j] = fin(r, quad, sum_i in[r][i] * M[i][j], pre(r, quad)) for the R rows of this workgroup.  `pre` is called
// BEFORE the reduction loop for every output the thread will finalise (its global loads fly during the loop).
// Narrow layers (no / 4 < MID_T) split the reduction over MID_T / (no / 4) thread slices that meet in `red`.
template <int R, class Pre, class Fin>
__device__ __forceinline__ void mid_matmul(const float *in, int ld, int ni, const float *__restrict__ M, int no, float *red, Pre pre,
                                           Fin fin) {
    const int tid = threadIdx.x, nq = no >> 2;
    if (nq >= MID_T) {
        for (int q = tid; q < nq; q += MID_T) {
            float4 pf[R], acc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) { pf[r] = pre(r, q); acc[r] = make_float4(0.f, 0.f, 0.f, 0.f); }
            mid_dot<R>(in, ld, M, no, q, 0, ni, acc);
#pragma unroll
            for (int r = 0; r < R; ++r) fin(r, q, acc[r], pf[r]);
        }
        return;
    }
    const int slices = min(MID_T / nq, 32), s = tid / nq, q = tid - s * nq;      // (a very narrow output keeps its final sum short)
    const int chunk = (((ni + slices - 1) / slices) + 3) & ~3;
    const int i_lo = min(s * chunk, ni), i_hi = min(i_lo + chunk, ni);
    float4 pf[R];
    bool mine[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {                            // the outputs this thread finalises: idx = tid + j * MID_T
        const int idx = tid + j * MID_T;
        mine[j] = idx < R * nq;
        pf[j] = mine[j] ? pre(idx / nq, idx % nq) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (s < slices) {
        mid_dot<R>(in, ld, M, no, q, i_lo, i_hi, acc);
#pragma unroll
        for (int r = 0; r < R; ++r) *reinterpret_cast<float4 *>(red + (s * R + r) * no + 4 * q) = acc[r];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < R; ++j) {
        if (!mine[j]) continue;
        const int idx = tid + j * MID_T, r = idx / nq, qq = idx - r * nq;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int ss = 0; ss < slices; ++ss) {                    // fixed order
            const float4 t = ld4(red + (ss * R + r) * no + 4 * qq);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        fin(r, qq, v, pf[j]);
    }
}

__device__ __forceinline__ float4 act4(float4 v, int act) {
    return make_float4(act_fwd(v.x, act), act_fwd(v.y, act), act_fwd(v.z, act), act_fwd(v.w, act));
}
__device__ __forceinline__ float4 dact4(float4 g, float4 y, int act) {       // g * act'(.) from the saved output y
    return make_float4(g.x * act_bwd_from_out(y.x, act), g.y * act_bwd_from_out(y.y, act), g.z * act_bwd_from_out(y.z, act),
                       g.w * act_bwd_from_out(y.w, act));
}

// four consecutive values (row, col .. col + 3; col a multiple of 4) of a [rows][n] tensor as their three bf16 terms into its TILED
// planes (x3tile.h x3_tiled_index), `pstride` = rows * n elements apart
__device__ __forceinline__ void store_planes(unsigned short *p, int64_t pstride, int rows, int row, int col, const float4 &v) {
    const int64_t idx = x3_tiled_index(rows, row, col);
    unsigned h0, m0, l0, h1, m1, l1;
    rg_split3(v.x, v.y, h0, m0, l0);
    rg_split3(v.z, v.w, h1, m1, l1);
    *reinterpret_cast<uint2 *>(p + idx) = uint2{h0, h1};
    *reinterpret_cast<uint2 *>(p + pstride + idx) = uint2{m0, m1};
    *reinterpret_cast<uint2 *>(p + 2 * pstride + idx) = uint2{l0, l1};
}

// maximum magnitude of the R x n block a pass leaves in LDS (rows past the batch hold zeros): one AMAX writer unit per workgroup
template <int R>
__device__ __forceinline__ void mid_amax(const float *rows, int ld, int n, float *scratch, unsigned *out) {
    if (out == nullptr) return;
    float m = 0.f;
    const int n4 = n >> 2;
    for (int i = threadIdx.x; i < R * n4; i += MID_T) m = fmaxf(m, amax4(ld4(rows + (i / n4) * ld + 4 * (i % n4))));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x < 64) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < MID_T / 64; ++w) t = fmaxf(t, scratch[w]);
        amax_publish(out, blockIdx.x, gridDim.x, t);
    }
}

#ifdef MID_STAMPS
__device__ unsigned long long g_mid_stamps[128 * 16];
#define MID_STAMP(slot) do { if (threadIdx.x == 0 && blockIdx.x < 128) g_mid_stamps[blockIdx.x * 16 + (slot)] = wall_clock64(); } while (0)
#else
#define MID_STAMP(slot)
#endif

// ================================================================================================ forward
template <int R>
__global__ __launch_bounds__(MID_T) void mid_forward_kernel(MidArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];  // bufA | bufB | red | heads scratch (R * 32)
    float *bufA = lds, *bufB = lds + R * p.ld, *red = bufB + R * p.ld, *outs = red + mid_red(R);
    const int tid = threadIdx.x, row0 = blockIdx.x * R;
    MID_STAMP(0);
    float warm[MID_WARM];
    mid_warm(p, warm);
    if (p.skip_enc0) {
        // the first encoder layer was multiplied by the tile GEMM (split reduction): finish it -- slices summed in order, bias,
        // activation, the saved output -- into bufA
        const MidLayer l = p.enc[0];
        const int n4 = l.n >> 2;
        for (int i = tid; i < R * n4; i += MID_T) {
            const int r = i / n4, c = i - r * n4, row = row0 + r;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < p.batch) {
                v = l.bias != nullptr ? ld4(l.bias + 4 * c) : v;
                const float *src = p.wide_partial + (int64_t)row * l.n + 4 * c;
                for (int sl = 0; sl < p.wide_slices; ++sl) {
                    const float4 t = ld4(src + sl * p.wide_slice_floats);
                    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
                }
                v = act4(v, l.act);
                *reinterpret_cast<float4 *>(l.y + (int64_t)row * l.n + 4 * c) = v;
            }
            *reinterpret_cast<float4 *>(bufA + r * p.ld + 4 * c) = v;
        }
    } else {   // conv features of this workgroup's rows -> bufA (rows past the batch: zeros)
        const int k4 = p.enc[0].k >> 2;
        for (int i = tid; i < R * k4; i += MID_T) {
            const int r = i / k4, c = i - r * k4;
            const float4 v = row0 + r < p.batch ? ld4(p.x0 + (int64_t)(row0 + r) * p.enc[0].k + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4 *>(bufA + r * p.ld + 4 * c) = v;
        }
    }
    __syncthreads();
    MID_STAMP(1);
    float *cur = bufA, *nxt = bufB;
    int stamp_no = 2;
    (void)stamp_no;
    auto run_layer = [&](const MidLayer l) {
        mid_matmul<R>(
            cur, p.ld, l.k, l.mf, l.n, red,
            [&](int, int q) { return l.bias != nullptr ? ld4(l.bias + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f); },
            [&](int r, int q, float4 v, float4 b) {
                v = act4(make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w), l.act);
                *reinterpret_cast<float4 *>(nxt + r * p.ld + 4 * q) = v;
                if (row0 + r < p.batch) {
                    *reinterpret_cast<float4 *>(l.y + (int64_t)(row0 + r) * l.n + 4 * q) = v;
                    if (l.y_planes != nullptr) store_planes(l.y_planes, (int64_t)p.batch * l.n, p.batch, row0 + r, 4 * q, v);
                }
            });
        __syncthreads();
        MID_STAMP(stamp_no); ++stamp_no;
        float *t = cur; cur = nxt; nxt = t;
    };
    for (int i = p.skip_enc0 ? 1 : 0; i < p.ne; ++i) run_layer(p.enc[i]);
    // heads: one more layer of the chain, (mu | log_std) = hidden x [h][2 zdim] + bias, into the scratch rows `outs`
    mid_matmul<R>(
        cur, p.ld, p.h, p.hf, 2 * p.zdim, red,
        [&](int, int q) { return ld4(p.hbias + 4 * q); },
        [&](int r, int q, float4 v, float4 b) {
            *reinterpret_cast<float4 *>(outs + r * 32 + 4 * q) = make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w);
        });
    __syncthreads();
    if (tid < R * p.zdim) {
        const int r = tid / p.zdim, j = tid - r * p.zdim, row = row0 + r;
        const int64_t idx = (int64_t)(row < p.batch ? row : 0) * p.zdim + j;
        const float m = outs[r * 32 + j], l = outs[r * 32 + j + p.zdim], s = expf(l);
        const float e = p.eps_out != nullptr ? rng_normal(p.rng, (uint64_t)idx) : p.eps[idx];
        const float zv = fmaf(e, s, m);
        nxt[r * p.ld + j] = row < p.batch ? zv : 0.f;
        if (row < p.batch) {
            p.mu[idx] = m; p.log_std[idx] = l; p.sigma[idx] = s; p.z[idx] = zv;
            if (p.eps_out != nullptr) p.eps_out[idx] = e;
        }
    }
    __syncthreads();
    { float *t = cur; cur = nxt; nxt = t; }
    MID_STAMP(stamp_no); ++stamp_no;
    const int nd_run = p.skip_dec_last ? p.nd - 1 : p.nd;      // (a skipped last layer: the tile GEMM multiplies dec[nd - 2]'s saved output)
    for (int i = 0; i < nd_run; ++i) run_layer(p.dec[i]);
    if (!p.skip_dec_last) mid_amax<R>(cur, p.ld, p.dec[p.nd - 1].n, red, p.amax_out);
    mid_warm_done(warm);
}

// ================================================================================================ backward
template <int R>
__global__ __launch_bounds__(MID_T) void mid_backward_kernel(MidArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *bufA = lds, *bufB = lds + R * p.ld, *red = bufB + R * p.ld, *dm = red + mid_red(R), *dl = dm + R * 16;
    const int tid = threadIdx.x, row0 = blockIdx.x * R;
    float warm[MID_WARM];
    mid_warm(p, warm);
    if (p.skip_dec_last) {
        // the last decoder layer's data gradient was multiplied by the tile GEMM (split reduction): the slices' sum is the gradient
        // at dec[nd - 2]'s OUTPUT; times act'(its saved output) = that layer's pre-activation gradient, kept for its weight gradient
        const MidLayer l = p.dec[p.nd - 2];
        const int n4 = l.n >> 2;
        for (int i = tid; i < R * n4; i += MID_T) {
            const int r = i / n4, c = i - r * n4, row = row0 + r;
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < p.batch) {
                const float *src = p.wide_partial + (int64_t)row * l.n + 4 * c;
                for (int sl = 0; sl < p.wide_slices; ++sl) {
                    const float4 t = ld4(src + sl * p.wide_slice_floats);
                    g.x += t.x; g.y += t.y; g.z += t.z; g.w += t.w;
                }
                g = dact4(g, ld4(l.y + (int64_t)row * l.n + 4 * c), l.act);
                *reinterpret_cast<float4 *>(l.gpre + (int64_t)row * l.n + 4 * c) = g;
            }
            *reinterpret_cast<float4 *>(bufA + r * p.ld + 4 * c) = g;
        }
    } else {   // gradient arriving at the last decoder Linear layer -> bufA as a pre-activation gradient
        const MidLayer l = p.dec[p.nd - 1];
        const int n4 = l.n >> 2;
        for (int i = tid; i < R * n4; i += MID_T) {
            const int r = i / n4, c = i - r * n4, row = row0 + r;
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < p.batch) {
                g = ld4(p.g_out + (int64_t)row * l.n + 4 * c);
                if (!p.g_is_pre) {
                    g = dact4(g, ld4(l.y + (int64_t)row * l.n + 4 * c), l.act);
                    *reinterpret_cast<float4 *>(l.gpre + (int64_t)row * l.n + 4 * c) = g;
                }
            }
            *reinterpret_cast<float4 *>(bufA + r * p.ld + 4 * c) = g;
        }
    }
    __syncthreads();
    float *cur = bufA, *nxt = bufB;
    // d(input of layer l) from d(pre-activation of l): [n] -> [k]; prev = the layer that produced the input (null: no activation)
    // y_prev / act_prev: saved output and activation of the layer that produced this layer's input (y_prev null: the input is
    // the conv feature map, gated by gate_relu when given); dst: where the input's pre-activation gradient goes
    auto back_layer = [&](const float *mb, int n, int k, const float *y_prev, int act_prev, const float *gate_relu, float *dst,
                          unsigned short *dst_planes = nullptr) {
        mid_matmul<R>(
            cur, p.ld, n, mb, k, red,
            [&](int r, int q) {
                const int row = row0 + r;
                const float *src = y_prev != nullptr ? y_prev : gate_relu;
                if (row >= p.batch || src == nullptr) return make_float4(1.f, 1.f, 1.f, 1.f);
                return ld4(src + (int64_t)row * k + 4 * q);
            },
            [&](int r, int q, float4 v, float4 y) {
                const int row = row0 + r;
                if (y_prev != nullptr) v = dact4(v, y, act_prev);
                else if (gate_relu != nullptr) v = make_float4(y.x > 0.f ? v.x : 0.f, y.y > 0.f ? v.y : 0.f, y.z > 0.f ? v.z : 0.f, y.w > 0.f ? v.w : 0.f);
                *reinterpret_cast<float4 *>(nxt + r * p.ld + 4 * q) = v;
                if (row < p.batch && dst != nullptr) {
                    *reinterpret_cast<float4 *>(dst + (int64_t)row * k + 4 * q) = v;
                    if (dst_planes != nullptr) store_planes(dst_planes, (int64_t)p.batch * k, p.batch, row, 4 * q, v);
                }
            });
        __syncthreads();
        float *t = cur; cur = nxt; nxt = t;
    };
    for (int i = p.nd - 1 - (p.skip_dec_last ? 1 : 0); i >= 1; --i)
        back_layer(p.dec[i].mb, p.dec[i].n, p.dec[i].k, p.dec[i - 1].y, p.dec[i - 1].act, nullptr, p.dec[i - 1].gpre);
    // dec[0]: its input is z (k = zdim, no activation; its backward matrix is zero-padded to a multiple of 4 columns)
    {
        const MidLayer l = p.dec[0];
        mid_matmul<R>(
            cur, p.ld, l.n, l.mb, l.kb, red, [&](int, int) { return make_float4(0.f, 0.f, 0.f, 0.f); },
            [&](int r, int q, float4 v, float4) { *reinterpret_cast<float4 *>(nxt + r * p.ld + 4 * q) = v; });
        __syncthreads();
        if (tid < R * p.zdim) {
            const int r = tid / p.zdim, j = tid - r * p.zdim, row = row0 + r;
            float gz = nxt[r * p.ld + j];
            // d(mu, log_std): decoder path + regulariser + KL (the formulas of heads_latent_bwd_kernel, heads.hip)
            const int64_t i = (int64_t)(row < p.batch ? row : 0) * p.zdim + j;
            const float g = p.g_loss[0];
            const float diff = p.kl[0] - (p.cap != nullptr ? p.cap[0] : 0.f);
            const float kk = g * p.beta * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) * p.inv_batch;
            if (p.dz_reg != nullptr) gz += g * p.reg_scale * p.dz_reg[i];
            if (p.dz_extra != nullptr) gz += p.dz_extra[i];
            const float s = p.sigma[i], mu = p.mu[i], e = p.eps[i];
            const float a = gz + kk * mu, b = (gz * e + kk * (s - 1.f / s)) * s;
            const bool on = row < p.batch;
            if (on) { p.d_mu[i] = a; p.d_ls[i] = b; }
            dm[r * 16 + j] = on ? a : 0.f;
            dl[r * 16 + j] = on ? b : 0.f;
        }
    }
    __syncthreads();
    // d hidden = (d_mu | d_ls) x [2 zdim][h], times act' of the last encoder Linear layer
    {
        const MidLayer last = p.enc[p.ne - 1];
        if (tid < R * 2 * p.zdim) {
            const int r = tid / (2 * p.zdim), j = tid - r * 2 * p.zdim;
            nxt[r * p.ld + j] = j < p.zdim ? dm[r * 16 + j] : dl[r * 16 + j - p.zdim];
        }
        __syncthreads();
        { float *t = cur; cur = nxt; nxt = t; }
        back_layer(p.hb, 2 * p.zdim, p.h, last.y, last.act, nullptr, last.gpre, last.g_planes);
    }
    for (int i = p.ne - 1; i >= 1; --i)
        back_layer(p.enc[i].mb, p.enc[i].n, p.enc[i].k, p.enc[i - 1].y, p.enc[i - 1].act, nullptr, p.enc[i - 1].gpre, p.enc[i - 1].g_planes);
    if (!p.skip_enc0) {          // (skipped: the tile GEMM multiplies enc[0]'s pre-activation gradient, stored just above)
        back_layer(p.enc[0].mb, p.enc[0].n, p.enc[0].k, nullptr, 0, p.gate0, p.d_x0);
        mid_amax<R>(cur, p.ld, p.enc[0].k, red, p.amax_out);
    }
    mid_warm_done(warm);
}

// ================================================================================================ weight layout prep
__global__ __launch_bounds__(256) void mid_prep_kernel(MidPrepArgs a) { mid_prep_block(a, blockIdx.x); }

// ================================================================================================ host side
static bool mid_layer_ok(const arvae_layer_t &l) {
    return dense_fits(&l.link) && !l.is_up && l.dropout == 0 && l.link.chi % 4 == 0 && l.link.clo % 4 == 0 && l.link.chi <= MID_MAX_W &&
           l.link.clo <= MID_MAX_W;
}

// trailing Linear layers of the encoder / leading Linear layers of the decoder that the block covers (0: block not usable)
bool mid_fusable(const arvae_image_vae_t *m, int *ne_out, int *nd_out) {
    // ON by default since the end of round 2 (ARVAE_MIDBLOCK=0 selects the twelve per-layer launches).  At B = 512 on MI355X:
    // forward 32 us, backward 39 us, prep 8 us against ~87 us for the launches it replaces -- the kernel time is a wash, but
    // the step has nine launches fewer and is 2.2 % faster in a same-box A/B (it measured equal while the step still had 44
    // launches).  Phase stamps (tools/stamp_mid.py): every layer of the chain costs >= 3.3 us however small (the 10 -> 256
    // layer included): a dependent round trip to the freshly written weights, two barriers and the saved-activation store per
    // layer, on 128 of the 256 CUs.
    static const bool off = diag_env("ARVAE_MIDBLOCK") != nullptr && diag_env("ARVAE_MIDBLOCK")[0] == '0';
    int ne = 0, nd = 0;
    while (ne < m->n_enc && ne < MID_MAX_LAYERS && mid_layer_ok(m->enc[m->n_enc - 1 - ne])) ++ne;
    while (nd < m->n_dec && nd < MID_MAX_LAYERS) {
        const arvae_layer_t &l = m->dec[nd];
        // dec[0] reads z: its input width is zdim (any value); the others must be multiples of 4 on both sides
        const bool ok = dense_fits(&l.link) && !l.is_up && l.dropout == 0 && l.link.clo % 4 == 0 && l.link.clo <= MID_MAX_W &&
                        (nd == 0 ? l.link.chi == m->zdim : l.link.chi % 4 == 0);
        if (!ok) break;
        ++nd;
    }
    // the whole Linear stack of either side must be inside the block, the hidden vector feeds the heads
    const bool enc_whole = ne >= 1 && (ne == m->n_enc || !dense_fits(&m->enc[m->n_enc - 1 - ne].link));
    const bool dec_whole = nd >= 1 && nd < m->n_dec && !dense_fits(&m->dec[nd].link);
    const arvae_link_t &hm = m->head_mu.link;
    const bool heads_ok = dense_fits(&hm) && dense_fits(&m->head_log_std.link) && hm.clo == m->zdim && m->head_log_std.link.clo == m->zdim &&
                          m->zdim <= 16 && m->zdim % 2 == 0 && hm.chi % 4 == 0 && hm.chi <= 512 && hm.hi_perm_c == 0 && hm.lo_perm_c == 0 &&
                          m->head_mu.act == ARVAE_ACT_NONE && m->head_log_std.act == ARVAE_ACT_NONE &&
                          m->head_log_std.link.hi_perm_c == 0 && m->head_log_std.link.lo_perm_c == 0;
    const bool ok = !off && enc_whole && dec_whole && heads_ok && ne < m->n_enc && m->enc[m->n_enc - 1].link.clo == hm.chi &&
                    m->dec[0].link.hi_perm_c == 0 && m->dec[0].link.lo_perm_c == 0;
    if (ne_out) *ne_out = ok ? ne : 0;
    if (nd_out) *nd_out = ok ? nd : 0;
    return ok;
}

static int64_t mid_layer_floats(const arvae_layer_t &l) {
    const int64_t k = l.link.chi, n = l.link.clo, kb = (k + 3) / 4 * 4;
    return (k * n + 3) / 4 * 4 + kb * n + (n + 3) / 4 * 4;       // mf | mb | bias, each 16-byte aligned
}

// The clustered kernels (midcluster.hip) cover the dSprites-shaped block: Linear K0 -> H -> H, heads H -> zdim, decoder
// zdim -> H -> H -> K0 with K0 = 512, H = 256, zdim <= 16 (imagevae/dsprites_vae.py:22-37)
static bool midc_topology(const arvae_image_vae_t *m, int ne, int nd) {
    if (ne != 2 || nd != 3 || m->zdim > 16) return false;
    const arvae_link_t &e0 = m->enc[m->n_enc - 2].link, &e1 = m->enc[m->n_enc - 1].link;
    const arvae_link_t &d0 = m->dec[0].link, &d1 = m->dec[1].link, &d2 = m->dec[2].link;
    return e0.chi == MC_K0 && e0.clo == MC_H && e1.chi == MC_H && e1.clo == MC_H && m->head_mu.link.chi == MC_H && d0.chi == m->zdim &&
           d0.clo == MC_H && d1.chi == MC_H && d1.clo == MC_H && d2.chi == MC_H && d2.clo == MC_K0 && m->head_mu.b_off >= 0 &&
           m->head_log_std.b_off >= 0;
}
constexpr int64_t MIDC_COUNTER_WORDS = MC_COUNTER_WORDS;     // arrival counters of 32 clusters + the ticket heads (midcluster.h)
// cluster-layout floats of one matrix: both axes rounded up to 16 (the reduce axis of the z-sized ones to 16, the heads' to 32)
static int64_t midc_mat_floats(int k, int n) { return (int64_t)((k + 15) / 16 * 16) * ((n + 15) / 16 * 16); }
static int64_t midc_floats(const arvae_image_vae_t *m, int ne, int nd) {
    if (!midc_topology(m, ne, nd)) return 0;
    int64_t total = MIDC_COUNTER_WORDS;
    for (int i = 0; i < ne; ++i) total += 2 * midc_mat_floats(m->enc[m->n_enc - ne + i].link.chi, m->enc[m->n_enc - ne + i].link.clo);
    for (int i = 0; i < nd; ++i) total += 2 * midc_mat_floats(m->dec[i].link.chi, m->dec[i].link.clo);
    return total + 2 * midc_mat_floats(m->head_mu.link.chi, 32) + 2 * 2 * 16384;     // + the two folded conv layers' matrices
}

static bool midc_use(bool cluster_ok, int batch, int flags);

// The conv layers on either side of the block that the clustered kernels can compute themselves (midcluster.h, McArgs.fold): the
// encoder's last conv layer (Conv2d 8x8 -> 4x4, 32 <-> 32 channels, ReLU, no dropout) whose input is a ReLU layer's output too,
// and the decoder's first transposed one (4x4 -> 8x8, ReLU, no dropout), on the block's 512-wide edges (imagevae/dsprites_vae.py:
// 18-20, 38-39).  Both or none.
static bool conv_fold_layer_ok(const arvae_layer_t &l, bool up) {
    const arvae_link_t &k = l.link;
    return (l.is_up != 0) == up && k.chi == 32 && k.clo == 32 && k.kh == 4 && k.kw == 4 && k.stride == 2 && k.pad == 1 && k.hh == 8 &&
           k.hw == 8 && k.lh == 4 && k.lw == 4 && k.hi_perm_c == 0 && k.lo_perm_c == 0 && l.act == ARVAE_ACT_RELU && l.dropout == 0;
}
static bool midc_fold_topology(const arvae_image_vae_t *m, int ne, int nd) {
    static const bool off = diag_env("ARVAE_MIDC_NO_FOLD") != nullptr;       // diagnostic build: the four launches of round 4
    const int e = m->n_enc - ne - 1;                             // the conv layer in front of the block
    if (off || !midc_topology(m, ne, nd) || e < 1 || nd + 1 >= m->n_dec) return false;
    const arvae_layer_t &before = m->enc[e - 1];
    return conv_fold_layer_ok(m->enc[e], false) && conv_fold_layer_ok(m->dec[nd], true) && before.act == ARVAE_ACT_RELU &&
           before.dropout == 0 && m->dec[nd - 1].act == ARVAE_ACT_RELU;
}
bool mid_fold_fits(const arvae_image_vae_t *m, int batch) {
    int ne, nd;
    if (!mid_fusable(m, &ne, &nd) || !midc_fold_topology(m, ne, nd)) return false;
    return midc_use(true, batch, m->flags);
}
// floats of weight-gradient slabs each folded layer needs: one per workgroup of the clustered grid
int64_t mid_fold_slab_floats(const arvae_image_vae_t *m, int batch) {
    int ne, nd;
    if (!mid_fusable(m, &ne, &nd) || !midc_fold_topology(m, ne, nd)) return 0;
    return (int64_t)((batch + MC_R - 1) / MC_R) * MC_S * (32 * 32 + 32);    // (a tap block + bias sums per member: reduce.h, SLAB_C32T)
}

// floats of workspace for the prepped matrices of the block's layers
int64_t mid_prep_floats(const arvae_image_vae_t *m) {
    int ne, nd;
    if (!mid_fusable(m, &ne, &nd)) return 0;
    int64_t total = 0;
    for (int i = 0; i < ne; ++i) total += mid_layer_floats(m->enc[m->n_enc - ne + i]);
    for (int i = 0; i < nd; ++i) total += mid_layer_floats(m->dec[i]);
    const int64_t h = m->head_mu.link.chi, z2 = 2 * m->zdim;
    return total + 2 * ((h * z2 + 3) / 4 * 4) + (z2 + 3) / 4 * 4 + midc_floats(m, ne, nd);  // heads: [h][2z] | [2z][h] | bias; cluster layouts
}

struct MidPlan {
    MidArgs args;
    MidPrepArgs prep;
    bool cluster;                // the clustered kernels can take this model (and `cl` holds their matrices)
    bool fold;                   // ... and compute the conv layers on either side of the block too (cl.cv_e / cl.cv_d are set)
    McArgs cl;
    size_t lds_bytes;
    int rows;                    // batch rows per workgroup: 8 when two row buffers of that height fit LDS beside the scratch, else 4
    // round 6: wide layers on the tile GEMMs (dense.hip wide_gemm_x3_kernel): the block's first encoder layer / last decoder layer
    bool wide_e, wide_d;
};

// fills the layer tables from the model description; y / gpre buffers are given per layer by the caller afterwards
static int mid_cu_count() { return device_cu_count(); }

// The clustered kernels take the pass when the model has their shape, the caller has not switched them off
// (ARVAE_VAE_NO_CLUSTER: a hand-off gave up earlier in this run, arvae_image_vae_t.status) and the whole grid can be resident at
// once by the runtime's own occupancy answer for these kernels (one 512-thread workgroup with ~117 KB of LDS per CU).  Residency
// is a matter of speed, not of correctness: places are handed out by tickets (midcluster.hip), so a grid that finds part of the
// device taken still completes.
static bool midc_use(bool cluster_ok, int batch, int flags) {
    static const bool rows_only = diag_env("ARVAE_MID_NO_CLUSTER") != nullptr;       // diagnostic build: the row kernels of this file
    const int64_t clusters = (batch + MC_R - 1) / MC_R;
    return !rows_only && cluster_ok && !(flags & ARVAE_VAE_NO_CLUSTER) && clusters <= MC_MAX_CLUSTERS &&
           clusters * MC_S <= midc_resident_capacity();
}
static void mid_describe(const arvae_image_vae_t *m, const float *params, float *prep_ws, MidPlan &pl, int batch = 512) {
    int ne, nd;
    mid_fusable(m, &ne, &nd);
    MidArgs &a = pl.args;
    a = MidArgs{};
    a.ne = ne; a.nd = nd; a.zdim = m->zdim; a.h = m->head_mu.link.chi;
    pl.prep.count = 0;
    int64_t off = 0;
    int blocks = 0, maxw = 0;
    auto add = [&](const arvae_layer_t &l, MidLayer &ml) {
        const int k = l.link.chi, n = l.link.clo, kb = (k + 3) / 4 * 4;
        ml.k = k; ml.n = n; ml.act = l.act; ml.kb = kb;
        float *mf = prep_ws + off, *mb = mf + ((int64_t)k * n + 3) / 4 * 4, *bias = mb + (int64_t)kb * n;
        off += mid_layer_floats(l);
        ml.mf = mf; ml.mb = mb; ml.bias = l.b_off >= 0 ? bias : nullptr;
        MidPrepJob &j = pl.prep.job[pl.prep.count];
        j = MidPrepJob{};
        j.w = params + l.w_off; j.b = l.b_off >= 0 ? params + l.b_off : nullptr;
        j.mf = mf; j.mb = mb; j.bias = l.b_off >= 0 ? bias : nullptr;
        j.k = k; j.n = n; j.kb = kb;
        j.kp = Perm{l.link.hi_perm_c, l.link.hi_perm_hw};
        j.np = Perm{l.link.lo_perm_c, l.link.lo_perm_hw};
        int b = (kb * n + 1023) / 1024;                           // ~4 elements per thread
        if (b > 256) b = 256;
        blocks += b;
        pl.prep.blk_end[pl.prep.count++] = blocks;
        if (k > maxw) maxw = k;
        if (n > maxw) maxw = n;
    };
    for (int i = 0; i < ne; ++i) add(m->enc[m->n_enc - ne + i], a.enc[i]);
    for (int i = 0; i < nd; ++i) add(m->dec[i], a.dec[i]);
    // Wide layers go to the tile GEMMs (not for the dSprites-shaped block: the clustered kernels take that one whole); they keep
    // ONE prepared copy, mb = W'[n_mem][k_mem], which the GEMMs read both ways
    {
        static const bool no_wide = diag_env("ARVAE_MID_NO_WIDE") != nullptr;     // diagnostic build: every layer on the row kernels
        const bool shape_ok = !no_wide && !midc_topology(m, ne, nd);
        pl.wide_e = shape_ok && ne >= 1 && a.enc[0].k >= MID_WIDE_MIN && a.enc[0].k % 4 == 0 && a.enc[0].n % 4 == 0;
        pl.wide_d = shape_ok && nd >= 2 && a.dec[nd - 1].n >= MID_WIDE_MIN && a.dec[nd - 1].k % 4 == 0 && a.dec[nd - 1].n % 4 == 0;
        // their ONE prepared copy: the three bf16 terms of W'[n_mem][k_mem] as planes, in the space of the fp32 layouts they replace
        auto to_planes = [&](MidLayer &ml, MidPrepJob &j) {
            const int n_pad = (ml.n + 31) / 32 * 32, kb_pad = (ml.k + 31) / 32 * 32;
            if ((int64_t)3 * n_pad * kb_pad * 2 > ((int64_t)ml.k * ml.n + (int64_t)ml.kb * ml.n) * 4) return false;
            unsigned short *planes = reinterpret_cast<unsigned short *>(j.mf);
            ml.w_planes = planes; ml.n_pad = n_pad; ml.kb_pad = kb_pad;
            j.planes = planes; j.n_pad = n_pad; j.kb_pad = kb_pad;
            ml.mf = ml.mb = nullptr;
            j.mf = j.mb = nullptr;
            return true;
        };
        if (pl.wide_e) pl.wide_e = to_planes(a.enc[0], pl.prep.job[0]);
        if (pl.wide_d) pl.wide_d = to_planes(a.dec[nd - 1], pl.prep.job[ne + nd - 1]);
    }
    {   // the two heads as one [h] -> [2 zdim] layer
        const int h = m->head_mu.link.chi, z2 = 2 * m->zdim;
        float *hf = prep_ws + off, *hb = hf + ((int64_t)h * z2 + 3) / 4 * 4, *hbias = hb + ((int64_t)h * z2 + 3) / 4 * 4;
        a.hf = hf; a.hb = hb; a.hbias = hbias;
        MidPrepJob &j = pl.prep.job[pl.prep.count];
        j = MidPrepJob{};
        j.w = params + m->head_mu.w_off; j.b = m->head_mu.b_off >= 0 ? params + m->head_mu.b_off : nullptr;
        j.w2 = params + m->head_log_std.w_off; j.b2 = m->head_log_std.b_off >= 0 ? params + m->head_log_std.b_off : nullptr;
        j.nsplit = m->zdim;
        j.mf = hf; j.mb = hb; j.bias = hbias;
        j.k = h; j.n = z2; j.kb = h;
        j.kp = Perm{0, 0}; j.np = Perm{0, 0};
        blocks += (h * z2 + 1023) / 1024;
        pl.prep.blk_end[pl.prep.count++] = blocks;
        off += 2 * (((int64_t)h * z2 + 3) / 4 * 4) + (z2 + 3) / 4 * 4;
    }
    pl.prep.counters = nullptr;
    pl.prep.counter_words = 0;
    pl.cluster = midc_topology(m, ne, nd);
    pl.cl = McArgs{};
    pl.fold = false;
    pl.prep.n_conv = 0;
    pl.prep.conv[0] = pl.prep.conv[1] = McConvPrep{nullptr, nullptr, nullptr};
    const bool use_c_early = pl.cluster && midc_use(true, batch, m->flags);
    if (pl.cluster) {
        // cluster layouts behind everything else in the prep workspace; job order: enc0, enc1, dec0, dec1, dec2, heads
        McMat *fw[6] = {&pl.cl.e0f, &pl.cl.e1f, &pl.cl.d0f, &pl.cl.d1f, &pl.cl.d2f, &pl.cl.hdf};
        McMat *bw[6] = {&pl.cl.e0b, &pl.cl.e1b, &pl.cl.d0b, &pl.cl.d1b, &pl.cl.d2b, &pl.cl.hdb};
        for (int q = 0; q < 6; ++q) {
            MidPrepJob &j = pl.prep.job[q];
            const bool heads = q == 5, first_dec = q == 2;
            const int kp = (j.k + 15) / 16 * 16, np = heads ? 32 : (j.n + 15) / 16 * 16;
            // partitioned over the 16 members where the output axis is wide; the z-sized products are whole in every member
            j.cf = prep_ws + off; off += (int64_t)kp * np;
            j.cb = prep_ws + off; off += (int64_t)kp * np;
            j.cf_kb = kp / 16; j.cf_s = (heads || first_dec) ? 1 : MC_S; j.cf_ct = np / 16 / j.cf_s;
            j.cb_kb = np / 16; j.cb_s = (heads || first_dec) ? 1 : MC_S; j.cb_ct = kp / 16 / j.cb_s;
            fw[q]->w = j.cf; fw[q]->bias = j.bias;
            bw[q]->w = j.cb; bw[q]->bias = nullptr;
        }
        // the folded conv layers' matrices (made whenever the clustered kernels run and the model has the layers)
        pl.fold = use_c_early && midc_fold_topology(m, ne, nd);
        pl.prep.n_conv = 0;
        pl.prep.conv[0] = pl.prep.conv[1] = McConvPrep{nullptr, nullptr, nullptr};
        {
            const arvae_layer_t *cl[2] = {&m->enc[m->n_enc - ne - 1], &m->dec[nd]};
            McConv *cv[2] = {&pl.cl.cv_e, &pl.cl.cv_d};
            for (int q = 0; q < 2; ++q) {
                float *down = prep_ws + off, *up = down + 16384;
                off += 2 * 16384;
                if (!pl.fold) continue;
                pl.prep.conv[q] = McConvPrep{params + cl[q]->w_off, down, up};
                *cv[q] = McConv{down, up, cl[q]->b_off >= 0 ? params + cl[q]->b_off : nullptr};
            }
            if (pl.fold) pl.prep.n_conv = 2;
        }
        pl.prep.counters = reinterpret_cast<unsigned *>(prep_ws + off);
        pl.prep.counter_words = (int)MIDC_COUNTER_WORDS;
        pl.cl.counters = pl.prep.counters;
        pl.cl.status = m->status;
        off += MIDC_COUNTER_WORDS;
        pl.cl.zdim = m->zdim;
        // one family of layouts per step: the cluster layouts when the clustered kernels take this batch, else the row kernels'
        const bool use_c = midc_use(true, batch, m->flags);
        for (int q = 0; q < 6; ++q) {
            MidPrepJob &j = pl.prep.job[q];
            if (use_c) j.mf = j.mb = nullptr;
            else j.cf = j.cb = nullptr;
        }
        if (!use_c) { pl.prep.counters = nullptr; pl.prep.counter_words = 0; }
        pl.cl.act_e0 = a.enc[0].act; pl.cl.act_e1 = a.enc[1].act;
        pl.cl.act_d0 = a.dec[0].act; pl.cl.act_d1 = a.dec[1].act; pl.cl.act_d2 = a.dec[2].act;
    }
    a.ld = ((maxw + 3) / 4) * 4 + 4;
    a.w_mu = params + m->head_mu.w_off; a.b_mu = m->head_mu.b_off >= 0 ? params + m->head_mu.b_off : nullptr;
    a.w_ls = params + m->head_log_std.w_off; a.b_ls = m->head_log_std.b_off >= 0 ? params + m->head_log_std.b_off : nullptr;
    // Batch rows per workgroup: every workgroup streams every matrix (~115 GB/s, the L2 -> CU rate of one CU), so the layer time is
    // that stream plus the FMA / LDS work of its rows -- as few rows as still give every CU at most ONE workgroup: 1 / 2 / 4 rows
    // for B <= 256 / 512 / more on 256 CUs (B = 512: forward 32.0 -> 28.8 us, backward 34.6 -> 31.0 us against 4 rows; 8 rows per
    // workgroup measured 49 vs 36 us forward: the FMA work per workgroup doubles).  ARVAE_MID_ROWS=n overrides.
    {
        static const int forced = diag_env("ARVAE_MID_ROWS") != nullptr ? atoi(diag_env("ARVAE_MID_ROWS")) : 0;
        const int cus = mid_cu_count();
        pl.rows = batch > 2 * cus ? 4 : batch > cus ? 2 : 1;
        if (forced == 1 || forced == 2 || forced == 4) pl.rows = forced;
    }
    pl.lds_bytes = (size_t)(2 * pl.rows * a.ld + mid_red(pl.rows) + pl.rows * 32) * sizeof(float);
}

// LDS row pitch and size for the layers a pass runs inside the row kernel (a wide layer the tile GEMMs took is not one of them)
static void mid_size_lds(MidPlan &pl, bool skip_enc0, bool skip_dec_last) {
    MidArgs &a = pl.args;
    int maxw = max(a.h, 2 * a.zdim);
    for (int i = 0; i < a.ne; ++i) {
        if (!(i == 0 && skip_enc0)) maxw = max(maxw, a.enc[i].k);
        maxw = max(maxw, a.enc[i].n);
    }
    for (int i = 0; i < a.nd; ++i) {
        maxw = max(maxw, a.dec[i].k);
        if (!(i == a.nd - 1 && skip_dec_last)) maxw = max(maxw, a.dec[i].n);
    }
    a.ld = ((maxw + 3) / 4) * 4 + 4;
    pl.lds_bytes = (size_t)(2 * pl.rows * a.ld + mid_red(pl.rows) + pl.rows * 32) * sizeof(float);
}

static int64_t mid_wide_plane_floats(int64_t batch, int64_t w) { return (3 * batch * w / 2 + 3) / 4 * 4; }
// workspace of the split reductions the tile GEMMs leave for the row kernels: WIDE_MAX_SLICES partial products of the narrow side
int64_t mid_wide_ws_floats(const arvae_image_vae_t *m, int batch) {
    int ne, nd;
    if (!mid_fusable(m, &ne, &nd) || midc_topology(m, ne, nd)) return 0;
    int64_t w = 0;
    if (ne >= 1 && m->enc[m->n_enc - ne].link.chi >= MID_WIDE_MIN) w = max(w, (int64_t)m->enc[m->n_enc - ne].link.clo);
    if (nd >= 2 && m->dec[nd - 1].link.clo >= MID_WIDE_MIN) w = max(w, (int64_t)m->dec[nd - 1].link.chi);
    // partial products | planes of the encoder layer's pre-activation gradient | planes of the decoder layer's input (1.5 floats per value)
    return w > 0 ? (int64_t)WIDE_MAX_SLICES * batch * w + 2 * mid_wide_plane_floats(batch, w) : 0;
}

static void midc_common(McArgs &c, const MidArgs &a, int batch) {
    c.batch = batch;
    c.clusters = (batch + MC_R - 1) / MC_R;
    c.heads = c.clusters % 4 == 0 ? 4 : (c.clusters % 2 == 0 ? 2 : 1);
    if (const char *hd = diag_env("ARVAE_MIDC_HEADS")) { const int v = atoi(hd); if (v >= 1 && v <= 8 && c.clusters % v == 0) c.heads = v; }
    c.debug_drop = diag_env("ARVAE_MIDC_DROP_ARRIVAL") != nullptr;
    c.debug_static = diag_env("ARVAE_MIDC_STATIC") != nullptr;
    c.wait_ticks = midc_wait_ticks();
    if (const char *ms = diag_env("ARVAE_MIDC_WAIT_MS")) c.wait_ticks = (unsigned long long)atoll(ms) * 100000ull;
    c.y_e0 = a.enc[0].y; c.y_e1 = a.enc[1].y; c.y_d0 = a.dec[0].y; c.y_d1 = a.dec[1].y; c.y_d2 = a.dec[2].y;
    c.g_e0 = a.enc[0].gpre; c.g_e1 = a.enc[1].gpre; c.g_d0 = a.dec[0].gpre; c.g_d1 = a.dec[1].gpre; c.g_d2 = a.dec[2].gpre;
}

static void mid_allow_lds() {
    static std::once_flag once;
    std::call_once(once, [] {
        const int bytes4 = (2 * 4 * (MID_MAX_W + 4) + mid_red(4) + 4 * 32) * (int)sizeof(float), bytes8 = 120 * 1024;
        static_assert((2 * 4 * (MID_MAX_W + 4) + mid_red(4) + 4 * 32) * sizeof(float) <= 160 * 1024, "the 4-row block must fit the LDS");
        (void)hipFuncSetAttribute((const void *)mid_forward_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes4);
        (void)hipFuncSetAttribute((const void *)mid_backward_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes4);
        (void)hipFuncSetAttribute((const void *)mid_forward_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes4);
        (void)hipFuncSetAttribute((const void *)mid_backward_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes4);
        (void)hipFuncSetAttribute((const void *)mid_forward_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes4);
        (void)hipFuncSetAttribute((const void *)mid_backward_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes4);
        (void)hipFuncSetAttribute((const void *)mid_forward_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes8);
        (void)hipFuncSetAttribute((const void *)mid_backward_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, bytes8);
    });
}

// the regions of the wide layers' workspace (mid_wide_ws_floats)
struct MidWideWs { float *partial; unsigned short *g_planes, *y_planes; };
static MidWideWs mid_wide_regions(const MidPlan &pl, int batch, float *ws) {
    const MidArgs &a = pl.args;
    MidWideWs r{ws, nullptr, nullptr};
    if (ws == nullptr) return r;
    int64_t w = 0;
    if (pl.wide_e) w = max(w, (int64_t)a.enc[0].n);
    if (pl.wide_d) w = max(w, (int64_t)a.dec[a.nd - 1].k);
    float *g = ws + (int64_t)WIDE_MAX_SLICES * batch * w;
    r.g_planes = reinterpret_cast<unsigned short *>(g);
    r.y_planes = reinterpret_cast<unsigned short *>(g + mid_wide_plane_floats(batch, w));
    return r;
}

// the prep launch's arguments alone (plan.hip hands them to conv32_weight_prep, which runs both preps as one launch)
void mid_prep_args(const arvae_image_vae_t *m, const float *params, float *prep_ws, MidPrepArgs *out, int batch) {
    MidPlan pl;
    mid_describe(m, params, prep_ws, pl, batch);
    *out = pl.prep;
}

// (1) weight layout prep unless prep_done, (2) the forward block.  enc_y / dec_y: saved outputs of the block's layers.
int mid_forward(const arvae_image_vae_t *m, int batch, const float *params, float *prep_ws, const float *x0, float *const *enc_y,
                float *const *dec_y, const float *eps, float *mu, float *log_std, float *sigma, float *z, hipStream_t s, bool prep_done,
                unsigned *amax_out, const MidFold *fold, float *wide_ws) {
    MidPlan pl;
    mid_describe(m, params, prep_ws, pl, batch);
    MidArgs &a = pl.args;
    // the wide layers' tile GEMMs (dense.hip): F1 = x0 . W'_e0^T as a split reduction this kernel's prologue finishes; F2 behind it
    WideGemm f1{}, f2{};
    int f1_slices = 1;
    const bool wide_e = pl.wide_e, wide_d = pl.wide_d;
    ARVAE_REQUIRE(!(wide_e || wide_d) || wide_ws != nullptr, "mid_forward: the wide layers' workspace is missing");
    const MidWideWs wws = mid_wide_regions(pl, batch, wide_ws);
    if (wide_e) {
        const MidLayer &l = a.enc[0];
        f1.a = x0; f1.lda = l.k; f1.b = l.w_planes; f1.ldb = l.n_pad; f1.b_pstride = (int64_t)l.n_pad * l.kb_pad; f1.b_planes = 1; f1.b_krows = 0;
        f1.M = batch; f1.N = l.n; f1.K = l.k;
        f1.out = wws.partial; f1.ldo = l.n; f1.slice_floats = (int64_t)batch * l.n;
        f1_slices = wide_gemm_slices(batch, l.n, l.k);
        ARVAE_REQUIRE(wide_gemm_fits(f1, f1_slices), "mid_forward: the first Linear layer does not fit the tile GEMM (alignment / size)");
    }
    if (wide_d) {
        const MidLayer &l = a.dec[a.nd - 1];
        a.dec[a.nd - 2].y_planes = wws.y_planes;             // the row kernel leaves its last layer's output pre-split for F2
        f2.a = wws.y_planes; f2.lda = batch; f2.a_pstride = (int64_t)batch * l.k; f2.a_planes = 1;
        f2.b = l.w_planes; f2.ldb = l.n_pad; f2.b_pstride = (int64_t)l.n_pad * l.kb_pad; f2.b_planes = 1; f2.b_krows = 0;
        f2.M = batch; f2.N = l.n; f2.K = l.k;
        f2.out = dec_y[a.nd - 1]; f2.ldo = l.n; f2.bias = l.bias; f2.act = l.act; f2.amax_out = amax_out;
        if (!wide_gemm_fits(f2, 1)) f2.amax_out = nullptr;       // (more tiles than AMAX entries: a launch of its own below)
        ARVAE_REQUIRE(wide_gemm_fits(f2, 1), "mid_forward: the last Linear layer does not fit the tile GEMM (alignment / size)");
    }
    a.skip_enc0 = wide_e; a.skip_dec_last = wide_d;
    if (wide_e) { a.wide_partial = wws.partial; a.wide_slices = f1_slices; a.wide_slice_floats = f1.slice_floats; }
    mid_size_lds(pl, wide_e, wide_d);
    // AMAX of the last output: one writer unit per workgroup when they fit the array, else a reduction launch of its own
    const bool amax_in_kernel = (batch + pl.rows - 1) / pl.rows <= AMAX_N;
    a.amax_out = amax_in_kernel ? amax_out : nullptr;
    for (int i = 0; i < a.ne; ++i) a.enc[i].y = enc_y[i];
    for (int i = 0; i < a.nd; ++i) a.dec[i].y = dec_y[i];
    a.batch = batch; a.x0 = x0;
    a.mu = mu; a.log_std = log_std; a.sigma = sigma; a.z = z; a.eps = eps;
    if (m->rng_eps) {
        a.eps_out = const_cast<float *>(eps);
        a.rng = RngStream{m->rng_seed, m->rng_offset, m->rng_dev_step, m->rng_step};
    }
    {   // forward streams enc[0].mf first; everything after it is requested up front
        auto lines = [](int64_t floats) { return (int)((floats + 31) / 32); };
        int &nw = a.n_warm;
        nw = 0;
        for (int i = 1; i < a.ne; ++i) { a.warm_ptr[nw] = a.enc[i].mf; a.warm_lines[nw++] = lines((int64_t)a.enc[i].k * a.enc[i].n); }
        a.warm_ptr[nw] = a.hf; a.warm_lines[nw++] = lines((int64_t)a.h * 2 * a.zdim);
        for (int i = 0; i < a.nd - (wide_d ? 1 : 0); ++i) { a.warm_ptr[nw] = a.dec[i].mf; a.warm_lines[nw++] = lines((int64_t)a.dec[i].k * a.dec[i].n); }
        if (diag_env("ARVAE_MID_NO_WARM") != nullptr) nw = 0;
    }
    mid_allow_lds();
    if (!prep_done) {
        ARVAE_LAUNCH(mid_prep_kernel, dim3(mid_prep_blocks(pl.prep)), dim3(256), 0, s, pl.prep);
        if (int rc = check_launch("mid_prep_kernel")) return rc;
    }
    if (wide_e)
        if (int rc = wide_gemm(f1, f1_slices, true, s)) return rc;
    if (midc_use(pl.cluster, batch, m->flags)) {
        McArgs &c = pl.cl;
        midc_common(c, a, batch);
        c.x0 = x0; c.mu = mu; c.log_std = log_std; c.sigma = sigma; c.z = z; c.eps = eps;
        c.eps_out = a.eps_out; c.rng = a.rng;
        c.amax_out = amax_out;                               // one writer unit per workgroup: at most 256 of them
        ARVAE_REQUIRE((fold != nullptr) == pl.fold, "mid_forward: the executor and the latent block disagree about the folded conv layers");
        if (fold != nullptr) {                               // x0 is written by this launch, y_d2 goes on to hi_d
            c.fold = 1;
            c.hi_e = fold->hi_e; c.x0_out = const_cast<float *>(x0);
            c.hi_d = fold->hi_d; c.hi_d_bits = fold->hi_d_bits; c.hi_d_amax = fold->hi_d_amax;
        }
        return midc_forward(c, s);
    }
    ARVAE_REQUIRE(fold == nullptr, "mid_forward: folded conv layers need the clustered kernels");
    if (pl.rows == 1) ARVAE_LAUNCH(mid_forward_kernel<1>, dim3(batch), dim3(MID_T), pl.lds_bytes, s, a);
    else if (pl.rows == 2) ARVAE_LAUNCH(mid_forward_kernel<2>, dim3((batch + 1) / 2), dim3(MID_T), pl.lds_bytes, s, a);
    else if (pl.rows == 8) ARVAE_LAUNCH(mid_forward_kernel<8>, dim3((batch + 7) / 8), dim3(MID_T), pl.lds_bytes, s, a);
    else ARVAE_LAUNCH(mid_forward_kernel<4>, dim3((batch + 3) / 4), dim3(MID_T), pl.lds_bytes, s, a);
    if (int rc = check_launch("mid_forward_kernel")) return rc;
    if (wide_d) {
        if (int rc = wide_gemm(f2, 1, false, s)) return rc;
        if (amax_out != nullptr && f2.amax_out == nullptr) return conv32_amax(dec_y[a.nd - 1], (int64_t)batch * a.dec[a.nd - 1].n, amax_out, s);
        return ARVAE_OK;
    }
    if (amax_out != nullptr && !amax_in_kernel) return conv32_amax(dec_y[a.nd - 1], (int64_t)batch * a.dec[a.nd - 1].n, amax_out, s);
    return ARVAE_OK;
}

int mid_backward(const arvae_image_vae_t *m, int batch, const float *params, float *prep_ws, float *const *enc_y, float *const *dec_y,
                 float *const *enc_g, float *const *dec_g, const float *g_out, int g_is_pre, const float *gate0, float *d_x0,
                 const float *eps, const float *mu, const float *sigma, const float *dz_reg, const float *dz_extra, const float *g_loss,
                 const float *kl, const float *cap, float beta, float reg_scale, float *d_mu, float *d_ls, hipStream_t s,
                 unsigned *amax_out, const MidFold *fold, float *wide_ws) {
    MidPlan pl;
    mid_describe(m, params, prep_ws, pl, batch);
    MidArgs &a = pl.args;
    // the wide layers' data gradients on the tile GEMMs (dense.hip): B1 = g_last . W'_d(last) as a split reduction this kernel's
    // prologue finishes (needs the gradient w.r.t. the last layer's PRE-activation: else that layer stays on the row kernel);
    // B2 = enc[0]'s pre-activation gradient . W'_e0 behind this kernel
    WideGemm b1{}, b2{};
    int b1_slices = 1;
    const bool wide_d = pl.wide_d, wide_e = pl.wide_e;
    ARVAE_REQUIRE(!(wide_e || wide_d) || wide_ws != nullptr, "mid_backward: the wide layers' workspace is missing");
    const MidWideWs wws = mid_wide_regions(pl, batch, wide_ws);
    if (wide_d) {
        const MidLayer &l = a.dec[a.nd - 1];
        if (!g_is_pre) {
            // the split reduction multiplies a PRE-activation gradient: make it (into the layer's keep buffer, where the weight
            // gradient looks for it) with the operand pass the per-layer path uses
            const arvae_operand_t op{g_out, dec_y[a.nd - 1], nullptr, l.act};
            if (int rc = arvae_operand_apply(&op, (int64_t)batch * l.n, dec_g[a.nd - 1], reinterpret_cast<arvae_stream_t>(s))) return rc;
            g_out = dec_g[a.nd - 1];
            g_is_pre = 1;
        }
        b1.a = g_out; b1.lda = l.n; b1.b = l.w_planes; b1.ldb = l.n_pad; b1.b_pstride = (int64_t)l.n_pad * l.kb_pad; b1.b_planes = 1; b1.b_krows = 1;
        b1.M = batch; b1.N = l.k; b1.K = l.n;
        b1.out = wws.partial; b1.ldo = l.k; b1.slice_floats = (int64_t)batch * l.k;
        b1_slices = wide_gemm_slices(batch, l.k, l.n);
        ARVAE_REQUIRE(wide_gemm_fits(b1, b1_slices), "mid_backward: the last Linear layer does not fit the tile GEMM (alignment / size)");
    }
    if (wide_e) {
        const MidLayer &l = a.enc[0];
        a.enc[0].g_planes = wws.g_planes;                    // the row kernel leaves this layer's pre-activation gradient pre-split for B2
        b2.a = wws.g_planes; b2.lda = batch; b2.a_pstride = (int64_t)batch * l.n; b2.a_planes = 1;
        b2.b = l.w_planes; b2.ldb = l.n_pad; b2.b_pstride = (int64_t)l.n_pad * l.kb_pad; b2.b_planes = 1; b2.b_krows = 1;
        b2.M = batch; b2.N = l.k; b2.K = l.n;
        b2.out = d_x0; b2.ldo = l.k; b2.act = ARVAE_ACT_NONE; b2.gate = gate0; b2.amax_out = amax_out;
        if (!wide_gemm_fits(b2, 1)) b2.amax_out = nullptr;
        ARVAE_REQUIRE(wide_gemm_fits(b2, 1), "mid_backward: the first Linear layer does not fit the tile GEMM (alignment / size)");
    }
    a.skip_enc0 = wide_e; a.skip_dec_last = wide_d;
    if (wide_d) { a.wide_partial = wws.partial; a.wide_slices = b1_slices; a.wide_slice_floats = b1.slice_floats; }
    mid_size_lds(pl, wide_e, wide_d);
    const bool amax_in_kernel = (batch + pl.rows - 1) / pl.rows <= AMAX_N;
    a.amax_out = amax_in_kernel ? amax_out : nullptr;
    for (int i = 0; i < a.ne; ++i) { a.enc[i].y = enc_y[i]; a.enc[i].gpre = enc_g[i]; }
    for (int i = 0; i < a.nd; ++i) { a.dec[i].y = dec_y[i]; a.dec[i].gpre = dec_g[i]; }
    a.batch = batch;
    a.g_out = g_out; a.g_is_pre = g_is_pre; a.gate0 = gate0; a.d_x0 = d_x0;
    a.eps = eps; a.mu = const_cast<float *>(mu); a.sigma = const_cast<float *>(sigma);
    a.dz_reg = dz_reg; a.dz_extra = dz_extra; a.g_loss = g_loss; a.kl = kl; a.cap = cap;
    a.beta = beta; a.inv_batch = 1.f / (float)batch; a.reg_scale = reg_scale;
    a.d_mu = d_mu; a.d_ls = d_ls;
    {   // backward streams dec[nd-1].mb first (dec[0].mb only when it is not that one), then the heads' and the encoder's matrices
        auto lines = [](int64_t floats) { return (int)((floats + 31) / 32); };
        int &nw = a.n_warm;
        nw = 0;
        for (int i = a.nd - 2 - (wide_d ? 1 : 0); i >= 0; --i) { a.warm_ptr[nw] = a.dec[i].mb; a.warm_lines[nw++] = lines((int64_t)a.dec[i].kb * a.dec[i].n); }
        a.warm_ptr[nw] = a.hb; a.warm_lines[nw++] = lines((int64_t)a.h * 2 * a.zdim);
        for (int i = a.ne - 1; i >= (wide_e ? 1 : 0); --i) { a.warm_ptr[nw] = a.enc[i].mb; a.warm_lines[nw++] = lines((int64_t)a.enc[i].kb * a.enc[i].n); }
        if (diag_env("ARVAE_MID_NO_WARM") != nullptr) nw = 0;
    }
    mid_allow_lds();
    if (wide_d)
        if (int rc = wide_gemm(b1, b1_slices, true, s)) return rc;
    if (midc_use(pl.cluster, batch, m->flags)) {
        McArgs &c = pl.cl;
        midc_common(c, a, batch);
        c.g_out = g_out; c.g_is_pre = g_is_pre; c.gate0 = gate0; c.d_x0 = d_x0;
        c.eps = eps; c.mu = a.mu; c.sigma = a.sigma;
        c.dz_reg = dz_reg; c.dz_extra = dz_extra; c.g_loss = g_loss; c.kl = kl; c.cap = cap;
        c.beta = beta; c.inv_batch = a.inv_batch; c.reg_scale = reg_scale; c.d_mu = d_mu; c.d_ls = d_ls;
        c.amax_out = amax_out;
        ARVAE_REQUIRE((fold != nullptr) == pl.fold, "mid_backward: the executor and the latent block disagree about the folded conv layers");
        if (fold != nullptr) {
            c.fold = 1;
            c.hi_e = fold->hi_e; c.g_hi_d = fold->g_hi_d; c.d_hi_e = fold->d_hi_e; c.d_hi_e_amax = fold->d_hi_e_amax;
            c.slab_e = fold->slab_e; c.slab_d = fold->slab_d;
        }
        return midc_backward(c, s);
    }
    ARVAE_REQUIRE(fold == nullptr, "mid_backward: folded conv layers need the clustered kernels");
    if (pl.rows == 1) ARVAE_LAUNCH(mid_backward_kernel<1>, dim3(batch), dim3(MID_T), pl.lds_bytes, s, a);
    else if (pl.rows == 2) ARVAE_LAUNCH(mid_backward_kernel<2>, dim3((batch + 1) / 2), dim3(MID_T), pl.lds_bytes, s, a);
    else if (pl.rows == 8) ARVAE_LAUNCH(mid_backward_kernel<8>, dim3((batch + 7) / 8), dim3(MID_T), pl.lds_bytes, s, a);
    else ARVAE_LAUNCH(mid_backward_kernel<4>, dim3((batch + 3) / 4), dim3(MID_T), pl.lds_bytes, s, a);
    if (int rc = check_launch("mid_backward_kernel")) return rc;
    if (wide_e) {
        if (int rc = wide_gemm(b2, 1, false, s)) return rc;
        if (amax_out != nullptr && b2.amax_out == nullptr) return conv32_amax(d_x0, (int64_t)batch * a.enc[0].k, amax_out, s);
        return ARVAE_OK;
    }
    if (amax_out != nullptr && !amax_in_kernel) return conv32_amax(d_x0, (int64_t)batch * a.enc[0].k, amax_out, s);
    return ARVAE_OK;
}

// The wide layers' weight gradients (dense.hip wide_wgrad_x3_kernel), called by the executor behind mid_backward: the encoder
// layer's from its pre-activation gradient (planes, left by mid_backward) and the block's input x0; the decoder layer's from the
// gradient at its pre-activation and its input (planes, left by mid_forward).  *took: bit 0 / 1 = the first encoder / last decoder
// layer was done here (the caller keeps them out of the grouped launch).
int mid_wide_wgrad(const arvae_image_vae_t *m, int batch, const float *params, float *prep_ws, float *wide_ws, const float *x0,
                   const float *g_last_pre, float *grads, hipStream_t s, int *took) {
    *took = 0;
    static const bool off = diag_env("ARVAE_MID_NO_WIDE_WGRAD") != nullptr;      // diagnostic build: the grouped 32 x 32-tile launch
    if (wide_ws == nullptr || off) return ARVAE_OK;
    MidPlan pl;
    mid_describe(m, params, prep_ws, pl, batch);
    if (!pl.wide_e && !pl.wide_d) return ARVAE_OK;
    const MidArgs &a = pl.args;
    const MidWideWs wws = mid_wide_regions(pl, batch, wide_ws);
    WideWgradJob jobs[2];
    int n = 0;
    if (pl.wide_e) {
        const arvae_layer_t &l = m->enc[m->n_enc - a.ne];
        const MidLayer &ml = a.enc[0];
        WideWgradJob j{};
        j.a = wws.g_planes; j.lda = batch; j.a_pstride = (int64_t)batch * ml.n; j.a_planes = 1;
        j.b = x0; j.ldb = ml.k; j.b_planes = 0;
        j.P = ml.n; j.Q = ml.k; j.R = batch;
        j.dw = grads + l.w_off; j.ldw = ml.k; j.dbias = l.b_off >= 0 ? grads + l.b_off : nullptr;
        j.p_perm = Perm{l.link.lo_perm_c, l.link.lo_perm_hw}; j.q_perm = Perm{l.link.hi_perm_c, l.link.hi_perm_hw};
        if (wide_wgrad_fits(j)) { jobs[n++] = j; *took |= 1; }
    }
    if (pl.wide_d) {
        const arvae_layer_t &l = m->dec[a.nd - 1];
        const MidLayer &ml = a.dec[a.nd - 1];
        WideWgradJob j{};
        j.a = g_last_pre; j.lda = ml.n; j.a_planes = 0;
        j.b = wws.y_planes; j.ldb = batch; j.b_pstride = (int64_t)batch * ml.k; j.b_planes = 1;
        j.P = ml.n; j.Q = ml.k; j.R = batch;
        j.dw = grads + l.w_off; j.ldw = ml.k; j.dbias = l.b_off >= 0 ? grads + l.b_off : nullptr;
        j.p_perm = Perm{l.link.lo_perm_c, l.link.lo_perm_hw}; j.q_perm = Perm{l.link.hi_perm_c, l.link.hi_perm_hw};
        if (wide_wgrad_fits(j)) { jobs[n++] = j; *took |= 2; }
    }
    return n > 0 ? wide_wgrad(jobs, n, s) : ARVAE_OK;
}

}  // namespace arvae

#ifdef MID_STAMPS
extern "C" int arvae_debug_mid_stamps(unsigned long long *out, int count) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(arvae::g_mid_stamps), sizeof(unsigned long long) * count);
}
#endif
