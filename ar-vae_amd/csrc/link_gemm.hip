// Strided-link kernels (conv / conv-transpose / linear; forward, data-gradient, weight-gradient)
// as gather-GEMMs on the gfx950 fp32 matrix core (v_mfma_f32_32x32x2_f32: exact fp32, same
// k-ordered fmaf chain as a scalar loop, 64 FLOP/clk/SIMD).
//
// One kernel template, three index policies:
//   Down  : lo[m=(n,ly,lx)][clo]  = sum_k hi(gather)[m][k=(ky,kx,chi)] * wt[k][clo]
//   Up    : hi[m=(n,yy,xx) of one stride-parity class][chi] = sum_k lo(gather)[m][k=(ty,tx,clo)] * wt[k][chi]
//   Wgrad : dwt[clo][n'=(ky,kx,chi)] += sum_{k=pixel} lo[k][clo] * hi(gather)[k][n']     (pixels split over
//           workgroups; partial tiles go to a slab and are summed in a fixed order -- no float atomics)
// Work decomposition: a workgroup of WM x WN wavefronts (64 lanes each) owns a (32*WM) x (32*WN)
// output tile; every wave accumulates ONE 32x32 tile in 16 accumulator registers.  Operands are
// gathered global -> registers (prefetched one K-tile ahead so the loads fly under the MFMAs) ->
// LDS in k-major order [BK][BM+1], from where each lane reads the single A and B value the
// 32x32x2 MFMA wants (lane = (row|col) + 32 * k-parity) with conflict-free ds_read_b32.
#include "diag.h"
#include "common.h"
#include "conv32_common.h"

namespace arvae {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;

// batch-sized nn.Linear kernels (dense.hip)
bool dense_fits(const arvae_link_t *l);
int dense_fwd(const arvae_link_t *l, const float *x, const float *w, const float *bias, int act, float *y, hipStream_t s);
int dense_dgrad(const arvae_link_t *l, const Operand &g, const float *w, const float *gate, float *dx, hipStream_t s);
int64_t dense_wgrad_ws_floats(const arvae_link_t *l);
int dense_wgrad(const arvae_link_t *l, const Operand &g, const float *x, float *dw, float *dbias, float *ws, hipStream_t s);

// 1-channel 64x64 image links (conv_c1.hip)
bool conv_c1_fits(const arvae_link_t *l);
bool conv_c1w_fits(const arvae_link_t *l);
int64_t conv_c1w_wgrad_ws_floats(const arvae_link_t *l);
int conv_c1w_wgrad(const arvae_link_t *l, const Operand &lo, const Operand &img, float *dwt, float *dbias, int bias_mode,
                   float *slab, hipStream_t s);
int conv_c1_down(const arvae_link_t *l, const Operand &img, const float *wt, const float *bias, int relu,
                 const float *gate, const uint16_t *gate_bits, uint16_t *bits_out, float *out, hipStream_t s, unsigned *amax_out);
int conv_c1_up(const arvae_link_t *l, const float *lo, const float *wt, const float *bias, float *out, hipStream_t s);
int64_t conv_c1_wgrad_ws_floats(const arvae_link_t *l);
int conv_c1_wgrad(const arvae_link_t *l, const Operand &lo, const Operand &img, float *dwt, float *dbias, int bias_mode,
                  float *slab, hipStream_t s);

// stride-1 wide-channel convolutions on the split-bf16 MFMA (conv64.hip)
bool conv64_fits(const arvae_link_t *l, bool up);
int64_t conv64_ws_floats(const arvae_link_t *l);
int conv64_down(const arvae_link_t *l, const Operand &hi, const float *wt, const float *bias, int act, const uint8_t *mask,
                float *lo, float *ws, hipStream_t s, const GateOp *gate, const unsigned *amax_in = nullptr, unsigned *amax_out = nullptr,
                float *prepped = nullptr);
int conv64_up(const arvae_link_t *l, const Operand &lo, const float *wt, const float *bias, int act, const uint8_t *mask,
              float *hi, float *ws, hipStream_t s, const GateOp *gate, const unsigned *amax_in = nullptr, unsigned *amax_out = nullptr,
              float *prepped = nullptr);
bool conv64_wgrad_fits(const arvae_link_t *l);
int64_t conv64_wgrad_ws_floats(const arvae_link_t *l);
int conv64_wgrad(const arvae_link_t *l, const Operand &lo, const Operand &hi, float *dwt, float *ws, hipStream_t s,
                 const unsigned *amax_lo, const unsigned *amax_hi, float *dbias, int bias_side, bool *bias_done);

// specialised 32-channel k4/s2/p1 kernels (conv32.hip)
bool conv32_fits(const arvae_link_t *l);
// (operands come with AMAX arrays and prepared weights: conv32_common.h; a per-layer caller has neither and makes them in `ws`)
int conv32_down(const arvae_link_t *l, const Operand &hi, const float *bias, int relu, const float *gate, const uint16_t *gate_bits,
                uint16_t *bits_out, float *out, hipStream_t s, const float *wprep, const unsigned *amax_in, unsigned *amax_out);
int conv32_up(const arvae_link_t *l, const Operand &lo, const float *bias, int relu, const float *gate, const uint16_t *gate_bits,
              uint16_t *bits_out, float *out, hipStream_t s, const float *wprep, const unsigned *amax_in, unsigned *amax_out);
int64_t conv32_wgrad_ws_floats(const arvae_link_t *l);
int conv32_wgrad(const arvae_link_t *l, const Operand &lo, const Operand &hi, float *dwt, float *dbias, int bias_mode,
                 float *slab, hipStream_t s, const unsigned *amax_lo, const unsigned *amax_hi);
int64_t conv32_prep_floats();
int64_t conv32_scratch_floats();
int conv32_weight_prep(const float *const *wts, float *const *preps, int n_layers, hipStream_t s);
int conv32_amax(const float *x, int64_t count, unsigned *out, hipStream_t s);
constexpr int64_t CONV32_AMAX_FLOATS = 1024;          // AMAX_N (conv32_common.h)

struct Geom {
    int n, hh, hw, chi, lh, lw, clo, kh, kw, stride, pad;
    int hi_pc, lo_pc;
    FastDiv d_chi, d_clo, d_kw, d_lw, d_lhlw, d_hi_phw, d_lo_phw;
    // Up only: taps per axis and class grid
    int tkh, tkw, yh, xw;
    FastDiv d_tkw, d_xw, d_yhxw;

    __device__ __forceinline__ int perm_hi(int f) const {
        if (hi_pc == 0) return f;
        uint32_t q, r;
        d_hi_phw.divmod((uint32_t)f, q, r);
        return (int)(r * hi_pc + q);
    }
    __device__ __forceinline__ int perm_lo(int f) const {
        if (lo_pc == 0) return f;
        uint32_t q, r;
        d_lo_phw.divmod((uint32_t)f, q, r);
        return (int)(r * lo_pc + q);
    }
};

struct Epilogue {
    const float *bias;
    const uint8_t *mask;
    float *out;
    int act;
    GateOp gate = GateOp{};      // down_single_channel_mfma_kernel only: result *= act'(gate.y) * 2 gate.mask at the output location
    unsigned *amax_out = nullptr;    // down_single_channel_mfma_kernel only: AMAX array of `out` (conv32_common.h), one writer unit per image
};

// gather context of a tensor position (n, y0, x0) and of a (ky, kx, channel) tap
struct PosCtx {
    int base, y0, x0;
    bool ok;
};
struct TapCtx {
    int ky, kx, coff;
    bool ok;
};
struct OffCtx {
    int off;
    bool ok;
};

// ------------------------------------------------------------------------------------------------
struct DownPolicy {
    static constexpr bool A_CONTIG_K = true, B_CONTIG_K = true;
    Geom g;
    Operand hi;
    const float *wt;
    Epilogue ep;
    int M, N, K;

    __device__ __forceinline__ void slice(int, int &m, int &n, int &kbeg, int &kend) const {
        m = M; n = N; kbeg = 0; kend = K;
    }
    typedef PosCtx RowA; typedef TapCtx KA; typedef OffCtx RowB; typedef OffCtx KB;
    __device__ __forceinline__ RowA rowA(int m, int) const {
        uint32_t img, rem, ly, lx;
        g.d_lhlw.divmod((uint32_t)m, img, rem);
        g.d_lw.divmod(rem, ly, lx);
        return RowA{(int)img * g.hh * g.hw * g.chi, (int)ly * g.stride - g.pad, (int)lx * g.stride - g.pad, m < M};
    }
    __device__ __forceinline__ KA kA(int k, int) const {
        uint32_t tap, c, ky, kx;
        g.d_chi.divmod((uint32_t)k, tap, c);
        g.d_kw.divmod(tap, ky, kx);
        return KA{(int)ky, (int)kx, g.perm_hi((int)c), k < K};
    }
    __device__ __forceinline__ float fetchA(const RowA &r, const KA &k) const {
        const int iy = r.y0 + k.ky, ix = r.x0 + k.kx;
        const bool ok = r.ok && k.ok && (unsigned)iy < (unsigned)g.hh && (unsigned)ix < (unsigned)g.hw;
        return ok ? hi.at(r.base + (iy * g.hw + ix) * g.chi + k.coff) : 0.f;
    }
    __device__ __forceinline__ RowB rowB(int col, int) const { return RowB{col * g.chi * g.kh * g.kw, col < N}; }
    __device__ __forceinline__ KB kB(int k, int) const {
        uint32_t tap, c;
        g.d_chi.divmod((uint32_t)k, tap, c);
        return KB{(int)c * g.kh * g.kw + (int)tap, k < K};
    }
    __device__ __forceinline__ float fetchB(const RowB &r, const KB &k) const {
        return (r.ok && k.ok) ? wt[r.off + k.off] : 0.f;
    }
    __device__ __forceinline__ int out_row(int m, int) const { return m * g.clo; }
    struct ColC { int off; float bias; };
    __device__ __forceinline__ ColC colC(int col, int) const {
        return ColC{g.perm_lo(col), ep.bias != nullptr ? ep.bias[col] : 0.f};
    }
    __device__ __forceinline__ void store(int rowoff, const ColC &c, float acc) const {
        const int idx = rowoff + c.off;
        float v = act_fwd(acc + c.bias, ep.act);
        if (ep.mask != nullptr) v *= 2.f * (float)ep.mask[idx];
        ep.out[idx] = v;
    }
};

// ------------------------------------------------------------------------------------------------
struct UpPolicy {
    static constexpr bool A_CONTIG_K = true, B_CONTIG_K = true;
    Geom g;
    Operand lo;
    const float *wt;
    Epilogue ep;
    int M, N, K;   // per stride-parity class

    __device__ __forceinline__ void slice(int, int &m, int &n, int &kbeg, int &kend) const {
        m = M; n = N; kbeg = 0; kend = K;
    }
    typedef PosCtx RowA; typedef TapCtx KA; typedef OffCtx RowB; typedef OffCtx KB;
    // z = cy * stride + cx ; ky0 = (cy+pad) % stride ; oy = (cy+pad) / stride
    __device__ __forceinline__ RowA rowA(int m, int z) const {
        const int cy = z / g.stride, cx = z - cy * g.stride;
        uint32_t img, rem, yy, xx;
        g.d_yhxw.divmod((uint32_t)m, img, rem);
        g.d_xw.divmod(rem, yy, xx);
        return RowA{(int)img * g.lh * g.lw * g.clo, (int)yy + (cy + g.pad) / g.stride, (int)xx + (cx + g.pad) / g.stride,
                    m < M};
    }
    __device__ __forceinline__ KA kA(int k, int) const {
        uint32_t tap, c, ty, tx;
        g.d_clo.divmod((uint32_t)k, tap, c);
        g.d_tkw.divmod(tap, ty, tx);
        return KA{(int)ty, (int)tx, g.perm_lo((int)c), k < K};
    }
    __device__ __forceinline__ float fetchA(const RowA &r, const KA &k) const {
        const int ly = r.y0 - k.ky, lx = r.x0 - k.kx;
        const bool ok = r.ok && k.ok && (unsigned)ly < (unsigned)g.lh && (unsigned)lx < (unsigned)g.lw;
        return ok ? lo.at(r.base + (ly * g.lw + lx) * g.clo + k.coff) : 0.f;
    }
    __device__ __forceinline__ RowB rowB(int col, int) const { return RowB{col * g.kh * g.kw, col < N}; }
    __device__ __forceinline__ KB kB(int k, int z) const {
        const int cy = z / g.stride, cx = z - cy * g.stride;
        uint32_t tap, c, ty, tx;
        g.d_clo.divmod((uint32_t)k, tap, c);
        g.d_tkw.divmod(tap, ty, tx);
        const int ky = (cy + g.pad) % g.stride + g.stride * (int)ty;
        const int kx = (cx + g.pad) % g.stride + g.stride * (int)tx;
        return KB{(int)c * g.chi * g.kh * g.kw + ky * g.kw + kx, k < K};
    }
    __device__ __forceinline__ float fetchB(const RowB &r, const KB &k) const {
        return (r.ok && k.ok) ? wt[r.off + k.off] : 0.f;
    }
    __device__ __forceinline__ int out_row(int m, int z) const {
        const int cy = z / g.stride, cx = z - cy * g.stride;
        uint32_t img, rem, yy, xx;
        g.d_yhxw.divmod((uint32_t)m, img, rem);
        g.d_xw.divmod(rem, yy, xx);
        return (((int)img * g.hh + (int)yy * g.stride + cy) * g.hw + (int)xx * g.stride + cx) * g.chi;
    }
    struct ColC { int off; float bias; };
    __device__ __forceinline__ ColC colC(int col, int) const {
        return ColC{g.perm_hi(col), ep.bias != nullptr ? ep.bias[col] : 0.f};
    }
    __device__ __forceinline__ void store(int rowoff, const ColC &c, float acc) const {
        const int idx = rowoff + c.off;
        float v = act_fwd(acc + c.bias, ep.act);
        if (ep.mask != nullptr) v *= 2.f * (float)ep.mask[idx];
        ep.out[idx] = v;
    }
};

// ------------------------------------------------------------------------------------------------
struct WgradPolicy {
    static constexpr bool A_CONTIG_K = false, B_CONTIG_K = false;
    Geom g;
    Operand lo, hi;
    float *dwt;     // zsplit == 1: accumulate here directly
    float *slab;    // zsplit  > 1: partial tiles [z][M][N], summed by wgrad_reduce_kernel in a fixed order
    int M, N, P, chunk, zsplit;   // M = clo, N = kh*kw*chi, P = n*lh*lw pixels, chunk = pixels per z-slice

    __device__ __forceinline__ void slice(int z, int &m, int &n, int &kbeg, int &kend) const {
        m = M; n = N; kbeg = z * chunk; kend = min(P, kbeg + chunk);
    }
    typedef OffCtx RowA; typedef OffCtx KA; typedef TapCtx RowB; typedef PosCtx KB;
    __device__ __forceinline__ RowA rowA(int m, int) const { return RowA{g.perm_lo(m), m < M}; }
    __device__ __forceinline__ KA kA(int pix, int) const { return KA{pix * g.clo, pix < P}; }   // kend checked by caller
    __device__ __forceinline__ float fetchA(const RowA &r, const KA &k) const {
        return (r.ok && k.ok) ? lo.at(k.off + r.off) : 0.f;
    }
    __device__ __forceinline__ RowB rowB(int col, int) const {
        uint32_t tap, c, ky, kx;
        g.d_chi.divmod((uint32_t)col, tap, c);
        g.d_kw.divmod(tap, ky, kx);
        return RowB{(int)ky, (int)kx, g.perm_hi((int)c), col < N};
    }
    __device__ __forceinline__ KB kB(int pix, int) const {
        uint32_t img, rem, ly, lx;
        g.d_lhlw.divmod((uint32_t)pix, img, rem);
        g.d_lw.divmod(rem, ly, lx);
        return KB{(int)img * g.hh * g.hw * g.chi, (int)ly * g.stride - g.pad, (int)lx * g.stride - g.pad, pix < P};
    }
    __device__ __forceinline__ float fetchB(const RowB &r, const KB &k) const {
        const int iy = k.y0 + r.ky, ix = k.x0 + r.kx;
        const bool ok = r.ok && k.ok && (unsigned)iy < (unsigned)g.hh && (unsigned)ix < (unsigned)g.hw;
        return ok ? hi.at(k.base + (iy * g.hw + ix) * g.chi + r.coff) : 0.f;
    }
    __device__ __forceinline__ int out_row(int m, int z) const {
        return zsplit > 1 ? (z * M + m) * N : m * g.chi * g.kh * g.kw;
    }
    struct ColC { int off; };
    __device__ __forceinline__ ColC colC(int col, int) const {
        if (zsplit > 1) return ColC{col};
        uint32_t tap, c;
        g.d_chi.divmod((uint32_t)col, tap, c);
        return ColC{(int)c * g.kh * g.kw + (int)tap};
    }
    __device__ __forceinline__ void store(int rowoff, const ColC &c, float acc) const {
        if (zsplit > 1)
            slab[rowoff + c.off] = acc;
        else
            dwt[rowoff + c.off] += acc;      // one owner per element: no atomics
    }
};

// dwt[clo][chi][ky][kx] += sum_z slab[z][clo][(ky,kx,chi)]   -- fixed summation order (bitwise reproducible)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ slab, int zsplit, int mn, int n,
                                                            int khkw, FastDiv d_n, FastDiv d_chi,
                                                            float *__restrict__ dwt) {
    __shared__ float red[4][64];
    const int il = threadIdx.x & 63, zg = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + il;
    float s = 0.f;
    if (i < mn)
        for (int z0 = zg; z0 < zsplit; z0 += 32) {           // 8 loads in flight, added in slab order
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int z = z0 + 4 * u;
                const float v = slab[(int64_t)(z < zsplit ? z : zg) * mn + i];
                t[u] = z < zsplit ? v : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t[u];
        }
    red[zg][il] = s;
    __syncthreads();
    if (zg == 0 && i < mn) {
        const float tot = (red[0][il] + red[1][il]) + (red[2][il] + red[3][il]);
        uint32_t m, col, tap, c;
        d_n.divmod((uint32_t)i, m, col);
        d_chi.divmod(col, tap, c);
        dwt[(int)m * n + (int)c * khkw + (int)tap] += tot;
    }
}

// ------------------------------------------------------------------------------------------------
template <int WM, int WN, class P>
__global__ __launch_bounds__(64 * WM * WN) void link_gemm_kernel(const P p) {
    constexpr int BM = 32 * WM, BN = 32 * WN, NT = 64 * WM * WN;
    constexpr int EA = BM * BK / NT, EB = BN * BK / NT;
    static_assert(NT % BK == 0 && NT % BM == 0 && NT % BN == 0, "tile/thread mapping");
    __shared__ float As[BK][BM + 1];
    __shared__ float Bs[BK][BN + 1];
    __shared__ int rowOff[BM];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int z = blockIdx.z;
    int M, N, kbeg, kend;
    p.slice(z, M, N, kbeg, kend);
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    if (m0 >= M || kbeg >= kend) return;

    if (t < BM) rowOff[t] = (m0 + t < M) ? p.out_row(m0 + t, z) : 0;

    // per-thread gather contexts that do not change over the K loop
    typename P::RowA rowsA[P::A_CONTIG_K ? EA : 1];
    typename P::RowB rowsB[P::B_CONTIG_K ? EB : 1];
    if constexpr (P::A_CONTIG_K) {
#pragma unroll
        for (int j = 0; j < EA; ++j) rowsA[j] = p.rowA(m0 + t / BK + j * (NT / BK), z);
    } else {
        rowsA[0] = p.rowA(m0 + t % BM, z);
    }
    if constexpr (P::B_CONTIG_K) {
#pragma unroll
        for (int j = 0; j < EB; ++j) rowsB[j] = p.rowB(n0 + t / BK + j * (NT / BK), z);
    } else {
        rowsB[0] = p.rowB(n0 + t % BN, z);
    }

    float ra[EA], rb[EB];
    auto gather = [&](int kt) {
        if constexpr (P::A_CONTIG_K) {
            const int k = kt + t % BK;
            const auto ka = p.kA(k < kend ? k : 0x3fffffff, z);
#pragma unroll
            for (int j = 0; j < EA; ++j) ra[j] = p.fetchA(rowsA[j], ka);
        } else {
#pragma unroll
            for (int j = 0; j < EA; ++j) {
                const int k = kt + t / BM + j * (NT / BM);
                ra[j] = p.fetchA(rowsA[0], p.kA(k < kend ? k : 0x3fffffff, z));
            }
        }
        if constexpr (P::B_CONTIG_K) {
            const int k = kt + t % BK;
            const auto kb = p.kB(k < kend ? k : 0x3fffffff, z);
#pragma unroll
            for (int j = 0; j < EB; ++j) rb[j] = p.fetchB(rowsB[j], kb);
        } else {
#pragma unroll
            for (int j = 0; j < EB; ++j) {
                const int k = kt + t / BN + j * (NT / BN);
                rb[j] = p.fetchB(rowsB[0], p.kB(k < kend ? k : 0x3fffffff, z));
            }
        }
    };
    auto stage = [&]() {
        if constexpr (P::A_CONTIG_K) {
#pragma unroll
            for (int j = 0; j < EA; ++j) As[t % BK][t / BK + j * (NT / BK)] = ra[j];
        } else {
#pragma unroll
            for (int j = 0; j < EA; ++j) As[t / BM + j * (NT / BM)][t % BM] = ra[j];
        }
        if constexpr (P::B_CONTIG_K) {
#pragma unroll
            for (int j = 0; j < EB; ++j) Bs[t % BK][t / BK + j * (NT / BK)] = rb[j];
        } else {
#pragma unroll
            for (int j = 0; j < EB; ++j) Bs[t / BN + j * (NT / BN)][t % BN] = rb[j];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    const int kh2 = lane >> 5, rc = lane & 31;
    gather(kbeg);
    for (int kt = kbeg; kt < kend; kt += BK) {
        __syncthreads();
        stage();
        __syncthreads();
        if (kt + BK < kend) gather(kt + BK);
#pragma unroll
        for (int s = 0; s < BK / 2; ++s) {
            const float a = As[2 * s + kh2][wm * 32 + rc];
            const float b = Bs[2 * s + kh2][wn * 32 + rc];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }

    const int col = n0 + wn * 32 + rc;
    if (col < N) {
        const auto cc = p.colC(col, z);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int r = (i & 3) + 8 * (i >> 2) + 4 * kh2;
            if (m0 + wm * 32 + r < M) p.store(rowOff[wm * 32 + r], cc, acc[i]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Up with a single hi channel (the decoder's last ConvTranspose2d -> logits): N = 1 would waste
// 31/32 of an MFMA tile, so this is a vector-ALU dot product: one lane per output pixel, the
// (ky,kx,clo) weights staged in LDS, lo pixels read as float4 channel chunks.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void up_single_channel_kernel(Geom g, const float *__restrict__ lo,
                                                                 const float *__restrict__ wt, Epilogue ep,
                                                                 int total) {
    extern __shared__ __attribute__((aligned(16))) float w_lds[];   // [kh*kw][clo]
    const int taps = g.kh * g.kw;
    for (int i = threadIdx.x; i < taps * g.clo; i += blockDim.x) {
        const int tap = i / g.clo, c = i - tap * g.clo;
        w_lds[i] = wt[c * taps + tap];   // wt[clo][chi=1][ky][kx]
    }
    __syncthreads();
    const float bias = ep.bias != nullptr ? ep.bias[0] : 0.f;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int hx = idx % g.hw;
        const int tmp = idx / g.hw;
        const int hy = tmp % g.hh;
        const int img = tmp / g.hh;
        float acc = 0.f;
        for (int ky = (hy + g.pad) % g.stride; ky < g.kh; ky += g.stride) {
            const int ly = (hy + g.pad - ky) / g.stride;
            if (hy + g.pad - ky < 0 || ly >= g.lh) continue;
            for (int kx = (hx + g.pad) % g.stride; kx < g.kw; kx += g.stride) {
                const int lx = (hx + g.pad - kx) / g.stride;
                if (hx + g.pad - kx < 0 || lx >= g.lw) continue;
                const float4 *src = reinterpret_cast<const float4 *>(lo + ((img * g.lh + ly) * g.lw + lx) * g.clo);
                const float4 *w4 = reinterpret_cast<const float4 *>(w_lds + (ky * g.kw + kx) * g.clo);
                for (int c = 0; c < g.clo / 4; ++c) {
                    const float4 a = src[c], b = w4[c];
                    acc = fmaf(a.x, b.x, acc);
                    acc = fmaf(a.y, b.y, acc);
                    acc = fmaf(a.z, b.z, acc);
                    acc = fmaf(a.w, b.w, acc);
                }
            }
        }
        float v = act_fwd(acc + bias, ep.act);
        if (ep.mask != nullptr) v *= 2.f * (float)ep.mask[idx];
        ep.out[idx] = v;
    }
}

// The same map on the matrix cores, one workgroup per image (used when the image's T fits LDS): first
// T[pos][tap] = sum_c lo[pos][c] * wt[c][tap] as a dense [positions x C] x [C x 16 taps] product on the 16x16x4 MFMA
// (lo read exactly once, 64-byte pieces per position straight into the MFMA lane layout), then every output pixel
// gathers its <= 16 contributions from the T image in LDS (col2im).  The lane-per-pixel version above re-reads each lo
// pixel 16 times through L1: 1.9 ms for the Morpho-MNIST logits layer (164 MB of lo) against ~0.1 ms here.
typedef float f32x4t __attribute__((ext_vector_type(4)));
template <int KQ>                                            // clo = 16 * KQ
__global__ __launch_bounds__(256) void up_single_channel_mfma_kernel(Geom g, const float *__restrict__ lo,
                                                                      const float *__restrict__ wt, Epilogue ep) {
    extern __shared__ __attribute__((aligned(16))) float t_lds[];   // [lh*lw][17]
    constexpr int TP = 17;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, quad = lane >> 4;
    const int taps = g.kh * g.kw, npos = g.lh * g.lw, clo = 16 * KQ;
    const int img = blockIdx.x;
    f32x4t b[KQ];
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq)
#pragma unroll
        for (int j = 0; j < 4; ++j) b[kq][j] = col < taps ? wt[(16 * kq + 4 * quad + j) * taps + col] : 0.f;
    const float *lo_img = lo + (int64_t)img * npos * clo;
    const int mtiles = (npos + 15) / 16;
    // two M-tiles per round, the NEXT round's 2 * KQ loads requested before this round's MFMAs (round 5: a wave's five rounds were
    // five exposed round trips to HBM; positions past the end clamp to the last one, so the requests need no condition)
    f32x4t a[2][KQ], an[2][KQ];
    auto request = [&](int mt0, f32x4t (&dst)[2][KQ]) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int pos = 16 * (mt0 + 4 * u) + col;
            const float *src = lo_img + (int64_t)(pos < npos ? pos : npos - 1) * clo + 4 * quad;
#pragma unroll
            for (int kq = 0; kq < KQ; ++kq) dst[u][kq] = *reinterpret_cast<const f32x4t *>(src + 16 * kq);
        }
    };
    request(wave, a);
    for (int mt0 = wave; mt0 < mtiles; mt0 += 8) {
        request(mt0 + 8, an);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int mt = mt0 + 4 * u;
            if (mt >= mtiles) break;
            f32x4t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kq = 0; kq < KQ; ++kq)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][kq][j], b[kq][j], acc, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int pos = 16 * mt + 4 * quad + i;
                if (pos < npos) t_lds[pos * TP + col] = acc[i];
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int kq = 0; kq < KQ; ++kq) a[u][kq] = an[u][kq];
    }
    __syncthreads();
    const float bias = ep.bias != nullptr ? ep.bias[0] : 0.f;
    const int hpix = g.hh * g.hw;
    for (int idx = threadIdx.x; idx < hpix; idx += 256) {
        const int hy = idx / g.hw, hx = idx - hy * g.hw;
        float acc = 0.f;
        for (int ky = (hy + g.pad) % g.stride; ky < g.kh; ky += g.stride) {
            const int ly = (hy + g.pad - ky) / g.stride;
            if (hy + g.pad - ky < 0 || ly >= g.lh) continue;
            for (int kx = (hx + g.pad) % g.stride; kx < g.kw; kx += g.stride) {
                const int lx = (hx + g.pad - kx) / g.stride;
                if (hx + g.pad - kx < 0 || lx >= g.lw) continue;
                acc += t_lds[(ly * g.lw + lx) * TP + ky * g.kw + kx];
            }
        }
        const int64_t o = (int64_t)img * hpix + idx;
        float v = act_fwd(acc + bias, ep.act);
        if (ep.mask != nullptr) v *= 2.f * (float)ep.mask[o];
        ep.out[o] = v;
    }
}

// Conv2d 1 -> 64 channels (and the data gradient of ConvTranspose2d 64 -> 1) on the 16x16x4 MFMA, one workgroup per
// image: the single-channel image sits in LDS, an M-tile is 16 output positions, K = the <= 16 taps (lane = (position,
// tap column) reads its patch value straight from the LDS image), N = the 64 channels in four tiles with the weights in
// registers: 36-113 us for the Morpho-MNIST layers against ~120 us on the generic gather-GEMM (their weight gradient
// stays there: 145 us, a one-workgroup-per-image MFMA version measured 370 us).
__global__ __launch_bounds__(256) void down_single_channel_mfma_kernel(Geom g, Operand hi, const float *__restrict__ wt,
                                                                        Epilogue ep) {
    extern __shared__ __attribute__((aligned(16))) float img_lds[];   // [hh*hw]
    // a wave's 16 positions x 64 channels meet here, so that the epilogue is 16-byte loads (gate values, four keep-mask bytes) and
    // 16-byte stores, a position's 256 bytes contiguous per 16 lanes -- with one dword store per (position, channel) and lane the
    // launch wrote its 164 MB at 2.5 TB/s
    __shared__ __attribute__((aligned(16))) float otile[4][16][68];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, quad = lane >> 4;
    const int taps = g.kh * g.kw, npos = g.lh * g.lw, hpix = g.hh * g.hw;
    const int img = blockIdx.x;
    if (hi.y == nullptr && hpix <= 4 * 256) {
        // a plain image: all of a thread's pixels requested at once (Operand::at's conditions make every iteration of the loop below
        // a round trip of its own)
        const __amdgpu_buffer_rsrc_t rs_img = make_rsrc(hi.v, (int64_t)gridDim.x * hpix * 4);
        float px[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = threadIdx.x + 256 * u;
            px[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_img, i < hpix ? (int)((img * hpix + i) * 4) : (int)OOB, 0, 0));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (threadIdx.x + 256 * u < hpix) img_lds[threadIdx.x + 256 * u] = px[u];
    } else {
        for (int i = threadIdx.x; i < hpix; i += 256) img_lds[i] = hi.at((int64_t)img * hpix + i);
    }
    // B[k = tap][n = channel]: lane (col = channel in tile, quad), k-step kk <-> tap = 4 kk + quad
    float b[4][4], bias[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) b[nt][kk] = 4 * kk + quad < taps ? wt[(16 * nt + col) * taps + 4 * kk + quad] : 0.f;
        bias[nt] = ep.bias != nullptr ? ep.bias[16 * nt + col] : 0.f;
    }
    // this lane's tap offsets inside the image (tap = 4 kk + quad)
    int toff[4], tky[4], tkx[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const int tap = 4 * kk + quad;
        tky[kk] = tap / g.kw; tkx[kk] = tap - tky[kk] * g.kw;
        toff[kk] = tky[kk] * g.hw + tkx[kk];
    }
    __syncthreads();
    const int mtiles = (npos + 15) / 16;
    float amax_run = 0.f;
    const ActCoef ac = act_coef(ep.act);
    const GateCoef gc = epilogue_coef(ep.gate.y, ep.gate.act, ep.gate.mask, ep.mask);
    // the epilogue's operands and stores as raw buffer operations with an out-of-range offset for "none" (round 5): under run-time
    // conditions (a mask? a gate? a position past the end?) every load sat in a block of its own and the compiler put a wait for the
    // wave's whole memory queue -- these requests and the previous tile's stores -- in front of the tile's first MFMA
    const uint8_t *mp = ep.gate.y != nullptr ? ep.gate.mask : ep.mask;          // (a gated launch has no forward keep-mask)
    const int64_t out_bytes = (int64_t)gridDim.x * npos * 64 * 4;
    const __amdgpu_buffer_rsrc_t rs_out = make_rsrc(ep.out, out_bytes);
    const __amdgpu_buffer_rsrc_t rs_mask = make_rsrc(mp != nullptr ? (const void *)mp : (const void *)ep.out, mp != nullptr ? out_bytes / 4 : 0);
    const __amdgpu_buffer_rsrc_t rs_gate = make_rsrc(ep.gate.y != nullptr ? (const void *)ep.gate.y : (const void *)ep.out, ep.gate.y != nullptr ? out_bytes : 0);
    const bool has_mask = mp != nullptr;
    for (int mt = wave; mt < mtiles; mt += 4) {
        const int pos = 16 * mt + col;                       // A row = position
        uint32_t ly, lx;
        g.d_lw.divmod((uint32_t)(pos < npos ? pos : npos - 1), ly, lx);
        const int y0 = (int)ly * g.stride - g.pad, x0 = (int)lx * g.stride - g.pad;
        f32x4t acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        // keep-mask bytes / gate values of this lane's four (position, 4 channels) slots: requested before the MFMAs
        const int e_pos = lane >> 4, e_c = 4 * (lane & 15);
        unsigned mk[4], oo[4];                               // (oo: the slot's element offset in the output, OOB / 4 for none)
        float4 gy[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = 16 * mt + e_pos + 4 * u;
            oo[u] = p < npos ? (unsigned)((img * npos + p) * 64 + e_c) : OOB / 4;
            const unsigned m = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs_mask, (int)oo[u], 0, 0);
            mk[u] = has_mask ? m : 0x01010101u;
            gy[u] = buf_load4(rs_gate, oo[u] == OOB / 4 ? OOB : oo[u] * 4u);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int y = y0 + tky[kk], x = x0 + tkx[kk];
            const bool ok = 4 * kk + quad < taps && y >= 0 && y < g.hh && x >= 0 && x < g.hw;
            const float a = ok ? img_lds[y0 * g.hw + x0 + toff[kk]] : 0.f;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[nt][kk], acc[nt], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) otile[wave][4 * quad + i][16 * nt + col] = acc[nt][i] + bias[nt];
        // (the tile is this wave's own: LDS operations of a wave complete in order)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 a4 = *reinterpret_cast<const float4 *>(&otile[wave][e_pos + 4 * u][e_c]);
            const bool live = oo[u] != OOB / 4;
            // activation, then gate / keep-mask in common.h's coefficient form (result *= d(y) * keep byte; no gate: d = 2 with a
            // keep-mask, else 1 with bytes of ones)
            float4 v = make_float4(act_fwd_coef(a4.x, ac), act_fwd_coef(a4.y, ac), act_fwd_coef(a4.z, ac), act_fwd_coef(a4.w, ac));
            const unsigned m = mk[u];
            v.x *= gate_deriv(gy[u].x, gc) * (float)(m & 255u);
            v.y *= gate_deriv(gy[u].y, gc) * (float)((m >> 8) & 255u);
            v.z *= gate_deriv(gy[u].z, gc) * (float)((m >> 16) & 255u);
            v.w *= gate_deriv(gy[u].w, gc) * (float)(m >> 24);
            amax_run = live ? fmaxf(amax_run, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)))) : amax_run;
            buf_store4(v, rs_out, live ? oo[u] * 4u : OOB);
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (ep.amax_out != nullptr) {                               // the wide convolution that reads `out` next scales it into fp16
        __shared__ float wm[4];
        amax_run = wave_max(amax_run);
        if (lane == 0) wm[wave] = amax_run;
        __syncthreads();
        if (threadIdx.x < 64) amax_publish(ep.amax_out, blockIdx.x, gridDim.x, fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])));
    }
}

static bool single_channel_mfma_fits(const arvae_link_t *l) {
    return l->chi == 1 && l->clo == 64 && l->kh * l->kw <= 16 && l->hi_perm_c == 0 && l->lo_perm_c == 0 &&
           (size_t)l->hh * l->hw * sizeof(float) <= 64 * 1024 && diag_env("ARVAE_C1_GENERIC") == nullptr &&
           (int64_t)l->n * l->lh * l->lw * 64 * 4 < 0x7fff0000ll;       // (32-bit byte offsets into the 64-channel tensor)
}

// bias gradients: out[feature(c)] += sum_rows g[row, c], two fixed-order stages (no float atomics)
//   stage 1: each workgroup reduces a row range into partial[block][c]   (channel index fastest: coalesced)
//   stage 2: one workgroup column-sums the partials the same way and applies the NCHW-flatten permutation
template <bool FINISH>
__global__ __launch_bounds__(256) void channel_sum_kernel(Operand g, int64_t rows, int channels,
                                                           int64_t rows_per_block, int perm_c, int perm_hw,
                                                           float *__restrict__ dst) {
    __shared__ float red[256];
    const int t = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(rows, r0 + rows_per_block);
    for (int cbase = 0; cbase < channels; cbase += 256) {
        const int cw = min(channels - cbase, 256);
        const int rstep = 256 / cw > 0 ? 256 / cw : 1;
        const int c = t % cw, rsub = t / cw;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        if (rsub < rstep) {
            int64_t r = r0 + rsub;
            const int64_t step = rstep;
            for (; r + 3 * step < r1; r += 4 * step) {       // 4 independent loads in flight per lane
                s0 += g.at(r * channels + cbase + c);
                s1 += g.at((r + step) * channels + cbase + c);
                s2 += g.at((r + 2 * step) * channels + cbase + c);
                s3 += g.at((r + 3 * step) * channels + cbase + c);
            }
            for (; r < r1; r += step) s0 += g.at(r * channels + cbase + c);
        }
        red[t] = (s0 + s1) + (s2 + s3);
        __syncthreads();
        if (t < cw) {
            float tot = 0.f;
            for (int j = 0; j < rstep; ++j) tot += red[j * cw + t];
            const int cm = cbase + t;
            if (FINISH) {
                int f = cm;                      // memory channel -> flattened NCHW feature
                if (perm_c > 0) f = (cm % perm_c) * perm_hw + cm / perm_c;
                dst[f] += tot;
            } else {
                dst[(int64_t)blockIdx.x * channels + cm] = tot;
            }
        }
        __syncthreads();
    }
}

// stage 1 for a plain operand (no activation derivative / keep-mask to fold) whose channel count is a multiple of 4 and divides
// 1024: 16-byte loads, eight rows in flight per lane -- the scalar kernel above walked the 25 MB gradients of the Morpho-MNIST
// stack at 0.8 TB/s (20-32 us per launch, four launches per step).  Partials as above; summation order fixed.
__global__ __launch_bounds__(256) void channel_sum4_kernel(const float4 *__restrict__ g, int64_t rows, int c4, int64_t rows_per_block,
                                                            float *__restrict__ dst) {
    __shared__ float4 red[256];
    const int t = threadIdx.x;
    const int rstep = 256 / c4;                           // rows per sweep of the workgroup
    const int c = t % c4, rsub = t / c4;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block, r1 = min(rows, r0 + rows_per_block);
    float4 s[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) s[u] = float4{0.f, 0.f, 0.f, 0.f};
    for (int64_t r = r0 + rsub; r < r1; r += 8 * rstep) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t rr = r + (int64_t)u * rstep;
            const float4 x = g[(rr < r1 ? rr : r0) * c4 + c];
            v[u] = rr < r1 ? x : float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s[u].x += v[u].x + v[u + 4].x; s[u].y += v[u].y + v[u + 4].y;
            s[u].z += v[u].z + v[u + 4].z; s[u].w += v[u].w + v[u + 4].w;
        }
    }
    red[t] = float4{(s[0].x + s[1].x) + (s[2].x + s[3].x), (s[0].y + s[1].y) + (s[2].y + s[3].y), (s[0].z + s[1].z) + (s[2].z + s[3].z),
                    (s[0].w + s[1].w) + (s[2].w + s[3].w)};
    __syncthreads();
    if (t < c4) {
        float4 tot = red[t];
        for (int j = 1; j < rstep; ++j) {
            const float4 x = red[j * c4 + t];
            tot.x += x.x; tot.y += x.y; tot.z += x.z; tot.w += x.w;
        }
        reinterpret_cast<float4 *>(dst + (int64_t)blockIdx.x * c4 * 4)[t] = tot;
    }
}

// stage 2 with the parallelism the partials allow: 16 channels per workgroup, 16 row lanes per channel, eight loads in flight
// per thread (one workgroup walking all the partials four loads at a time took 23 us for 1024 x 64 partials: five such launches
// per Morpho-MNIST step)
__global__ __launch_bounds__(256) void channel_sum_finish_kernel(const float *__restrict__ part, int blocks, int channels, int perm_c,
                                                                  int perm_hw, float *__restrict__ dst) {
    __shared__ float red[256];
    const int t = threadIdx.x, cl = t & 15, rsub = t >> 4;
    const int c = blockIdx.x * 16 + cl;
    const bool cok = c < channels;
    const float *base = part + (cok ? c : 0);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int r = rsub; r < blocks; r += 16 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int rr = r + 16 * u;
            const float x = base[(int64_t)(rr < blocks ? rr : 0) * channels];
            v[u] = rr < blocks ? x : 0.f;
        }
        s0 += v[0] + v[4];
        s1 += v[1] + v[5];
        s2 += v[2] + v[6];
        s3 += v[3] + v[7];
    }
    red[t] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (t < 16 && cok) {
        float tot = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) tot += red[j * 16 + t];
        int f = c;                               // memory channel -> flattened NCHW feature
        if (perm_c > 0) f = (c % perm_c) * perm_hw + c / perm_c;
        dst[f] += tot;
    }
}

// ------------------------------------------------------------------------------------------------
static int make_geom(const arvae_link_t *l, Geom &g) {
    ARVAE_REQUIRE(l != nullptr, "link: null descriptor");
    ARVAE_REQUIRE(l->n > 0 && l->hh > 0 && l->hw > 0 && l->chi > 0 && l->lh > 0 && l->lw > 0 && l->clo > 0,
                  "link: non-positive extent");
    ARVAE_REQUIRE(l->kh > 0 && l->kw > 0 && l->stride > 0 && l->pad >= 0, "link: bad kernel/stride/pad");
    ARVAE_REQUIRE((l->hh + 2 * l->pad - l->kh) / l->stride + 1 == l->lh &&
                      (l->hw + 2 * l->pad - l->kw) / l->stride + 1 == l->lw,
                  "link: lo extent %dx%d does not match hi %dx%d k%dx%d s%d p%d", l->lh, l->lw, l->hh, l->hw, l->kh,
                  l->kw, l->stride, l->pad);
    ARVAE_REQUIRE((int64_t)l->n * l->hh * l->hw * l->chi < (1ll << 31) &&
                      (int64_t)l->n * l->lh * l->lw * l->clo < (1ll << 31),
                  "link: tensor exceeds 2^31 elements");
    ARVAE_REQUIRE(l->hi_perm_c == 0 || l->hi_perm_c * l->hi_perm_hw == l->chi, "link: hi perm does not cover chi");
    ARVAE_REQUIRE(l->lo_perm_c == 0 || l->lo_perm_c * l->lo_perm_hw == l->clo, "link: lo perm does not cover clo");
    g.n = l->n; g.hh = l->hh; g.hw = l->hw; g.chi = l->chi; g.lh = l->lh; g.lw = l->lw; g.clo = l->clo;
    g.kh = l->kh; g.kw = l->kw; g.stride = l->stride; g.pad = l->pad;
    g.hi_pc = l->hi_perm_c; g.lo_pc = l->lo_perm_c;
    g.d_chi = FastDiv(l->chi); g.d_clo = FastDiv(l->clo); g.d_kw = FastDiv(l->kw); g.d_lw = FastDiv(l->lw);
    g.d_lhlw = FastDiv(l->lh * l->lw);
    g.d_hi_phw = FastDiv(l->hi_perm_c ? l->hi_perm_hw : 1);
    g.d_lo_phw = FastDiv(l->lo_perm_c ? l->lo_perm_hw : 1);
    g.tkh = g.tkw = g.yh = g.xw = 1;
    g.d_tkw = g.d_xw = g.d_yhxw = FastDiv(1);
    return ARVAE_OK;
}

template <class P>
static int launch_gemm(const P &p, int M, int N, int zdim, bool wide_m, hipStream_t s, const char *what) {
    // wide_m: many rows, N small  -> 4x1 waves (128x32) when N<=32, else 2x2 (64x64)
    // !wide_m (wgrad): few rows   -> 1x4 waves (32x128) when M<=32, else 2x2
    if (wide_m && N <= 32) {
        dim3 grid((M + 127) / 128, (N + 31) / 32, zdim);
        ARVAE_LAUNCH((link_gemm_kernel<4, 1, P>), grid, dim3(256), 0, s, p);
    } else if (!wide_m && M <= 32) {
        dim3 grid((M + 31) / 32, (N + 127) / 128, zdim);
        ARVAE_LAUNCH((link_gemm_kernel<1, 4, P>), grid, dim3(256), 0, s, p);
    } else {
        dim3 grid((M + 63) / 64, (N + 63) / 64, zdim);
        ARVAE_LAUNCH((link_gemm_kernel<2, 2, P>), grid, dim3(256), 0, s, p);
    }
    return check_launch(what);
}

}  // namespace arvae

using namespace arvae;

// experiment switch: Linear layers over >= 2048 rows (the MeasureVAE's whole-sequence GEMMs) on the LDS-staged generic kernels
static bool dense_rows_generic(const arvae_link_t *l, int which) {
    static const int mode = diag_env("ARVAE_DENSE_GENERIC") ? atoi(diag_env("ARVAE_DENSE_GENERIC")) : 0;
    return l->n >= 2048 && (mode & which) != 0;
}

extern "C" int64_t arvae_link_ws_floats(const arvae_link_t *link) {
    if (link == nullptr) return 0;
    if (conv32_fits(link)) return conv32_scratch_floats();       // one layer's prepared weights + the input's AMAX array
    return (conv64_fits(link, false) || conv64_fits(link, true)) ? conv64_ws_floats(link) : 0;
}

namespace arvae {
// a per-layer call on the 32-channel kernels: split this layer's weights and take the input's maxima into the caller's scratch
static int conv32_make_operands(const float *wt, const float *x, int64_t count, float *ws, hipStream_t st, const float **wprep,
                                const unsigned **amax) {
    ARVAE_REQUIRE(ws != nullptr, "link_down / link_up: workspace of arvae_link_ws_floats() floats needed");
    float *prep = ws;
    unsigned *am = reinterpret_cast<unsigned *>(ws + (conv32_prep_floats() + 3) / 4 * 4);
    if (int rc = conv32_weight_prep(&wt, &prep, 1, st)) return rc;
    if (int rc = conv32_amax(x, count, am, st)) return rc;
    *wprep = prep;
    *amax = am;
    return ARVAE_OK;
}
}  // namespace arvae

namespace arvae {
// data gradient of ConvTranspose2d(64 -> 1) with the producing layer's activation derivative / keep-mask in the epilogue
// (plan.hip: the next layer then reads a plain pre-activation gradient and needs no operand pass)
bool single_channel_down_gated_fits(const arvae_link_t *l) { return single_channel_mfma_fits(l); }
int single_channel_down_gated(const arvae_link_t *link, const Operand &hi, const float *wt, const GateOp *gate, float *lo, hipStream_t s,
                              unsigned *amax_out) {
    DownPolicy p;
    if (int rc = make_geom(link, p.g)) return rc;
    Epilogue ep{nullptr, nullptr, lo, ARVAE_ACT_NONE};
    ep.gate = *gate;
    ep.amax_out = link->n <= 1024 ? amax_out : nullptr;         // (AMAX_N writer units)
    ARVAE_LAUNCH(down_single_channel_mfma_kernel, dim3(link->n), dim3(256), sizeof(float) * link->hh * link->hw, s, p.g, hi, wt, ep);
    return check_launch("link_down(single channel, mfma)");
}
// the forward form (bias, activation, keep-mask) for the whole-model executor, which wants the result's maxima published
bool single_channel_down_fits(const arvae_link_t *l) { return single_channel_mfma_fits(l) && l->n <= 1024; }
int single_channel_down(const arvae_link_t *link, const Operand &hi, const float *wt, const float *bias, int act, const uint8_t *mask,
                        float *lo, hipStream_t s, unsigned *amax_out) {
    DownPolicy p;
    if (int rc = make_geom(link, p.g)) return rc;
    Epilogue ep{bias, mask, lo, act};
    ep.amax_out = amax_out;
    ARVAE_LAUNCH(down_single_channel_mfma_kernel, dim3(link->n), dim3(256), sizeof(float) * link->hh * link->hw, s, p.g, hi, wt, ep);
    return check_launch("link_down(single channel, mfma)");
}
}  // namespace arvae

extern "C" int arvae_link_down(const arvae_link_t *link, const arvae_operand_t *hi, const float *wt,
                               const float *bias, int32_t out_act, const uint8_t *out_mask, float *lo, float *ws,
                               arvae_stream_t stream) {
    DownPolicy p;
    if (int rc = make_geom(link, p.g)) return rc;
    ARVAE_REQUIRE(hi && hi->v && wt && lo, "link_down: null pointer");
    if (dense_fits(link) && out_mask == nullptr && hi->y == nullptr && !dense_rows_generic(link, 1))
        return dense_fwd(link, hi->v, wt, bias, out_act, lo, as_stream(stream));
    if (conv32_fits(link) && out_mask == nullptr && hi->y == nullptr && out_act != ARVAE_ACT_SELU) {
        const float *wprep;
        const unsigned *amax;
        if (int rc = conv32_make_operands(wt, hi->v, (int64_t)link->n * link->hh * link->hw * link->chi, ws, as_stream(stream), &wprep, &amax))
            return rc;
        return conv32_down(link, make_operand(hi), bias, out_act == ARVAE_ACT_RELU, nullptr, nullptr, nullptr, lo, as_stream(stream), wprep,
                           amax, nullptr);
    }
    if (conv_c1_fits(link) && out_mask == nullptr && hi->mask == nullptr && out_act != ARVAE_ACT_SELU &&
        hi->act != ARVAE_ACT_SELU)
        return conv_c1_down(link, make_operand(hi), wt, bias, out_act == ARVAE_ACT_RELU, nullptr, nullptr, nullptr, lo, as_stream(stream), nullptr);
    if (conv64_fits(link, false))
        return conv64_down(link, make_operand(hi), wt, bias, out_act, out_mask, lo, ws, as_stream(stream), nullptr);
    if (single_channel_mfma_fits(link)) {
        ARVAE_LAUNCH(down_single_channel_mfma_kernel, dim3(link->n), dim3(256), sizeof(float) * link->hh * link->hw,
                           as_stream(stream), p.g, make_operand(hi), wt, Epilogue{bias, out_mask, lo, out_act});
        return check_launch("link_down(single channel, mfma)");
    }
    p.hi = make_operand(hi);
    p.wt = wt;
    p.ep = Epilogue{bias, out_mask, lo, out_act};
    p.M = link->n * link->lh * link->lw;
    p.N = link->clo;
    p.K = link->kh * link->kw * link->chi;
    return launch_gemm(p, p.M, p.N, 1, true, as_stream(stream), "link_down");
}

extern "C" int arvae_link_up(const arvae_link_t *link, const arvae_operand_t *lo, const float *wt,
                             const float *bias, int32_t out_act, const uint8_t *out_mask, float *hi, float *ws,
                             arvae_stream_t stream) {
    UpPolicy p;
    if (int rc = make_geom(link, p.g)) return rc;
    ARVAE_REQUIRE(lo && lo->v && wt && hi, "link_up: null pointer");
    const int s = link->stride;
    ARVAE_REQUIRE(link->kh % s == 0 && link->kw % s == 0, "link_up: kernel %dx%d not a multiple of stride %d",
                  link->kh, link->kw, s);
    ARVAE_REQUIRE(link->hh % s == 0 && link->hw % s == 0, "link_up: hi extent not a multiple of the stride");
    hipStream_t st = as_stream(stream);
    if (dense_fits(link) && out_mask == nullptr && bias == nullptr && out_act == ARVAE_ACT_NONE && !dense_rows_generic(link, 2))
        return dense_dgrad(link, make_operand(lo), wt, nullptr, hi, st);
    if (conv32_fits(link) && out_mask == nullptr && lo->y == nullptr && out_act != ARVAE_ACT_SELU) {
        const float *wprep;
        const unsigned *amax;
        if (int rc = conv32_make_operands(wt, lo->v, (int64_t)link->n * link->lh * link->lw * link->clo, ws, st, &wprep, &amax)) return rc;
#ifdef ARVAE_STAMPS
        static const float *stamp_gate = nullptr;                // diagnostic build: time the gated variant too
        if (diag_env("ARVAE_STAMP_GATE") != nullptr) {
            if (stamp_gate == nullptr) (void)hipMalloc((void **)&stamp_gate, (size_t)link->n * link->hh * link->hw * link->chi * 4);
            return conv32_up(link, make_operand(lo), nullptr, 0, stamp_gate, nullptr, nullptr, hi, st, wprep, amax, nullptr);
        }
#endif
        return conv32_up(link, make_operand(lo), bias, out_act == ARVAE_ACT_RELU, nullptr, nullptr, nullptr, hi, st, wprep, amax, nullptr);
    }
    if (conv_c1_fits(link) && lo->y == nullptr && out_mask == nullptr && out_act == ARVAE_ACT_NONE)
        return conv_c1_up(link, lo->v, wt, bias, hi, st);
    if (conv64_fits(link, true))
        return conv64_up(link, make_operand(lo), wt, bias, out_act, out_mask, hi, ws, st, nullptr);
    if (link->chi == 1 && lo->y == nullptr && link->clo % 4 == 0 && link->lo_perm_c == 0) {
        const int total = link->n * link->hh * link->hw;
        Epilogue ep{bias, out_mask, hi, out_act};
        const size_t t_bytes = sizeof(float) * 17 * link->lh * link->lw;
        if ((link->clo == 64 || link->clo == 32) && link->kh * link->kw <= 16 && t_bytes <= 96 * 1024 &&
            diag_env("ARVAE_UP1_NAIVE") == nullptr) {
            if (link->clo == 64) ARVAE_LAUNCH(up_single_channel_mfma_kernel<4>, dim3(link->n), dim3(256), t_bytes, st, p.g, lo->v, wt, ep);
            else ARVAE_LAUNCH(up_single_channel_mfma_kernel<2>, dim3(link->n), dim3(256), t_bytes, st, p.g, lo->v, wt, ep);
            return check_launch("link_up(single channel, mfma)");
        }
        const int blocks = min((total + 255) / 256, 256 * 8);
        const size_t lds = sizeof(float) * link->kh * link->kw * link->clo;
        ARVAE_LAUNCH(up_single_channel_kernel, dim3(blocks), dim3(256), lds, st, p.g, lo->v, wt, ep, total);
        return check_launch("link_up(single channel)");
    }
    p.lo = make_operand(lo);
    p.wt = wt;
    p.ep = Epilogue{bias, out_mask, hi, out_act};
    p.g.tkh = link->kh / s;
    p.g.tkw = link->kw / s;
    p.g.yh = link->hh / s;
    p.g.xw = link->hw / s;
    p.g.d_tkw = FastDiv(p.g.tkw);
    p.g.d_xw = FastDiv(p.g.xw);
    p.g.d_yhxw = FastDiv(p.g.yh * p.g.xw);
    p.M = link->n * p.g.yh * p.g.xw;
    p.N = link->chi;
    p.K = p.g.tkh * p.g.tkw * link->clo;
    return launch_gemm(p, p.M, p.N, s * s, true, st, "link_up");
}

static void wgrad_split(const arvae_link_t *link, int &m, int &n, int &pix, int &zsplit, int &chunk) {
    m = link->clo;
    n = link->kh * link->kw * link->chi;
    pix = link->n * link->lh * link->lw;
    const int bm = m <= 32 ? 32 : 64, bn = m <= 32 ? 128 : 64;
    const int tiles = ((m + bm - 1) / bm) * ((n + bn - 1) / bn);
    zsplit = (1024 + tiles - 1) / tiles;          // ~4 workgroups per CU
    const int max_split = (pix + BK - 1) / BK;
    if (zsplit > max_split) zsplit = max_split;
    if (zsplit < 1) zsplit = 1;
    chunk = (((pix + zsplit - 1) / zsplit) + BK - 1) / BK * BK;
    zsplit = (pix + chunk - 1) / chunk;
}

static void channel_sum_split(int64_t rows, int64_t &blocks, int64_t &rpb);
static int channel_sum_launch(const Operand &g, int64_t rows, int channels, int perm_c, int perm_hw, float *out,
                              float *ws, hipStream_t st);

namespace arvae {
// weight + bias gradients of a wide stride-1 link (conv64.hip); amax_*: AMAX arrays of the operands when the caller has them
// (plain operands only), else null
int link_wgrad_conv64(const arvae_link_t *link, const Operand &lo, const Operand &hi, float *dwt, float *dbias, int bias_side, float *ws,
                      hipStream_t st, const unsigned *amax_lo, const unsigned *amax_hi) {
    bool bias_done = false;
    if (int rc = conv64_wgrad(link, lo, hi, dwt, ws, st, amax_lo, amax_hi, dbias, bias_side, &bias_done)) return rc;
    if (bias_done) return ARVAE_OK;
    if (bias_side == 1) return channel_sum_launch(lo, (int64_t)link->n * link->lh * link->lw, link->clo, 0, 0, dbias, ws, st);
    if (bias_side == 2) return channel_sum_launch(hi, (int64_t)link->n * link->hh * link->hw, link->chi, 0, 0, dbias, ws, st);
    return ARVAE_OK;
}
}  // namespace arvae

static bool wgrad_fast(const arvae_link_t *l, const arvae_operand_t *lo, const arvae_operand_t *hi) {
    // the 32-channel kernels take plain operands only (a gradient that still needs act'(y) goes the generic way)
    return conv32_fits(l) && (lo == nullptr || lo->y == nullptr) && (hi == nullptr || hi->y == nullptr);
}

extern "C" int64_t arvae_link_wgrad_ws_floats(const arvae_link_t *link) {
    if (link == nullptr || link->n <= 0) return 0;
    int m, n, pix, zsplit, chunk;
    wgrad_split(link, m, n, pix, zsplit, chunk);
    int64_t need = zsplit > 1 ? (int64_t)zsplit * m * n : 0;
    // bias sums share the workspace (stream ordered): the larger of the lo- and hi-side partial buffers
    int64_t blocks, rpb;
    channel_sum_split((int64_t)link->n * link->lh * link->lw, blocks, rpb);
    if (blocks * link->clo > need) need = blocks * link->clo;
    channel_sum_split((int64_t)link->n * link->hh * link->hw, blocks, rpb);
    if (blocks * link->chi > need) need = blocks * link->chi;
    if (dense_fits(link) && dense_wgrad_ws_floats(link) > need) need = dense_wgrad_ws_floats(link);
    if (conv64_wgrad_fits(link) && conv64_wgrad_ws_floats(link) > need) need = conv64_wgrad_ws_floats(link);
    if (conv32_fits(link) && conv32_wgrad_ws_floats(link) + 2 * CONV32_AMAX_FLOATS > need)        // slabs + the operands' AMAX arrays
        need = conv32_wgrad_ws_floats(link) + 2 * CONV32_AMAX_FLOATS;
    if (conv_c1_fits(link) && conv_c1_wgrad_ws_floats(link) > need) need = conv_c1_wgrad_ws_floats(link);
    if (conv_c1w_fits(link) && conv_c1w_wgrad_ws_floats(link) > need) need = conv_c1w_wgrad_ws_floats(link);
    return need;
}

extern "C" int arvae_link_wgrad(const arvae_link_t *link, const arvae_operand_t *lo, const arvae_operand_t *hi,
                                float *dwt, float *dbias, int32_t bias_side, float *ws, arvae_stream_t stream) {
    WgradPolicy p;
    if (int rc = make_geom(link, p.g)) return rc;
    ARVAE_REQUIRE(lo && lo->v && hi && hi->v && dwt, "link_wgrad: null pointer");
    ARVAE_REQUIRE(bias_side >= 0 && bias_side <= 2 && (bias_side == 0 || dbias != nullptr), "link_wgrad: bad bias request");
    ARVAE_REQUIRE(ws != nullptr || arvae_link_wgrad_ws_floats(link) == 0,
                  "link_wgrad: workspace of arvae_link_wgrad_ws_floats() floats needed");
    hipStream_t st = as_stream(stream);
    if (dense_fits(link) && hi->y == nullptr && bias_side != 2 && !dense_rows_generic(link, 4))
        return dense_wgrad(link, make_operand(lo), hi->v, dwt, bias_side == 1 ? dbias : nullptr, ws, st);
    if (conv_c1_fits(link) && lo->mask == nullptr && hi->mask == nullptr && lo->act != ARVAE_ACT_SELU &&
        hi->act != ARVAE_ACT_SELU)
        return conv_c1_wgrad(link, make_operand(lo), make_operand(hi), dwt, dbias, bias_side, ws, st);
    if (conv_c1w_fits(link))
        return conv_c1w_wgrad(link, make_operand(lo), make_operand(hi), dwt, dbias, bias_side, ws, st);
    if (wgrad_fast(link, lo, hi)) {
        unsigned *am = reinterpret_cast<unsigned *>(ws + conv32_wgrad_ws_floats(link));
        if (int rc = conv32_amax(lo->v, (int64_t)link->n * link->lh * link->lw * link->clo, am, st)) return rc;
        if (int rc = conv32_amax(hi->v, (int64_t)link->n * link->hh * link->hw * link->chi, am + CONV32_AMAX_FLOATS, st)) return rc;
        return conv32_wgrad(link, make_operand(lo), make_operand(hi), dwt, dbias, bias_side, ws, st, am, am + CONV32_AMAX_FLOATS);
    }
    p.lo = make_operand(lo);
    p.hi = make_operand(hi);
    if (conv64_wgrad_fits(link)) return link_wgrad_conv64(link, p.lo, p.hi, dwt, dbias, bias_side, ws, st, nullptr, nullptr);
    p.dwt = dwt;
    p.slab = ws;
    wgrad_split(link, p.M, p.N, p.P, p.zsplit, p.chunk);
    ARVAE_REQUIRE((int64_t)p.zsplit * p.M * p.N < (1ll << 31), "link_wgrad: slab too large");
    if (int rc = launch_gemm(p, p.M, p.N, p.zsplit, false, st, "link_wgrad")) return rc;
    if (p.zsplit > 1) {
        const int mn = p.M * p.N;
        ARVAE_LAUNCH(wgrad_reduce_kernel, dim3((mn + 63) / 64), dim3(256), 0, st, ws, p.zsplit, mn, p.N,
                           link->kh * link->kw, FastDiv(p.N), FastDiv(link->chi), dwt);
        if (int rc = check_launch("link_wgrad(reduce)")) return rc;
    }
    if (bias_side == 1)
        return channel_sum_launch(p.lo, (int64_t)link->n * link->lh * link->lw, link->clo, link->lo_perm_c,
                                  link->lo_perm_hw, dbias, ws, st);
    if (bias_side == 2)
        return channel_sum_launch(p.hi, (int64_t)link->n * link->hh * link->hw, link->chi, link->hi_perm_c,
                                  link->hi_perm_hw, dbias, ws, st);
    return ARVAE_OK;
}

static void channel_sum_split(int64_t rows, int64_t &blocks, int64_t &rpb) {
    blocks = (rows + 15) / 16;
    if (blocks > 1024) blocks = 1024;          // measured: 2048 blocks made the one-workgroup finish stage as slow as the sums
    if (blocks < 1) blocks = 1;
    rpb = (rows + blocks - 1) / blocks;
    blocks = (rows + rpb - 1) / rpb;
}

extern "C" int64_t arvae_channel_sum_ws_floats(int64_t rows, int32_t channels) {
    if (rows <= 0 || channels <= 0) return 0;
    int64_t blocks, rpb;
    channel_sum_split(rows, blocks, rpb);
    return blocks * channels;
}

static int channel_sum_launch(const Operand &g, int64_t rows, int channels, int perm_c, int perm_hw, float *out,
                              float *ws, hipStream_t st) {
    int64_t blocks, rpb;
    channel_sum_split(rows, blocks, rpb);
    const bool plain = g.y == nullptr && g.scale == nullptr;
    if (plain && channels % 4 == 0 && channels <= 1024 && 256 % (channels / 4) == 0 && (reinterpret_cast<uintptr_t>(g.v) & 15) == 0)
        ARVAE_LAUNCH(channel_sum4_kernel, dim3((unsigned)blocks), dim3(256), 0, st, reinterpret_cast<const float4 *>(g.v), rows, channels / 4,
                     rpb, ws);
    else
        ARVAE_LAUNCH(channel_sum_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, g, rows, channels, rpb, 0, 0, ws);
    if (int rc = check_launch("channel_sum")) return rc;
    ARVAE_LAUNCH(channel_sum_finish_kernel, dim3((unsigned)((channels + 15) / 16)), dim3(256), 0, st, ws, (int)blocks, channels, perm_c,
                 perm_hw, out);
    return check_launch("channel_sum(finish)");
}

extern "C" int arvae_channel_sum(const arvae_operand_t *g, int64_t rows, int32_t channels, int32_t perm_c,
                                 int32_t perm_hw, float *out, float *ws, arvae_stream_t stream) {
    ARVAE_REQUIRE(g && g->v && out && ws, "channel_sum: null pointer");
    ARVAE_REQUIRE(rows > 0 && channels > 0, "channel_sum: empty tensor");
    ARVAE_REQUIRE(perm_c == 0 || perm_c * perm_hw == channels, "channel_sum: perm does not cover channels");
    return channel_sum_launch(make_operand(g), rows, channels, perm_c, perm_hw, out, ws, as_stream(stream));
}

// out[i] = operand value with the activation derivative / keep-mask folded in (a "plain" copy of a gradient operand,
// for the long-batch Linear kernels that take plain operands only)
__global__ __launch_bounds__(256) void operand_apply_kernel(Operand g, int64_t count, float *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) out[i] = g.at(i);
}

extern "C" int arvae_operand_apply(const arvae_operand_t *g, int64_t count, float *out, arvae_stream_t stream) {
    ARVAE_REQUIRE(g && g->v && out && count > 0, "operand_apply: bad argument");
    const int64_t blocks = (count + 255) / 256;
    ARVAE_LAUNCH(operand_apply_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, as_stream(stream),
                       make_operand(g), count, out);
    return check_launch("operand_apply_kernel");
}
