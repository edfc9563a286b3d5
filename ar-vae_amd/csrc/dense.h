// Argument blocks of the Linear-layer kernels (dense.hip), shared with the whole-model executor (plan.hip).
#pragma once
#include "common.h"

namespace arvae {

struct Perm {           // feature f = c*hw + p  <->  memory column p*c_count + c ; c_count == 0: identity
    int c_count, hw;
    __device__ __forceinline__ int to_mem(int f) const { return c_count == 0 ? f : (f % hw) * c_count + f / hw; }
    __device__ __forceinline__ int to_feat(int m) const { return c_count == 0 ? m : (m % c_count) * hw + m / c_count; }
};

struct DenseArgs {
    Operand a;          // forward: X plain; dgrad / wgrad: G (gradient operand)
    const float *x;     // wgrad: layer input X
    const float *w;     // [n_out][n_in]
    const float *bias;
    float *out;         // forward: Y ; dgrad: dX ; wgrad: dW (accumulated)
    float *dbias;       // wgrad: accumulated, may be null
    const float *gate;  // dgrad: optional saved activation of the dX location: dX *= (gate > 0)
    int batch, n_in, n_out, act;
    Perm in_perm, out_perm;
    int store = 0;      // wgrad: write the tile (and bias sums) instead of adding to what is there
};

constexpr int DENSE_SPLIT_ROWS = 128;        // rows per workgroup slice of the row-split weight gradient
constexpr int DENSE_SPLIT_MIN_ROWS = 2048;   // shorter reduction axes stay on the one-workgroup-per-tile kernel

constexpr int DENSE_BATCH_MAX = 16;
struct DenseWgradBatch {
    int count;
    int tile_end[DENSE_BATCH_MAX];      // running number of 32x32 tiles after job j
    DenseArgs job[DENSE_BATCH_MAX];
};

// C[M][N] = sum_k A[m][k] B(k, n) on the three-term bf16 MFMA, 64 x 64 tiles (dense.hip, wide_gemm_x3_kernel): the wide Linear
// layers of the latent block.  A: fp32 [M][lda] (k contiguous), or (a_planes) its three bf16 terms as planes [3][M][lda] that are
// a_pstride elements apart.  B: ALWAYS planes, b_pstride apart, zero-padded along the reduction axis to a multiple of 32:
// b_krows == 0: [N][ldb] (k contiguous); b_krows == 1: [K][ldb] (n contiguous) -- the same copy read the other way.
constexpr int WIDE_MAX_SLICES = 16;
struct WideGemm {
    const void *a, *b;
    int64_t lda, ldb, a_pstride, b_pstride;
    int M, N, K, kslice, a_planes, b_planes, b_krows;
    float *out;                 // finished: [M][ldo]; split: workspace, slice z at out + z * slice_floats
    int64_t ldo, slice_floats;
    const float *bias;          // finished: per output column, may be null
    int act;                    // finished: activation (ARVAE_ACT_*)
    const float *gate;          // finished: optional ReLU gate: out = gate[m][n] > 0 ? value : 0
    unsigned *amax_out;         // finished: AMAX array of the result (conv32_common.h), may be null
    int dbg;                    // tools/probes/wide_gemm.py only (0 in the product): 1 no MFMAs, 2 no LDS commits, 4 no result stores
};
// dW'[p][q] (+)= sum_m A(p, m) B(q, m), operands [batch][features] ("K x rows"), exactly one of them as bf16 planes; rows / columns
// of dW' land at p_perm / q_perm.to_feat(.) of the [rows][ldw] gradient; dbias (may be null) += row sums of A
struct WideWgradJob {
    const void *a, *b;
    int64_t lda, ldb, a_pstride, b_pstride;
    int a_planes, b_planes, P, Q, R;
    float *dw, *dbias;
    int64_t ldw;
    Perm p_perm, q_perm;
};
struct WideWgradBatch {
    int count, wg_end[2];
    WideWgradJob job[2];
};
bool wide_wgrad_fits(const WideWgradJob &j);
int wide_wgrad(const WideWgradJob *jobs, int count, hipStream_t s);
int wide_gemm_slices(int M, int N, int K);
bool wide_gemm_fits(const WideGemm &g, int slices);
int wide_gemm(WideGemm g, int slices, bool partial, hipStream_t s);

bool dense_fits(const arvae_link_t *l);
int dense_fwd(const arvae_link_t *l, const float *x, const float *w, const float *bias, int act, float *y, hipStream_t s);
int dense_dgrad(const arvae_link_t *l, const Operand &g, const float *w, const float *gate, float *dx, hipStream_t s);
int64_t dense_wgrad_ws_floats(const arvae_link_t *l);
int dense_wgrad(const arvae_link_t *l, const Operand &g, const float *x, float *dw, float *dbias, float *ws, hipStream_t s);
// long-batch weight gradients of one pass as one launch + one reduction (dense.hip); the queue is opaque to callers
struct LongWgradQueue;
LongWgradQueue *dense_wgrad_long_new();
void dense_wgrad_long_delete(LongWgradQueue *q);
void dense_wgrad_long_begin(LongWgradQueue *q, float *ws, int64_t ws_cap);
int64_t dense_wgrad_long_ws_floats(const arvae_link_t *l);
bool dense_wgrad_long_defer(LongWgradQueue *q, const arvae_link_t *l, const Operand &g, const float *x, float *dw, float *dbias);
int dense_wgrad_long_flush(LongWgradQueue *q, hipStream_t s);
bool dense_wgrad_defer(DenseWgradBatch *b, const arvae_link_t *l, const Operand &g, const float *x, float *dw, float *dbias);
int dense_wgrad_flush(DenseWgradBatch *b, hipStream_t s);
struct SlabReduceBatch;
int dense_wgrad_slab_flush(DenseWgradBatch *b, SlabReduceBatch *r, hipStream_t s);      // both closing queues of a backward pass, one launch

}  // namespace arvae
