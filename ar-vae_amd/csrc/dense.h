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
