// Fixed-order reduction of the per-workgroup weight-gradient slabs written by the conv kernels (conv32.hip,
// conv_c1.hip).  Several layers' reductions can be queued and run as ONE launch at the end of the backward pass.
#pragma once
#include "common.h"

namespace arvae {

enum { SLAB_C32 = 0, SLAB_C1 = 1, SLAB_C1W = 2, SLAB_C32T = 3 };
constexpr int SLAB_C32_FLOATS = 16 * 32 * 32 + 32;      // [ky][kx][clo][chi] + 32 bias sums
constexpr int SLAB_C1_FLOATS = 32 * 16 + 32 + 1;        // [clo][tap] + 32 lo sums + 1 image sum
constexpr int SLAB_C1W_FLOATS = 64 * 16 + 64 + 1;       // the same for the 64-channel single-channel links (conv_c1.hip, wgrad_c1w)
// SLAB_C32T (round 5: the conv layers the clustered latent block computes, midcluster.hip): one slab per CLUSTER, sixteen tap
// blocks of [clo][chi] + 32 bias sums each (a member's share): n_wg = clusters
constexpr int SLAB_C32T_TAP = 32 * 32 + 32, SLAB_C32T_FLOATS = 16 * SLAB_C32T_TAP;
constexpr int SLAB_BATCH_MAX = 8;

struct SlabJob {
    const float *slab;
    float *dwt, *dbias;
    int n_wg, kind, bias_mode;
};
struct SlabReduceBatch {
    int count;
    int block_end[SLAB_BATCH_MAX];      // running number of workgroups after job j
    SlabJob job[SLAB_BATCH_MAX];
};

// run one job now / queue it / run everything queued
int slab_reduce(const SlabJob &job, hipStream_t s);
bool slab_reduce_defer(SlabReduceBatch *b, const SlabJob &job);
int slab_reduce_flush(SlabReduceBatch *b, hipStream_t s);

}  // namespace arvae
