#include "reduce.h"

namespace arvae {

__global__ __launch_bounds__(64 * RED_Z) void slab_reduce_kernel(SlabJob j) {
    __shared__ float red[RED_Z][RED_OUT];
    slab_reduce_block(j, blockIdx.x, red);
}

__global__ __launch_bounds__(64 * RED_Z) void slab_reduce_batch_kernel(SlabReduceBatch b) {
    __shared__ float red[RED_Z][RED_OUT];
    int j = 0, start = 0;
#pragma unroll
    for (int q = 0; q + 1 < SLAB_BATCH_MAX; ++q)
        if (q + 1 < b.count && (int)blockIdx.x >= b.block_end[q]) { j = q + 1; start = b.block_end[q]; }
    switch (j) {                         // constant indices into the by-value argument block
#define ARVAE_JOB(J) case J: slab_reduce_block(b.job[J], blockIdx.x - start, red); break;
        ARVAE_JOB(0) ARVAE_JOB(1) ARVAE_JOB(2) ARVAE_JOB(3) ARVAE_JOB(4) ARVAE_JOB(5) ARVAE_JOB(6) ARVAE_JOB(7)
#undef ARVAE_JOB
        default: break;
    }
}

static int job_blocks(const SlabJob &j) {
    return ((j.kind == SLAB_C32 ? SLAB_C32_FLOATS : j.kind == SLAB_C32T ? SLAB_C32T_FLOATS : j.kind == SLAB_C1 ? SLAB_C1_FLOATS : SLAB_C1W_FLOATS) + RED_OUT - 1) / RED_OUT;
}

int slab_reduce(const SlabJob &job, hipStream_t s) {
    ARVAE_LAUNCH(slab_reduce_kernel, dim3(job_blocks(job)), dim3(64 * RED_Z), 0, s, job);
    return check_launch(job.kind == SLAB_C32 ? "wgrad32_reduce_kernel" : "wgrad_c1_reduce_kernel");
}

bool slab_reduce_defer(SlabReduceBatch *b, const SlabJob &job) {
    if (b->count >= SLAB_BATCH_MAX) return false;
    b->block_end[b->count] = (b->count > 0 ? b->block_end[b->count - 1] : 0) + job_blocks(job);
    b->job[b->count++] = job;
    return true;
}

int slab_reduce_flush(SlabReduceBatch *b, hipStream_t s) {
    if (b->count == 0) return ARVAE_OK;
    ARVAE_LAUNCH(slab_reduce_batch_kernel, dim3(b->block_end[b->count - 1]), dim3(64 * RED_Z), 0, s, *b);
    b->count = 0;
    return check_launch("slab_reduce_batch_kernel");
}

}  // namespace arvae
