#include "reduce.h"

namespace arvae {

// 64 outputs x 8 slab groups per workgroup: a wave reads 256 contiguous bytes of one slab per load, and ~1500
// workgroups keep enough loads in flight to stream the slabs at HBM speed (four groups of 256 threads: 15.7 us for the
// dSprites step's 44 MB, eight: 14.8 -- the paired launches left 128-139 slabs per layer, two rounds of loads per thread)
constexpr int RED_OUT = 64, RED_Z = 8;

__device__ __forceinline__ void slab_reduce_block(const SlabJob &j, int block, float (*red)[RED_OUT]) {
    const int slab_floats = j.kind == SLAB_C32 ? SLAB_C32_FLOATS : j.kind == SLAB_C32T ? SLAB_C32T_FLOATS : j.kind == SLAB_C1 ? SLAB_C1_FLOATS : SLAB_C1W_FLOATS;
    const int c1_ch = j.kind == SLAB_C1 ? 32 : 64;
    const int il = threadIdx.x & (RED_OUT - 1), zg = threadIdx.x / RED_OUT;
    const int i = block * RED_OUT + il;
    const int ic = i < slab_floats ? i : 0;
    // eight loads in flight per thread (the kernel is pure memory latency: with four it spent 91 % of its wave cycles
    // waiting), summed into four chains in a fixed association
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = zg;
    const float *base = j.slab + ic;
    // tap slabs: a bias sum is spread over the sixteen tap blocks of every slab (each member's share): tap 0's thread groups sum
    // all of them, (tap block, slab) pairs dealt over the groups in a fixed order; the other taps' bias entries are not outputs
    const bool tap_bias = j.kind == SLAB_C32T && ic % SLAB_C32T_TAP >= 32 * 32;
    if (tap_bias) {
        if (ic < SLAB_C32T_TAP) {
            const int nq = 16 * j.n_wg;
            for (int q = zg; q < nq; q += 8 * RED_Z) {           // eight loads in flight (pairs past the end: slab 0's own entry, times 0)
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int qq = q + u * RED_Z, qc = qq < nq ? qq : 0;
                    v[u] = base[(int64_t)(qc >> 4) * slab_floats + (qc & 15) * SLAB_C32T_TAP] * (qq < nq ? 1.f : 0.f);
                }
                s0 += v[0] + v[4];
                s1 += v[1] + v[5];
                s2 += v[2] + v[6];
                s3 += v[3] + v[7];
            }
        }
        z = j.n_wg;                                              // (skip the plain loops below)
    }
    for (; z + 7 * RED_Z < j.n_wg; z += 8 * RED_Z) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = base[(int64_t)(z + u * RED_Z) * slab_floats];
        s0 += v[0] + v[4];
        s1 += v[1] + v[5];
        s2 += v[2] + v[6];
        s3 += v[3] + v[7];
    }
    for (; z < j.n_wg; z += RED_Z) s0 += base[(int64_t)z * slab_floats];
    red[zg][il] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (zg == 0 && i < slab_floats) {
        float tot = (red[0][il] + red[1][il]) + (red[2][il] + red[3][il]);
        tot += (red[4][il] + red[5][il]) + (red[6][il] + red[7][il]);
        if (j.kind == SLAB_C32T) {
            const int tap = i / SLAB_C32T_TAP, w = i - tap * SLAB_C32T_TAP;
            if (w < 32 * 32) {
                j.dwt[w * 16 + tap] += tot;                      // w = clo * 32 + chi: dwt[clo][chi][ky][kx]
            } else if (tap == 0 && j.dbias != nullptr) {
                j.dbias[w - 32 * 32] += tot;
            }
        } else if (j.kind == SLAB_C32) {
            if (i < 16 * 32 * 32) {
                const int chi = i & 31, clo = (i >> 5) & 31, tap = i >> 10;
                j.dwt[(clo * 32 + chi) * 16 + tap] += tot;      // dwt[clo][chi][ky][kx]
            } else if (j.dbias != nullptr) {
                j.dbias[i - 16 * 32 * 32] += tot;
            }
        } else {
            if (i < c1_ch * 16) j.dwt[i] += tot;                 // wt[clo][0][ky][kx] is exactly [clo][tap]
            else if (i < c1_ch * 16 + c1_ch) { if (j.bias_mode == 1) j.dbias[i - c1_ch * 16] += tot; }
            else if (j.bias_mode == 2) j.dbias[0] += tot;
        }
    }
}

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
