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

// run one job now / queue it / run everything queued
int slab_reduce(const SlabJob &job, hipStream_t s);
bool slab_reduce_defer(SlabReduceBatch *b, const SlabJob &job);
int slab_reduce_flush(SlabReduceBatch *b, hipStream_t s);

}  // namespace arvae
