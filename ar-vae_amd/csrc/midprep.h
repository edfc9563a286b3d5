// Weight layout prep of the latent block (midblock.hip) as a device function: midblock.hip launches it alone, conv32.hip as the
// second half of ONE prep launch per step together with the 32-channel conv weight split (prep_all_kernel).
#pragma once
#include "common.h"
#include "dense.h"

namespace arvae {

constexpr int MID_MAX_LAYERS = 4;    // Linear layers on either side of the latent

struct MidPrepJob {
    const float *w, *b;          // reference layout [n][k], bias [n]
    const float *w2, *b2;        // rows n >= nsplit come from here (the two heads share one prepped matrix); null: one source
    int nsplit;
    float *mf, *mb, *bias;
    int k, n, kb;                // kb: row length of mb (k rounded up to a multiple of 4, zero-padded)
    Perm kp, np;                 // memory order <-> feature order of the input / output axis
    // the same matrix in CLUSTER LAYOUT (midcluster.h, McMat) for the clustered latent block, or cf == cb == null:
    // cf: reduce axis k in cf_kb blocks of 16, outputs n in cf_s slices of cf_ct tiles of 16; cb: reduce axis n, outputs k
    float *cf, *cb;
    int cf_kb, cf_ct, cf_s, cb_kb, cb_ct, cb_s;
};
struct MidPrepArgs {
    MidPrepJob job[2 * MID_MAX_LAYERS + 1];
    int count, blk_end[2 * MID_MAX_LAYERS + 1];
    unsigned *counters;          // arrival counters of the clustered latent block (midcluster.hip): zeroed here, every step
    int counter_words;
};

__device__ __forceinline__ void mid_prep_block(const MidPrepArgs &a, int block) {
    int j = 0, start = 0;
#pragma unroll
    for (int q = 0; q + 1 < 2 * MID_MAX_LAYERS + 1; ++q)
        if (q + 1 < a.count && block >= a.blk_end[q]) { j = q + 1; start = a.blk_end[q]; }
    const MidPrepJob &p = a.job[j];
    if (block == 0)
        for (int e = threadIdx.x; e < a.counter_words; e += 256) a.counters[e] = 0u;
    const int total = p.kb * p.n, stride = (a.blk_end[j] - start) * 256;
    auto src = [&](int nf, int kf) {                              // W[nf][kf] of the layer (two stacked sources for the heads)
        return (p.w2 != nullptr && nf >= p.nsplit) ? p.w2[(int64_t)(nf - p.nsplit) * p.k + kf] : p.w[(int64_t)nf * p.k + kf];
    };
    // (a job carries the row-kernel layouts mf / mb or the cluster layouts cf / cb, whichever kernels will run: never both)
    for (int e = (block - start) * 256 + threadIdx.x; e < (p.mf != nullptr ? total : p.n); e += stride) {
        if (p.mf != nullptr) {
            if (e < p.k * p.n) {
                const int km = e / p.n, nm = e - km * p.n;        // forward matrix [k][n], written in order
                p.mf[e] = src(p.np.to_feat(nm), p.kp.to_feat(km));
            }
            const int nm2 = e / p.kb, km2 = e - nm2 * p.kb;      // backward matrix [n][kb], written in order
            p.mb[e] = km2 < p.k ? src(p.np.to_feat(nm2), p.kp.to_feat(km2)) : 0.f;
        }
        if (e < p.n && p.bias != nullptr) {
            const int nf = p.np.to_feat(e);
            const float *bs = (p.w2 != nullptr && nf >= p.nsplit) ? p.b2 : p.b;
            p.bias[e] = bs != nullptr ? bs[(p.w2 != nullptr && nf >= p.nsplit) ? nf - p.nsplit : nf] : 0.f;
        }
    }
    if (p.cf == nullptr) return;
    // cluster layouts: element e = (((slice * CT + ct) * KB + b) * 64 + lane) * 4 + j  <-  M[16 b + 4 (lane / 16) + j][column]
    // with column = (slice * CT + ct) * 16 + lane % 16; written in order, zero past the matrix
    const int cf_total = p.cf_s * p.cf_ct * p.cf_kb * 256, cb_total = p.cb_s * p.cb_ct * p.cb_kb * 256;
    for (int e = (block - start) * 256 + threadIdx.x; e < max(cf_total, cb_total); e += stride) {
        const int j = e & 3, lane = (e >> 2) & 63, blk = e >> 8;
        if (e < cf_total) {
            const int b = blk % p.cf_kb, tcol = blk / p.cf_kb;
            const int km = 16 * b + 4 * (lane >> 4) + j, nm = 16 * tcol + (lane & 15);
            p.cf[e] = (km < p.k && nm < p.n) ? src(p.np.to_feat(nm), p.kp.to_feat(km)) : 0.f;
        }
        if (e < cb_total) {
            const int b = blk % p.cb_kb, tcol = blk / p.cb_kb;
            const int nm = 16 * b + 4 * (lane >> 4) + j, km = 16 * tcol + (lane & 15);
            p.cb[e] = (km < p.k && nm < p.n) ? src(p.np.to_feat(nm), p.kp.to_feat(km)) : 0.f;
        }
    }
}

}  // namespace arvae
