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
};
struct MidPrepArgs {
    MidPrepJob job[2 * MID_MAX_LAYERS + 1];
    int count, blk_end[2 * MID_MAX_LAYERS + 1];
};

__device__ __forceinline__ void mid_prep_block(const MidPrepArgs &a, int block) {
    int j = 0, start = 0;
#pragma unroll
    for (int q = 0; q + 1 < 2 * MID_MAX_LAYERS + 1; ++q)
        if (q + 1 < a.count && block >= a.blk_end[q]) { j = q + 1; start = a.blk_end[q]; }
    const MidPrepJob &p = a.job[j];
    const int total = p.kb * p.n, stride = (a.blk_end[j] - start) * 256;
    for (int e = (block - start) * 256 + threadIdx.x; e < total; e += stride) {
        auto src = [&](int nf, int kf) {                          // W[nf][kf] of the layer (two stacked sources for the heads)
            return (p.w2 != nullptr && nf >= p.nsplit) ? p.w2[(int64_t)(nf - p.nsplit) * p.k + kf] : p.w[(int64_t)nf * p.k + kf];
        };
        if (e < p.k * p.n) {
            const int km = e / p.n, nm = e - km * p.n;            // forward matrix [k][n], written in order
            p.mf[e] = src(p.np.to_feat(nm), p.kp.to_feat(km));
        }
        const int nm2 = e / p.kb, km2 = e - nm2 * p.kb;          // backward matrix [n][kb], written in order
        p.mb[e] = km2 < p.k ? src(p.np.to_feat(nm2), p.kp.to_feat(km2)) : 0.f;
        if (e < p.n && p.bias != nullptr) {
            const int nf = p.np.to_feat(e);
            const float *bs = (p.w2 != nullptr && nf >= p.nsplit) ? p.b2 : p.b;
            p.bias[e] = bs != nullptr ? bs[(p.w2 != nullptr && nf >= p.nsplit) ? nf - p.nsplit : nf] : 0.f;
        }
    }
}

}  // namespace arvae
