// Weight layout prep of the latent block (midblock.hip) as a device function: midblock.hip launches it alone, conv32.hip as the
// second half of ONE prep launch per step together with the 32-channel conv weight split (prep_all_kernel).
#pragma once
#include "common.h"
#include "dense.h"
#include "x3tile.h"

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
    // a wide layer the tile GEMMs multiply (dense.hip wide_gemm_x3_kernel; round 6): W'[n_mem][k_mem] as its three bf16 terms,
    // planes [3][n_pad][kb_pad] (both axes zero-padded to multiples of 32: either may be the reduction axis), or null
    unsigned short *planes;
    int n_pad, kb_pad;
};
// A 32-channel k4 / s2 / p1 link between a 4x4 and an 8x8 map that the clustered latent block computes itself (round 5: the conv
// layer in front of the block and the transposed one behind it, midcluster.hip): its weight wt[clo][chi][ky][kx] as the two
// matrices of that kernel, each 512 x 32 in CLUSTER LAYOUT (one slice, two column tiles, 32 k blocks):
//   down[k][clo], k = (ky * 4 + kx) * 32 + chi                    the DOWN map: a lo pixel from its 4 x 4 window of hi pixels
//   up[k][chi],   k = ((a * 2 + b) * 4 + ty * 2 + tx) * 32 + clo  the UP map, per parity class (a, b) of the hi pixel (2 i + a,
//                 2 j + b): tap t = 0 is lo row i (ky = a + 1), t = 1 the other contributing row (ky = a ? 0 : 3); columns alike;
//                 stored class after class: [class][ct][b 8][lane][4]
struct McConvPrep {
    const float *wt;             // null: no such layer
    float *down, *up;            // 16384 floats each
};
constexpr int MC_CONV_PREP_BLOCKS = 16;      // per layer: 2 x 16384 values, 8 per thread
__device__ __forceinline__ int mc_up_k(int a, int t) { return t == 0 ? a + 1 : (a ? 0 : 3); }       // kernel row / column of tap t

struct MidPrepArgs {
    MidPrepJob job[2 * MID_MAX_LAYERS + 1];
    int count, blk_end[2 * MID_MAX_LAYERS + 1];
    unsigned *counters;          // arrival counters of the clustered latent block (midcluster.hip): zeroed here, every step
    int counter_words;
    McConvPrep conv[2];          // blocks behind the jobs': MC_CONV_PREP_BLOCKS per layer that is present
    int n_conv;
};
inline int mid_prep_blocks(const MidPrepArgs &a) { return a.blk_end[a.count - 1] + MC_CONV_PREP_BLOCKS * a.n_conv; }

// what the executor (plan.hip) hands the latent block when the clustered kernels also compute the conv layers on either side of
// it (midblock.hip mid_fold_fits; midcluster.h McArgs.fold): the tensors beyond those layers
struct MidFold {
    const float *hi_e;           // forward + backward: the conv layer's input [batch][8][8][32] (a ReLU layer's saved output)
    float *hi_d;                 // forward: the transposed conv layer's output [batch][8][8][32], its sign bits and its AMAX array
    unsigned char *hi_d_bits;
    unsigned *hi_d_amax;
    const float *g_hi_d;         // backward: gradient w.r.t. hi_d's pre-activation
    float *d_hi_e;               // backward: gradient w.r.t. hi_e's pre-activation, and its AMAX array
    unsigned *d_hi_e_amax;
    float *slab_e, *slab_d;      // backward: the two layers' weight-gradient slabs (mid_fold_slab_floats each)
};

__device__ __forceinline__ void mc_conv_prep_block(const McConvPrep &c, int block) {
    for (int e = block * 256 + threadIdx.x; e < 2 * 16384; e += MC_CONV_PREP_BLOCKS * 256) {
        const bool up = e >= 16384;
        const int f = e & 16383, j = f & 3, lane = (f >> 2) & 63, blk = f >> 8;
        if (!up) {
            const int b = blk & 31, ct = blk >> 5;
            const int k = 16 * b + 4 * (lane >> 4) + j, n = 16 * ct + (lane & 15);
            const int tap = k >> 5, chi = k & 31;
            c.down[f] = c.wt[(n * 32 + chi) * 16 + tap];
        } else {
            const int b = blk & 7, ct = (blk >> 3) & 1, cls = blk >> 4;
            const int k = 16 * b + 4 * (lane >> 4) + j, n = 16 * ct + (lane & 15);
            const int t = k >> 5, clo = k & 31;
            const int ky = mc_up_k(cls >> 1, t >> 1), kx = mc_up_k(cls & 1, t & 1);
            c.up[f] = c.wt[(clo * 32 + n) * 16 + ky * 4 + kx];
        }
    }
}

__device__ __forceinline__ void mid_prep_block(const MidPrepArgs &a, int block) {
    if (block >= a.blk_end[a.count - 1]) {                      // the folded conv layers' matrices
        const int cb = block - a.blk_end[a.count - 1];
        if (cb / MC_CONV_PREP_BLOCKS == 0) mc_conv_prep_block(a.conv[0], cb);
        else mc_conv_prep_block(a.conv[1], cb - MC_CONV_PREP_BLOCKS);
        return;
    }
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
    // (mb alone: a wide layer the tile GEMMs multiply -- dense.hip wide_gemm_x3_kernel reads the one [n][kb] copy both ways)
    for (int e = (block - start) * 256 + threadIdx.x; e < (p.mb != nullptr ? total : p.n); e += stride) {
        if (p.mf != nullptr && e < p.k * p.n) {
            const int km = e / p.n, nm = e - km * p.n;            // forward matrix [k][n], written in order
            p.mf[e] = src(p.np.to_feat(nm), p.kp.to_feat(km));
        }
        if (p.mb != nullptr) {
            const int nm2 = e / p.kb, km2 = e - nm2 * p.kb;      // backward matrix [n][kb], written in order
            p.mb[e] = km2 < p.k ? src(p.np.to_feat(nm2), p.kp.to_feat(km2)) : 0.f;
        }
        if (e < p.n && p.bias != nullptr) {
            const int nf = p.np.to_feat(e);
            const float *bs = (p.w2 != nullptr && nf >= p.nsplit) ? p.b2 : p.b;
            p.bias[e] = bs != nullptr ? bs[(p.w2 != nullptr && nf >= p.nsplit) ? nf - p.nsplit : nf] : 0.f;
        }
    }
    if (p.planes != nullptr) {
        // TILED planes [3][kb_pad / 32][n_pad][32] (x3tile.h x3_tiled_index); pairs of consecutive k_mem: one 4-byte store per plane,
        // written in order
        const int pairs = p.n_pad * (p.kb_pad >> 1);
        unsigned *pl = reinterpret_cast<unsigned *>(p.planes);
        for (int e = (block - start) * 256 + threadIdx.x; e < pairs; e += stride) {
            const int kblk = e / (p.n_pad * 16), rem = e - kblk * (p.n_pad * 16);
            const int nm = rem >> 4, km = kblk * 32 + 2 * (rem & 15);
            const bool n_ok = nm < p.n;
            const int nf = p.np.to_feat(n_ok ? nm : 0);
            const float x0 = (n_ok && km < p.k) ? src(nf, p.kp.to_feat(km)) : 0.f;
            const float x1 = (n_ok && km + 1 < p.k) ? src(nf, p.kp.to_feat(km + 1)) : 0.f;
            unsigned h, m, l;
            rg_split3(x0, x1, h, m, l);
            pl[e] = h;
            pl[e + pairs] = m;
            pl[e + 2 * pairs] = l;
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
