// Whole-model forward / backward executors for MeasureVAE (see include/arvae_hip.h, arvae_measure_vae_*): one host call
// enqueues every kernel of a pass on the caller's stream -- the GRU encoder, the latent head, the hierarchical decoder and
// all loss terms forward; the hand-chained adjoints of the same launches backward, parameter gradients accumulating straight
// into the gradient arena.  Host-side sequencing only: the math lives in the sequence / dense / loss kernels, reached through
// the entry points a per-layer caller uses (the Python path of ar-vae_amd/measure_vae.py issues the same launches one by one
// through autograd; tests/test_measure_executor.py holds the two against each other).
//
// Reference graph: measurevae/encoder.py:8-124, measurevae/decoder.py:309-525, measurevae/measure_vae.py:97-131,
// measurevae/measure_vae_trainer.py:85-140 (loss), utils/trainer.py:140 (backward).
#include "diag.h"
#include "common.h"
#include <mutex>
#include "dense.h"
#include "regloss.h"
#include "gru_mask.h"
#include "attributes.h"

namespace arvae {

// loss-term pieces (losses.hip)
int token_recon_partials(const float *weights, const int64_t *score, int batch, int beats, int tpb, int32_t vocab, float *ws,
                         float *dweights, hipStream_t s, int *nb_out, const AttrArgs *attr);
int token_recon_blocks(int64_t rows);
bool embed_fwd_with_beat(const int64_t *idx, const float *table, int32_t batch, int32_t steps, int32_t dim, int32_t vocab, int32_t time_major,
                         float *out, const BeatInput &beat, hipStream_t s, int *rc);
// arvae_tick_gi_fwd that also copies the tokens it reads (sequence.hip)
int tick_gi_fwd_copy(const float *g_small, const int64_t *tokens, const float *bias, int32_t batch, int32_t beats, int32_t ticks_per_beat,
                     int32_t vocab, int32_t cols, float *gi, int64_t *copy_to, hipStream_t s);
// several draws as one launch (rng.hip)
int philox_draws(int n_draws, const int *kind, void *const *out, const int64_t *count, const float *keep_prob, const uint32_t *offset,
                 uint64_t seed, uint32_t step, const uint32_t *dev_step, hipStream_t s);
int reg_partials(const float *z_rows, const float *lab_rows, int64_t n_rows, const float *z_cols, const float *lab_cols,
                 int64_t n_cols, int64_t ldz, int64_t ldl, const RegDims &rd, int32_t r, float delta, float *ws,
                 hipStream_t s, const struct VaeFinishArgs *park = nullptr, struct VaeFinishArgs *park_dst = nullptr);
int vae_finish(const float *rec_partial, int nb, int64_t batch, int64_t pix, const float *mu, const float *sigma,
               int64_t zdim, float beta, const float *cap, const float *reg_ws, int64_t n_cols, int64_t ldz,
               const int32_t *dims, int32_t r, float gamma, float delta, float reg_scale, float *dz, float *rec_out,
               float *kld_out, float *reg_out, float *scalars, hipStream_t s, int64_t rec_rows = 0);

static inline int64_t up4(int64_t v) { return (v + 3) / 4 * 4; }

static arvae_link_t dense_link(int n, int n_in, int n_out) {
    arvae_link_t l{};
    l.n = n;
    l.hh = l.hw = l.lh = l.lw = l.kh = l.kw = 1;
    l.stride = 1;
    l.chi = n_in;
    l.clo = n_out;
    return l;
}

// ---- glue kernels ----------------------------------------------------------------------------------
// rows of the tick RNN's sequence launches are ordered (tick-in-beat j, beat, measure b); the reference's tensors are ordered
// (tick t = tpb*beat + j, b) or (b, t).  y = alpha * x * mask with x in sequence order and the keep-mask in (t, b) order.
__global__ __launch_bounds__(256) void scale_mask_tick_kernel(const float *__restrict__ x, const uint8_t *__restrict__ mask, float alpha,
                                                               int batch, int beats, int tpb, int hid4, float *__restrict__ y) {
    const int64_t total = (int64_t)tpb * beats * batch * hid4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % hid4);
        int64_t r = i / hid4;
        const int b = (int)(r % batch);
        r /= batch;
        const int beat = (int)(r % beats), j = (int)(r / beats);
        const int64_t mrow = ((int64_t)(beat * tpb + j) * batch + b) * hid4 + c;
        const float4 v = reinterpret_cast<const float4 *>(x)[i];
        const uchar4 m = reinterpret_cast<const uchar4 *>(mask)[mrow];
        reinterpret_cast<float4 *>(y)[i] = make_float4(alpha * v.x * (float)m.x, alpha * v.y * (float)m.y, alpha * v.z * (float)m.z,
                                                       alpha * v.w * (float)m.w);
    }
}

// out = g[0] * d * (y > 0): the upstream scalar and the ReLU of the note projection folded into the cross-entropy gradient
__global__ __launch_bounds__(256) void relu_gate_scale_kernel(const float *__restrict__ d, const float *__restrict__ y,
                                                               const float *__restrict__ g, int64_t count4, float *__restrict__ out) {
    const float s = g[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count4; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4 *>(d)[i], a = reinterpret_cast<const float4 *>(y)[i];
        reinterpret_cast<float4 *>(out)[i] = make_float4(a.x > 0.f ? s * v.x : 0.f, a.y > 0.f ? s * v.y : 0.f, a.z > 0.f ? s * v.z : 0.f,
                                                         a.w > 0.f ? s * v.w : 0.f);
    }
}
__global__ __launch_bounds__(256) void relu_gate_scale1_kernel(const float *__restrict__ d, const float *__restrict__ y,
                                                                const float *__restrict__ g, int64_t count, float *__restrict__ out) {
    const float s = g[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256)
        out[i] = y[i] > 0.f ? s * d[i] : 0.f;
}

// y[r][:] = x[r][:] + bias[:]
__global__ __launch_bounds__(256) void add_bias_rows_kernel(const float4 *__restrict__ x, const float4 *__restrict__ bias, int64_t rows,
                                                             int cols4, float4 *__restrict__ y) {
    const int64_t total = rows * cols4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const float4 v = x[i], bb = bias[i % cols4];
        y[i] = make_float4(v.x + bb.x, v.y + bb.y, v.z + bb.z, v.w + bb.w);
    }
}

// gradient of the loss w.r.t. (mu, log_std): the decoder path g_z (already times the upstream scalar), the regulariser's unit
// gradient dz_reg and the beta-KL term; sigma = exp(log_std), z = mu + eps * sigma (measure_vae.py:115-123, utils/trainer.py:354-367)
__global__ __launch_bounds__(256) void measure_latent_bwd_kernel(const float *__restrict__ g_z, const float *__restrict__ dz_reg,
                                                                  const float *__restrict__ mu, const float *__restrict__ sigma,
                                                                  const float *__restrict__ eps, const float *__restrict__ g_loss,
                                                                  const float *__restrict__ kl, const float *__restrict__ cap, float beta,
                                                                  float inv_batch, float reg_scale, int64_t count, float *__restrict__ d_mu,
                                                                  float *__restrict__ d_ls) {
    const float g = g_loss[0];
    const float diff = kl[0] - (cap != nullptr ? cap[0] : 0.f);
    const float k = g * beta * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) * inv_batch;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        float gz = g_z[i];
        if (dz_reg != nullptr) gz += g * reg_scale * dz_reg[i];
        const float s = sigma[i];
        d_mu[i] = gz + k * mu[i];
        d_ls[i] = (gz * eps[i] + k * (s - 1.f / s)) * s;
    }
}

// ---- the latent head's second layers + the reparameterised sample as ONE launch (forward), and their data gradients (backward).
// The two heads' first layers are one product h12 = [hmu | hls] (rows of 2 * hw floats); per row
//     mu = W_mu hmu + b_mu,   log_std = W_ls hls + b_ls,   sigma = exp(log_std),   z = mu + eps * sigma      (measure_vae.py:100-123)
// were a column split, two 5 us Linear launches and the sample; backward, the (d mu, d log_std) kernel, two data-gradient launches
// and a column concatenation.  MH_ROWS rows per workgroup, the 2 zdim weight rows in LDS (forward) or a thread's two weight
// columns in registers (backward); exact fp32 FMA chains.
constexpr int MH_ROWS = 4, MH_ZMAX = 32;
struct MeasureHeadsFwd {
    const float *h12, *w_mu, *b_mu, *w_ls, *b_ls, *eps;
    float *hmu, *hls, *mu, *log_std, *sigma, *z;      // hmu / hls: the halves of h12 as the weight gradients read them
    int batch, hw, zdim;                               // hw = width of one head's hidden vector (a multiple of 4)
};
__global__ __launch_bounds__(256) void measure_heads_fwd_kernel(MeasureHeadsFwd p) {
    extern __shared__ __attribute__((aligned(16))) float mh_lds[];
    const int ld = 2 * p.hw, ws = p.hw + 4, h4 = p.hw >> 2;
    float *hs = mh_lds, *wl = hs + MH_ROWS * ld, *outs = wl + 2 * p.zdim * ws;       // rows | 2 zdim weight rows | products
    const int row0 = blockIdx.x * MH_ROWS;
    for (int i = threadIdx.x; i < MH_ROWS * 2 * h4; i += 256) {
        const int r = i / (2 * h4), c4 = i - r * 2 * h4, row = row0 + r;
        const int rr = row < p.batch ? row : p.batch - 1;                               // clamped: unconditional load
        const float4 v = reinterpret_cast<const float4 *>(p.h12 + (int64_t)rr * ld)[c4];
        reinterpret_cast<float4 *>(hs + r * ld)[c4] = v;
        if (row < p.batch) {
            float *dst = c4 < h4 ? p.hmu + (int64_t)row * p.hw + 4 * c4 : p.hls + (int64_t)row * p.hw + 4 * (c4 - h4);
            *reinterpret_cast<float4 *>(dst) = v;
        }
    }
    for (int base = 0; base < 2 * p.zdim * h4; base += 8 * 256) {       // eight independent 16-byte loads per thread and round trip
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = min(base + u * 256 + (int)threadIdx.x, 2 * p.zdim * h4 - 1), j = i / h4, k = i - j * h4;
            const float *src = j < p.zdim ? p.w_mu + (int64_t)j * p.hw : p.w_ls + (int64_t)(j - p.zdim) * p.hw;
            v[u] = reinterpret_cast<const float4 *>(src)[k];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = base + u * 256 + (int)threadIdx.x, j = i / h4, k = i - j * h4;
            if (i < 2 * p.zdim * h4) reinterpret_cast<float4 *>(wl + j * ws)[k] = v[u];
        }
    }
    const int r = threadIdx.x >> 6, j = threadIdx.x & 63, row = row0 + r;
    const bool col_ok = j < 2 * p.zdim, lat = j < p.zdim && row < p.batch;
    const int64_t idx = lat ? (int64_t)row * p.zdim + j : 0;
    const float e = p.eps[idx];
    const int jc = col_ok ? j : 0;
    const float *bp = jc < p.zdim ? p.b_mu : p.b_ls;
    const float bias = bp != nullptr ? bp[jc < p.zdim ? jc : jc - p.zdim] : 0.f;
    __syncthreads();
    {
        const float4 *w = reinterpret_cast<const float4 *>(wl + jc * ws);
        const float4 *x = reinterpret_cast<const float4 *>(hs + r * ld + (jc < p.zdim ? 0 : p.hw));
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int k = 0; k < h4; ++k) {
            const float4 a = x[k], b = w[k];
            acc.x = fmaf(a.x, b.x, acc.x); acc.y = fmaf(a.y, b.y, acc.y);
            acc.z = fmaf(a.z, b.z, acc.z); acc.w = fmaf(a.w, b.w, acc.w);
        }
        outs[r * 64 + j] = (acc.x + acc.y) + (acc.z + acc.w) + bias;
    }
    __syncthreads();
    if (lat) {
        const float m = outs[r * 64 + j], l = outs[r * 64 + j + p.zdim];
        const float sg = expf(l);
        p.mu[idx] = m;
        p.log_std[idx] = l;
        p.sigma[idx] = sg;
        p.z[idx] = fmaf(e, sg, m);
    }
}

struct MeasureHeadsBwd {
    const float *g_z, *dz_reg, *mu, *sigma, *eps, *g_loss, *kl, *cap, *w_mu, *w_ls;
    float beta, inv_batch, reg_scale;
    float *d_mu, *d_ls, *d_h12;                        // d_h12 rows: [d hmu | d hls]
    int batch, hw, zdim;
};
// thread = one column of each head's hidden vector (hw <= 256 columns): its two weight columns in registers, the rows'
// (d mu, d log_std) through LDS
__global__ __launch_bounds__(256) void measure_heads_bwd_kernel(MeasureHeadsBwd p) {
    __shared__ float dm[MH_ROWS][MH_ZMAX], dl[MH_ROWS][MH_ZMAX];
    const int row0 = blockIdx.x * MH_ROWS, k = threadIdx.x;
    const bool kok = k < p.hw;
    float wm[MH_ZMAX], wls[MH_ZMAX];
#pragma unroll
    for (int j = 0; j < MH_ZMAX; ++j) {
        const bool ok = kok && j < p.zdim;
        wm[j] = ok ? p.w_mu[(int64_t)j * p.hw + k] : 0.f;
        wls[j] = ok ? p.w_ls[(int64_t)j * p.hw + k] : 0.f;
    }
    if (threadIdx.x < MH_ROWS * MH_ZMAX) {
        const int r = threadIdx.x / MH_ZMAX, j = threadIdx.x % MH_ZMAX, row = row0 + r;
        float a = 0.f, b = 0.f;
        if (row < p.batch && j < p.zdim) {
            const int64_t i = (int64_t)row * p.zdim + j;
            const float g = p.g_loss[0];
            const float diff = p.kl[0] - (p.cap != nullptr ? p.cap[0] : 0.f);
            const float kk = g * p.beta * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) * p.inv_batch;
            float gz = p.g_z[i];
            if (p.dz_reg != nullptr) gz += g * p.reg_scale * p.dz_reg[i];
            const float sg = p.sigma[i];
            a = gz + kk * p.mu[i];
            b = (gz * p.eps[i] + kk * (sg - 1.f / sg)) * sg;
            p.d_mu[i] = a;
            p.d_ls[i] = b;
        }
        dm[r][j] = a;
        dl[r][j] = b;
    }
    __syncthreads();
    if (!kok) return;
#pragma unroll
    for (int r = 0; r < MH_ROWS; ++r) {
        if (row0 + r >= p.batch) break;
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int j = 0; j < MH_ZMAX; ++j) {
            a = fmaf(dm[r][j], wm[j], a);
            b = fmaf(dl[r][j], wls[j], b);
        }
        float *dst = p.d_h12 + (int64_t)(row0 + r) * 2 * p.hw;
        dst[k] = a;
        dst[p.hw + k] = b;
    }
}
static bool measure_heads_fit(int hw, int zdim) { return hw >= 4 && hw <= 256 && (hw & 3) == 0 && zdim >= 1 && zdim <= MH_ZMAX; }

// the beat RNN's constant input b_0 (decoder.py:436-440): its copies x0b[rows] (what the weight gradient reads) and its projection
// gi[b][c] = b_0 * w[c] + bias[c], the same row for every measure -- one launch instead of a broadcast and a 1-wide Linear layer
__global__ __launch_bounds__(256) void beat_input_kernel(BeatInput p) {
    beat_input_items(p, (int64_t)blockIdx.x * 256 + threadIdx.x, (int64_t)gridDim.x * 256);
}
// Three small sums that nothing in the pass waits for, as ONE launch at its end (round 5; each was a ~5 us launch of its own at the point
// where its operand appeared): the tick RNN's first bias gradient (column sums of the note rows), the gradient of b_0 (a sum over
// beats x batch numbers) and the encoder table's gradient added to the arena.  Workgroups [0, cs_blocks) the column sums, one the sum,
// the rest the addition; each job's own fixed order is what it was.
struct GradTail {
    const float *cs_x; int cs_rows, cs_cols; float *cs_dst; int cs_blocks;
    const float *sum_x; int sum_n; float *sum_dst;
    const float *add_x; int add_n; float *add_dst;
};
__global__ __launch_bounds__(256) void grad_tail_kernel(GradTail t) {
    __shared__ float red[256];
    const int b = blockIdx.x;
    if (b < t.cs_blocks) {
        const int c = b * 256 + threadIdx.x;
        if (c >= t.cs_cols) return;
        float a = 0.f;
        for (int r0 = 0; r0 < t.cs_rows; r0 += 16) {             // sixteen independent loads per round trip, summed in row order
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = t.cs_x[(int64_t)min(r0 + u, t.cs_rows - 1) * t.cs_cols + c];
#pragma unroll
            for (int u = 0; u < 16; ++u) a += r0 + u < t.cs_rows ? v[u] : 0.f;
        }
        t.cs_dst[c] += a;
    } else if (b == t.cs_blocks) {
        float a = 0.f;
        for (int i = threadIdx.x; i < t.sum_n; i += 256) a += t.sum_x[i];
        red[threadIdx.x] = a;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) t.sum_dst[0] += red[0];
    } else {
        const int i = (b - t.cs_blocks - 1) * 256 + threadIdx.x;
        if (i < t.add_n) t.add_dst[i] += t.add_x[i];
    }
}

static inline unsigned blocks_for(int64_t items, int cap = 2048) {
    const int64_t b = (items + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ---- workspace layout ------------------------------------------------------------------------------
struct MvWs {
    // encoder
    float *ptab, *gi0, *out0, *sv0, *mid, *gi1, *out1, *sv1, *hidden, *h12, *hmu, *hls, *log_std;
    // decoder
    float *flatb, *x0b, *gi0b, *out0b, *svb0, *midb, *gi1b, *beat_out, *svb1;
    float *both, *xs, *gsm, *gib, *frws;
    float *gi0t, *out0t, *svt0, *midt, *gi1t, *out1t, *svt1, *probs;
    int64_t *tgt;
    // loss terms
    float *dprobs, *rec_ws, *ce_out, *kld_out, *labels, *reg_ws, *reg_out, *dz_reg;
    // backward
    float *gpre, *d_seq_h, *dgi_t1, *dgh_t1, *hprev_t1, *dgi_t0, *dgh_t0, *hprev_t0, *d_mid_t, *dg_small, *tick_ws, *dx_small, *d_both;
    float *d_rows_h, *d_mid_b, *dgi_b1, *dgh_b1, *hprev_b1, *dgi_b0, *dgh_b0, *hprev_b0, *d_x0, *d_flatb, *d_z, *d_mu, *d_ls, *d_hmu, *d_hls, *d_h12, *d_hidden;
    float *dgi_e1, *dgh_e1, *hprev_e1, *dgi_e0, *dgh_e0, *hprev_e0, *d_mid_e, *d_out0_e, *dptab, *embed_ws, *d_table, *wg_ws, *wg_long, *cs_ws;
    int64_t wg_long_floats;
};

struct MvDims {
    int b, t, nb, tpb, v, e, he, hd, z;
    int tb, rb, rt, ns;      // encoder rows T*B, beat rows nb*B, tick rows tpb*nb*B, rows of the small tick product
};

static MvDims dims_of(const arvae_measure_vae_t *m, int batch) {
    MvDims d{};
    d.b = batch; d.t = m->steps; d.nb = m->beats; d.tpb = m->ticks_per_beat; d.v = m->vocab; d.e = m->emb;
    d.he = m->enc_hidden; d.hd = m->dec_hidden; d.z = m->zdim;
    d.tb = d.t * batch; d.rb = d.nb * batch; d.rt = d.tpb * d.nb * batch; d.ns = d.v + 1 + d.rb;
    return d;
}

static int64_t long_wgrad_ws(int rows, int n_in, int n_out) {
    const arvae_link_t l = dense_link(rows, n_in, n_out);
    return dense_wgrad_ws_floats(&l);
}

static int64_t carve(const arvae_measure_vae_t *m, int batch, float *base, MvWs *w) {
    const MvDims d = dims_of(m, batch);
    int64_t off = 0;
    auto take = [&](int64_t n) {
        float *p = base != nullptr ? base + off : nullptr;
        off += up4(n);
        return p;
    };
    const int64_t TB = d.tb, RB = d.rb, RT = d.rt, B = d.b;
    w->ptab = take((int64_t)d.v * 6 * d.he);
    w->gi0 = take(TB * 6 * d.he);
    w->out0 = take(TB * 2 * d.he);
    w->sv0 = take(2 * TB * 4 * d.he);
    w->mid = take(TB * 2 * d.he);
    w->gi1 = take(TB * 6 * d.he);
    w->out1 = take(TB * 2 * d.he);
    w->sv1 = take(2 * TB * 4 * d.he);
    w->hidden = take(B * 4 * d.he);
    w->h12 = take(B * 4 * d.he);
    w->hmu = take(B * 2 * d.he);
    w->hls = take(B * 2 * d.he);
    w->log_std = take(B * d.z);
    w->flatb = take(B * 2 * d.hd);
    w->x0b = take(RB);
    w->gi0b = take(B * 3 * d.hd);
    w->out0b = take(RB * d.hd);
    w->svb0 = take(RB * 4 * d.hd);
    w->midb = take(RB * d.hd);
    w->gi1b = take(RB * 3 * d.hd);
    w->beat_out = take(RB * d.hd);
    w->svb1 = take(RB * 4 * d.hd);
    w->both = take(RB * 3 * d.hd);
    w->xs = take((int64_t)d.ns * (d.e + d.hd));
    w->gsm = take((int64_t)d.ns * 3 * d.hd);
    w->gib = take(RB * 3 * d.hd);
    w->frws = take(arvae_tick_free_run_ws_floats(d.hd));
    w->gi0t = take(RT * 3 * d.hd);
    w->out0t = take(RT * d.hd);
    w->svt0 = take(RT * 4 * d.hd);
    w->midt = take(RT * d.hd);
    w->gi1t = take(RT * 3 * d.hd);
    w->out1t = take(RT * d.hd);
    w->svt1 = take(RT * 4 * d.hd);
    w->probs = take(RT * d.v);
    w->tgt = reinterpret_cast<int64_t *>(take(2 * RT));
    w->dprobs = take(RT * d.v);
    w->rec_ws = take(arvae_recon_ws_floats(RT));
    w->ce_out = take(4);
    w->kld_out = take(4);
    w->labels = take(B * 4);
    w->reg_ws = take(arvae_reg_loss_ws_floats(B, m->n_reg > 0 ? m->n_reg : 1));
    w->reg_out = take(4);
    w->dz_reg = take(B * d.z);
    // backward
    w->gpre = take(RT * d.v);
    w->d_seq_h = take(RT * d.hd);
    w->dgi_t1 = take(RT * 3 * d.hd);
    w->dgh_t1 = take(RT * 3 * d.hd);
    w->hprev_t1 = take(RT * d.hd);
    w->dgi_t0 = take(RT * 3 * d.hd);
    w->dgh_t0 = take(RT * 3 * d.hd);
    w->hprev_t0 = take(RT * d.hd);
    w->d_mid_t = take(RT * d.hd);
    w->dg_small = take((int64_t)d.ns * 3 * d.hd);
    w->tick_ws = take(arvae_tick_gi_bwd_ws_floats(d.v, 3 * d.hd));
    w->dx_small = take((int64_t)d.ns * (d.e + d.hd));
    w->d_both = take(RB * 3 * d.hd);
    w->d_rows_h = take(RB * d.hd);
    w->d_mid_b = take(RB * d.hd);
    w->dgi_b1 = take(RB * 3 * d.hd);
    w->dgh_b1 = take(RB * 3 * d.hd);
    w->hprev_b1 = take(RB * d.hd);
    w->dgi_b0 = take(RB * 3 * d.hd);
    w->dgh_b0 = take(RB * 3 * d.hd);
    w->hprev_b0 = take(RB * d.hd);
    w->d_x0 = take(RB);
    w->d_flatb = take(B * 2 * d.hd);
    w->d_z = take(B * d.z);
    w->d_mu = take(B * d.z);
    w->d_ls = take(B * d.z);
    w->d_hmu = take(B * 2 * d.he);
    w->d_hls = take(B * 2 * d.he);
    w->d_h12 = take(B * 4 * d.he);
    w->d_hidden = take(B * 4 * d.he);
    w->dgi_e1 = take(TB * 6 * d.he);
    w->dgh_e1 = take(2 * TB * 3 * d.he);
    w->hprev_e1 = take(2 * TB * d.he);
    w->dgi_e0 = take(TB * 6 * d.he);
    w->dgh_e0 = take(2 * TB * 3 * d.he);
    w->hprev_e0 = take(2 * TB * d.he);
    w->d_mid_e = take(TB * 2 * d.he);
    w->d_out0_e = take(TB * 2 * d.he);
    w->dptab = take((int64_t)d.v * 6 * d.he);
    w->embed_ws = take(arvae_embed_bwd_ws_floats(d.b, d.t, 6 * d.he, d.v));
    w->d_table = take((int64_t)d.v * d.e);
    int64_t wg = 0;
    auto wmax = [&](int64_t v) { if (v > wg) wg = v; };
    wmax(long_wgrad_ws(d.tb, d.he, 3 * d.he));
    wmax(long_wgrad_ws(d.tb, 2 * d.he, 6 * d.he));
    wmax(long_wgrad_ws(d.rt, d.hd, 3 * d.hd));
    wmax(long_wgrad_ws(d.rt, d.hd, d.v));
    wmax(long_wgrad_ws(d.rb, d.hd, 3 * d.hd));
    wmax(long_wgrad_ws(d.ns, d.e + d.hd, 3 * d.hd));
    w->wg_ws = take(wg);
    // every whole-sequence weight gradient of the pass keeps its row slices until the one reduction at the end
    int64_t lw = 0;
    auto ladd = [&](int rows, int n_in, int n_out, int times) {
        const arvae_link_t l = dense_link(rows, n_in, n_out);
        lw += times * dense_wgrad_long_ws_floats(&l);
    };
    ladd(d.tb, d.he, 3 * d.he, 4);
    ladd(d.tb, 2 * d.he, 6 * d.he, 1);
    ladd(d.rt, d.hd, 3 * d.hd, 3);
    ladd(d.rt, d.hd, d.v, 1);
    ladd(d.rb, d.hd, 3 * d.hd, 5);
    ladd(d.ns, d.e + d.hd, 3 * d.hd, 1);
    w->wg_long_floats = lw;
    w->wg_long = take(lw);
    int64_t cs = arvae_channel_sum_ws_floats(d.v + 1, 3 * d.hd);
    if (arvae_channel_sum_ws_floats(d.rb, 1) > cs) cs = arvae_channel_sum_ws_floats(d.rb, 1);
    w->cs_ws = take(cs);
    return off;
}

static int check_model(const arvae_measure_vae_t *m, int batch, const char *what) {
    ARVAE_REQUIRE(m != nullptr && batch >= 1, "%s: null model or empty batch", what);
    ARVAE_REQUIRE(m->steps == m->beats * m->ticks_per_beat && m->steps >= 1, "%s: steps must be beats * ticks_per_beat", what);
    ARVAE_REQUIRE(arvae_gru_seq_supported(m->enc_hidden) && arvae_gru_seq_supported(m->dec_hidden),
                  "%s: hidden sizes %d / %d are not built as sequence kernels (32, 64, 128)", what, m->enc_hidden, m->dec_hidden);
    ARVAE_REQUIRE(m->vocab >= 1 && m->emb >= 1 && m->zdim >= 1, "%s: empty vocabulary / embedding / latent", what);
    ARVAE_REQUIRE(m->n_reg >= 0 && m->n_reg <= 16, "%s: at most 16 regularised dims", what);
    ARVAE_REQUIRE((int64_t)(m->vocab + 1) * 1024 + ((int64_t)m->steps * batch + 15) / 16 * 8 <= 65536,
                  "%s: vocabulary of %d notes at batch %d exceeds the segment sums' LDS", what, m->vocab, batch);
    return ARVAE_OK;
}

#define MV_TRY(expr)                 \
    do {                             \
        if (int rc_ = (expr)) return rc_; \
    } while (0)

static arvae_operand_t plain(const float *v) { return arvae_operand_t{v, nullptr, nullptr, ARVAE_ACT_NONE}; }
static arvae_operand_t gated(const float *v, const float *y, int act) { return arvae_operand_t{v, y, nullptr, act}; }

// y = act(x W^T + b) for `rows` rows
static int lin_fwd(int rows, int n_in, int n_out, const float *x, const float *w, const float *bias, int act, float *y, hipStream_t s) {
    const arvae_link_t l = dense_link(rows, n_in, n_out);
    return dense_fwd(&l, x, w, bias, act, y, s);
}
// dx = g W
static int lin_dgrad(int rows, int n_in, int n_out, const arvae_operand_t &g, const float *w, float *dx, hipStream_t s) {
    const arvae_link_t l = dense_link(rows, n_in, n_out);
    return dense_dgrad(&l, make_operand(&g), w, nullptr, dx, s);
}
// dw += g^T x, db += column sums of g: queued for the pass's one batched launch when the batch is short, else launched now
struct WgradQueues {
    DenseWgradBatch batch{};             // batch-sized layers: one launch of 32 x 32 tiles (dense_wgrad_batch_kernel)
    LongWgradQueue *rows = nullptr;      // whole-sequence layers: one row-sliced launch + one reduction
    float *ws = nullptr;                 // workspace of a long-batch gradient the row queue refuses
    WgradQueues() : rows(dense_wgrad_long_new()) {}
    ~WgradQueues() { dense_wgrad_long_delete(rows); }
};
static int lin_wgrad(WgradQueues *wq, int rows, int n_in, int n_out, const arvae_operand_t &g, const float *x, float *dw, float *db,
                     hipStream_t s) {
    const arvae_link_t l = dense_link(rows, n_in, n_out);
    DenseWgradBatch *q = &wq->batch;
    if (rows >= DENSE_SPLIT_MIN_ROWS) {
        if (dense_wgrad_long_defer(wq->rows, &l, make_operand(&g), x, dw, db)) return ARVAE_OK;
        return dense_wgrad(&l, make_operand(&g), x, dw, db, wq->ws, s);
    }
    if (dense_wgrad_defer(q, &l, make_operand(&g), x, dw, db)) return ARVAE_OK;
    if (int rc = dense_wgrad_flush(q, s)) return rc;            // the queue is full: launch what it holds, start the next one
    q->count = 0;
    return dense_wgrad_defer(q, &l, make_operand(&g), x, dw, db) ? ARVAE_OK : fail(ARVAE_E_INVALID, "measure_vae_backward: weight-gradient queue");
}

}  // namespace arvae

using namespace arvae;

extern "C" int64_t arvae_measure_vae_ws_floats(const arvae_measure_vae_t *model, int32_t batch) {
    if (check_model(model, batch, "measure_vae_ws_floats") != ARVAE_OK) return -1;
    MvWs w{};
    return carve(model, batch, nullptr, &w);
}

extern "C" int arvae_measure_vae_forward(const arvae_measure_vae_t *m, int32_t batch, const float *params, const int64_t *score,
                                         float *eps, uint8_t *enc_mask, uint8_t *dec_mask, int32_t teacher_forced,
                                         const float *capacity, const arvae_measure_tables_t *tables, float *ws, float *scalars,
                                         float *mu, float *sigma, float *z, int64_t *tokens, float *labels, int32_t defer_finish,
                                         arvae_stream_t stream) {
    MV_TRY(check_model(m, batch, "measure_vae_forward"));
    ARVAE_REQUIRE(params && score && eps && ws && scalars && mu && sigma && z && tokens, "measure_vae_forward: null pointer");
    ARVAE_REQUIRE((enc_mask == nullptr) == (dec_mask == nullptr), "measure_vae_forward: give both keep-masks (training) or neither (evaluation)");
    ARVAE_REQUIRE(m->n_reg == 0 || tables != nullptr, "measure_vae_forward: the regulariser needs the attribute tables");
    ARVAE_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15) == 0, "measure_vae_forward: workspace must be 16-byte aligned");
    const bool dropping = enc_mask != nullptr;
    ARVAE_REQUIRE(!dropping || (m->enc_dropout >= 0.f && m->enc_dropout < 1.f && m->dec_dropout >= 0.f && m->dec_dropout < 1.f),
                  "measure_vae_forward: dropout probabilities must be in [0, 1)");
    hipStream_t s = as_stream(stream);
    MvWs w{};
    carve(m, batch, ws, &w);
    const MvDims d = dims_of(m, batch);
    const float *P = params;
    const int He = d.he, Hd = d.hd;

    // ---- draws (csrc/rng.h): keep-masks and eps exactly as ops.keep_mask / ops.normal_noise make them, in the Python path's order
    if (m->rng_draw) {                                         // one launch (rng.hip: philox_draws)
        int kind[3], n = 0;
        void *out[3];
        int64_t cnt[3];
        float keep[3];
        uint32_t off[3];
        auto add = [&](int k, void *o, int64_t c, float kp, uint32_t of) { kind[n] = k; out[n] = o; cnt[n] = c; keep[n] = kp; off[n] = of; ++n; };
        if (dropping) add(1, enc_mask, (int64_t)d.tb * 2 * He, 1.f - m->enc_dropout, m->rng_offset[0]);
        add(0, eps, (int64_t)d.b * d.z, 1.f, m->rng_offset[1]);
        if (dropping) add(1, dec_mask, (int64_t)(d.nb + d.t) * d.b * Hd, 1.f - m->dec_dropout, m->rng_offset[2]);
        MV_TRY(philox_draws(n, kind, out, cnt, keep, off, m->rng_seed, m->rng_step, m->rng_dev_step, s));
    }
    const float enc_keep = dropping ? 1.f / (1.f - m->enc_dropout) : 1.f, dec_keep = dropping ? 1.f / (1.f - m->dec_dropout) : 1.f;
    const uint8_t *beat_mask = dec_mask, *tick_mask = dropping ? dec_mask + (int64_t)d.nb * d.b * Hd : nullptr;
    // the dropout between two stacked GRU layers inside the lower layer's launches (gru_mask.h) instead of a launch of its own
    const bool fuse_masks = dropping && gru_seq_masks_supported() && diag_env("ARVAE_GRU_MASK_APART") == nullptr;

    // ---- encoder (encoder.py:108-124): layer 0's input projection by lookup, both directions side by side
    MV_TRY(lin_fwd(d.v, d.e, 6 * He, P + m->enc_table, P + m->enc_w_ih[0], P + m->enc_b_ih[0], ARVAE_ACT_NONE, w.ptab, s));
    // (the beat RNN's constant input -- a function of the parameters alone -- rides in this lookup's grid: attributes.h)
    bool beat_done = false;
    {
        const BeatInput bi{P + m->b0, P + m->beat_w_ih[0], P + m->beat_b_ih[0], d.b, 3 * Hd, d.rb, w.x0b, w.gi0b};
        int rc = ARVAE_OK;
        beat_done = embed_fwd_with_beat(score, w.ptab, d.b, d.t, 6 * He, d.v, 1, w.gi0, bi, s, &rc);
        if (beat_done) MV_TRY(rc);
        else MV_TRY(arvae_embed_fwd(score, w.ptab, d.b, d.t, 6 * He, d.v, 1, w.gi0, stream));
    }
    arvae_gru_seq_t q[2];
    for (int layer = 0; layer < 2; ++layer) {
        const float *gi = layer == 0 ? w.gi0 : w.gi1;
        float *out = layer == 0 ? w.out0 : w.out1, *sv = layer == 0 ? w.sv0 : w.sv1;
        if (layer == 1) {
            const float *src = w.out0;
            if (dropping) {                                   // (fused: layer 0's recurrence wrote the masked copy itself, gru_mask.h)
                if (!fuse_masks) MV_TRY(arvae_scale_mask(w.out0, enc_mask, enc_keep, (int64_t)d.tb * 2 * He, 0, w.mid, stream));
                src = w.mid;
            }
            MV_TRY(lin_fwd(d.tb, 2 * He, 6 * He, src, P + m->enc_w_ih[1], P + m->enc_b_ih[1], ARVAE_ACT_NONE, w.gi1, s));
        }
        GruSeqMask qm[2] = {};
        for (int dir = 0; dir < 2; ++dir) {
            arvae_gru_seq_t &g = q[dir];
            g = arvae_gru_seq_t{};
            g.gi = gi + dir * 3 * He;
            g.gi_tstride = (int64_t)d.b * 6 * He;
            g.gi_rstride = 6 * He;
            g.w_hh = P + m->enc_w_hh[layer][dir];
            g.b_hh = P + m->enc_b_hh[layer][dir];
            g.h_all = out + dir * He;
            g.h_stride = 2 * He;
            g.saved = sv + (int64_t)dir * d.tb * 4 * He;
            g.reverse = dir;
            g.h_fin = w.hidden + (2 * layer + dir) * He;      // h_n of nn.GRU: (l0 fwd, l0 rev, l1 fwd, l1 rev)
            g.h_fin_stride = 4 * He;
            if (layer == 0 && fuse_masks)                     // the dropout in front of layer 1: keep bytes (t, b, 2 He), this direction's half
                qm[dir] = GruSeqMask{enc_mask + dir * He, enc_keep, w.mid + dir * He, 2 * He, (int64_t)d.b * 2 * He, 2 * He, 0, 0};
        }
        MV_TRY(gru_seq_fwd_masked(q, qm, 2, d.t, d.b, He, stream));
    }
    // the two heads' first layers as one product, then mu / log_std and the reparameterised sample
    MV_TRY(lin_fwd(d.b, 4 * He, 4 * He, w.hidden, P + m->head_w0, P + m->head_b0, ARVAE_ACT_SELU, w.h12, s));
    if (measure_heads_fit(2 * He, d.z)) {                    // one launch (measure_heads_fwd_kernel)
        MeasureHeadsFwd hf{w.h12, P + m->mean_w2, P + m->mean_b2, P + m->lstd_w2, P + m->lstd_b2, eps, w.hmu, w.hls, mu, w.log_std, sigma, z,
                           d.b, 2 * He, d.z};
        const int lds = (MH_ROWS * 4 * He + 2 * d.z * (2 * He + 4) + MH_ROWS * 64) * 4;
        static std::once_flag attr;
        std::call_once(attr, [] { (void)hipFuncSetAttribute((const void *)measure_heads_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); });
        ARVAE_LAUNCH(measure_heads_fwd_kernel, dim3((d.b + MH_ROWS - 1) / MH_ROWS), dim3(256), lds, s, hf);
        MV_TRY(check_launch("measure_heads_fwd_kernel"));
    } else {
        MV_TRY(arvae_split_cols(w.h12, d.b, 2 * He, 2 * He, w.hmu, w.hls, 0, stream));
        MV_TRY(lin_fwd(d.b, 2 * He, d.z, w.hmu, P + m->mean_w2, P + m->mean_b2, ARVAE_ACT_NONE, mu, s));
        MV_TRY(lin_fwd(d.b, 2 * He, d.z, w.hls, P + m->lstd_w2, P + m->lstd_b2, ARVAE_ACT_NONE, w.log_std, s));
        MV_TRY(arvae_latent_fwd(mu, w.log_std, eps, (int64_t)d.b * d.z, sigma, z, stream));
    }

    // ---- beat RNN (decoder.py:436-457): the same input b_0 at every beat
    MV_TRY(lin_fwd(d.b, d.z, 2 * Hd, z, P + m->z2beat_w, P + m->z2beat_b, ARVAE_ACT_SELU, w.flatb, s));
    if (!beat_done) {
        ARVAE_LAUNCH(beat_input_kernel, dim3(blocks_for((int64_t)d.b * 3 * Hd + d.rb)), dim3(256), 0, s,
                     BeatInput{P + m->b0, P + m->beat_w_ih[0], P + m->beat_b_ih[0], d.b, 3 * Hd, d.rb, w.x0b, w.gi0b});
        MV_TRY(check_launch("beat_input_kernel"));
    }
    arvae_gru_seq_t g{};
    g.gi = w.gi0b; g.gi_tstride = 0;
    g.w_hh = P + m->beat_w_hh[0]; g.b_hh = P + m->beat_b_hh[0]; g.h0 = w.flatb; g.h0_stride = 2 * Hd;   // view(B, 2, H)[:, 0]
    g.h_all = w.out0b; g.h_stride = Hd; g.saved = w.svb0;
    GruSeqMask gm{};
    if (fuse_masks) gm = GruSeqMask{beat_mask, dec_keep, w.midb, Hd, (int64_t)d.b * Hd, Hd, 0, 0};
    MV_TRY(gru_seq_fwd_masked(&g, &gm, 1, d.nb, d.b, Hd, stream));
    const float *midb = w.out0b;
    if (dropping) {
        if (!fuse_masks) MV_TRY(arvae_scale_mask(w.out0b, beat_mask, dec_keep, (int64_t)d.rb * Hd, 0, w.midb, stream));
        midb = w.midb;
    }
    MV_TRY(lin_fwd(d.rb, Hd, 3 * Hd, midb, P + m->beat_w_ih[1], P + m->beat_b_ih[1], ARVAE_ACT_NONE, w.gi1b, s));
    g = arvae_gru_seq_t{};
    g.gi = w.gi1b; g.gi_tstride = (int64_t)d.b * 3 * Hd;
    g.w_hh = P + m->beat_w_hh[1]; g.b_hh = P + m->beat_b_hh[1]; g.h0 = w.flatb + Hd; g.h0_stride = 2 * Hd;
    g.h_all = w.beat_out; g.h_stride = Hd; g.saved = w.svb1;
    MV_TRY(arvae_gru_seq_fwd(&g, 1, d.nb, d.b, Hd, stream));

    // ---- tick RNN (decoder.py:459-525): the four beats as one 6-step sequence over beats*batch rows
    MV_TRY(lin_fwd(d.rb, Hd, 3 * Hd, w.beat_out, P + m->tick_init_w, P + m->tick_init_b, ARVAE_ACT_SELU, w.both, s));
    // its columns: [layer-0 initial state | layer-1 initial state | beat embedding], read in place through row strides
    const float *h0t0 = w.both, *h0t1 = w.both + Hd, *beat_emb = w.both + 2 * Hd;
    const int64_t both_ld = 3 * Hd;
    // layer 0's input projection by lookup: W_ih0 applied once to the vocabulary's embeddings, x_0 and the beat embeddings
    MV_TRY(arvae_tick_rows_fwd(P + m->dec_table, P + m->x0, beat_emb, both_ld, d.v, d.e, Hd, d.rb, w.xs, stream));
    MV_TRY(lin_fwd(d.ns, d.e + Hd, 3 * Hd, w.xs, P + m->tick_w_ih[0], nullptr, ARVAE_ACT_NONE, w.gsm, s));
    if (!teacher_forced) {
        // argmax feedback (not differentiated, decoder.py:506-516): the same small product holds the note table's and the beat
        // embeddings' projections the free-running launch reads
        ARVAE_REQUIRE(arvae_tick_free_run_supported(Hd, d.v), "measure_vae_forward: free-running decoder not built for hidden %d / %d notes",
                      Hd, d.v);
        ARVAE_LAUNCH(add_bias_rows_kernel, dim3(blocks_for((int64_t)d.rb * 3 * Hd / 4)), dim3(256), 0, s,
                     reinterpret_cast<const float4 *>(w.gsm + (int64_t)(d.v + 1) * 3 * Hd),
                     reinterpret_cast<const float4 *>(P + m->tick_b_ih[0]), (int64_t)d.rb, 3 * Hd / 4, reinterpret_cast<float4 *>(w.gib));
        MV_TRY(check_launch("add_bias_rows_kernel"));
        arvae_tick_weights_t tw{P + m->tick_w_hh[0], P + m->tick_b_hh[0], P + m->tick_w_ih[1], P + m->tick_b_ih[1],
                                P + m->tick_w_hh[1], P + m->tick_b_hh[1], P + m->out_w, P + m->out_b};
        MV_TRY(arvae_tick_free_run(&tw, h0t0, h0t1, both_ld, w.gib, w.gsm, tick_mask, dec_keep, d.b, d.nb, d.tpb, Hd, d.v, tokens, w.frws, stream));
    }
    // (teacher forcing: the notes fed back are the score's; their copy into `tokens` rides in the lookup launch)
    if (teacher_forced) {
        ARVAE_REQUIRE(tokens != score, "measure_vae_forward: tokens must not alias the score");
        MV_TRY(tick_gi_fwd_copy(w.gsm, score, P + m->tick_b_ih[0], d.b, d.nb, d.tpb, d.v, 3 * Hd, w.gi0t, tokens, s));
    } else {
        MV_TRY(arvae_tick_gi_fwd(w.gsm, tokens, P + m->tick_b_ih[0], d.b, d.nb, d.tpb, d.v, 3 * Hd, w.gi0t, stream));
    }
    g = arvae_gru_seq_t{};
    g.gi = w.gi0t; g.gi_tstride = (int64_t)d.rb * 3 * Hd;
    g.w_hh = P + m->tick_w_hh[0]; g.b_hh = P + m->tick_b_hh[0]; g.h0 = h0t0; g.h0_stride = both_ld;
    g.h_all = w.out0t; g.h_stride = Hd; g.saved = w.svt0;
    // (the tick sequences' rows are (beat, measure) over ticks-in-beat steps; the keep bytes are ordered (tick = tpb * beat + j, measure))
    gm = GruSeqMask{};
    if (fuse_masks) gm = GruSeqMask{tick_mask, dec_keep, w.midt, Hd, (int64_t)d.b * Hd, Hd, (int64_t)d.tpb * d.b * Hd, d.b};
    MV_TRY(gru_seq_fwd_masked(&g, &gm, 1, d.tpb, d.rb, Hd, stream));
    const float *midt = w.out0t;
    if (dropping) {
        if (!fuse_masks) {
            ARVAE_LAUNCH(scale_mask_tick_kernel, dim3(blocks_for((int64_t)d.rt * Hd / 4)), dim3(256), 0, s, w.out0t, tick_mask, dec_keep, d.b,
                         d.nb, d.tpb, Hd / 4, w.midt);
            MV_TRY(check_launch("scale_mask_tick_kernel"));
        }
        midt = w.midt;
    }
    MV_TRY(lin_fwd(d.rt, Hd, 3 * Hd, midt, P + m->tick_w_ih[1], P + m->tick_b_ih[1], ARVAE_ACT_NONE, w.gi1t, s));
    g = arvae_gru_seq_t{};
    g.gi = w.gi1t; g.gi_tstride = (int64_t)d.rb * 3 * Hd;
    g.w_hh = P + m->tick_w_hh[1]; g.b_hh = P + m->tick_b_hh[1]; g.h0 = h0t1; g.h0_stride = both_ld;
    g.h_all = w.out1t; g.h_stride = Hd; g.saved = w.svt1;
    MV_TRY(arvae_gru_seq_fwd(&g, 1, d.tpb, d.rb, Hd, stream));
    MV_TRY(lin_fwd(d.rt, Hd, d.v, w.out1t, P + m->out_w, P + m->out_b, ARVAE_ACT_RELU, w.probs, s));

    // ---- loss terms (measure_vae_trainer.py:85-140): cross entropy over the 24*B rows (in the sequence launches' row order, targets
    // looked up through it), the attribute labels and the regulariser's pair sums, then ONE finishing launch: the partial sums,
    // beta-KL, the regulariser's gradient and the pass's scalars
    int nb = 0;
    float *lab = labels != nullptr ? labels : w.labels;
    // (the attribute labels -- a function of the score alone -- ride in the cross-entropy launch's grid: attributes.h)
    const AttrArgs attr{score, d.b, d.t, tables != nullptr ? tables->midi_lut : nullptr, tables != nullptr ? tables->is_note : nullptr,
                        tables != nullptr ? tables->is_density_note : nullptr, d.v, tables != nullptr ? tables->rhythm_weights : nullptr,
                        tables != nullptr ? tables->rhythm_norm : 1.f, m->n_reg > 0 ? lab : nullptr};
    MV_TRY(token_recon_partials(w.probs, score, d.b, d.nb, d.tpb, d.v, w.rec_ws, w.dprobs, s, &nb, &attr));
    if (defer_finish) return ARVAE_OK;                   // data parallel: arvae_measure_vae_finish, once z and the labels are gathered
    if (m->n_reg > 0) {
        RegDims rd;
        for (int i = 0; i < 16; ++i) rd.d[i] = i < m->n_reg ? m->reg_dims[i] : 0;
        for (int i = 0; i < m->n_reg; ++i)
            ARVAE_REQUIRE(m->reg_dims[i] >= 0 && m->reg_dims[i] < d.z && m->reg_dims[i] < 4, "measure_vae_forward: regularised dim %d outside z / the attributes",
                          m->reg_dims[i]);
        MV_TRY(reg_partials(z, lab, d.b, z, lab, d.b, d.z, 4, rd, m->n_reg, m->delta, w.reg_ws, s));
    }
    return vae_finish(w.rec_ws, nb, d.b, d.rt, mu, sigma, d.z, m->beta, capacity, m->n_reg > 0 ? w.reg_ws : nullptr, d.b, d.z, m->reg_dims,
                      m->n_reg, m->gamma, m->delta, 1.f, w.dz_reg, w.ce_out, w.kld_out, w.reg_out, scalars, s, d.rt);
}

extern "C" int arvae_measure_vae_finish(const arvae_measure_vae_t *m, int32_t batch, const float *capacity, const float *z_cols,
                                        const float *lab_cols, int64_t n_cols, float reg_scale, float *ws, float *scalars, const float *mu,
                                        const float *sigma, const float *z, const float *labels, arvae_stream_t stream) {
    MV_TRY(check_model(m, batch, "measure_vae_finish"));
    ARVAE_REQUIRE(ws && scalars && mu && sigma && z, "measure_vae_finish: null pointer");
    ARVAE_REQUIRE(m->n_reg == 0 || (z_cols && lab_cols && labels && n_cols >= batch), "measure_vae_finish: the regulariser needs the gathered columns");
    hipStream_t s = as_stream(stream);
    MvWs w{};
    carve(m, batch, ws, &w);
    const MvDims d = dims_of(m, batch);
    if (m->n_reg > 0) {
        RegDims rd;
        for (int i = 0; i < 16; ++i) rd.d[i] = i < m->n_reg ? m->reg_dims[i] : 0;
        for (int i = 0; i < m->n_reg; ++i)
            ARVAE_REQUIRE(m->reg_dims[i] >= 0 && m->reg_dims[i] < d.z && m->reg_dims[i] < 4, "measure_vae_finish: regularised dim %d outside z / the attributes",
                          m->reg_dims[i]);
        MV_TRY(reg_partials(z, labels, d.b, z_cols, lab_cols, n_cols, d.z, 4, rd, m->n_reg, m->delta, w.reg_ws, s));
    }
    const int nb = token_recon_blocks(d.rt);
    return vae_finish(w.rec_ws, nb, d.b, d.rt, mu, sigma, d.z, m->beta, capacity, m->n_reg > 0 ? w.reg_ws : nullptr, n_cols, d.z, m->reg_dims,
                      m->n_reg, m->gamma, m->delta, reg_scale, w.dz_reg, w.ce_out, w.kld_out, w.reg_out, scalars, s, d.rt);
}

extern "C" int arvae_measure_vae_backward(const arvae_measure_vae_t *m, int32_t batch, const float *params, float *grads,
                                          const int64_t *score, const float *eps, const uint8_t *enc_mask, const uint8_t *dec_mask,
                                          const float *capacity, const float *mu, const float *sigma, const float *z,
                                          const int64_t *tokens, const float *scalars, const float *g_loss, float reg_scale, float *ws,
                                          arvae_stream_t stream) {
    MV_TRY(check_model(m, batch, "measure_vae_backward"));
    ARVAE_REQUIRE(params && grads && score && eps && mu && sigma && z && tokens && scalars && g_loss && ws, "measure_vae_backward: null pointer");
    ARVAE_REQUIRE((enc_mask == nullptr) == (dec_mask == nullptr), "measure_vae_backward: give both keep-masks or neither");
    hipStream_t s = as_stream(stream);
    MvWs w{};
    carve(m, batch, ws, &w);
    const MvDims d = dims_of(m, batch);
    const float *P = params;
    float *G = grads;
    const int He = d.he, Hd = d.hd;
    const bool dropping = enc_mask != nullptr;
    const float enc_keep = dropping ? 1.f / (1.f - m->enc_dropout) : 1.f, dec_keep = dropping ? 1.f / (1.f - m->dec_dropout) : 1.f;
    const uint8_t *beat_mask = dec_mask, *tick_mask = dropping ? dec_mask + (int64_t)d.nb * d.b * Hd : nullptr;
    // the lower layers' recurrences multiply the gradient they load by keep * mask themselves (gru_mask.h)
    const bool fuse_masks = dropping && gru_seq_masks_supported() && diag_env("ARVAE_GRU_MASK_APART") == nullptr;
    WgradQueues queue;
    dense_wgrad_long_begin(queue.rows, w.wg_long, w.wg_long_floats);
    queue.ws = w.wg_ws;

    // ---- note projection: d probs (cross entropy, unit upstream) times the upstream scalar, through the ReLU
    const int64_t np = (int64_t)d.rt * d.v;
    if (np % 4 == 0)
        ARVAE_LAUNCH(relu_gate_scale_kernel, dim3(blocks_for(np / 4)), dim3(256), 0, s, w.dprobs, w.probs, g_loss, np / 4, w.gpre);
    else
        ARVAE_LAUNCH(relu_gate_scale1_kernel, dim3(blocks_for(np)), dim3(256), 0, s, w.dprobs, w.probs, g_loss, np, w.gpre);
    MV_TRY(check_launch("relu_gate_scale_kernel"));
    MV_TRY(lin_dgrad(d.rt, Hd, d.v, plain(w.gpre), P + m->out_w, w.d_seq_h, s));
    MV_TRY(lin_wgrad(&queue, d.rt, Hd, d.v, plain(w.gpre), w.out1t, G + m->out_w, G + m->out_b, s));

    // ---- tick RNN, layer 1 then layer 0
    const float *midt = dropping ? w.midt : w.out0t;
    arvae_gru_seq_t g{};
    g.w_hh = P + m->tick_w_hh[1]; g.h0 = w.both + Hd; g.h0_stride = 3 * Hd; g.h_all = w.out1t; g.h_stride = Hd; g.saved = w.svt1;
    g.dh_all = w.d_seq_h; g.dh_stride = Hd; g.dgi = w.dgi_t1; g.dgh = w.dgh_t1; g.h_prev_out = w.hprev_t1;
    g.dh0 = w.d_both + Hd; g.dh0_stride = 3 * Hd;           // the initial states' gradients land in their columns of d_both
    MV_TRY(arvae_gru_seq_bwd(&g, 1, d.tpb, d.rb, Hd, stream));
    MV_TRY(lin_wgrad(&queue, d.rt, Hd, 3 * Hd, plain(w.dgh_t1), w.hprev_t1, G + m->tick_w_hh[1], G + m->tick_b_hh[1], s));
    MV_TRY(lin_dgrad(d.rt, Hd, 3 * Hd, plain(w.dgi_t1), P + m->tick_w_ih[1], w.d_mid_t, s));
    MV_TRY(lin_wgrad(&queue, d.rt, Hd, 3 * Hd, plain(w.dgi_t1), midt, G + m->tick_w_ih[1], G + m->tick_b_ih[1], s));
    const float *d_out0t = w.d_mid_t;
    GruSeqMask gm{};
    if (fuse_masks) {
        gm = GruSeqMask{tick_mask, dec_keep, nullptr, 0, (int64_t)d.b * Hd, Hd, (int64_t)d.tpb * d.b * Hd, d.b};
    } else if (dropping) {
        ARVAE_LAUNCH(scale_mask_tick_kernel, dim3(blocks_for((int64_t)d.rt * Hd / 4)), dim3(256), 0, s, w.d_mid_t, tick_mask, dec_keep, d.b,
                     d.nb, d.tpb, Hd / 4, w.d_seq_h);
        MV_TRY(check_launch("scale_mask_tick_kernel"));
        d_out0t = w.d_seq_h;
    }
    g = arvae_gru_seq_t{};
    g.w_hh = P + m->tick_w_hh[0]; g.h0 = w.both; g.h0_stride = 3 * Hd; g.h_all = w.out0t; g.h_stride = Hd; g.saved = w.svt0;
    g.dh_all = d_out0t; g.dh_stride = Hd; g.dgi = w.dgi_t0; g.dgh = w.dgh_t0; g.h_prev_out = w.hprev_t0;
    g.dh0 = w.d_both; g.dh0_stride = 3 * Hd;
    MV_TRY(gru_seq_bwd_masked(&g, &gm, 1, d.tpb, d.rb, Hd, stream));
    MV_TRY(lin_wgrad(&queue, d.rt, Hd, 3 * Hd, plain(w.dgh_t0), w.hprev_t0, G + m->tick_w_hh[0], G + m->tick_b_hh[0], s));
    // layer 0's input projection: per-tick gradients summed per previous note and per beat row, then the small product's adjoints
    MV_TRY(arvae_tick_gi_bwd(w.dgi_t0, tokens, d.b, d.nb, d.tpb, d.v, 3 * Hd, w.dg_small, w.tick_ws, stream));
    // (every tick row carries the bias once and exactly one note entry: the bias gradient is the column sum of the note rows)
    // (its column sums -- the bias gradient -- wait for the pass's closing launch: grad_tail_kernel)
    MV_TRY(lin_wgrad(&queue, d.ns, d.e + Hd, 3 * Hd, plain(w.dg_small), w.xs, G + m->tick_w_ih[0], nullptr, s));
    MV_TRY(lin_dgrad(d.ns, d.e + Hd, 3 * Hd, plain(w.dg_small), P + m->tick_w_ih[0], w.dx_small, s));
    MV_TRY(arvae_tick_rows_bwd(w.dx_small, d.v, d.e, Hd, d.rb, G + m->dec_table, G + m->x0, w.d_both + 2 * Hd, 3 * Hd, stream));
    // initial states + beat-embedding input: one SELU layer on the beat outputs (d_both is complete: three column blocks)
    MV_TRY(lin_dgrad(d.rb, Hd, 3 * Hd, gated(w.d_both, w.both, ARVAE_ACT_SELU), P + m->tick_init_w, w.d_rows_h, s));
    MV_TRY(lin_wgrad(&queue, d.rb, Hd, 3 * Hd, gated(w.d_both, w.both, ARVAE_ACT_SELU), w.beat_out, G + m->tick_init_w, G + m->tick_init_b,
                     s));

    // ---- beat RNN, layer 1 then layer 0 (their batch-sized weight gradients wait in the queue: separate buffers per layer)
    const float *midb = dropping ? w.midb : w.out0b;
    g = arvae_gru_seq_t{};
    g.w_hh = P + m->beat_w_hh[1]; g.h0 = w.flatb + Hd; g.h0_stride = 2 * Hd; g.h_all = w.beat_out; g.h_stride = Hd; g.saved = w.svb1;
    g.dh_all = w.d_rows_h; g.dh_stride = Hd; g.dgi = w.dgi_b1; g.dgh = w.dgh_b1; g.h_prev_out = w.hprev_b1;
    g.dh0 = w.d_flatb + Hd; g.dh0_stride = 2 * Hd;
    MV_TRY(arvae_gru_seq_bwd(&g, 1, d.nb, d.b, Hd, stream));
    MV_TRY(lin_wgrad(&queue, d.rb, Hd, 3 * Hd, plain(w.dgh_b1), w.hprev_b1, G + m->beat_w_hh[1], G + m->beat_b_hh[1], s));
    MV_TRY(lin_dgrad(d.rb, Hd, 3 * Hd, plain(w.dgi_b1), P + m->beat_w_ih[1], w.d_mid_b, s));
    MV_TRY(lin_wgrad(&queue, d.rb, Hd, 3 * Hd, plain(w.dgi_b1), midb, G + m->beat_w_ih[1], G + m->beat_b_ih[1], s));
    const float *d_out0b = w.d_mid_b;
    gm = GruSeqMask{};
    if (fuse_masks) {
        gm = GruSeqMask{beat_mask, dec_keep, nullptr, 0, (int64_t)d.b * Hd, Hd, 0, 0};
    } else if (dropping) {
        MV_TRY(arvae_scale_mask(w.d_mid_b, beat_mask, dec_keep, (int64_t)d.rb * Hd, 0, w.d_rows_h, stream));
        d_out0b = w.d_rows_h;
    }
    g = arvae_gru_seq_t{};
    g.w_hh = P + m->beat_w_hh[0]; g.h0 = w.flatb; g.h0_stride = 2 * Hd; g.h_all = w.out0b; g.h_stride = Hd; g.saved = w.svb0;
    g.dh_all = d_out0b; g.dh_stride = Hd; g.dgi = w.dgi_b0; g.dgh = w.dgh_b0; g.h_prev_out = w.hprev_b0;
    g.dh0 = w.d_flatb; g.dh0_stride = 2 * Hd;
    MV_TRY(gru_seq_bwd_masked(&g, &gm, 1, d.nb, d.b, Hd, stream));
    MV_TRY(lin_wgrad(&queue, d.rb, Hd, 3 * Hd, plain(w.dgh_b0), w.hprev_b0, G + m->beat_w_hh[0], G + m->beat_b_hh[0], s));
    // the constant input b_0 (decoder.py:436-440): the projection's gradients over all beats*batch rows (x0b holds b_0 once per row)
    MV_TRY(lin_wgrad(&queue, d.rb, 1, 3 * Hd, plain(w.dgi_b0), w.x0b, G + m->beat_w_ih[0], G + m->beat_b_ih[0], s));
    MV_TRY(lin_dgrad(d.rb, 1, 3 * Hd, plain(w.dgi_b0), P + m->beat_w_ih[0], w.d_x0, s));
    // (the gradient of b_0, the sum of d_x0, in grad_tail_kernel)
    MV_TRY(lin_dgrad(d.b, d.z, 2 * Hd, gated(w.d_flatb, w.flatb, ARVAE_ACT_SELU), P + m->z2beat_w, w.d_z, s));
    MV_TRY(lin_wgrad(&queue, d.b, d.z, 2 * Hd, gated(w.d_flatb, w.flatb, ARVAE_ACT_SELU), z, G + m->z2beat_w, G + m->z2beat_b, s));

    // ---- latent head: decoder path + regulariser + beta-KL -> (d mu, d log_std), then the heads' two layers
    const bool heads_fused = measure_heads_fit(2 * He, d.z);
    if (heads_fused) {                                        // (d mu, d log_std) and both heads' data gradients: one launch
        MeasureHeadsBwd hb{w.d_z, m->n_reg > 0 ? w.dz_reg : nullptr, mu, sigma, eps, g_loss, scalars + ARVAE_VAE_KL, capacity,
                           P + m->mean_w2, P + m->lstd_w2, m->beta, 1.f / (float)d.b, reg_scale, w.d_mu, w.d_ls, w.d_h12, d.b, 2 * He, d.z};
        ARVAE_LAUNCH(measure_heads_bwd_kernel, dim3((d.b + MH_ROWS - 1) / MH_ROWS), dim3(256), 0, s, hb);
        MV_TRY(check_launch("measure_heads_bwd_kernel"));
    } else {
        ARVAE_LAUNCH(measure_latent_bwd_kernel, dim3(blocks_for((int64_t)d.b * d.z)), dim3(256), 0, s, w.d_z, m->n_reg > 0 ? w.dz_reg : nullptr, mu,
                     sigma, eps, g_loss, scalars + ARVAE_VAE_KL, capacity, m->beta, 1.f / (float)d.b, reg_scale, (int64_t)d.b * d.z, w.d_mu, w.d_ls);
        MV_TRY(check_launch("measure_latent_bwd_kernel"));
        MV_TRY(lin_dgrad(d.b, 2 * He, d.z, plain(w.d_mu), P + m->mean_w2, w.d_hmu, s));
        MV_TRY(lin_dgrad(d.b, 2 * He, d.z, plain(w.d_ls), P + m->lstd_w2, w.d_hls, s));
    }
    MV_TRY(lin_wgrad(&queue, d.b, 2 * He, d.z, plain(w.d_mu), w.hmu, G + m->mean_w2, G + m->mean_b2, s));
    MV_TRY(lin_wgrad(&queue, d.b, 2 * He, d.z, plain(w.d_ls), w.hls, G + m->lstd_w2, G + m->lstd_b2, s));
    if (!heads_fused) MV_TRY(arvae_concat_cols(w.d_hmu, w.d_hls, d.b, 2 * He, 2 * He, w.d_h12, stream));
    MV_TRY(lin_dgrad(d.b, 4 * He, 4 * He, gated(w.d_h12, w.h12, ARVAE_ACT_SELU), P + m->head_w0, w.d_hidden, s));
    MV_TRY(lin_wgrad(&queue, d.b, 4 * He, 4 * He, gated(w.d_h12, w.h12, ARVAE_ACT_SELU), w.hidden, G + m->head_w0, G + m->head_b0, s));

    // ---- encoder, layer 1 then layer 0: the final states' gradients enter each direction at its last processed step
    arvae_gru_seq_t q[2];
    for (int layer = 1; layer >= 0; --layer) {
        const float *out = layer == 0 ? w.out0 : w.out1, *sv = layer == 0 ? w.sv0 : w.sv1;
        float *dgi = layer == 0 ? w.dgi_e0 : w.dgi_e1, *dgh = layer == 0 ? w.dgh_e0 : w.dgh_e1, *hprev = layer == 0 ? w.hprev_e0 : w.hprev_e1;
        const float *d_out = nullptr;
        GruSeqMask qm[2] = {};
        if (layer == 0) {
            d_out = w.d_mid_e;
            if (fuse_masks) {
                for (int dir = 0; dir < 2; ++dir)
                    qm[dir] = GruSeqMask{enc_mask + dir * He, enc_keep, nullptr, 0, (int64_t)d.b * 2 * He, 2 * He, 0, 0};
            } else if (dropping) {
                MV_TRY(arvae_scale_mask(w.d_mid_e, enc_mask, enc_keep, (int64_t)d.tb * 2 * He, 0, w.d_out0_e, stream));
                d_out = w.d_out0_e;
            }
        }
        for (int dir = 0; dir < 2; ++dir) {
            arvae_gru_seq_t &e = q[dir];
            e = arvae_gru_seq_t{};
            e.w_hh = P + m->enc_w_hh[layer][dir];
            e.h_all = const_cast<float *>(out) + dir * He;
            e.h_stride = 2 * He;
            e.saved = const_cast<float *>(sv) + (int64_t)dir * d.tb * 4 * He;
            e.reverse = dir;
            if (d_out != nullptr) {
                e.dh_all = d_out + dir * He;
                e.dh_stride = 2 * He;
            }
            e.dh_last = w.d_hidden + (2 * layer + dir) * He;
            e.dh_last_stride = 4 * He;
            e.dgi = dgi + dir * 3 * He;
            e.dgi_rstride = 6 * He;
            e.dgh = dgh + (int64_t)dir * d.tb * 3 * He;
            e.h_prev_out = hprev + (int64_t)dir * d.tb * He;
        }
        MV_TRY(gru_seq_bwd_masked(q, qm, 2, d.t, d.b, He, stream));
        for (int dir = 0; dir < 2; ++dir)
            MV_TRY(lin_wgrad(&queue, d.tb, He, 3 * He, plain(dgh + (int64_t)dir * d.tb * 3 * He), hprev + (int64_t)dir * d.tb * He,
                             G + m->enc_w_hh[layer][dir], G + m->enc_b_hh[layer][dir], s));
        if (layer == 1) {
            const float *src = dropping ? w.mid : w.out0;
            MV_TRY(lin_wgrad(&queue, d.tb, 2 * He, 6 * He, plain(dgi), src, G + m->enc_w_ih[1], G + m->enc_b_ih[1], s));
            MV_TRY(lin_dgrad(d.tb, 2 * He, 6 * He, plain(dgi), P + m->enc_w_ih[1], w.d_mid_e, s));
        }
    }
    // layer 0's projection table: per-position gradients summed per token, then the small product's adjoints
    MV_TRY(arvae_embed_bwd(score, w.dgi_e0, d.b, d.t, 6 * He, d.v, 1, w.dptab, 0, w.embed_ws, stream));
    MV_TRY(lin_dgrad(d.v, d.e, 6 * He, plain(w.dptab), P + m->enc_w_ih[0], w.d_table, s));
    {
        GradTail t{w.dg_small, d.v + 1, 3 * Hd, G + m->tick_b_ih[0], (3 * Hd + 255) / 256, w.d_x0, d.rb, G + m->b0, w.d_table, d.v * d.e, G + m->enc_table};
        ARVAE_LAUNCH(grad_tail_kernel, dim3(t.cs_blocks + 1 + (t.add_n + 255) / 256), dim3(256), 0, s, t);
        MV_TRY(check_launch("grad_tail_kernel"));
    }
    MV_TRY(lin_wgrad(&queue, d.v, d.e, 6 * He, plain(w.dptab), P + m->enc_table, G + m->enc_w_ih[0], G + m->enc_b_ih[0], s));
    MV_TRY(dense_wgrad_long_flush(queue.rows, s));
    return dense_wgrad_flush(&queue.batch, s);
}
