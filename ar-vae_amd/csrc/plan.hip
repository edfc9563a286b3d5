// Whole-model forward / backward executors for the conv AR-VAEs (see include/arvae_hip.h): one host call
// enqueues every kernel of a pass on the caller's stream.  Host-side sequencing only -- the math lives in
// the link / dense / loss kernels, reached through the same C-ABI entry points a per-layer caller uses.
#include "diag.h"
#include "common.h"
#include "dense.h"
#include "reduce.h"
#include "midprep.h"
#include "regloss.h"
#include "vae_finish.h"
#include "conv32_common.h"

namespace arvae {

// fast kernels with a gated epilogue (conv32.hip / conv_c1.hip / dense.hip)
bool conv32_fits(const arvae_link_t *l);
bool conv_c1_fits(const arvae_link_t *l);
bool conv64_fits(const arvae_link_t *l, bool up);
bool conv64s_fits(const arvae_link_t *l, bool up);
int64_t conv64s_ws_floats();
int conv64s_prep_batch(const float *const *wts, float *const *outs, const int *transposed, const int *q, int count, hipStream_t s);
bool conv64_wgrad_fits(const arvae_link_t *l);
int link_wgrad_conv64(const arvae_link_t *link, const Operand &lo, const Operand &hi, float *dwt, float *dbias, int bias_side, float *ws,
                      hipStream_t st, const unsigned *amax_lo, const unsigned *amax_hi);
bool single_channel_down_gated_fits(const arvae_link_t *l);
int single_channel_down_gated(const arvae_link_t *link, const Operand &hi, const float *wt, const GateOp *gate, float *lo, hipStream_t s,
                              unsigned *amax_out);
bool single_channel_down_fits(const arvae_link_t *l);
int single_channel_down(const arvae_link_t *link, const Operand &hi, const float *wt, const float *bias, int act, const uint8_t *mask,
                        float *lo, hipStream_t s, unsigned *amax_out);
int conv64_down(const arvae_link_t *l, const Operand &hi, const float *wt, const float *bias, int act, const uint8_t *mask,
                float *lo, float *ws, hipStream_t s, const GateOp *gate, const unsigned *amax_in = nullptr, unsigned *amax_out = nullptr,
                float *prepped = nullptr);
int conv64_up(const arvae_link_t *l, const Operand &lo, const float *wt, const float *bias, int act, const uint8_t *mask,
              float *hi, float *ws, hipStream_t s, const GateOp *gate, const unsigned *amax_in = nullptr, unsigned *amax_out = nullptr,
              float *prepped = nullptr);
// (32-channel conv kernels, conv32.hip: operands come with their AMAX arrays, conv32_common.h)
int conv32_down(const arvae_link_t *l, const Operand &hi, const float *bias, int relu, const float *gate, const uint16_t *gate_bits,
                uint16_t *bits_out, float *out, hipStream_t s, const float *wprep, const unsigned *amax_in, unsigned *amax_out);
int conv32_up(const arvae_link_t *l, const Operand &lo, const float *bias, int relu, const float *gate, const uint16_t *gate_bits,
              uint16_t *bits_out, float *out, hipStream_t s, const float *wprep, const unsigned *amax_in, unsigned *amax_out);
int conv32_amax(const float *x, int64_t count, unsigned *out, hipStream_t s);
int conv_c1_up_recon(const arvae_link_t *l, const float *lo, const float *wt, const float *bias, float *out, const float *x,
                     int dist, float *partial, float *dlogits, hipStream_t s, int *nb_out, const VaeFinishArgs *fin = nullptr,
                     VaeFinishArgs *fin_dst = nullptr);
int vae_finish_deferred(const VaeFinishArgs *fin_dev, hipStream_t s);
VaeFinishArgs vae_finish_args(const float *rec_partial, int nb, int64_t batch, int64_t pix, const float *mu, const float *sigma,
                              int64_t zdim, float beta, const float *cap, const float *reg_ws, int64_t n_cols, int64_t ldz,
                              const int32_t *dims, int32_t r, float gamma, float delta, float reg_scale, float *dz, float *rec_out,
                              float *kld_out, float *reg_out, float *scalars, int64_t rec_rows);
bool conv32_up_reg_fits(const arvae_link_t *l);
bool conv32_down_chain_fits(const arvae_link_t *a, const arvae_link_t *b);
int conv32_down_chain(const arvae_link_t *a, const arvae_link_t *b, const float *hi, const unsigned *amax_in, const float *bias_a,
                      uint16_t *bits_a, float *out_a, const float *wprep_a, unsigned *amax_a, const float *bias_b, uint16_t *bits_b,
                      float *out_b, const float *wprep_b, unsigned *amax_b, hipStream_t s);
int conv32_up_reg(const arvae_link_t *l, const Operand &lo, const float *bias, uint16_t *bits_out, float *out, const float *wprep,
                  const unsigned *amax_in, unsigned *amax_out, const RegArgs &reg, int r, hipStream_t s);
int conv_c1_up_recon_blocks(const arvae_link_t *l);
int recon_partial_blocks(int64_t count);
int64_t conv32_prep_floats();
int conv32_weight_prep(const float *const *wts, float *const *preps, int n_layers, hipStream_t s);
int conv32_weight_prep_with_mid(const float *const *wts, float *const *preps, int n_layers, const MidPrepArgs &mid, hipStream_t s);
void mid_prep_args(const arvae_image_vae_t *m, const float *params, float *prep_ws, MidPrepArgs *out, int batch);
bool conv32_pair_fits(const arvae_link_t *l, bool up, const float *gate, const uint16_t *gate_bits, int bias_mode);
int conv32_pair(const arvae_link_t *l, bool up, const float *g, const float *x_in, const float *gate, const uint16_t *gate_bits,
                 float *d_in, const float *wprep, float *dwt, float *dbias, float *slab, hipStream_t s, SlabJob *job,
                 const unsigned *amax_g, const unsigned *amax_x, unsigned *amax_out, const float *c1_img, float *c1_slab, SlabJob *c1_job);
bool conv32_pair_c1_fits(const arvae_link_t *l, bool up, const uint16_t *gate_bits);
int conv32_wgrad_partial(const arvae_link_t *l, const Operand &lo, const Operand &hi, float *dwt, float *dbias, int bias_mode,
                         float *slab, hipStream_t s, SlabJob *job, const unsigned *amax_lo, const unsigned *amax_hi);
bool conv_c1_pair_fits(const arvae_link_t *l);
int conv_c1_pair(const arvae_link_t *l, const Operand &g_img, const float *wt, const float *gate, const uint16_t *gate_bits, float *d_lo,
                 const Operand &w_lo, float *dwt, float *dbias, int bias_mode, float *slab, hipStream_t s, SlabJob *job,
                 unsigned *amax_out, const VaeFinishArgs *finish = nullptr);
int conv_c1_wgrad_partial(const arvae_link_t *l, const Operand &lo, const Operand &img, float *dwt, float *dbias,
                          int bias_mode, float *slab, hipStream_t s, SlabJob *job);
bool dense_wgrad_c1_fits(const DenseWgradBatch *b);
int dense_wgrad_flush_with_c1(DenseWgradBatch *b, const arvae_link_t *l, const Operand &lo, const Operand &img, float *dwt, float *dbias,
                              int bias_mode, float *slab, hipStream_t s, SlabJob *job);
int conv_c1_down(const arvae_link_t *l, const Operand &img, const float *wt, const float *bias, int relu,
                 const float *gate, const uint16_t *gate_bits, uint16_t *bits_out, float *out, hipStream_t s, unsigned *amax_out);
int conv_c1_down_with_prep(const arvae_link_t *l, const Operand &img, const float *wt, const float *bias, int relu, uint16_t *bits_out,
                           float *out, const float *const *prep_wts, float *const *preps, int n_prep, const MidPrepArgs &mid,
                           hipStream_t s, unsigned *amax_out);

// fused encoder heads + reparameterisation (heads.hip)
bool heads_fusable(const arvae_layer_t *hm, const arvae_layer_t *hl, int zdim);
int heads_latent_fwd(const arvae_layer_t *hm, const arvae_layer_t *hl, int batch, int zdim, const float *params,
                     const float *hidden, const float *eps, float *mu, float *log_std, float *sigma, float *z, hipStream_t s,
                     const arvae_image_vae_t *rng_model = nullptr, const arvae_layer_t *next = nullptr, float *next_out = nullptr);
bool heads_next_fusable(const arvae_layer_t *l, int zdim);
int heads_latent_bwd(const arvae_layer_t *hm, const arvae_layer_t *hl, int batch, int zdim, const float *params,
                     const float *g_z, const float *dz_reg, const float *dz_extra, const float *mu, const float *sigma,
                     const float *eps, const float *g_loss, const float *kl, const float *cap, float beta, float reg_scale,
                     const float *gate, float *d_mu, float *d_ls, float *d_hidden, hipStream_t s, const arvae_layer_t *next = nullptr,
                     const float *next_g = nullptr);

// the latent block (trailing Linear layers of the encoder, heads + reparameterisation, leading Linear layers of the
// decoder) as one launch per pass (midblock.hip)
bool mid_fusable(const arvae_image_vae_t *m, int *ne_out, int *nd_out);
int64_t mid_prep_floats(const arvae_image_vae_t *m);
int mid_forward(const arvae_image_vae_t *m, int batch, const float *params, float *prep_ws, const float *x0, float *const *enc_y,
                float *const *dec_y, const float *eps, float *mu, float *log_std, float *sigma, float *z, hipStream_t s, bool prep_done,
                unsigned *amax_out, const MidFold *fold, float *wide_ws);
int64_t mid_wide_ws_floats(const arvae_image_vae_t *m, int batch);      // split-reduction workspace of the wide layers' tile GEMMs
int mid_wide_wgrad(const arvae_image_vae_t *m, int batch, const float *params, float *prep_ws, float *wide_ws, const float *x0,
                   const float *g_last_pre, float *grads, hipStream_t s, int *took);
// the conv layers on either side of the latent block computed by the block's clustered kernels (midblock.hip)
bool mid_fold_fits(const arvae_image_vae_t *m, int batch);
int64_t mid_fold_slab_floats(const arvae_image_vae_t *m, int batch);
int mid_backward(const arvae_image_vae_t *m, int batch, const float *params, float *prep_ws, float *const *enc_y, float *const *dec_y,
                 float *const *enc_g, float *const *dec_g, const float *g_out, int g_is_pre, const float *gate0, float *d_x0,
                 const float *eps, const float *mu, const float *sigma, const float *dz_reg, const float *dz_extra, const float *g_loss,
                 const float *kl, const float *cap, float beta, float reg_scale, float *d_mu, float *d_ls, hipStream_t s,
                 unsigned *amax_out, const MidFold *fold, float *wide_ws);

// loss-term pieces (losses.hip)
int recon_partials(const float *logits, const float *x, int64_t count, int64_t batch, int32_t dist, float *ws,
                   float *dlogits, hipStream_t s, int *nb_out);
int reg_partials(const float *z_rows, const float *lab_rows, int64_t n_rows, const float *z_cols, const float *lab_cols,
                 int64_t n_cols, int64_t ldz, int64_t ldl, const RegDims &rd, int32_t r, float delta, float *ws,
                 hipStream_t s, const VaeFinishArgs *park = nullptr, VaeFinishArgs *park_dst = nullptr);
int vae_finish(const float *rec_partial, int nb, int64_t batch, int64_t pix, const float *mu, const float *sigma,
               int64_t zdim, float beta, const float *cap, const float *reg_ws, int64_t n_cols, int64_t ldz,
               const int32_t *dims, int32_t r, float gamma, float delta, float reg_scale, float *dz, float *rec_out,
               float *kld_out, float *reg_out, float *scalars, hipStream_t s, int64_t rec_rows = 0);

// ---- small glue kernels ---------------------------------------------------------------------------
// gradient of the loss w.r.t. (mu, log_std) from: the decoder path g_z (already times g), the
// regularisation gradient dz_reg (unit upstream, scaled by g*reg_scale here), an optional external
// z gradient, and the beta-KL term; sigma = exp(log_std), z = mu + eps*sigma.
__global__ __launch_bounds__(256) void latent_bwd_full_kernel(const float *__restrict__ g_z, const float *__restrict__ dz_reg,
                                                               const float *__restrict__ dz_extra, const float *__restrict__ mu,
                                                               const float *__restrict__ sigma, const float *__restrict__ eps,
                                                               const float *__restrict__ g_loss, const float *__restrict__ kl,
                                                               const float *__restrict__ cap, float beta, float inv_batch,
                                                               float reg_scale, int64_t count, float *__restrict__ d_mu,
                                                               float *__restrict__ d_ls) {
    const float g = g_loss[0];
    const float diff = kl[0] - (cap != nullptr ? cap[0] : 0.f);
    const float k = g * beta * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) * inv_batch;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        float gz = g_z[i];
        if (dz_reg != nullptr) gz += g * reg_scale * dz_reg[i];
        if (dz_extra != nullptr) gz += dz_extra[i];
        const float s = sigma[i];
        d_mu[i] = gz + k * mu[i];
        d_ls[i] = (gz * eps[i] + k * (s - 1.f / s)) * s;
    }
}

// a += b, optionally gated by the saved ReLU output the gradient belongs to
__global__ __launch_bounds__(256) void add_inplace_kernel(float *__restrict__ a, const float *__restrict__ b,
                                                           const float *__restrict__ gate, int64_t count) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        const float v = a[i] + b[i];
        a[i] = (gate == nullptr || gate[i] > 0.f) ? v : 0.f;
    }
}

// ---- workspace layout ------------------------------------------------------------------------------
static inline int64_t up4(int64_t v) { return (v + 3) / 4 * 4; }

static inline int64_t out_elems(const arvae_layer_t &l, int64_t n) {
    return l.is_up ? n * l.link.hh * l.link.hw * l.link.chi : n * l.link.lh * l.link.lw * l.link.clo;
}
static inline int64_t in_elems(const arvae_layer_t &l, int64_t n) {
    return l.is_up ? n * l.link.lh * l.link.lw * l.link.clo : n * l.link.hh * l.link.hw * l.link.chi;
}

struct Layout {
    int64_t enc_out[ARVAE_MAX_LAYERS], dec_out[ARVAE_MAX_LAYERS];   // dec_out[last] unused (logits are external)
    // the gradient w.r.t. each layer's output (kept until the end of the backward pass)
    int64_t enc_keep[ARVAE_MAX_LAYERS], dec_keep[ARVAE_MAX_LAYERS];
    // conv layers with a slab kernel: their own slab, so that all the reductions can run as one launch at the end
    int64_t enc_slab[ARVAE_MAX_LAYERS], dec_slab[ARVAE_MAX_LAYERS];
    // ReLU conv layers on the fast kernels also leave the sign bits of their output (relu_bits16, 4 bytes per pixel):
    // the backward pass gates with those instead of re-reading the 128-byte-per-pixel activation
    int64_t enc_bits[ARVAE_MAX_LAYERS], dec_bits[ARVAE_MAX_LAYERS];
    // 32-channel conv layers: the weights split into bf16 terms in per-lane order, rebuilt at the start of every forward
    // pass by ONE launch and used by the layer's forward and data-gradient kernels (-1: not such a layer)
    int64_t enc_wprep[ARVAE_MAX_LAYERS], dec_wprep[ARVAE_MAX_LAYERS];
    // wide (64-channel, row-staged) conv layers: the split weights of both orientations ([0] Conv2d-forward, [1] transposed), made by
    // ONE launch at the start of the forward pass and used by the layer's forward and data-gradient launches (-1: not such a layer)
    int64_t enc_wide[ARVAE_MAX_LAYERS][2], dec_wide[ARVAE_MAX_LAYERS][2];
    // AMAX arrays (conv32_common.h: the partial maxima a tensor carries for the 32-channel kernels that scale it into fp16) of
    // every layer's output, of the gradient w.r.t. it, of the two ping-pong gradient buffers, and one for a tensor that arrives
    // without (the image, or what a kernel outside the conv32 / conv_c1 / latent-block family wrote)
    int64_t enc_amax[ARVAE_MAX_LAYERS], dec_amax[ARVAE_MAX_LAYERS], enc_gamax[ARVAE_MAX_LAYERS], dec_gamax[ARVAE_MAX_LAYERS];
    int64_t ga_amax, gb_amax, tmp_amax, tmp2_amax;
    int64_t log_std, dlogits, dz_reg, d_mu, d_ls, g_a, g_b, slab, link_ws, mid_prep, mid_wide, fin_args, rec_ws, reg_ws, rec_out, kld_out, reg_out, total;
    int64_t slab_floats;
};

static int make_layout(const arvae_image_vae_t *m, int64_t n, int64_t n_cols, Layout &L) {
    ARVAE_REQUIRE(m != nullptr && n > 0, "image_vae: null model or empty batch");
    ARVAE_REQUIRE(m->n_enc >= 1 && m->n_enc <= ARVAE_MAX_LAYERS && m->n_dec >= 1 && m->n_dec <= ARVAE_MAX_LAYERS,
                  "image_vae: layer counts out of range");
    ARVAE_REQUIRE(m->zdim > 0 && m->n_reg >= 0 && m->n_reg <= 16, "image_vae: bad zdim / n_reg");
    int64_t off = 0, gmax = 0, slab = 0, lws = 0;
    auto take = [&](int64_t count) { const int64_t o = off; off += up4(count); return o; };
    auto visit = [&](const arvae_layer_t &l) {
        arvae_link_t lk = l.link;
        lk.n = (int32_t)n;
        const int64_t s = arvae_link_wgrad_ws_floats(&lk);
        if (s > slab) slab = s;
        if (arvae_link_ws_floats(&lk) > lws) lws = arvae_link_ws_floats(&lk);
        if (out_elems(l, n) > gmax) gmax = out_elems(l, n);
        if (in_elems(l, n) > gmax) gmax = in_elems(l, n);
    };
    auto wide = [&](const arvae_layer_t &l, int64_t (&slot)[2]) {
        arvae_link_t lk = l.link;
        lk.n = (int32_t)n;
        const bool other = dense_fits(&lk) || conv32_fits(&lk) || conv_c1_fits(&lk);
        for (int t = 0; t < 2; ++t) slot[t] = (!other && conv64_fits(&lk, t == 1) && conv64s_fits(&lk, t == 1)) ? take(conv64s_ws_floats()) : -1;
    };
    for (int i = 0; i < ARVAE_MAX_LAYERS; ++i) L.enc_wide[i][0] = L.enc_wide[i][1] = L.dec_wide[i][0] = L.dec_wide[i][1] = -1;
    for (int i = 0; i < m->n_enc; ++i) wide(m->enc[i], L.enc_wide[i]);
    for (int i = 0; i < m->n_dec; ++i) wide(m->dec[i], L.dec_wide[i]);
    for (int i = 0; i < m->n_enc; ++i) { L.enc_out[i] = take(out_elems(m->enc[i], n)); visit(m->enc[i]); }
    for (int i = 0; i < m->n_dec; ++i) {
        L.dec_out[i] = (i + 1 < m->n_dec) ? take(out_elems(m->dec[i], n)) : -1;
        visit(m->dec[i]);
    }
    visit(m->head_mu);
    visit(m->head_log_std);
    // (a layer the clustered latent block may compute itself leaves one slab per workgroup of that grid: mid_fold_slab_floats)
    const int64_t fold_slab = mid_fold_slab_floats(m, (int)n);
    auto own_slab = [&](const arvae_layer_t &l) -> int64_t {
        arvae_link_t lk = l.link;
        lk.n = (int32_t)n;
        if (!(conv32_fits(&lk) || conv_c1_fits(&lk))) return -1;
        const int64_t need = arvae_link_wgrad_ws_floats(&lk);
        return take((conv32_fits(&lk) && lk.lh == 4 && fold_slab > need) ? fold_slab : need);
    };
    auto own_bits = [&](const arvae_layer_t &l, int64_t wprep) -> int64_t {
        arvae_link_t lk = l.link;
        lk.n = (int32_t)n;
        const bool fast = (conv32_fits(&lk) && wprep >= 0) || (conv_c1_fits(&lk) && !l.is_up);
        static const bool off = diag_env("ARVAE_NO_RELU_BITS") != nullptr;      // diagnostic: gate with the float activations
        return (fast && !off && l.act == ARVAE_ACT_RELU && !l.dropout) ? take(out_elems(l, n) / 32) : -1;
    };
    int n_prep = 0;
    auto own_prep = [&](const arvae_layer_t &l) -> int64_t {
        arvae_link_t lk = l.link;
        lk.n = (int32_t)n;
        return (conv32_fits(&lk) && n_prep++ < 8) ? take(conv32_prep_floats()) : -1;
    };
    for (int i = 0; i < m->n_enc; ++i) L.enc_wprep[i] = own_prep(m->enc[i]);
    for (int i = 0; i < m->n_dec; ++i) L.dec_wprep[i] = own_prep(m->dec[i]);
    for (int i = 0; i < m->n_enc; ++i) L.enc_bits[i] = own_bits(m->enc[i], L.enc_wprep[i]);
    for (int i = 0; i < m->n_dec; ++i) L.dec_bits[i] = own_bits(m->dec[i], L.dec_wprep[i]);
    for (int i = 0; i < m->n_enc; ++i) L.enc_slab[i] = own_slab(m->enc[i]);
    for (int i = 0; i < m->n_dec; ++i) L.dec_slab[i] = own_slab(m->dec[i]);
    // every layer's output gradient has its own buffer: weight gradients run on a second stream (and the Linear ones
    // at the very end), so a gradient must not be overwritten two layers later
    for (int i = 0; i < m->n_enc; ++i) L.enc_keep[i] = take(out_elems(m->enc[i], n));
    for (int i = 0; i < m->n_dec; ++i) L.dec_keep[i] = take(out_elems(m->dec[i], n));
    for (int i = 0; i < m->n_enc; ++i) { L.enc_amax[i] = take(AMAX_N); L.enc_gamax[i] = take(AMAX_N); }
    for (int i = 0; i < m->n_dec; ++i) { L.dec_amax[i] = take(AMAX_N); L.dec_gamax[i] = take(AMAX_N); }
    L.ga_amax = take(AMAX_N);
    L.gb_amax = take(AMAX_N);
    L.tmp_amax = take(AMAX_N);
    L.tmp2_amax = take(AMAX_N);
    const int64_t bz = n * m->zdim;
    L.log_std = take(bz);
    L.dlogits = take(out_elems(m->dec[m->n_dec - 1], n));
    L.dz_reg = take(bz);
    L.d_mu = take(bz);
    L.d_ls = take(bz);
    L.g_a = take(gmax);
    L.g_b = take(gmax);
    L.slab_floats = slab;
    L.slab = take(slab);
    L.link_ws = take(lws);               // scratch of one arvae_link_down / _up call at a time (stream-ordered)
    L.mid_prep = take(mid_prep_floats(m));   // the latent block's matrices in the layout its kernels stream (midblock.hip)
    {
        const int64_t w = mid_wide_ws_floats(m, (int)n);     // partial products of the wide Linear layers' split reductions (dense.hip)
        L.mid_wide = w > 0 ? take(w) : -1;
    }
    static_assert(sizeof(VaeFinishArgs) <= 64 * sizeof(float), "the parked finishing step's arguments fit their workspace slot");
    L.fin_args = take(64);                // a deferred finishing step's arguments (ARVAE_VAE_DEFER_FINISH, vae_finish.h)
    L.rec_ws = take(arvae_recon_ws_floats(out_elems(m->dec[m->n_dec - 1], n)));
    L.reg_ws = take(arvae_reg_loss_ws_floats(n, m->n_reg > 0 ? m->n_reg : 1));
    L.rec_out = take(4);
    L.kld_out = take(4);
    L.reg_out = take(4);
    L.total = off;
    (void)n_cols;
    return ARVAE_OK;
}

// milestones (include/arvae_hip.h): record the caller's event on the pass's stream
static inline void mark(void *event, hipStream_t st) {
    if (event != nullptr) (void)hipEventRecord(reinterpret_cast<hipEvent_t>(event), st);
}

// the last decoder layer is the single-channel transposed conv whose launch also sums the reconstruction term (conv_c1.hip
// up_c1_kernel<DIST, true>): the models a deferred finishing step exists for
static bool recon_is_fused(const arvae_image_vae_t *m) {
    const arvae_layer_t &last = m->dec[m->n_dec - 1];
    return last.is_up && last.act == ARVAE_ACT_NONE && last.dropout == 0 && conv_c1_fits(&last.link) && arvae_recon_ws_floats(0) >= 2 * 1024;
}
// ARVAE_VAE_DEFER_FINISH honoured: the forward pass parks the finishing step's arguments (its last launch stores them), the
// backward pass's first launch -- or a launch of its own right behind it -- runs it
static bool finish_deferred(const arvae_image_vae_t *m) { return (m->flags & ARVAE_VAE_DEFER_FINISH) != 0 && recon_is_fused(m); }

static arvae_operand_t plain(const float *v) { return arvae_operand_t{v, nullptr, nullptr, ARVAE_ACT_NONE}; }

// in_amax: AMAX array of `in` or null (then one is made in tmp_amax when a 32-channel kernel needs it); out_amax: where the
// maxima of `out` go when the kernel that runs can deliver them (*out_has tells)
static int layer_forward(const arvae_layer_t &l, int32_t n, const float *params, const float *in, const uint8_t *mask,
                         float *out, uint16_t *bits_out, float *link_ws, arvae_stream_t st, const float *wprep,
                         const unsigned *in_amax, unsigned *tmp_amax, unsigned *out_amax, bool *out_has, float *wide_prep = nullptr) {
    arvae_link_t lk = l.link;
    lk.n = n;
    const arvae_operand_t op = plain(in);
    const float *w = params + l.w_off, *b = l.b_off >= 0 ? params + l.b_off : nullptr;
    *out_has = false;
    if (bits_out != nullptr) {                           // make_layout grants bits only to ReLU layers on these kernels
        hipStream_t hs = as_stream(st);
        *out_has = true;
        if (conv32_fits(&lk)) {
            if (in_amax == nullptr) {
                if (int rc = conv32_amax(in, in_elems(l, n), tmp_amax, hs)) return rc;
                in_amax = tmp_amax;
            }
            return l.is_up ? conv32_up(&lk, make_operand(&op), b, 1, nullptr, nullptr, bits_out, out, hs, wprep, in_amax, out_amax)
                           : conv32_down(&lk, make_operand(&op), b, 1, nullptr, nullptr, bits_out, out, hs, wprep, in_amax, out_amax);
        }
        return conv_c1_down(&lk, make_operand(&op), w, b, 1, nullptr, nullptr, bits_out, out, hs, out_amax);
    }
    // the wide stride-1 convolutions (conv64s.hip runs the two-term fp16 arithmetic too): the input's maxima come along or are
    // taken once, into the input's own array (the layer's weight gradient reads them again); the row-staged kernel publishes
    // the output's
    if (conv64_fits(&lk, l.is_up != 0) && tmp_amax != nullptr && !dense_fits(&lk) && !conv32_fits(&lk) && !conv_c1_fits(&lk)) {
        hipStream_t hs = as_stream(st);
        if (in_amax == nullptr && conv64s_fits(&lk, l.is_up != 0)) {
            if (int rc = conv32_amax(in, in_elems(l, n), tmp_amax, hs)) return rc;
            in_amax = tmp_amax;
        }
        *out_has = out_amax != nullptr;                  // (both the row-staged and the gathering kernel publish their output's maxima)
        return l.is_up ? conv64_up(&lk, make_operand(&op), w, b, l.act, mask, out, link_ws, hs, nullptr, in_amax, *out_has ? out_amax : nullptr, wide_prep)
                       : conv64_down(&lk, make_operand(&op), w, b, l.act, mask, out, link_ws, hs, nullptr, in_amax, *out_has ? out_amax : nullptr,
                                     wide_prep);
    }
    // the single-channel first layer of the wide stack (Conv2d(1, 64)): the same kernel arvae_link_down picks, with the maxima
    if (!l.is_up && out_amax != nullptr && !dense_fits(&lk) && !conv32_fits(&lk) && !conv_c1_fits(&lk) && !conv64_fits(&lk, false) &&
        single_channel_down_fits(&lk)) {
        *out_has = true;
        return single_channel_down(&lk, make_operand(&op), w, b, l.act, mask, out, as_stream(st), out_amax);
    }
    return l.is_up ? arvae_link_up(&lk, &op, w, b, l.act, mask, out, link_ws, st)
                   : arvae_link_down(&lk, &op, w, b, l.act, mask, out, link_ws, st);
}

// One layer of the backward pass.
//   g      : gradient arriving at this layer: w.r.t. its pre-activation (g_is_pre) or w.r.t. its output
//   gate   : when non-null, the saved ReLU output of the PRODUCER of `in`; the data gradient is then
//            written as d_in * (gate > 0), i.e. already w.r.t. the producer's pre-activation, so that the
//            producer's dgrad and wgrad read ONE plain tensor instead of re-deriving ReLU' twice.
//   gate_op: the same for any activation / dropout mask of that producer (GateOp, common.h); honoured by the wide stride-1
//            convolution kernels (conv64.hip), which then also spare the next layer its in-place operand pass
//   *gated : set when the gate was applied (a fast kernel with a gated epilogue was available)
static int layer_backward(const arvae_layer_t &l, int32_t n, const float *params, float *grads, const float *in,
                          const float *out, const uint8_t *mask, const float *g, bool g_is_pre, const float *gate,
                          float *d_in, bool *gated, float *slab, float *link_ws, DenseWgradBatch *defer, float *own_slab,
                          SlabReduceBatch *rdefer, arvae_stream_t st, const float *g_scale = nullptr,
                          const uint16_t *gate_bits = nullptr, const float *wprep = nullptr, const GateOp *gate_op = nullptr,
                          const unsigned *g_amax = nullptr, const unsigned *in_amax = nullptr, unsigned *tmp_amax = nullptr,
                          unsigned *tmp2_amax = nullptr, unsigned *din_amax = nullptr, bool *din_has = nullptr, float *wide_prep = nullptr,
                          const float *c1_img = nullptr, float *c1_slab = nullptr, SlabJob *c1_job = nullptr,
                          const VaeFinishArgs *finish = nullptr, bool *finish_taken = nullptr) {
    // finish (device pointer) / finish_taken: the forward pass's deferred finishing step, for the launch that can carry it
    // c1_img / c1_slab / c1_job: this is the layer behind a single-channel first layer whose backward pass wants nothing but its
    // weight gradient: the paired launch of this layer computes that too and writes NO data gradient (conv32.hip, C1Wgrad);
    // c1_job->slab != nullptr afterwards says it did
    // g_amax / in_amax: AMAX arrays (conv32_common.h) of g and of `in`, or null -- a 32-channel kernel that needs one then gets it
    // made in tmp_amax / tmp2_amax; din_amax: where the maxima of d_in go when the kernel that writes it delivers them (*din_has)
    arvae_link_t lk = l.link;
    lk.n = n;
    bool din_has_local = false;
    if (din_has == nullptr) din_has = &din_has_local;
    *din_has = false;
    const bool c32 = conv32_fits(&lk) && wprep != nullptr && tmp_amax != nullptr && tmp2_amax != nullptr;
    auto need_g = [&]() -> int {
        if (g_amax != nullptr) return ARVAE_OK;
        g_amax = tmp_amax;
        return conv32_amax(g, out_elems(l, n), tmp_amax, as_stream(st));
    };
    auto need_in = [&]() -> int {
        if (in_amax != nullptr) return ARVAE_OK;
        in_amax = tmp2_amax;
        return conv32_amax(in, in_elems(l, n), tmp2_amax, as_stream(st));
    };
    arvae_operand_t gop = g_is_pre ? plain(g) : arvae_operand_t{g, out, mask, l.act};
    // The wide stride-1 convolutions gather their operands once per tap: fold the activation derivative / keep-mask into
    // the upstream gradient once, in place (it is this executor's scratch and has no other reader), and hand the data
    // gradient, the weight gradient and the bias sums ONE plain tensor instead of three tensors each (MNIST: 780 -> 520 us
    // per data-gradient launch).
    if (!g_is_pre && (mask != nullptr || l.act != ARVAE_ACT_NONE) && (conv64_fits(&lk, false) || conv64_fits(&lk, true))) {
        const int64_t count = (int64_t)n * (l.is_up ? (int64_t)lk.hh * lk.hw * lk.chi : (int64_t)lk.lh * lk.lw * lk.clo);
        if (int rc = arvae_operand_apply(&gop, count, const_cast<float *>(g), st)) return rc;
        gop = plain(g);
        g_amax = nullptr;                                    // (of what was there before)
    }
    const arvae_operand_t xin = plain(in);
    const float *w = params + l.w_off;
    float *dw = grads + l.w_off, *db = l.b_off >= 0 ? grads + l.b_off : nullptr;
    hipStream_t hs = as_stream(st);
    const arvae_stream_t wst = st;
    hipStream_t whs = hs;                               // (a second stream for the weight gradients measured 1-9 % slower in
                                                        // rounds 1 and 2 -- the big kernels cannot share a CU -- and was removed)
    if (gated != nullptr) *gated = false;
    // 32-channel layers: gated data gradient and weight-gradient partials in one launch (conv32.hip, pair4_* / pair_*_wgrad_kernel)
    if (d_in != nullptr && gated != nullptr && rdefer != nullptr && own_slab != nullptr && gop.y == nullptr && g_scale == nullptr &&
        rdefer->count < SLAB_BATCH_MAX && c32 &&
        conv32_pair_fits(&lk, l.is_up != 0, gate, gate_bits, db ? (l.is_up ? 2 : 1) : 0)) {
        SlabJob job;
        if (int rc = need_g()) return rc;
        if (int rc = need_in()) return rc;
        const bool c1 = c1_img != nullptr && c1_slab != nullptr && c1_job != nullptr && conv32_pair_c1_fits(&lk, l.is_up != 0, gate_bits);
        if (int rc = conv32_pair(&lk, l.is_up != 0, gop.v, in, gate, gate_bits, d_in, wprep, dw, db, own_slab, hs, &job, g_amax, in_amax,
                                  c1 ? nullptr : din_amax, c1 ? c1_img : nullptr, c1 ? c1_slab : nullptr, c1 ? c1_job : nullptr))
            return rc;
        slab_reduce_defer(rdefer, job);
        if (c1) { *gated = true; *din_has = false; return ARVAE_OK; }
        *gated = true;
        *din_has = din_amax != nullptr;
        return ARVAE_OK;
    }
    const bool simple = gop.mask == nullptr && gop.act != ARVAE_ACT_SELU;
    // the single-channel forward-UP link (last decoder layer): gated data gradient and weight-gradient partials in one launch
    if (d_in != nullptr && gated != nullptr && l.is_up && gate != nullptr && simple && gop.y == nullptr && rdefer != nullptr &&
        own_slab != nullptr && rdefer->count < SLAB_BATCH_MAX && conv_c1_pair_fits(&lk)) {
        Operand g_op = make_operand(&gop);
        g_op.scale = g_scale;
        SlabJob job;
        if (int rc = conv_c1_pair(&lk, g_op, w, gate_bits ? nullptr : gate, gate_bits, d_in, make_operand(&xin), dw, db, db ? 2 : 0, own_slab,
                                  hs, &job, din_amax, finish))
            return rc;
        if (finish != nullptr && finish_taken != nullptr) *finish_taken = true;
        slab_reduce_defer(rdefer, job);
        *gated = true;
        *din_has = din_amax != nullptr;
        return ARVAE_OK;
    }
    if (d_in != nullptr) {
        int rc;
        if (l.is_up) {                                   // forward UP  -> data gradient is a DOWN map
            if (gate != nullptr && gop.y == nullptr && c32) {
                if (int rc2 = need_g()) return rc2;
                rc = conv32_down(&lk, make_operand(&gop), nullptr, 0, gate_bits ? nullptr : gate, gate_bits, nullptr, d_in, hs, wprep,
                                 g_amax, din_amax);
                *gated = true;
                *din_has = din_amax != nullptr;
            } else if (gate != nullptr && simple && conv_c1_fits(&lk)) {
                Operand g_op = make_operand(&gop);
                g_op.scale = g_scale;
                rc = conv_c1_down(&lk, g_op, w, nullptr, 0, gate_bits ? nullptr : gate, gate_bits, nullptr, d_in, hs, din_amax);
                *gated = true;
                *din_has = din_amax != nullptr;
            } else if (gate_op != nullptr && !conv64_fits(&lk, false) && !conv_c1_fits(&lk) && single_channel_down_gated_fits(&lk)) {
                rc = single_channel_down_gated(&lk, make_operand(&gop), w, gate_op, d_in, hs, din_amax);
                *gated = true;
                *din_has = din_amax != nullptr && lk.n <= 1024;
            } else if (gate_op != nullptr && conv64s_fits(&lk, false)) {       // (the gathering kernel's scattered epilogue loses more than the operand pass costs)
                if (gop.y == nullptr && tmp_amax != nullptr)
                    if (int rc2 = need_g()) return rc2;
                rc = conv64_down(&lk, make_operand(&gop), w, nullptr, ARVAE_ACT_NONE, nullptr, d_in, link_ws, hs, gate_op,
                                 gop.y == nullptr ? g_amax : nullptr, din_amax, wide_prep);
                *gated = true;
                *din_has = din_amax != nullptr;
            } else {
                rc = arvae_link_down(&lk, &gop, w, nullptr, ARVAE_ACT_NONE, nullptr, d_in, link_ws, st);
            }
        } else {                                         // forward DOWN -> data gradient is an UP map
            if (gate != nullptr && gop.y == nullptr && c32) {
                if (int rc2 = need_g()) return rc2;
                rc = conv32_up(&lk, make_operand(&gop), nullptr, 0, gate_bits ? nullptr : gate, gate_bits, nullptr, d_in, hs, wprep,
                               g_amax, din_amax);
                *gated = true;
                *din_has = din_amax != nullptr;
            } else if (gate != nullptr && dense_fits(&lk)) {
                rc = dense_dgrad(&lk, make_operand(&gop), w, gate, d_in, hs);
                *gated = true;
            } else if (gate_op != nullptr && conv64_fits(&lk, true)) {         // (conv64s.hip or the gathering kernel: both take the gate)
                // (both kernels scale a plain source by its maxima: the row-staged one and, since round 4, the gathering one)
                if (gop.y == nullptr && tmp_amax != nullptr)
                    if (int rc2 = need_g()) return rc2;
                rc = conv64_up(&lk, make_operand(&gop), w, nullptr, ARVAE_ACT_NONE, nullptr, d_in, link_ws, hs, gate_op,
                               gop.y == nullptr ? g_amax : nullptr, din_amax, wide_prep);
                *gated = true;
                *din_has = din_amax != nullptr;
            } else {
                rc = arvae_link_up(&lk, &gop, w, nullptr, ARVAE_ACT_NONE, nullptr, d_in, link_ws, st);
            }
        }
        if (rc) return rc;
    }
    // conv layers with a slab kernel and plain operands: partial sums now, reduction queued for the end of the pass
    if (rdefer != nullptr && own_slab != nullptr && gop.y == nullptr && rdefer->count < SLAB_BATCH_MAX &&
        ((conv32_fits(&lk) && tmp_amax != nullptr && tmp2_amax != nullptr) || conv_c1_fits(&lk))) {
        Operand lo_op = make_operand(l.is_up ? &xin : &gop), hi_op = make_operand(l.is_up ? &gop : &xin);
        (l.is_up ? hi_op : lo_op).scale = g_scale;                // only the conv_c1 kernels honour it (checked by the caller)
        const int bias_mode = db ? (l.is_up ? 2 : 1) : 0;
        SlabJob job;
        if (conv32_fits(&lk)) {
            if (int rc = need_g()) return rc;
            if (int rc = need_in()) return rc;
        }
        // (the single-channel FIRST layer closes the pass: the Linear weight gradients queued so far ride in its launch, dense.hip)
        const int rc = conv32_fits(&lk) ? conv32_wgrad_partial(&lk, lo_op, hi_op, dw, db, bias_mode, own_slab, whs, &job,
                                                               l.is_up ? in_amax : g_amax, l.is_up ? g_amax : in_amax)
                       : (d_in == nullptr && !l.is_up && dense_wgrad_c1_fits(defer))
                           ? dense_wgrad_flush_with_c1(defer, &lk, lo_op, hi_op, dw, db, bias_mode, own_slab, whs, &job)
                           : conv_c1_wgrad_partial(&lk, lo_op, hi_op, dw, db, bias_mode, own_slab, whs, &job);
        if (rc) return rc;
        slab_reduce_defer(rdefer, job);
        return ARVAE_OK;
    }
    // the wide stride-1 layers' weight gradient (conv64.hip) takes the operands' maxima when they are plain tensors
    if (conv64_wgrad_fits(&lk) && !dense_fits(&lk) && !conv_c1_fits(&lk) && !conv32_fits(&lk) && tmp_amax != nullptr && tmp2_amax != nullptr) {
        const bool g_plain = gop.y == nullptr;
        if (g_plain)
            if (int rc = need_g()) return rc;
        if (int rc = need_in()) return rc;
        const unsigned *ga = g_plain ? g_amax : nullptr;
        return l.is_up ? link_wgrad_conv64(&lk, make_operand(&xin), make_operand(&gop), dw, db, db ? 2 : 0, slab, whs, in_amax, ga)
                       : link_wgrad_conv64(&lk, make_operand(&gop), make_operand(&xin), dw, db, db ? 1 : 0, slab, whs, ga, in_amax);
    }
    if (l.is_up) return arvae_link_wgrad(&lk, &xin, &gop, dw, db, db ? 2 : 0, slab, wst);
    if (defer != nullptr && dense_fits(&lk) && dense_wgrad_defer(defer, &lk, make_operand(&gop), in, dw, db)) return ARVAE_OK;
    return arvae_link_wgrad(&lk, &gop, &xin, dw, db, db ? 1 : 0, slab, wst);
}

// Do the latent block's clustered kernels also compute the conv layers on either side of it in this pass (midblock.hip
// mid_fold_fits)?  One answer for the forward and the backward pass of a step: it depends on the model, the batch, the caller's
// flags and the workspace layout only.
static bool fold_conv_layers(const arvae_image_vae_t *m, const Layout &L, int batch, const uint8_t *const *masks, int mid_ne, int mid_nd) {
    const int e = m->n_enc - mid_ne - 1;
    return masks == nullptr && e >= 1 && mid_nd + 1 < m->n_dec && mid_fold_fits(m, batch) && L.dec_bits[mid_nd] >= 0 &&
           L.enc_slab[e] >= 0 && L.dec_slab[mid_nd] >= 0 && L.dec_wprep[mid_nd + 1] >= 0;
}

static int count_masks(const arvae_image_vae_t *m) {
    int c = 0;
    for (int i = 0; i < m->n_enc; ++i) c += m->enc[i].dropout != 0;
    for (int i = 0; i < m->n_dec; ++i) c += m->dec[i].dropout != 0;
    return c;
}

}  // namespace arvae

using namespace arvae;

extern "C" int64_t arvae_image_vae_ws_floats(const arvae_image_vae_t *model, int32_t batch, int64_t n_cols) {
    Layout L;
    if (make_layout(model, batch, n_cols, L)) return -1;
    return L.total;
}

extern "C" int arvae_image_vae_forward(const arvae_image_vae_t *m, int32_t batch, const float *params, const float *x,
                                       const float *labels, int64_t ld_labels, const float *eps,
                                       const uint8_t *const *masks, const float *capacity, const float *z_cols,
                                       const float *lab_cols, int64_t n_cols, float reg_scale, float *ws,
                                       float *scalars, float *mu, float *sigma, float *z, float *logits,
                                       arvae_stream_t stream) {
    Layout L;
    if (int rc = make_layout(m, batch, n_cols, L)) return rc;
    ARVAE_REQUIRE(params && x && eps && ws && scalars && mu && sigma && z && logits, "image_vae_forward: null pointer");
    ARVAE_REQUIRE(m->n_reg == 0 || n_cols < 0 || labels != nullptr, "image_vae_forward: labels needed for the reg loss");
    hipStream_t st = as_stream(stream);
    auto U = [&](int64_t off) { return reinterpret_cast<unsigned *>(ws + off); };
    int mi = 0;
    bool mid_prepped = false;
    int mid_ne = 0, mid_nd = 0;
    const bool mid = mid_fusable(m, &mid_ne, &mid_nd);
    int first = 0;                                           // first encoder layer the loop below still has to run
    {   // split the 32-channel conv weights once for this step's forward and backward kernels
        const float *wts[8];
        float *preps[8];
        int np = 0;
        for (int i = 0; i < m->n_enc; ++i)
            if (L.enc_wprep[i] >= 0) { wts[np] = params + m->enc[i].w_off; preps[np++] = ws + L.enc_wprep[i]; }
        for (int i = 0; i < m->n_dec; ++i)
            if (L.dec_wprep[i] >= 0) { wts[np] = params + m->dec[i].w_off; preps[np++] = ws + L.dec_wprep[i]; }
        // (together with the latent block's matrix layouts when that block runs: one prep launch per step -- or none: when
        // the first encoder layer is the single-channel convolution, the prep rides in ITS grid, conv_c1.hip)
        if (np > 0 && mid && diag_env("ARVAE_SPLIT_PREP") == nullptr) {
            MidPrepArgs margs;
            mid_prep_args(m, params, ws + L.mid_prep, &margs, batch);
            const arvae_layer_t &l0 = m->enc[0];
            arvae_link_t lk0 = l0.link;
            lk0.n = batch;
            static const bool no_pair = diag_env("ARVAE_NO_PAIR_PREP") != nullptr;     // diagnostic: the prep as its own launch
            if (!no_pair && m->n_enc - mid_ne > 1 && L.enc_bits[0] >= 0 && L.enc_wprep[0] < 0 && !l0.is_up && conv_c1_fits(&lk0) &&
                !(masks != nullptr && l0.dropout)) {
                const arvae_operand_t op = plain(x);
                if (int rc = conv_c1_down_with_prep(&lk0, make_operand(&op), params + l0.w_off, l0.b_off >= 0 ? params + l0.b_off : nullptr,
                                                    1, reinterpret_cast<uint16_t *>(ws + L.enc_bits[0]), ws + L.enc_out[0], wts, preps, np,
                                                    margs, st, U(L.enc_amax[0])))
                    return rc;
                first = 1;
            } else if (int rc = conv32_weight_prep_with_mid(wts, preps, np, margs, st)) return rc;
            mid_prepped = true;
        } else if (int rc = conv32_weight_prep(wts, preps, np, st)) return rc;
    }
    {   // split the wide conv layers' weights, both orientations, once for this step's forward and data-gradient launches
        const float *wts[4 * ARVAE_MAX_LAYERS];
        float *outs[4 * ARVAE_MAX_LAYERS];
        int tr[4 * ARVAE_MAX_LAYERS], qs[4 * ARVAE_MAX_LAYERS], nj = 0;
        auto add = [&](const arvae_layer_t &l, const int64_t (&slot)[2]) {
            for (int t = 0; t < 2; ++t)
                if (slot[t] >= 0) { wts[nj] = params + l.w_off; outs[nj] = ws + slot[t]; tr[nj] = t; qs[nj] = t ? l.link.chi : l.link.clo; ++nj; }
        };
        for (int i = 0; i < m->n_enc; ++i) add(m->enc[i], L.enc_wide[i]);
        for (int i = 0; i < m->n_dec; ++i) add(m->dec[i], L.dec_wide[i]);
        if (int rc = conv64s_prep_batch(wts, outs, tr, qs, nj, st)) return rc;
    }
    // the conv layers on either side of the latent block inside its launches (midcluster.hip): the encoder loop stops one layer
    // earlier, the decoder loop starts one layer later
    const bool fold = mid && fold_conv_layers(m, L, batch, masks, mid_ne, mid_nd);
    // encoder
    const float *h = first ? ws + L.enc_out[0] : x;
    const unsigned *h_amax = first ? U(L.enc_amax[0]) : nullptr;   // AMAX array of h, when the kernel that wrote h delivered one
    mi += first ? (m->enc[0].dropout != 0) : 0;
    for (int i = first; i < m->n_enc - mid_ne - (fold ? 1 : 0); ++i) {
        const uint8_t *mask = (masks != nullptr && m->enc[i].dropout) ? masks[mi] : nullptr;
        mi += m->enc[i].dropout != 0;
        uint16_t *bits = L.enc_bits[i] >= 0 ? reinterpret_cast<uint16_t *>(ws + L.enc_bits[i]) : nullptr;
        bool has = false;
        // two stacked 32-channel ReLU layers (16x16 then 8x8 output) as ONE launch (conv32.hip chain_down_kernel)
        if (i + 1 < m->n_enc - mid_ne - (fold ? 1 : 0) && mask == nullptr && !(masks != nullptr && m->enc[i + 1].dropout) && bits != nullptr &&
            L.enc_bits[i + 1] >= 0 && L.enc_wprep[i] >= 0 && L.enc_wprep[i + 1] >= 0 && h_amax != nullptr && !m->enc[i].is_up && !m->enc[i + 1].is_up) {
            arvae_link_t la = m->enc[i].link, lb = m->enc[i + 1].link;
            la.n = lb.n = batch;
            if (conv32_down_chain_fits(&la, &lb)) {
                const arvae_layer_t &a = m->enc[i], &b = m->enc[i + 1];
                if (int rc = conv32_down_chain(&la, &lb, h, h_amax, a.b_off >= 0 ? params + a.b_off : nullptr, bits, ws + L.enc_out[i],
                                               ws + L.enc_wprep[i], U(L.enc_amax[i]), b.b_off >= 0 ? params + b.b_off : nullptr,
                                               reinterpret_cast<uint16_t *>(ws + L.enc_bits[i + 1]), ws + L.enc_out[i + 1],
                                               ws + L.enc_wprep[i + 1], U(L.enc_amax[i + 1]), st))
                    return rc;
                ++i;
                h = ws + L.enc_out[i];
                h_amax = U(L.enc_amax[i]);
                continue;
            }
        }
        // (a 32-channel layer's input keeps its AMAX array for the weight gradient: a missing one is made in the input's own slot)
        if (int rc = layer_forward(m->enc[i], batch, params, h, mask, ws + L.enc_out[i], bits, ws + L.link_ws, stream,
                                   L.enc_wprep[i] >= 0 ? ws + L.enc_wprep[i] : nullptr, h_amax, i > 0 ? U(L.enc_amax[i - 1]) : U(L.tmp_amax),
                                   U(L.enc_amax[i]), &has, L.enc_wide[i][m->enc[i].is_up ? 1 : 0] >= 0 ? ws + L.enc_wide[i][m->enc[i].is_up ? 1 : 0] : nullptr))
            return rc;
        h = ws + L.enc_out[i];
        h_amax = has ? U(L.enc_amax[i]) : nullptr;
    }
    const int64_t bz = (int64_t)batch * m->zdim;
    bool heads_next = false;
    if (mid) {                                               // Linear stack + heads + reparameterisation + Linear stack: one launch
        float *enc_y[ARVAE_MAX_LAYERS], *dec_y[ARVAE_MAX_LAYERS];
        for (int i = 0; i < mid_ne; ++i) enc_y[i] = ws + L.enc_out[m->n_enc - mid_ne + i];
        for (int i = 0; i < mid_nd; ++i) dec_y[i] = ws + L.dec_out[i];
        MidFold mf{};
        if (fold) {
            mf.hi_e = h;                                         // the conv layer's input; its output is the block's x0
            mf.hi_d = ws + L.dec_out[mid_nd];
            mf.hi_d_bits = reinterpret_cast<unsigned char *>(ws + L.dec_bits[mid_nd]);
            mf.hi_d_amax = U(L.dec_amax[mid_nd]);
        }
        if (int rc = mid_forward(m, batch, params, ws + L.mid_prep, fold ? ws + L.enc_out[m->n_enc - mid_ne - 1] : h, enc_y, dec_y, eps, mu,
                                 ws + L.log_std, sigma, z, st, mid_prepped, U(L.dec_amax[mid_nd - 1]), fold ? &mf : nullptr,
                                 L.mid_wide >= 0 ? ws + L.mid_wide : nullptr))
            return rc;
        h = fold ? ws + L.dec_out[mid_nd] : dec_y[mid_nd - 1];
        h_amax = fold ? U(L.dec_amax[mid_nd]) : U(L.dec_amax[mid_nd - 1]);
    } else if (heads_fusable(&m->head_mu, &m->head_log_std, m->zdim)) {
        // the decoder's first Linear layer rides in the heads kernel when it can (heads.hip)
        heads_next = m->n_dec > 1 && heads_next_fusable(&m->dec[0], m->zdim) && !(masks != nullptr && m->dec[0].dropout);
        if (int rc = heads_latent_fwd(&m->head_mu, &m->head_log_std, batch, m->zdim, params, h, eps, mu, ws + L.log_std,
                                      sigma, z, st, m, heads_next ? &m->dec[0] : nullptr, heads_next ? ws + L.dec_out[0] : nullptr))
            return rc;
    } else {
        if (m->rng_eps)                                      // no fused heads kernel for this model: draw eps first
            if (int rc = arvae_philox_normal(const_cast<float *>(eps), bz, m->rng_seed, m->rng_offset, m->rng_step,
                                             m->rng_dev_step, stream))
                return rc;
        bool has = false;
        if (int rc = layer_forward(m->head_mu, batch, params, h, nullptr, mu, nullptr, ws + L.link_ws, stream, nullptr, nullptr, nullptr, nullptr, &has)) return rc;
        if (int rc = layer_forward(m->head_log_std, batch, params, h, nullptr, ws + L.log_std, nullptr, ws + L.link_ws, stream, nullptr, nullptr, nullptr, nullptr, &has)) return rc;
        if (int rc = arvae_latent_fwd(mu, ws + L.log_std, eps, bz, sigma, z, stream)) return rc;
    }
    if (m->milestones != nullptr) mark(m->milestones->z_ready, st);   // mu / sigma / z are final: the caller's all-gather may start
    // decoder
    if (!mid) { h = heads_next ? ws + L.dec_out[0] : z; h_amax = nullptr; }
    int nb = 0;
    const bool recon_fused = recon_is_fused(m);
    // the regulariser needs z and the labels only: when this rank's batch is the whole batch its workgroups ride in the grid
    // of the first decoder convolution (conv32.hip, up32x_reg_kernel) instead of a launch of their own after the decoder
    const bool reg_here = m->n_reg > 0 && n_cols >= 0;
    const bool defer_finish = finish_deferred(m) && n_cols != -2;
    bool reg_done = false;
    RegArgs reg_args{};
    if (reg_here) {
        for (int i = 0; i < m->n_reg; ++i)
            ARVAE_REQUIRE(m->reg_dims[i] >= 0 && m->reg_dims[i] < m->zdim && m->reg_dims[i] < ld_labels,
                          "image_vae_forward: reg dim %d outside z/labels", m->reg_dims[i]);
        const float *zc = z_cols != nullptr ? z_cols : z;
        const float *lc = lab_cols != nullptr ? lab_cols : labels;
        reg_args = RegArgs{z, labels, batch, zc, lc, z_cols != nullptr ? n_cols : (int64_t)batch, m->zdim, ld_labels, RegDims{},
                           m->delta, ws + L.reg_ws, ws + L.reg_ws + (int64_t)batch * m->n_reg};
        for (int i = 0; i < 16; ++i) reg_args.dims.d[i] = i < m->n_reg ? m->reg_dims[i] : 0;
    }
    for (int i = mid ? mid_nd + (fold ? 1 : 0) : (heads_next ? 1 : 0); i < m->n_dec; ++i) {
        const uint8_t *mask = (masks != nullptr && m->dec[i].dropout) ? masks[mi] : nullptr;
        mi += m->dec[i].dropout != 0;
        float *out = (i + 1 < m->n_dec) ? ws + L.dec_out[i] : logits;
        arvae_link_t lki = m->dec[i].link;
        lki.n = batch;
        if (reg_here && !reg_done && z_cols == nullptr && i + 1 < m->n_dec && m->dec[i].is_up && m->dec[i].act == ARVAE_ACT_RELU &&
            mask == nullptr && L.dec_bits[i] >= 0 && L.dec_wprep[i] >= 0 && h_amax != nullptr && conv32_up_reg_fits(&lki)) {
            const arvae_layer_t &l = m->dec[i];
            const arvae_operand_t op = plain(h);
            if (int rc = conv32_up_reg(&lki, make_operand(&op), l.b_off >= 0 ? params + l.b_off : nullptr,
                                       reinterpret_cast<uint16_t *>(ws + L.dec_bits[i]), out, ws + L.dec_wprep[i], h_amax, U(L.dec_amax[i]),
                                       reg_args, m->n_reg, st))
                return rc;
            reg_done = true;
            h_amax = U(L.dec_amax[i]);
        } else if (i + 1 == m->n_dec && recon_fused) {             // last layer: logits + reconstruction partials in one kernel
            arvae_link_t lk = m->dec[i].link;
            lk.n = batch;
            const arvae_layer_t &l = m->dec[i];
            // a training step may leave the finishing step to its backward pass (ARVAE_VAE_DEFER_FINISH): this launch then parks
            // that step's arguments in the workspace and poisons the scalars (vae_finish.h)
            VaeFinishArgs fa{};
            if (defer_finish) {
                const int64_t nc_d = reg_here ? reg_args.n_cols : (int64_t)batch;
                fa = vae_finish_args(ws + L.rec_ws, conv_c1_up_recon_blocks(&lk), batch, out_elems(m->dec[m->n_dec - 1], batch), mu, sigma,
                                     m->zdim, m->beta, capacity, reg_here ? ws + L.reg_ws : nullptr, nc_d, m->zdim, m->reg_dims, m->n_reg,
                                     m->gamma, m->delta, reg_scale, ws + L.dz_reg, ws + L.rec_out, ws + L.kld_out, ws + L.reg_out, scalars, 0);
            }
            if (int rc = conv_c1_up_recon(&lk, h, params + l.w_off, l.b_off >= 0 ? params + l.b_off : nullptr, logits, x,
                                          m->recon_dist, ws + L.rec_ws, ws + L.dlogits, st, &nb, defer_finish ? &fa : nullptr,
                                          reinterpret_cast<VaeFinishArgs *>(ws + L.fin_args)))
                return rc;
            h_amax = nullptr;
        } else {
            uint16_t *bits = L.dec_bits[i] >= 0 ? reinterpret_cast<uint16_t *>(ws + L.dec_bits[i]) : nullptr;
            bool has = false;
            if (int rc = layer_forward(m->dec[i], batch, params, h, mask, out, bits, ws + L.link_ws, stream,
                                       L.dec_wprep[i] >= 0 ? ws + L.dec_wprep[i] : nullptr, h_amax, i > 0 ? U(L.dec_amax[i - 1]) : U(L.tmp_amax),
                                       U(L.dec_amax[i]), &has, L.dec_wide[i][m->dec[i].is_up ? 1 : 0] >= 0 ? ws + L.dec_wide[i][m->dec[i].is_up ? 1 : 0] : nullptr))
                return rc;
            h_amax = has ? U(L.dec_amax[i]) : nullptr;
        }
        h = out;
    }
    // loss terms: per-block partials of the reconstruction term and the regulariser, then one finishing workgroup
    const int64_t pix = out_elems(m->dec[m->n_dec - 1], batch);
    if (!recon_fused)
        if (int rc = recon_partials(logits, x, pix, batch, m->recon_dist, ws + L.rec_ws, ws + L.dlogits, st, &nb)) return rc;
    if (n_cols == -2) return ARVAE_OK;                       // the caller finishes the pass itself: arvae_image_vae_finish
    const int64_t nc = reg_here ? reg_args.n_cols : (int64_t)batch;
    if (reg_here && !reg_done)
        if (int rc = reg_partials(z, labels, batch, reg_args.zc, reg_args.lc, nc, m->zdim, ld_labels, reg_args.dims, m->n_reg, m->delta,
                                  ws + L.reg_ws, st))
            return rc;
    if (defer_finish) return ARVAE_OK;                       // (arvae_image_vae_backward runs it: its arguments are parked in the workspace)
    return vae_finish(ws + L.rec_ws, nb, batch, pix, mu, sigma, m->zdim, m->beta, capacity,
                      reg_here ? ws + L.reg_ws : nullptr, nc, m->zdim, m->reg_dims, m->n_reg, m->gamma, m->delta, reg_scale,
                      ws + L.dz_reg, ws + L.rec_out, ws + L.kld_out, ws + L.reg_out, scalars, st);
}

extern "C" int arvae_image_vae_finish(const arvae_image_vae_t *m, int32_t batch, const float *labels, int64_t ld_labels,
                                      const float *capacity, const float *z_cols, const float *lab_cols, int64_t n_cols,
                                      float reg_scale, float *ws, float *scalars, const float *mu, const float *sigma,
                                      const float *z, arvae_stream_t stream) {
    Layout L;
    if (int rc = make_layout(m, batch, n_cols, L)) return rc;
    ARVAE_REQUIRE(ws && scalars && mu && sigma && z, "image_vae_finish: null pointer");
    ARVAE_REQUIRE(m->n_reg == 0 || (labels && z_cols && lab_cols && n_cols >= batch), "image_vae_finish: gathered columns needed");
    hipStream_t st = as_stream(stream);
    const arvae_layer_t &last = m->dec[m->n_dec - 1];
    arvae_link_t lk = last.link;
    lk.n = batch;
    const int64_t pix = out_elems(last, batch);
    // as arvae_image_vae_forward decides: the last layer's own reconstruction partials, or the stand-alone kernel's
    const bool recon_fused = last.is_up && last.act == ARVAE_ACT_NONE && last.dropout == 0 && conv_c1_fits(&lk) &&
                             arvae_recon_ws_floats(0) >= 2 * 1024;
    const int nb = recon_fused ? conv_c1_up_recon_blocks(&lk) : recon_partial_blocks(pix);
    const bool reg = m->n_reg > 0;
    if (reg) {
        RegDims rd;
        for (int i = 0; i < 16; ++i) rd.d[i] = i < m->n_reg ? m->reg_dims[i] : 0;
        for (int i = 0; i < m->n_reg; ++i)
            ARVAE_REQUIRE(m->reg_dims[i] >= 0 && m->reg_dims[i] < m->zdim && m->reg_dims[i] < ld_labels,
                          "image_vae_finish: reg dim %d outside z/labels", m->reg_dims[i]);
        // the gathered columns index dims 0 .. zdim-1 / 0 .. ld_labels-1 like the local arrays (whole rows are gathered)
        // (a training step may leave the finishing step to its backward pass -- ARVAE_VAE_DEFER_FINISH, as in the single-rank forward
        // pass: the regulariser's launch, the last of this call, then parks that step's arguments and poisons the scalars)
        if (finish_deferred(m)) {
            const VaeFinishArgs fa = vae_finish_args(ws + L.rec_ws, nb, batch, pix, mu, sigma, m->zdim, m->beta, capacity, ws + L.reg_ws, n_cols, m->zdim,
                                                     m->reg_dims, m->n_reg, m->gamma, m->delta, reg_scale, ws + L.dz_reg, ws + L.rec_out,
                                                     ws + L.kld_out, ws + L.reg_out, scalars, 0);
            return reg_partials(z, labels, batch, z_cols, lab_cols, n_cols, m->zdim, ld_labels, rd, m->n_reg, m->delta, ws + L.reg_ws, st, &fa,
                                reinterpret_cast<VaeFinishArgs *>(ws + L.fin_args));
        }
        if (int rc = reg_partials(z, labels, batch, z_cols, lab_cols, n_cols, m->zdim, ld_labels, rd, m->n_reg, m->delta, ws + L.reg_ws, st))
            return rc;
    }
    return vae_finish(ws + L.rec_ws, nb, batch, pix, mu, sigma, m->zdim, m->beta, capacity, reg ? ws + L.reg_ws : nullptr,
                      reg ? n_cols : (int64_t)batch, m->zdim, m->reg_dims, m->n_reg, m->gamma, m->delta, reg_scale, ws + L.dz_reg,
                      ws + L.rec_out, ws + L.kld_out, ws + L.reg_out, scalars, st);
}

extern "C" int arvae_image_vae_backward(const arvae_image_vae_t *m, int32_t batch, const float *params, float *grads,
                                        const float *x, const float *eps, const uint8_t *const *masks,
                                        const float *capacity, const float *mu, const float *sigma, const float *z,
                                        const float *logits, const float *g_loss, const float *dz_extra,
                                        int32_t reg_fused, float reg_scale, float *ws, arvae_stream_t stream) {
    Layout L;
    if (int rc = make_layout(m, batch, 0, L)) return rc;
    ARVAE_REQUIRE(params && grads && x && eps && mu && sigma && z && logits && g_loss && ws,
                  "image_vae_backward: null pointer");
    hipStream_t st = as_stream(stream);
    auto U = [&](int64_t off) { return reinterpret_cast<unsigned *>(ws + off); };
    // AMAX array (conv32_common.h) that belongs to a gradient buffer of this pass
    auto grad_amax = [&](const float *p) -> unsigned * {
        if (p == nullptr) return nullptr;
        for (int i = 0; i < m->n_enc; ++i) if (p == ws + L.enc_keep[i]) return U(L.enc_gamax[i]);
        for (int i = 0; i < m->n_dec; ++i) if (p == ws + L.dec_keep[i]) return U(L.dec_gamax[i]);
        return p == ws + L.g_a ? U(L.ga_amax) : p == ws + L.g_b ? U(L.gb_amax) : nullptr;
    };
    // the forward pass left a valid AMAX array with the input of every 32-channel layer that ran on the fast kernels
    auto in_amax_of = [&](bool dec, int i) -> const unsigned * {
        if (i == 0) return nullptr;
        const bool fast = (dec ? L.dec_bits[i] : L.enc_bits[i]) >= 0 && (dec ? L.dec_wprep[i] : L.enc_wprep[i]) >= 0;
        const arvae_layer_t &l = dec ? m->dec[i] : m->enc[i];
        arvae_link_t lk = l.link;
        lk.n = batch;
        const bool wide = conv64s_fits(&lk, l.is_up != 0) && !dense_fits(&lk) && !conv32_fits(&lk) && !conv_c1_fits(&lk);
        return (fast || wide) ? U(dec ? L.dec_amax[i - 1] : L.enc_amax[i - 1]) : nullptr;
    };
    const unsigned *cur_amax = nullptr;                   // AMAX array of `cur`, when the kernel that wrote it delivered one
    // keep-mask index of every dropout layer, in forward order
    int enc_mask[ARVAE_MAX_LAYERS], dec_mask[ARVAE_MAX_LAYERS], mi = 0;
    for (int i = 0; i < m->n_enc; ++i) enc_mask[i] = m->enc[i].dropout ? mi++ : -1;
    for (int i = 0; i < m->n_dec; ++i) dec_mask[i] = m->dec[i].dropout ? mi++ : -1;
    auto mask_of = [&](int idx) -> const uint8_t * { return (masks != nullptr && idx >= 0) ? masks[idx] : nullptr; };

    float *const pp_a = ws + L.g_a, *const pp_b = ws + L.g_b, *slab = L.slab_floats ? ws + L.slab : nullptr;
    DenseWgradBatch defer;
    defer.count = 0;
    SlabReduceBatch rdefer;
    rdefer.count = 0;
    hipStream_t flush_stream = st;
    // where the gradient for `keep` (a Linear layer's output, or -1) is written: its own buffer, or the ping-pong
    // buffer that does not hold the gradient being consumed
    auto grad_dst = [&](int64_t keep, const float *busy) -> float * {
        if (keep >= 0) return ws + keep;
        return busy == pp_a ? pp_b : pp_a;
    };
    const int64_t pix = out_elems(m->dec[m->n_dec - 1], batch);
    const int64_t bz = (int64_t)batch * m->zdim;
    // d loss / d logits was left unscaled by the forward pass.  When the last layer runs on the conv_c1 kernels they
    // multiply by the upstream gradient while loading it; otherwise one elementwise pass makes the scaled copy.
    float *cur;
    const float *first_scale = nullptr;
    // A layer's ReLU can be folded into the data-gradient epilogue of its consumer when no dropout mask sits
    // between them; the gradient handed down is then w.r.t. the pre-activation.
    auto relu_gate = [&](const arvae_layer_t &producer, int mask_idx, const float *saved) -> const float * {
        return (producer.act == ARVAE_ACT_RELU && mask_of(mask_idx) == nullptr) ? saved : nullptr;
    };
    // the general form (any activation, dropout): used where relu_gate() has nothing to offer
    auto general_gate = [&](const arvae_layer_t &producer, int mask_idx, const float *saved, GateOp &go) -> const GateOp * {
        if (producer.act == ARVAE_ACT_NONE && mask_of(mask_idx) == nullptr) return nullptr;
        if (producer.act == ARVAE_ACT_RELU && mask_of(mask_idx) == nullptr) return nullptr;      // relu_gate covers it
        go.y = saved; go.mask = mask_of(mask_idx); go.act = producer.act;
        return &go;
    };
    bool pre = true;                                     // the last decoder layer has no activation
    {
        const int li = m->n_dec - 1;
        const arvae_layer_t &last = m->dec[li];
        arvae_link_t lk = last.link;
        lk.n = batch;
        const bool fold = last.is_up && conv_c1_fits(&lk) && mask_of(dec_mask[li]) == nullptr && li > 0 &&
                          relu_gate(m->dec[li - 1], dec_mask[li - 1], ws + L.dec_out[li - 1]) != nullptr && L.dec_slab[li] >= 0;
        if (fold) {
            cur = ws + L.dlogits;
            first_scale = g_loss;
        } else {
            cur = grad_dst(L.dec_keep[li], nullptr);
            if (int rc = arvae_scale_by_scalar(g_loss, ws + L.dlogits, pix, cur, stream)) return rc;
        }
    }
    int mid_ne = 0, mid_nd = 0;
    const bool mid = mid_fusable(m, &mid_ne, &mid_nd);
    // (as the forward pass decided: the conv layers on either side of the latent block inside its launches)
    const bool fold = mid && fold_conv_layers(m, L, batch, masks, mid_ne, mid_nd);
    // the forward pass left its finishing step (loss scalars, KL mean, the regulariser's z-gradient: the latent block below reads
    // the last two) to this call (ARVAE_VAE_DEFER_FINISH)
    bool finish_pending = finish_deferred(m);
    const VaeFinishArgs *fin_dev = reinterpret_cast<const VaeFinishArgs *>(ws + L.fin_args);
    // decoder, last layer first (down to the latent block when that runs as one launch)
    const float *heads_next_g = nullptr;
    for (int i = m->n_dec - 1; i >= (mid ? mid_nd + (fold ? 1 : 0) : 0); --i) {
        const float *in = i > 0 ? ws + L.dec_out[i - 1] : z;
        const float *out = (i + 1 < m->n_dec) ? ws + L.dec_out[i] : logits;
        const float *gate = i > 0 ? relu_gate(m->dec[i - 1], dec_mask[i - 1], in) : nullptr;
        GateOp go;
        const GateOp *gate_op = i > 0 ? general_gate(m->dec[i - 1], dec_mask[i - 1], in, go) : nullptr;
        float *dst = grad_dst(i > 0 ? L.dec_keep[i - 1] : -1, cur);
        bool gated = false;
        // the first decoder layer's data gradient (d z) is computed inside the heads kernel (heads.hip) when the gradient
        // that arrives here is already w.r.t. the layer's pre-activation: only its weight gradient is queued
        if (i == 0 && !mid && pre && m->n_dec > 1 && heads_fusable(&m->head_mu, &m->head_log_std, m->zdim) &&
            heads_next_fusable(&m->dec[0], m->zdim) && mask_of(dec_mask[0]) == nullptr) {
            heads_next_g = cur;
            dst = nullptr;
        }
        bool din_has = false;
        // (the last decoder layer's launch is the pass's first: it carries a deferred finishing step when it is the paired one)
        const bool carry = finish_pending && i == m->n_dec - 1;
        bool taken = false;
        if (int rc = layer_backward(m->dec[i], batch, params, grads, in, out, mask_of(dec_mask[i]), cur, pre, gate, dst,
                                    &gated, slab, ws + L.link_ws, &defer, L.dec_slab[i] >= 0 ? ws + L.dec_slab[i] : nullptr, &rdefer, stream,
                                    i == m->n_dec - 1 ? first_scale : nullptr,
                                    (gate != nullptr && i > 0 && L.dec_bits[i - 1] >= 0)
                                        ? reinterpret_cast<const uint16_t *>(ws + L.dec_bits[i - 1]) : nullptr,
                                    L.dec_wprep[i] >= 0 ? ws + L.dec_wprep[i] : nullptr, gate_op, cur_amax, in_amax_of(true, i),
                                    U(L.tmp_amax), U(L.tmp2_amax), grad_amax(dst), &din_has,
                                    L.dec_wide[i][m->dec[i].is_up ? 0 : 1] >= 0 ? ws + L.dec_wide[i][m->dec[i].is_up ? 0 : 1] : nullptr,
                                    nullptr, nullptr, nullptr, carry ? fin_dev : nullptr, carry ? &taken : nullptr))
            return rc;
        if (carry) {
            if (!taken)
                if (int rc = vae_finish_deferred(fin_dev, st)) return rc;
            finish_pending = false;
        }
        pre = gated;
        if (heads_next_g == nullptr) { cur = dst; cur_amax = din_has ? grad_amax(dst) : nullptr; }
    }
    // data-parallel caller: finish the decoder's conv gradients now (their all-reduce then runs under the rest of the pass)
    const arvae_milestones *ms = m->milestones;
    bool dec_marked = false, lin_marked = false;
    if (ms != nullptr && ms->dec_grads != nullptr && defer.count == 0) {
        // (defer.count == 0: no Linear layer of the decoder went the per-layer way, so "decoder conv layers" is what is queued)
        if (int rc = slab_reduce_flush(&rdefer, st)) return rc;
        mark(ms->dec_grads, st);
        dec_marked = true;
    }
    // latent head (cur = gradient w.r.t. z from the decoder) and the two encoder heads:
    // d_hidden = W_mu^T d_mu + W_ls^T d_ls   (gated by the last encoder layer's ReLU when possible)
    const float *hidden = ws + L.enc_out[m->n_enc - 1];
    // regulariser gradient w.r.t. z, unit upstream (scaled by g * reg_scale in the latent kernel): the forward's own
    // (reg_fused 1), or the caller's row-block evaluation against gathered columns (reg_fused 2, in dz_extra)
    const float *dz_reg = (reg_fused == 1 && m->n_reg > 0) ? ws + L.dz_reg : reg_fused == 2 ? dz_extra : nullptr;
    if (reg_fused == 2) {
        ARVAE_REQUIRE(dz_extra != nullptr, "image_vae_backward: reg_fused 2 needs the unit regulariser gradient in dz_extra");
        dz_extra = nullptr;
    }
    const float *head_gate = relu_gate(m->enc[m->n_enc - 1], enc_mask[m->n_enc - 1], hidden);
    float *d_hidden = grad_dst(L.enc_keep[m->n_enc - 1], nullptr);
    int enc_from = m->n_enc - 1;                             // first encoder layer the per-layer loop below still has to visit
    if (mid) {
        // Linear stack of the decoder <- z <- heads <- Linear stack of the encoder: one launch (midblock.hip); it leaves each
        // layer's pre-activation gradient in that layer's keep buffer for the grouped weight-gradient launch
        const int e0 = m->n_enc - mid_ne;                    // index of the block's first encoder layer; e0 - 1 is a conv layer
        float *enc_y[ARVAE_MAX_LAYERS], *dec_y[ARVAE_MAX_LAYERS], *enc_g[ARVAE_MAX_LAYERS], *dec_g[ARVAE_MAX_LAYERS];
        for (int i = 0; i < mid_ne; ++i) { enc_y[i] = ws + L.enc_out[e0 + i]; enc_g[i] = ws + L.enc_keep[e0 + i]; }
        for (int i = 0; i < mid_nd; ++i) { dec_y[i] = ws + L.dec_out[i]; dec_g[i] = ws + L.dec_keep[i]; }
        const float *x0 = ws + L.enc_out[e0 - 1];
        const float *gate0 = relu_gate(m->enc[e0 - 1], enc_mask[e0 - 1], x0);
        float *d_x0 = ws + L.enc_keep[e0 - 1];
        const float *g_last = cur;                           // gradient arriving at the last Linear layer of the decoder
        MidFold mf{};
        if (fold) {
            // `cur` is the gradient at the transposed conv layer's output, left by the layer behind it: the block takes it from
            // there (it must be w.r.t. the pre-activation: that layer's data-gradient kernel gates with the sign bits the
            // forward launch wrote) and hands the gradient at the conv layer's INPUT on
            ARVAE_REQUIRE(pre, "image_vae_backward: the folded conv layer needs a gated upstream gradient");
            mf.hi_e = ws + L.enc_out[e0 - 2];
            mf.g_hi_d = cur;
            mf.d_hi_e = ws + L.enc_keep[e0 - 2];
            mf.d_hi_e_amax = grad_amax(mf.d_hi_e);
            mf.slab_e = ws + L.enc_slab[e0 - 1];
            mf.slab_d = ws + L.dec_slab[mid_nd];
        }
        if (int rc = mid_backward(m, batch, params, ws + L.mid_prep, enc_y, dec_y, enc_g, dec_g, g_last, (pre || fold) ? 1 : 0, gate0, d_x0, eps,
                                  mu, sigma, dz_reg, dz_extra, g_loss, ws + L.kld_out + 1, capacity, m->beta, reg_scale, ws + L.d_mu,
                                  ws + L.d_ls, st, grad_amax(d_x0), fold ? &mf : nullptr, L.mid_wide >= 0 ? ws + L.mid_wide : nullptr))
            return rc;
        if (fold) {                                          // the two layers' weight-gradient slabs: one per workgroup of the block's grid
            const int n_wg = (batch + 31) / 32;                  // one slab of sixteen tap blocks per cluster (reduce.h, SLAB_C32T)
            const arvae_layer_t &ce = m->enc[e0 - 1], &cd = m->dec[mid_nd];
            SlabJob je{mf.slab_e, grads + ce.w_off, ce.b_off >= 0 ? grads + ce.b_off : nullptr, n_wg, SLAB_C32T, ce.b_off >= 0 ? 1 : 0};
            SlabJob jd{mf.slab_d, grads + cd.w_off, cd.b_off >= 0 ? grads + cd.b_off : nullptr, n_wg, SLAB_C32T, cd.b_off >= 0 ? 2 : 0};
            ARVAE_REQUIRE(slab_reduce_defer(&rdefer, jd) && slab_reduce_defer(&rdefer, je), "image_vae_backward: too many slab reductions queued");
        }
        // weight gradients of the block's layers: (pre-activation gradient, layer input) pairs for the grouped launch
        auto queue = [&](const arvae_layer_t &l, const float *g, const float *in) -> int {
            arvae_link_t lk = l.link;
            lk.n = batch;
            float *dw = grads + l.w_off, *db = l.b_off >= 0 ? grads + l.b_off : nullptr;
            const arvae_operand_t gop = plain(g), xin = plain(in);
            if (dense_wgrad_defer(&defer, &lk, make_operand(&gop), in, dw, db)) return ARVAE_OK;
            return arvae_link_wgrad(&lk, &gop, &xin, dw, db, db ? 1 : 0, slab, stream);
        };
        // (the wide layers' own weight-gradient launch first: dense.hip wide_wgrad_x3_kernel, Morpho-MNIST's 2888-wide layers)
        int wide_took = 0;
        if (L.mid_wide >= 0)
            if (int rc = mid_wide_wgrad(m, batch, params, ws + L.mid_prep, ws + L.mid_wide, x0, (pre && !fold) ? g_last : dec_g[mid_nd - 1], grads,
                                        st, &wide_took))
                return rc;
        for (int i = mid_nd - 1; i >= 0; --i) {
            if (i == mid_nd - 1 && (wide_took & 2)) continue;
            if (int rc = queue(m->dec[i], (i == mid_nd - 1 && pre && !fold) ? g_last : dec_g[i], i > 0 ? dec_y[i - 1] : z)) return rc;
        }
        if (int rc = queue(m->head_mu, ws + L.d_mu, hidden)) return rc;
        if (int rc = queue(m->head_log_std, ws + L.d_ls, hidden)) return rc;
        for (int i = mid_ne - 1; i >= 0; --i) {
            if (i == 0 && (wide_took & 1)) continue;
            if (int rc = queue(m->enc[e0 + i], enc_g[i], i > 0 ? enc_y[i - 1] : x0)) return rc;
        }
        cur = fold ? mf.d_hi_e : d_x0;
        cur_amax = grad_amax(cur);
        pre = fold ? true : gate0 != nullptr;                // (folded: gated by the conv layer's saved input inside the launch)
        enc_from = fold ? e0 - 2 : e0 - 1;
        if (ms != nullptr && ms->linear_grads != nullptr) {   // every Linear weight gradient is queued
            if (int rc = dense_wgrad_flush(&defer, st)) return rc;
            mark(ms->linear_grads, st);
            lin_marked = true;
        }
    } else if (heads_fusable(&m->head_mu, &m->head_log_std, m->zdim)) {
        if (int rc = heads_latent_bwd(&m->head_mu, &m->head_log_std, batch, m->zdim, params, heads_next_g != nullptr ? nullptr : cur,
                                      dz_reg, dz_extra, mu, sigma, eps, g_loss, ws + L.kld_out + 1, capacity, m->beta, reg_scale,
                                      head_gate, ws + L.d_mu, ws + L.d_ls, d_hidden, st,
                                      heads_next_g != nullptr ? &m->dec[0] : nullptr, heads_next_g))
            return rc;
        const arvae_layer_t *heads[2] = {&m->head_mu, &m->head_log_std};
        const float *hg[2] = {ws + L.d_mu, ws + L.d_ls};
        for (int k = 0; k < 2; ++k) {                            // weight gradients of the heads join the grouped launch
            arvae_link_t lk = heads[k]->link;
            lk.n = batch;
            float *dw = grads + heads[k]->w_off, *db = heads[k]->b_off >= 0 ? grads + heads[k]->b_off : nullptr;
            const arvae_operand_t gop = plain(hg[k]), xin = plain(hidden);
            if (!dense_wgrad_defer(&defer, &lk, make_operand(&gop), hidden, dw, db))
                if (int rc = arvae_link_wgrad(&lk, &gop, &xin, dw, db, db ? 1 : 0, slab, stream)) return rc;
        }
        cur = d_hidden;
        cur_amax = nullptr;
        pre = head_gate != nullptr;
    } else {
        int64_t blocks = (bz + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        ARVAE_LAUNCH(latent_bwd_full_kernel, dim3((unsigned)blocks), dim3(256), 0, st, cur, dz_reg, dz_extra, mu, sigma,
                           eps, g_loss, ws + L.kld_out + 1, capacity, m->beta, 1.f / (float)batch, reg_scale, bz,
                           ws + L.d_mu, ws + L.d_ls);
        if (int rc = check_launch("image_vae_backward(latent)")) return rc;
        cur = d_hidden;
        float *other = grad_dst(-1, cur);
        if (int rc = layer_backward(m->head_mu, batch, params, grads, hidden, nullptr, nullptr, ws + L.d_mu, true, nullptr,
                                    cur, nullptr, slab, ws + L.link_ws, &defer, nullptr, nullptr, stream))
            return rc;
        if (int rc = layer_backward(m->head_log_std, batch, params, grads, hidden, nullptr, nullptr, ws + L.d_ls, true,
                                    nullptr, other, nullptr, slab, ws + L.link_ws, &defer, nullptr, nullptr, stream))
            return rc;
        const int64_t hn = in_elems(m->head_mu, batch);
        int64_t blocks2 = (hn + 255) / 256;
        if (blocks2 > 2048) blocks2 = 2048;
        ARVAE_LAUNCH(add_inplace_kernel, dim3((unsigned)blocks2), dim3(256), 0, st, cur, other, head_gate, hn);
        if (int rc = check_launch("image_vae_backward(add)")) return rc;
        pre = head_gate != nullptr;
        cur_amax = nullptr;
    }
    // encoder, last layer first; the image itself needs no gradient
    // The first layer's backward pass is its weight gradient alone.  When it is the single-channel convolution and the layer behind
    // it runs the paired 16x16 launch, that launch computes it from the data gradient it holds and stores no data gradient at all
    // (conv32.hip, C1Wgrad: 67 MB less written and 67 MB less read per step at B = 512)
    SlabJob c1_job{};
    bool c1_possible = false;
    if (enc_from >= 1 && masks == nullptr && L.enc_slab[0] >= 0 && L.enc_bits[0] >= 0) {
        arvae_link_t lk0 = m->enc[0].link;
        lk0.n = batch;
        c1_possible = !m->enc[0].is_up && m->enc[0].act == ARVAE_ACT_RELU && m->enc[0].dropout == 0 && conv_c1_fits(&lk0) && lk0.hh == 64 &&
                      lk0.hw == 64 && lk0.clo == 32 && m->enc[0].b_off >= 0;
    }
    for (int i = enc_from; i >= 0; --i) {
        if (i == 0 && c1_job.slab != nullptr) {                 // done inside layer 1's launch: queue its slab reduction
            c1_job.dwt = grads + m->enc[0].w_off;
            c1_job.dbias = grads + m->enc[0].b_off;
            ARVAE_REQUIRE(slab_reduce_defer(&rdefer, c1_job), "image_vae_backward: too many slab reductions queued");
            break;
        }
        const float *in = i > 0 ? ws + L.enc_out[i - 1] : x;
        const float *gate = i > 0 ? relu_gate(m->enc[i - 1], enc_mask[i - 1], in) : nullptr;
        GateOp go;
        const GateOp *gate_op = i > 0 ? general_gate(m->enc[i - 1], enc_mask[i - 1], in, go) : nullptr;
        float *dst = i > 0 ? grad_dst(L.enc_keep[i - 1], cur) : nullptr;
        bool gated = false;
        bool din_has = false;
        if (int rc = layer_backward(m->enc[i], batch, params, grads, in, ws + L.enc_out[i], mask_of(enc_mask[i]), cur, pre,
                                    gate, dst, &gated, slab, ws + L.link_ws, &defer, L.enc_slab[i] >= 0 ? ws + L.enc_slab[i] : nullptr, &rdefer,
                                    stream, nullptr,
                                    (gate != nullptr && i > 0 && L.enc_bits[i - 1] >= 0)
                                        ? reinterpret_cast<const uint16_t *>(ws + L.enc_bits[i - 1]) : nullptr,
                                    L.enc_wprep[i] >= 0 ? ws + L.enc_wprep[i] : nullptr, gate_op, cur_amax, in_amax_of(false, i),
                                    U(L.tmp_amax), U(L.tmp2_amax), grad_amax(dst), &din_has,
                                    L.enc_wide[i][m->enc[i].is_up ? 0 : 1] >= 0 ? ws + L.enc_wide[i][m->enc[i].is_up ? 0 : 1] : nullptr,
                                    (i == 1 && c1_possible) ? x : nullptr, (i == 1 && c1_possible) ? ws + L.enc_slab[0] : nullptr,
                                    (i == 1 && c1_possible) ? &c1_job : nullptr))
            return rc;
        pre = gated;
        cur = dst;
        cur_amax = din_has ? grad_amax(dst) : nullptr;
    }
    // The two closing kernels are independent (disjoint gradients): the slab reduction streams ~55 MB from HBM while the
    // grouped Linear weight gradients are latency-bound in L2.  Side by side on a second stream they measured 22 us SLOWER
    // per step than back to back (fork / join events cost more than the overlap returns; round 2); as ONE grid with the tiles
    // dispatched first they co-reside on the CUs (dense.hip, dense_wgrad_slab_kernel: round 6).
    if (int rc = dense_wgrad_slab_flush(&defer, &rdefer, flush_stream)) return rc;
    if (ms != nullptr) {                                 // milestones with no earlier point: everything is final here
        if (!dec_marked) mark(ms->dec_grads, st);
        if (!lin_marked) mark(ms->linear_grads, st);
    }
    return ARVAE_OK;
}
