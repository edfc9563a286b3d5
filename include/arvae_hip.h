/* arvae_hip.h -- C-ABI of libarvae_hip.so, the MI355X (gfx950) AR-VAE training-path library.
 *
 * The reference (ashispati/ar-vae) has no FFI / plugin interface: its hot path is stock PyTorch ops
 * called from Python classes.  This header is therefore the boundary a
 * maintainer binds INSTEAD of those torch calls; every entry point cites the reference call site it
 * replaces (paths relative to the reference repo).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller unless the
 *    parameter is documented as "host".  The library allocates no device memory (every scratch buffer is a caller
 *    workspace whose size an arvae_*_ws_floats function reports) and no call depends on state left by another, so the
 *    forward (main thread) and backward (autograd thread) may call concurrently.  What is process-wide: the once-per-kernel
 *    registration of dynamic LDS sizes (hipFuncSetAttribute behind std::call_once), the device's CU count (queried once), the
 *    run-time binding of RCCL (arvae_comm_*) and the opt-in timeline of arvae_profile_begin/_end.  The library reads NO
 *    environment variable: its diagnostic switches exist only in the -DARVAE_DIAG build (libarvae_hip_diag.so, csrc/diag.h).
 *  - `stream` is a hipStream_t passed as void*; calls only enqueue work and never synchronise.
 *  - return value: 0 = ok, <0 = error (ARVAE_E_*); arvae_last_error_string() gives the thread-local text.
 *  - tensors are fp32, contiguous, CHANNELS-LAST: activations [N, H, W, C]; a 1-channel image
 *    [N,1,H,W] is bit-identical in both layouts.  Weights keep the reference's state_dict layouts.
 *  - "accumulates" means the kernel ADDS into the buffer (caller zeroes it: Trainer.zero_grad()).
 */
#ifndef ARVAE_HIP_H
#define ARVAE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARVAE_ABI_VERSION 11  /* 11: arvae_philox_keep_masks (several Dropout masks, one launch); arvae_adam_step(status): the update is skipped while the sticky status word is set (a pass that reported a failed hand-off never reaches the weights), ARVAE_STATUS_* re-coded so that the word survives a float SUM all-reduce beside the gradients; 10: arvae_comm_init(timeout_ms); arvae_image_vae_t.status / .flags (a sticky device status word: an in-launch hand-off between workgroups that gives up says so there instead of hanging; ARVAE_VAE_NO_CLUSTER keeps the pass on kernels without such hand-offs); 9: arvae_measure_vae_* (whole-model MeasureVAE step), row strides for h0 / dh0 / the beat embeddings (arvae_gru_seq_t, arvae_tick_*); 8: arvae_gru_seq_t.gi_rstride / dgi_rstride / h_fin (merged input projections of a bidirectional layer, final states written by the sequence launch); 7: arvae_comm_* (the data-parallel step's collectives: RCCL on the launch stream, owned by the library); 6: the 32-channel k4 s2 p1 links need caller workspace too (arvae_link_ws_floats / arvae_link_wgrad_ws_floats: the layer's weights as scaled fp16 terms and the operands' maxima); 5: arvae_adam_step(zero_grad), arvae_image_vae_finish, arvae_image_vae_t.milestones (events the executors record for the data-parallel caller's collectives); 4: arvae_philox_* and in-kernel eps (arvae_image_vae_t.rng_*), arvae_tick_free_run_supported, caller workspace for arvae_link_down/up (arvae_link_ws_floats); 3: arvae_gru_seq_*, embed_bwd workspace; 2: arvae_image_vae_backward reg_fused == 2 (unit regulariser gradient in dz_extra) */

#define ARVAE_OK 0
#define ARVAE_E_INVALID (-1)  /* bad argument (null pointer, size out of range, unsupported shape) */
#define ARVAE_E_LAUNCH (-2)   /* hipLaunch / runtime error; text in arvae_last_error_string()        */
#define ARVAE_E_NODEVICE (-3) /* no HIP device visible to this process                              */
#define ARVAE_E_COMM (-4)     /* RCCL missing or a collective / communicator call failed            */

/* activation fused into a producing kernel / differentiated in a consuming one */
#define ARVAE_ACT_NONE 0
#define ARVAE_ACT_RELU 1 /* nn.ReLU  (imagevae/dsprites_vae.py:13-46)                              */
#define ARVAE_ACT_SELU 2 /* nn.SELU  (imagevae/mnist_vae.py:17-45, measurevae/encoder.py:39-51)    */

#define ARVAE_RECON_BERNOULLI 0
#define ARVAE_RECON_GAUSSIAN 1

typedef void *arvae_stream_t;

int arvae_abi_version(void);
const char *arvae_last_error_string(void);
/* number of HIP devices visible (0 on a CPU-only host; never initialises a context) */
int arvae_device_count(void);

/* Opt-in kernel timeline for benchmarking (off by default, no cost when off): after arvae_profile_begin
 * every kernel the library launches on `stream` is followed by a HIP event on that stream; a kernel's
 * duration is the time since the previous event.  arvae_profile_end stops recording, synchronises on
 * the last event and writes one text line per kernel label: "<label>\t<launches>\t<total ms>\n";
 * it returns the buffer size needed.  Labels name the __global__ function (e.g. "up32<16>"). */
int arvae_profile_begin(arvae_stream_t stream);
int64_t arvae_profile_end(char *out, int64_t cap);

/* ------------------------------------------------------------------------------------------------
 * Strided "link" between a HI-resolution tensor hi[N,HH,HW,CHI] and a LO-resolution tensor
 * lo[N,LH,LW,CLO] through a weight wt[CLO][CHI][KH][KW]:
 *     lo(n,ly,lx,clo) <-> hi(n, ly*stride - pad + ky, lx*stride - pad + kx, chi)
 * One link serves six roles of the reference's layers:
 *   nn.Conv2d            (weight [Cout,Cin,KH,KW]: CLO=Cout, CHI=Cin)   forward = DOWN, dgrad = UP
 *   nn.ConvTranspose2d   (weight [Cin,Cout,KH,KW]: CLO=Cin,  CHI=Cout)  forward = UP,   dgrad = DOWN
 *   nn.Linear            (weight [out,in]: KH=KW=HH=HW=LH=LW=1, CLO=out, CHI=in) forward = DOWN
 * and the weight gradient of all three = WGRAD.
 * Replaces: imagevae/dsprites_vae.py:12-46, imagevae/mnist_vae.py:16-47 (layers) as executed by
 * imagevae/mnist_vae.py:59-72 (encode/decode) and their autograd backward (utils/trainer.py:140).
 *
 * hi_perm_c/hi_perm_hw (and lo_*): when >0 the channel index of that tensor is a FLATTENED NCHW
 * feature index f = c*hw + p of a [C=perm_c, HW=perm_hw] map that is stored channels-last, i.e. it
 * lives at memory channel p*perm_c + c.  This is how `hidden.view(B,-1)` (mnist_vae.py:61) and
 * `.view(B,-1,inter,inter)` (mnist_vae.py:70) are honoured without a transpose pass.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t n;
    int32_t hh, hw, chi;
    int32_t lh, lw, clo;
    int32_t kh, kw, stride, pad;
    int32_t hi_perm_c, hi_perm_hw;
    int32_t lo_perm_c, lo_perm_hw;
} arvae_link_t;

/* An operand read by a kernel.  Plain tensor: y = mask = NULL.  Gradient operand: the value used is
 *   v * act'(y) * (mask ? 2*mask : 1)
 * where y is the SAVED OUTPUT of the layer that produced the tensor v is the gradient of (act' is
 * evaluated from the output: ReLU y>0; SELU y>0 ? scale : y + scale*alpha; with dropout the saved
 * output is act(.)*2*mask so act' is taken at y/2) and mask the uint8 keep-mask of nn.Dropout(0.5). */
typedef struct {
    const float *v;
    const float *y;
    const uint8_t *mask;
    int32_t act;
} arvae_operand_t;

/* Floats of caller workspace arvae_link_down / arvae_link_up need for this link (0 for most geometries; the wide stride-1
 * convolutions re-order their weights to [out channel][tap][in channel] there before the product; the 32-channel k4 s2 p1
 * links split their weights into scaled fp16 terms there and keep the input's per-workgroup maxima).  Independent of n. */
int64_t arvae_link_ws_floats(const arvae_link_t *link);

/* lo = epilogue( sum_{ky,kx,chi} hi * wt + bias[clo] );  epilogue = act, then *2*out_mask if given.
 * bias may be NULL.  ws: arvae_link_ws_floats(link) floats, 16-byte aligned (may be NULL when that is 0); its contents are
 * scratch of this call only. */
int arvae_link_down(const arvae_link_t *link, const arvae_operand_t *hi, const float *wt,
                    const float *bias, int32_t out_act, const uint8_t *out_mask, float *lo, float *ws,
                    arvae_stream_t stream);

/* hi = epilogue( sum_{ky,kx,clo} lo * wt + bias[chi] ), the adjoint map of arvae_link_down. */
int arvae_link_up(const arvae_link_t *link, const arvae_operand_t *lo, const float *wt,
                  const float *bias, int32_t out_act, const uint8_t *out_mask, float *hi, float *ws,
                  arvae_stream_t stream);

/* dwt[clo][chi][ky][kx] += sum_{n,ly,lx} lo * hi   (accumulates).  The batch is split over workgroups;
 * partial tiles go to `ws` (arvae_link_wgrad_ws_floats(link) floats, may be 0 -> ws unused) and are summed
 * in a fixed order: bitwise reproducible, no float atomics.
 * The bias gradient rides along: bias_side 1: dbias[clo] += sum_pixels lo (nn.Conv2d / nn.Linear, whose
 * output is the lo tensor); 2: dbias[chi] += sum_pixels hi (nn.ConvTranspose2d); 0: none (dbias unused). */
int64_t arvae_link_wgrad_ws_floats(const arvae_link_t *link);
int arvae_link_wgrad(const arvae_link_t *link, const arvae_operand_t *lo, const arvae_operand_t *hi,
                     float *dwt, float *dbias, int32_t bias_side, float *ws, arvae_stream_t stream);

/* out[c] += sum_rows operand[row, c]  for a [rows, channels] channels-last view (bias gradients).
 * perm_c/perm_hw as in arvae_link_t (out is indexed by the flattened NCHW feature).
 * ws: arvae_channel_sum_ws_floats(rows, channels) floats. */
int64_t arvae_channel_sum_ws_floats(int64_t rows, int32_t channels);
int arvae_channel_sum(const arvae_operand_t *g, int64_t rows, int32_t channels, int32_t perm_c,
                      int32_t perm_hw, float *out, float *ws, arvae_stream_t stream);

/* Weight (and bias) gradients of several nn.Linear layers in one launch: dw[n_out][n_in] += g^T x, dbias += column sums
 * of g (NULL: skip), g = the layer's output gradient as an operand (activation derivative / keep-mask folded in). */
typedef struct arvae_dense_wgrad_job {
    arvae_operand_t g;        /* [rows][n_out] */
    const float *x;           /* [rows][n_in] layer input */
    float *dw;                /* accumulated */
    float *dbias;             /* accumulated, or NULL */
    int32_t rows, n_in, n_out, reserved;
} arvae_dense_wgrad_job_t;
int arvae_dense_wgrad_batch(const arvae_dense_wgrad_job_t *jobs, int32_t njobs, arvae_stream_t stream);

/* out[i] = value of the operand with act'(y) and the keep-mask folded in: a plain copy of a gradient operand. */
int arvae_operand_apply(const arvae_operand_t *g, int64_t count, float *out, arvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Latent head.  Replaces z_dist = Normal(mu, exp(log_std)); z = z_dist.rsample()
 * (imagevae/mnist_vae.py:65,74-87; measurevae/encoder.py:123, measure_vae.py:115-123) with the noise
 * eps as an explicit input:  sigma = exp(log_std),  z = mu + eps * sigma.
 * ------------------------------------------------------------------------------------------------ */
int arvae_latent_fwd(const float *mu, const float *log_std, const float *eps, int64_t count,
                     float *sigma, float *z, arvae_stream_t stream);
/* d_mu += ... is NOT accumulated: d_mu = g_z, d_log_std = (g_z*eps + g_sigma)*sigma (g_sigma may be NULL) */
int arvae_latent_bwd(const float *g_z, const float *g_sigma, const float *eps, const float *sigma,
                     int64_t count, float *d_mu, float *d_log_std, arvae_stream_t stream);

/* beta-KL term.  Replaces Trainer.compute_kld_loss (utils/trainer.py:354-367):
 *   kl = mean_b sum_z 0.5*((s/s0)^2 + ((mu-m0)/s0)^2 - 1 - log((s/s0)^2));  out[0] = beta*|kl - c|,
 *   out[1] = kl.  prior_mu/prior_sigma NULL = standard normal.  `capacity` is a 1-element device
 *   tensor (the reference keeps c as a tensor: image_vae_trainer.py:94) or NULL for c = 0.
 * bwd: d_mu, d_sigma = g[0] * d out[0] / d(mu, sigma)  (g is a 1-element device tensor). */
int arvae_kld_fwd(const float *mu, const float *sigma, const float *prior_mu, const float *prior_sigma,
                  int64_t batch, int64_t zdim, float beta, const float *capacity, float *out,
                  arvae_stream_t stream);
int arvae_kld_bwd(const float *g, const float *mu, const float *sigma, const float *prior_mu,
                  const float *prior_sigma, int64_t batch, int64_t zdim, float beta, const float *kl_out,
                  const float *capacity, float *d_mu, float *d_sigma, arvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Attribute-regularisation loss over all pairs.  Replaces the per-dimension Python loop
 * (imagevae/image_vae_trainer.py:171-180, measurevae/measure_vae_trainer.py:135-139) over
 * Trainer.compute_reg_loss / reg_loss_sign (utils/trainer.py:369-403), all R dims in one launch:
 *   loss = sum_r gamma/NC^2 * sum_{i in rows, j in cols} | tanh(delta*(z_i - z_j)) - sign(a_i - a_j) |
 *   dz[i, dims[r]] = 2*gamma*delta/NC^2 * sum_j (1 - t_ij^2) * sgn(t_ij - s_ij)      (other columns 0)
 * Rows are this rank's samples, cols the (all-gathered) global batch; single GPU: cols == rows.
 * dims: HOST array of R latent/label column indices (z[:,d] pairs with labels[:,d]).
 * ws: device scratch of arvae_reg_loss_ws_floats(n_rows, R) floats.  loss_out: 1 float.
 * dz: [n_rows, ldz] fully written (gradient for unit upstream gradient), may be NULL.
 * ------------------------------------------------------------------------------------------------ */
int64_t arvae_reg_loss_ws_floats(int64_t n_rows, int32_t r);
int arvae_reg_loss(const float *z_rows, const float *lab_rows, int64_t n_rows, const float *z_cols,
                   const float *lab_cols, int64_t n_cols, int64_t ldz, int64_t ldl, const int32_t *dims,
                   int32_t r, float gamma, float delta, float *ws, float *loss_out, float *dz,
                   arvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Reconstruction terms.
 * Image: replaces ImageVAETrainer.reconstruction_loss + mean_accuracy
 * (imagevae/image_vae_trainer.py:623-655).  out[0] = sum_all(term)/batch, out[1] = pixel accuracy.
 * dlogits (optional) = d out[0] / d logits.  ws: arvae_recon_ws_floats(count) floats.
 * ------------------------------------------------------------------------------------------------ */
int64_t arvae_recon_ws_floats(int64_t count);
int arvae_image_recon(const float *logits, const float *x, int64_t count, int64_t batch, int32_t dist,
                      float *ws, float *out, float *dlogits, arvae_stream_t stream);

/* Measure: replaces Trainer.mean_crossentropy_loss + Trainer.mean_accuracy
 * (utils/trainer.py:247-282) on [rows, V] ReLU-ed logits and int64 targets.
 * out[0] = mean CE, out[1] = top-1 accuracy (lowest index on ties).  dweights optional. */
int arvae_token_recon(const float *weights, const int64_t *targets, int64_t rows, int32_t vocab,
                      float *ws, float *out, float *dweights, arvae_stream_t stream);

/* y[i] = g[0] * x[i]   (chain rule through a scalar loss term) */
int arvae_scale_by_scalar(const float *g, const float *x, int64_t count, float *y, arvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Adam over one flat fp32 arena.  Replaces torch.optim.Adam(...).step() (utils/trainer.py:31-34,
 * 170-174) with defaults beta1=.9 beta2=.999 eps=1e-8, no weight decay, no amsgrad:
 *   m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;  p -= (lr/(1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * Hyper-parameters are doubles (as torch holds them: 1-beta2 is formed in double, then rounded).
 * `step` is the 1-based step number t.  grad_scale multiplies g first (1/world_size after a SUM
 * all-reduce; 1 otherwise).  zero_grad != 0: g is cleared once it has been consumed, i.e. the update and the NEXT
 * step's Trainer.zero_grad() (utils/trainer.py:136) are one kernel.
 * status (optional, NULL: none): device words the caller owns.  status[0] is the sticky status word of the passes that
 * produced g (arvae_image_vae_t.status, ARVAE_STATUS_*); while it is non-zero the launch changes NOTHING in p, m, v -- the
 * gradient of a pass whose in-launch hand-off gave up is undefined and must not reach the weights (the reference would have
 * raised at its per-step host read of the loss, utils/trainer.py:145-147; this build reads the word once per epoch) -- it
 * still clears g when zero_grad is set, and adds one to status[4], the count of skipped updates: the caller that finally
 * reads the word takes that many steps back off its step counter, so the bias corrections continue where the last
 * APPLIED update left them.
 * ------------------------------------------------------------------------------------------------ */
int arvae_adam_step(float *p, float *g, float *m, float *v, int64_t count, int64_t step, double lr, double beta1,
                    double beta2, double eps, float grad_scale, int32_t zero_grad, uint32_t *status,
                    arvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * MeasureVAE building blocks (GRU encoder / hierarchical GRU decoder over 24-tick measures).
 * The GEMMs of nn.GRU (W_ih x and W_hh h, measurevae/encoder.py:27-34,115; decoder.py:338-368,504) run on
 * arvae_link_down with a dense link; these are the non-GEMM pieces.
 * ------------------------------------------------------------------------------------------------ */
/* GRU gate math of one time step.  gi = W_ih x + b_ih, gh = W_hh h + b_hh as [batch, 3*hidden] (gates r|z|n);
 * h_prev [batch, hidden] or NULL (zeros).  saved (optional, for the backward): 4*batch*hidden floats (r, z, n, gh_n). */
int arvae_gru_gates_fwd(const float *gi, const float *gh, const float *h_prev, int32_t batch, int32_t hidden,
                        float *h_new, float *saved, arvae_stream_t stream);
/* dh = d/d h_new.  Writes dgi, dgh [batch, 3*hidden] and dh_prev = dh * z (the direct path; the caller adds W_hh^T dgh). */
int arvae_gru_gates_bwd(const float *dh, const float *saved, const float *h_prev, int32_t batch, int32_t hidden,
                        float *dgi, float *dgh, float *dh_prev, arvae_stream_t stream);

/* Whole-sequence GRU layer: all `steps` time steps of one nn.GRU layer (measurevae/encoder.py:27-34,113-118 two-layer
 * bidirectional over 24 ticks; measurevae/decoder.py:338-368 beat RNN, :436-525 tick RNN) in one launch, forward and
 * backward-through-time.  `gi` holds the input projections W_ih x + b_ih of every step (one dense launch beforehand);
 * the recurrence h -> W_hh h + b_hh -> gates runs inside the kernel, 16 batch rows per workgroup with W_hh in registers.
 * Up to 4 independent parameter sets / directions per launch (e.g. the forward and reverse directions of a layer).
 * Built for hidden sizes 32, 64 and 128 (arvae_gru_seq_supported); the per-step kernels above serve any other size. */
typedef struct arvae_gru_seq {
    const float *gi;      /* [steps][rows][3*hidden], gi_tstride floats between steps (0: one block reused every step) */
    int64_t gi_tstride;
    const float *w_hh;    /* [3*hidden][hidden] */
    const float *b_hh;    /* [3*hidden] */
    const float *h0;      /* [rows][hidden] or NULL (zeros) */
    float *h_all;         /* output: h of (step t, row r) at h_all[(t*rows + r) * h_stride + 0..hidden) */
    int64_t h_stride;
    float *saved;         /* [steps][rows][hidden][4] (layout private to the kernels): written by fwd, read by bwd */
    int32_t reverse;      /* process t = steps-1 .. 0 (the `_reverse` direction of a bidirectional layer) */
    int32_t reserved;
    const float *dh_all;  /* bwd: gradient w.r.t. h_all, addressed like h_all with dh_stride; NULL = zeros */
    int64_t dh_stride;
    float *dgi;           /* bwd out: [steps][rows][3*hidden] gradient w.r.t. gi */
    float *dgh;           /* bwd out: [steps][rows][3*hidden] gradient w.r.t. W_hh h + b_hh */
    float *dh0;           /* bwd out: [rows][hidden] gradient w.r.t. h0, or NULL */
    const float *dh_last; /* bwd: gradient w.r.t. the final state (h_n of nn.GRU: h of the last processed step), row r at
                             dh_last + r*dh_last_stride; NULL = zeros */
    int64_t dh_last_stride;
    float *h_prev_out;    /* bwd out, optional: [steps][rows][hidden], the state entering each step (operand of dW_hh) */
    /* both directions of a bidirectional layer share ONE input projection [steps][rows][2 * 3*hidden] (measurevae/encoder.py:27-34:
     * nn.GRU applies W_ih of both directions to the same input): floats between two rows of gi / of dgi; 0 = 3*hidden */
    int64_t gi_rstride, dgi_rstride;
    float *h_fin;         /* fwd out, optional: the state after the last processed step (nn.GRU's h_n), row r at h_fin + r*h_fin_stride */
    int64_t h_fin_stride;
    /* floats between two rows of h0 / of dh0; 0 = hidden.  Lets the initial states be column blocks of the Linear layer's output that
     * makes them (measurevae/decoder.py:388-406 view(B, 2, H)) and their gradients column blocks of that layer's output gradient */
    int64_t h0_stride, dh0_stride;
} arvae_gru_seq_t;
int arvae_gru_seq_supported(int32_t hidden);
int arvae_gru_seq_fwd(const arvae_gru_seq_t *seqs, int32_t nseq, int32_t steps, int32_t rows, int32_t hidden,
                      arvae_stream_t stream);
int arvae_gru_seq_bwd(const arvae_gru_seq_t *seqs, int32_t nseq, int32_t steps, int32_t rows, int32_t hidden,
                      arvae_stream_t stream);

/* Free-running pass of the tick decoder (measurevae/decoder.py:459-525 without teacher forcing): 2-layer GRU, note
 * projection + ReLU, top-1 note (lowest index on ties) embedded as the next input, hidden state restarted from
 * h0_l0 / h0_l1 [beats*batch][hidden] (row = beat*batch + b) at every beat.  One launch, returns only the tokens
 * [batch][beats*ticks_per_beat]; the caller evaluates the differentiable graph on them with arvae_gru_seq_*.
 * The layer-0 input projection arrives pre-multiplied: gib [beats*batch][3*hidden] = W_ih0[:, E:] beat_emb + b_ih0 and
 * ptab [vocab+1][3*hidden] = W_ih0[:, :E] applied to the embedding table, row `vocab` = the learned start vector x_0.
 * mask: optional keep-mask [beats*ticks_per_beat][batch][hidden] of nn.GRU's inter-layer dropout, scaled by keep_scale.
 * ws: arvae_tick_free_run_ws_floats(hidden) floats, 16-byte aligned (the recurrent weights re-laid out for streaming). */
typedef struct arvae_tick_weights {
    const float *w_hh0, *b_hh0;     /* rnn_tick layer 0 recurrent weights */
    const float *w_ih1, *b_ih1, *w_hh1, *b_hh1;   /* layer 1 */
    const float *w_out, *b_out;     /* tick_emb_to_note_emb [vocab][hidden] */
} arvae_tick_weights_t;
int64_t arvae_tick_free_run_ws_floats(int32_t hidden);
/* 1 when arvae_tick_free_run is built for this (hidden, vocab): hidden in {32, 64, 128} and vocab <= min(64, 16*(hidden/16));
 * otherwise the caller runs the decoder tick by tick (arvae_gru_gates_fwd + arvae_row_argmax), as the reference does
 * (measurevae/decoder.py:469-525). */
int arvae_tick_free_run_supported(int32_t hidden, int32_t vocab);
int arvae_tick_free_run(const arvae_tick_weights_t *weights, const float *h0_l0, const float *h0_l1, int64_t h0_stride /* 0 = hidden */,
                        const float *gib,
                        const float *ptab, const uint8_t *mask, float keep_scale, int32_t batch, int32_t beats,
                        int32_t ticks_per_beat, int32_t hidden, int32_t vocab, int64_t *tokens, float *ws,
                        arvae_stream_t stream);

/* nn.Embedding (measurevae/encoder.py:36-37,111; decoder.py:18,516): out row (b,t) = table[idx[b][t]];
 * time_major: rows ordered (t, b) instead of (b, t).  embed_bwd adds to (accumulate != 0) or overwrites dtable, in a fixed summation order, and needs
 * a workspace of arvae_embed_bwd_ws_floats() floats.  `table` may also be a per-vocabulary PROJECTION table: a Linear layer
 * applied to embedded tokens is a lookup of table W^T + b (the encoder's layer-0 input projection of both directions,
 * 768 columns at hidden 128: encoder.py:27-37,111-114 computes it per position) -- any `dim`; rows wider than 128 take a
 * column-parallel segment-sum kernel in embed_bwd (vocab * 1 KB + 256 B of LDS <= 64 KB). */
int arvae_embed_fwd(const int64_t *idx, const float *table, int32_t batch, int32_t steps, int32_t dim, int32_t vocab,
                    int32_t time_major, float *out, arvae_stream_t stream);
int64_t arvae_embed_bwd_ws_floats(int32_t batch, int32_t steps, int32_t dim, int32_t vocab);
int arvae_embed_bwd(const int64_t *idx, const float *g, int32_t batch, int32_t steps, int32_t dim, int32_t vocab,
                    int32_t time_major, float *dtable, int32_t accumulate /* 0: dtable is overwritten */, float *ws,
                    arvae_stream_t stream);

/* The tick RNN's layer-0 input projection, re-associated (measurevae/decoder.py:459-505 applies W_ih0 to [embedding of the previous
 * note | beat embedding] at each of the 24 ticks x batch positions).  The note half takes vocab + 1 distinct values (the vocabulary's
 * embeddings and the learned start vector x_0), the beat half one per (beat, measure), so ONE small product G = X W_ih0^T over
 *     x_small [vocab + 1 + rows][emb + hidden] = [table | 0] (vocab rows), [x_0 | 0] (1 row), [0 | beat_emb] (rows = beats*batch)
 * (arvae_tick_rows_fwd builds it; G by arvae_link_down) gives every tick's projection as a gather:
 *     gi[(j*beats + beat)*batch + b][:] = G[prev][:] + G[vocab + 1 + beat*batch + b][:] + bias[:],
 *     prev = tokens[b][ticks_per_beat*beat + j - 1], or row `vocab` (x_0) at tick 0            (arvae_tick_gi_fwd)
 * in the row order the tick RNN's sequence launches take (arvae_gru_seq_*: ticks_per_beat steps over beats*batch rows).
 * Backward: arvae_tick_gi_bwd sums the per-tick gradients dgi into dg_small [vocab + 1 + rows][cols] (OVERWRITTEN: per previous
 * note in a fixed order, per beat row over its ticks); dx_small = dg_small W_ih0 (arvae_link_up) is split by arvae_tick_rows_bwd
 * into the table's and x_0's gradients (ADDED; either may be NULL) and the beat embedding's (written).
 * cols (= 3 * hidden of the tick RNN) must be a multiple of 4; ws: arvae_tick_gi_bwd_ws_floats(vocab, cols) floats. */
int arvae_tick_rows_fwd(const float *table, const float *x0, const float *beat_emb, int64_t beat_emb_stride /* floats between rows, 0 = hidden */,
                        int32_t vocab, int32_t emb, int32_t hidden, int32_t rows, float *x_small, arvae_stream_t stream);
int arvae_tick_rows_bwd(const float *dx_small, int32_t vocab, int32_t emb, int32_t hidden, int32_t rows, float *dtable, float *dx0,
                        float *dbeat_emb, int64_t dbeat_emb_stride /* 0 = hidden */, arvae_stream_t stream);
int arvae_tick_gi_fwd(const float *g_small, const int64_t *tokens, const float *bias, int32_t batch, int32_t beats,
                      int32_t ticks_per_beat, int32_t vocab, int32_t cols, float *gi, arvae_stream_t stream);
int64_t arvae_tick_gi_bwd_ws_floats(int32_t vocab, int32_t cols);
int arvae_tick_gi_bwd(const float *dgi, const int64_t *tokens, int32_t batch, int32_t beats, int32_t ticks_per_beat, int32_t vocab,
                      int32_t cols, float *dg_small, float *ws, arvae_stream_t stream);

/* top-1 index per row, lowest index on ties (the decoder's argmax feedback, measurevae/decoder.py:506-507) */
int arvae_row_argmax(const float *w, int32_t rows, int32_t cols, int64_t *idx, arvae_stream_t stream);

/* out[r] = [a[r] | b[r]] (torch.cat along dim 1/2, decoder.py:503) and its adjoint (db optionally accumulated) */
int arvae_concat_cols(const float *a, const float *b, int64_t rows, int32_t ca, int32_t cb, float *out,
                      arvae_stream_t stream);
int arvae_split_cols(const float *g, int64_t rows, int32_t ca, int32_t cb, float *da, float *db, int32_t accumulate_b,
                     arvae_stream_t stream);

/* y = alpha * x * mask (mask NULL = 1), optionally added to y: inter-layer GRU dropout and gradient sums */
int arvae_scale_mask(const float *x, const uint8_t *mask, float alpha, int64_t count, int32_t accumulate, float *y,
                     arvae_stream_t stream);
/* y[r][:] = v[:]  (the learned start vectors x_0 / b_0 expanded over the batch, decoder.py:459-463,485-489) */
int arvae_broadcast_rows(const float *v, int64_t rows, int32_t cols, float *y, arvae_stream_t stream);

/* Minibatch assembly from a device-resident uint8 image set: out[b][:] = scale * src[idx[b]][:] as fp32.
 * Replaces the host-side float32 copy of the whole dSprites / MNIST set and the DataLoader's per-batch H2D copy
 * (data/dataloaders/dsprites_dataset.py:38-53 `images.astype('float32')`, mnist_dataset.py:60-82 `/ 255.0`).
 * src [n_rows, row_elems] uint8 (row_elems % 4 == 0), idx [count] int64 row numbers (out-of-range rows give zeros). */
int arvae_gather_rows_u8(const uint8_t *src, int64_t n_rows, int64_t row_elems, const int64_t *idx, int64_t count,
                         float scale, float *out, arvae_stream_t stream);

/* Attribute labels of a batch of measures.  Replaces MeasureVAETrainer.compute_attribute_labels
 * (measurevae/measure_vae_trainer.py:167-186 -> data/dataloaders/bar_dataset.py:338-500):
 *   out[b] = [rhythmic complexity, pitch range/26, note density, contour/26]
 * midi_lut / is_note / is_density_note: per-vocabulary tables (is_density_note also counts the `None` symbol,
 * bar_dataset.py:348-356); rhythm_weights[steps] = RHY_COMPLEXITY_COEFFS (bar_dataset_helpers.py:21-30),
 * rhythm_norm their sum. */
int arvae_measure_attributes(const int64_t *score, int32_t batch, int32_t steps, const int32_t *midi_lut,
                             const uint8_t *is_note, const uint8_t *is_density_note, int32_t vocab,
                             const float *rhythm_weights, float rhythm_norm, float *out, arvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Whole-model step for the conv VAEs: ONE call enqueues every kernel of the forward pass (+ loss terms)
 * or of the backward pass, so the host does no per-layer work.  Replaces, as a unit,
 *   ImageVAETrainer.loss_and_acc_for_batch (imagevae/image_vae_trainer.py:137-217)
 *     = MnistVAE.forward (imagevae/mnist_vae.py:89-105) + reconstruction_loss + compute_kld_loss
 *       + the compute_reg_loss loop + mean_accuracy,
 *   and loss.backward() (utils/trainer.py:140) for that graph.
 * The model is described layer by layer; weights/biases are float offsets into ONE parameter arena and
 * gradients accumulate at the same offsets of a gradient arena (the trainer's flat Adam arenas).
 * ------------------------------------------------------------------------------------------------ */
#define ARVAE_MAX_LAYERS 8

typedef struct {
    arvae_link_t link;    /* geometry; link.n is ignored (the batch is a call argument)               */
    int32_t is_up;        /* 0: forward is DOWN (nn.Conv2d / nn.Linear), 1: UP (nn.ConvTranspose2d)   */
    int32_t act;          /* ARVAE_ACT_* applied after the bias                                        */
    int32_t dropout;      /* 1: nn.Dropout(0.5) follows (a keep-mask is consumed in train mode)        */
    int32_t reserved;
    int64_t w_off, b_off; /* float offsets of weight and bias in the parameter / gradient arenas       */
} arvae_layer_t;

struct arvae_milestones;
typedef struct {
    int32_t n_enc, n_dec;                 /* encoder layers up to the hidden vector; decoder layers z -> logits */
    arvae_layer_t enc[ARVAE_MAX_LAYERS];
    arvae_layer_t dec[ARVAE_MAX_LAYERS];
    arvae_layer_t head_mu, head_log_std;  /* dense heads on the encoder's hidden vector               */
    int32_t zdim;
    int32_t recon_dist;                   /* ARVAE_RECON_*                                             */
    int32_t n_reg;                        /* regularised dims (0: no attribute regularisation)         */
    int32_t reg_dims[16];
    float beta, gamma, delta;
    /* rng_eps != 0: arvae_image_vae_forward DRAWS the reparameterisation noise itself (Philox4x32-10, arvae_philox_normal's
     * stream for the same seed / offset / step / dev_step) inside the fused heads kernel and WRITES it to `eps`, which the
     * backward pass then reads; 0: `eps` is an input (parity runs with explicit noise). */
    int32_t rng_eps;
    uint32_t rng_offset, rng_step;
    uint64_t rng_seed;
    const uint32_t *rng_dev_step;
    /* Optional (NULL: none).  hipEvent_t handles, created by the caller, which the executors RECORD on `stream` as soon as a
     * result that other work may start from is complete -- so that a data-parallel caller can start its RCCL collectives on
     * a side stream while the rest of the pass still runs (the reference is single-process: SURVEY.md section 8(e), "one
     * all-gather between encoder forward and the reg kernel", "all-reduce overlapped with the tail of backward").  Every
     * non-NULL event is recorded exactly once per call (at the end of the pass if the executor has no earlier point for it).
     *   z_ready       arvae_image_vae_forward : mu / sigma / z are final (the decoder's launches follow)
     *   dec_grads     arvae_image_vae_backward: the gradients of every conv layer of the DECODER are final in `grads`
     *   linear_grads  arvae_image_vae_backward: the gradients of every Linear layer and of the two heads are final
     * With dec_grads / linear_grads set, the backward pass finishes those gradients early (the decoder layers' slab
     * reduction and the grouped Linear weight gradients are launched where their inputs are complete instead of at the end
     * of the pass: one launch more). */
    const struct arvae_milestones *milestones;
    /* Optional (NULL: none).  ONE device word the caller owns and zeroes once; the passes only ever OR bits into it
     * (ARVAE_STATUS_*).  The latent block of the dSprites-shaped model runs on clusters of workgroups that hand activations to
     * each other inside a launch (csrc/midcluster.hip); a hand-off whose partners never arrive -- they cannot all become
     * resident, e.g. many processes on one device -- gives up after a bounded poll, sets its bit and the launch ends: the
     * results of THAT pass are then undefined.  The caller reads the word where it synchronises anyway (the reference reads
     * its loss every step, utils/trainer.py:145-147; this build once per epoch), raises, and sets ARVAE_VAE_NO_CLUSTER in
     * `flags` for every later call. */
    uint32_t *status;
    int32_t flags;        /* ARVAE_VAE_* */
    int32_t reserved;
} arvae_image_vae_t;

/* The word is zero or a NORMAL POSITIVE fp32 bit pattern (2.0 .. 3.75: ARVAE_STATUS_SET + code bits in the top of the
 * mantissa), so that a data-parallel caller can keep it in the tail of the buffer it SUM all-reduces as floats
 * (ar-vae_amd/optim.py: the guard slot behind the gradient arena): a failure on one rank then reaches every rank's word
 * with the gradients it polluted, and every rank's arvae_adam_step skips the same update.  After such a sum only
 * "non-zero" is meaningful; on the rank that failed (and on a single rank) the code bits read back exactly. */
#define ARVAE_STATUS_SET 0x40000000u
#define ARVAE_STATUS_HANDOFF_FWD (ARVAE_STATUS_SET | (1u << 20))    /* a hand-off of the clustered latent block's forward launch gave up */
#define ARVAE_STATUS_HANDOFF_BWD (ARVAE_STATUS_SET | (2u << 20))    /* ... of its backward launch                                         */
#define ARVAE_STATUS_HANDOFF_TICKET (ARVAE_STATUS_SET | (4u << 20)) /* a workgroup found no place in any cluster (corrupt ticket heads)   */
#define ARVAE_VAE_NO_CLUSTER 1         /* flags: the latent block on the row kernels (no in-launch hand-offs)  */
/* flags (ABI 11), a TRAINING step only: arvae_image_vae_forward leaves the pass's finishing step -- the sums that become
 * scalars[] (loss, its terms, accuracy), the KL mean and the regulariser's z-gradient, ~9 us of one workgroup with the chip idle
 * -- to arvae_image_vae_backward, whose first launch carries it as one more workgroup (or which launches it first thing where that
 * launch is of another kind).  Until the backward call has run, scalars[0 .. 7] read as NaN.  Set the same flag for both calls
 * of a step; never for a forward pass no backward pass follows. */
#define ARVAE_VAE_DEFER_FINISH 2

typedef struct arvae_milestones {
    void *z_ready, *dec_grads, *linear_grads;
} arvae_milestones_t;

/* scalars written by the forward pass (device array of ARVAE_VAE_NSCALARS floats) */
#define ARVAE_VAE_LOSS 0   /* recon + dist + reg_scale*reg                                              */
#define ARVAE_VAE_RECON 1
#define ARVAE_VAE_DIST 2   /* beta * |kl - c|                                                            */
#define ARVAE_VAE_REG 3    /* reg_scale * (sum over dims of the row-block regularisation loss)          */
#define ARVAE_VAE_ACC 4
#define ARVAE_VAE_KL 5
#define ARVAE_VAE_NSCALARS 8

int64_t arvae_image_vae_ws_floats(const arvae_image_vae_t *model, int32_t batch, int64_t n_cols);

/* Forward + loss terms.  x [batch, H, W, 1]; labels [batch, ld_labels]; eps [batch, zdim] (WRITTEN when model->rng_eps);
 * masks: HOST array with one device uint8 keep-mask per dropout layer (encoder first), or NULL (eval).
 * z_cols/lab_cols [n_cols, ...]: all-gathered columns for the data-parallel row-block regularisation
 * (NULL: this batch is the whole batch); they index dims 0..n_reg-1 compactly when given.
 * n_cols: 0 = the regulariser runs on this batch alone; > 0 with z_cols/lab_cols; -1 = no regulariser in this call (the
 * caller evaluates it on z); -2 = neither the regulariser nor the scalars: arvae_image_vae_finish completes the pass.
 * reg_scale multiplies the regularisation term (world size under data parallelism, else 1).
 * Outputs: scalars[ARVAE_VAE_NSCALARS], mu/sigma/z [batch, zdim], logits [batch, H, W, 1].
 * Everything the backward pass needs stays in ws. */
int arvae_image_vae_forward(const arvae_image_vae_t *model, int32_t batch, const float *params, const float *x,
                            const float *labels, int64_t ld_labels, const float *eps,
                            const uint8_t *const *masks, const float *capacity, const float *z_cols,
                            const float *lab_cols, int64_t n_cols, float reg_scale, float *ws, float *scalars,
                            float *mu, float *sigma, float *z, float *logits, arvae_stream_t stream);

/* Data-parallel completion of a forward pass that was called with n_cols == -2 ("the caller finishes"): the regulariser of
 * this rank's rows against the columns GATHERED from every rank (whole rows: z_cols [n_cols, zdim], lab_cols [n_cols,
 * ld_labels], n_cols = world size x batch), then the pass's scalars exactly as the forward pass
 * would have written them: scalars[ARVAE_VAE_REG] = reg_scale x the row-block term.  The backward pass follows with
 * reg_fused == 1 and the same reg_scale.  (The reference evaluates the term on one process: utils/trainer.py:369-403.) */
int arvae_image_vae_finish(const arvae_image_vae_t *model, int32_t batch, const float *labels, int64_t ld_labels,
                           const float *capacity, const float *z_cols, const float *lab_cols, int64_t n_cols,
                           float reg_scale, float *ws, float *scalars, const float *mu, const float *sigma, const float *z,
                           arvae_stream_t stream);

/* Backward of scalars[ARVAE_VAE_LOSS] times g_loss[0] (device scalar): parameter gradients ACCUMULATE into
 * grads at the layers' offsets.  Must follow arvae_image_vae_forward on the same ws, with the same
 * x / eps / masks / capacity and that call's mu / sigma / z / logits outputs.
 * reg_fused: 1 when the forward evaluated the regularisation term itself (n_cols >= 0); 0: no regulariser here;
 *            2: dz_extra holds the regulariser's gradient w.r.t. z for UNIT upstream (arvae_reg_loss's dz, evaluated
 *            by a data-parallel caller on its row block against the gathered columns): it is scaled by
 *            g_loss[0] * reg_scale here.
 * dz_extra: with reg_fused 0 or 1 an optional extra gradient w.r.t. z [batch, zdim], ALREADY multiplied by the
 *            upstream gradient. */
int arvae_image_vae_backward(const arvae_image_vae_t *model, int32_t batch, const float *params, float *grads,
                             const float *x, const float *eps, const uint8_t *const *masks,
                             const float *capacity, const float *mu, const float *sigma, const float *z,
                             const float *logits, const float *g_loss, const float *dz_extra,
                             int32_t reg_fused, float reg_scale, float *ws, arvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Whole-model step for MeasureVAE: ONE call enqueues every kernel of the forward pass (+ loss terms) or of the backward pass
 * (about 45 / 45 launches at the reference configuration), so the host does no per-layer work.  Replaces, as a unit,
 *   MeasureVAETrainer.loss_and_acc_for_batch (measurevae/measure_vae_trainer.py:85-140)
 *     = MeasureVAE.forward (measurevae/measure_vae.py:97-131: Encoder.forward, encoder.py:108-124; HierarchicalDecoder.forward,
 *       decoder.py:408-525) + mean_crossentropy_loss + compute_kld_loss + compute_attribute_labels + the compute_reg_loss loop
 *       + mean_accuracy,
 *   and loss.backward() (utils/trainer.py:140) for that graph.
 * The model is the reference configuration's shape: a 2-layer bidirectional GRU encoder, two 2-layer heads, a 2-layer beat GRU and
 * a 2-layer tick GRU with teacher-forced or argmax-fed inputs.  Parameters are float offsets into ONE parameter arena, gradients
 * ACCUMULATE at the same offsets of a gradient arena (the trainer's flat Adam arenas).  Three pairs of tensors must lie back to
 * back in the arenas, because one product applies both (ar-vae_amd/measure_vae.py MeasureVAE.arena_parameters):
 *   per encoder layer  weight_ih_l{k} | weight_ih_l{k}_reverse   and   bias_ih_l{k} | bias_ih_l{k}_reverse,
 *   linear_mean.0 | linear_log_std.0 (weights, biases),  beat_emb_to_tick_rnn_hidden.0 | beat_emb_to_tick_rnn_input.0.
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t vocab, emb;                       /* note vocabulary, note_embedding_dim                                         */
    int32_t enc_hidden, dec_hidden, zdim;     /* hidden sizes must be built as sequence kernels (arvae_gru_seq_supported)     */
    int32_t steps, beats, ticks_per_beat;     /* 24 = 4 * 6 in the reference; steps == beats * ticks_per_beat                 */
    /* float offsets into the parameter / gradient arenas */
    int64_t enc_table;                        /* encoder.note_embedding_layer.weight [vocab][emb]                             */
    int64_t enc_w_ih[2], enc_b_ih[2];         /* per layer: the forward direction's; the reverse direction's follows directly  */
    int64_t enc_w_hh[2][2], enc_b_hh[2][2];   /* [layer][direction]                                                           */
    int64_t head_w0, head_b0;                 /* linear_mean.0 followed directly by linear_log_std.0                          */
    int64_t mean_w2, mean_b2, lstd_w2, lstd_b2;
    int64_t dec_table, x0, b0;                /* decoder.note_embedding_layer.weight, x_0 [emb], b_0 [1]                       */
    int64_t z2beat_w, z2beat_b;               /* z_to_beat_rnn_input.0                                                        */
    int64_t beat_w_ih[2], beat_b_ih[2], beat_w_hh[2], beat_b_hh[2];
    int64_t tick_init_w, tick_init_b;         /* beat_emb_to_tick_rnn_hidden.0 followed directly by beat_emb_to_tick_rnn_input.0 */
    int64_t tick_w_ih[2], tick_b_ih[2], tick_w_hh[2], tick_b_hh[2];
    int64_t out_w, out_b;                     /* tick_emb_to_note_emb.0                                                       */
    float enc_dropout, dec_dropout;           /* p of nn.GRU(dropout = p); applied when the call is given keep-masks          */
    int32_t n_reg;                            /* regularised dims (0: no attribute regularisation)                            */
    int32_t reg_dims[16];                     /* z[:, d] pairs with attribute column d (d < 4)                                */
    float beta, gamma, delta;
    /* rng_draw != 0: the forward pass DRAWS the encoder keep-mask, eps and the decoder keep-masks itself (arvae_philox_* streams
     * rng_offset[0..2] of rng_seed / rng_step / rng_dev_step) and WRITES them to the caller's buffers, which the backward pass then
     * reads; 0: they are inputs (parity runs with explicit noise and masks). */
    int32_t rng_draw;
    uint32_t rng_offset[3], rng_step;
    uint64_t rng_seed;
    const uint32_t *rng_dev_step;
} arvae_measure_vae_t;

/* the per-vocabulary tables of arvae_measure_attributes */
typedef struct {
    const int32_t *midi_lut;
    const uint8_t *is_note, *is_density_note;
    const float *rhythm_weights;
    float rhythm_norm;
} arvae_measure_tables_t;

int64_t arvae_measure_vae_ws_floats(const arvae_measure_vae_t *model, int32_t batch);

/* Forward + loss terms.  score [batch][steps] int64; eps [batch][zdim]; enc_mask [steps][batch][2*enc_hidden] and dec_mask
 * [beats + steps][batch][dec_hidden] uint8 keep-masks of the GRUs' inter-layer dropout (beat RNN's first), both NULL = evaluation
 * mode (no dropout); eps / masks are WRITTEN when model->rng_draw.  teacher_forced: the tick RNN is fed the score's notes
 * (decoder.py:427-428,510-512), else its own argmax (one free-running launch, not differentiated: decoder.py:506-516).
 * capacity: 1-element device tensor or NULL (c = 0).  tables: needed when model->n_reg > 0.
 * Outputs: scalars[ARVAE_VAE_NSCALARS] (loss = cross entropy + beta |KL - c| + sum_d gamma reg_d; ARVAE_VAE_ACC = top-1 accuracy),
 * mu / sigma / z [batch][zdim], tokens [batch][steps] = the notes fed back (the score when teacher_forced).
 * Everything the backward pass needs stays in ws (arvae_measure_vae_ws_floats floats, 16-byte aligned). */
int arvae_measure_vae_forward(const arvae_measure_vae_t *model, int32_t batch, const float *params, const int64_t *score,
                              float *eps, uint8_t *enc_mask, uint8_t *dec_mask, int32_t teacher_forced, const float *capacity,
                              const arvae_measure_tables_t *tables, float *ws, float *scalars, float *mu, float *sigma, float *z,
                              int64_t *tokens, float *labels /* [batch][4] attribute labels out, or NULL */, int32_t defer_finish,
                              arvae_stream_t stream);

/* Data-parallel completion of a forward pass called with defer_finish != 0 (which stops before the regulariser and the scalars): the
 * regulariser of this rank's rows against the columns GATHERED from every rank (z_cols [n_cols][zdim], lab_cols [n_cols][4],
 * n_cols = world size x batch: SURVEY.md section 8(e)), then the pass's scalars exactly as the forward pass would have written them,
 * scalars[ARVAE_VAE_REG] = reg_scale x the row-block term (reg_scale = world size).  The backward pass follows with the same reg_scale.
 * (The reference evaluates the term on one process: utils/trainer.py:369-403, measurevae/measure_vae_trainer.py:129-137.) */
int arvae_measure_vae_finish(const arvae_measure_vae_t *model, int32_t batch, const float *capacity, const float *z_cols,
                             const float *lab_cols, int64_t n_cols, float reg_scale, float *ws, float *scalars, const float *mu,
                             const float *sigma, const float *z, const float *labels, arvae_stream_t stream);

/* Backward of scalars[ARVAE_VAE_LOSS] times g_loss[0] (device scalar): parameter gradients ACCUMULATE into grads at the model's
 * offsets.  Must follow arvae_measure_vae_forward on the same ws, with the same score / eps / masks / capacity and that call's
 * mu / sigma / z / tokens / scalars outputs. */
int arvae_measure_vae_backward(const arvae_measure_vae_t *model, int32_t batch, const float *params, float *grads,
                               const int64_t *score, const float *eps, const uint8_t *enc_mask, const uint8_t *dec_mask,
                               const float *capacity, const float *mu, const float *sigma, const float *z, const int64_t *tokens,
                               const float *scalars, const float *g_loss, float reg_scale /* 1, or the world size after _finish */,
                               float *ws, arvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Random draws of the path: eps of z_dist.rsample() (imagevae/mnist_vae.py:79, measurevae/measure_vae.py:116) and the
 * keep-masks of nn.Dropout(0.5) / nn.GRU(dropout=0.5) (imagevae/mnist_vae.py:16-47, measurevae/encoder.py:27-34,
 * decoder.py:338-368) from a counter-based Philox4x32-10 generator: element i of a draw is a pure function of
 * (seed, offset = index of the draw inside the step, step + *dev_step, i).  dev_step (optional device word) lets a captured
 * HIP graph advance the stream by itself.  The reference draws from torch's global generator instead; parity runs pass
 * explicit eps / masks, so only the distribution has to agree (tests: known-answer vectors, moments, determinism).
 * arvae_philox_normal: one N(0,1) value per element (Box-Muller).  arvae_philox_keep_mask: uint8 1 with probability
 * keep_prob (quantised to 1/256), out 16-byte aligned.  arvae_philox_keep_masks: up to 8 such masks (the five Dropout(0.5)
 * layers of imagevae/mnist_vae.py:16-47) as ONE launch; mask j is draw offsets[j] of the step: the same bytes as
 * arvae_philox_keep_mask(outs[j], counts[j], keep_prob, seed, offsets[j], step, dev_step).
 * ------------------------------------------------------------------------------------------------ */
int arvae_philox_normal(float *out, int64_t count, uint64_t seed, uint32_t offset, uint32_t step, const uint32_t *dev_step,
                        arvae_stream_t stream);
int arvae_philox_keep_mask(uint8_t *out, int64_t count, float keep_prob, uint64_t seed, uint32_t offset, uint32_t step,
                           const uint32_t *dev_step, arvae_stream_t stream);
int arvae_philox_keep_masks(int32_t n_masks, uint8_t *const *outs, const int64_t *counts, float keep_prob, uint64_t seed,
                            const uint32_t *offsets, uint32_t step, const uint32_t *dev_step, arvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Debug-mode failure checks (off by default; the Python side runs them with ARVAE_CHECK=1 and raises ValueError as the
 * reference does): number of NaN / infinite values of a parameter arena (measurevae/encoder.py:101-106,
 * decoder.py:420-425) and number of note indices outside [lo, hi) (Decoder.check_index, decoder.py:30-41), ADDED to the
 * device word `flag` (the caller zeroes it; integer adds, so the count is exact).
 * ------------------------------------------------------------------------------------------------ */
int arvae_count_nonfinite(const float *values, int64_t count, int32_t *flag, arvae_stream_t stream);
int arvae_count_out_of_range(const int64_t *indices, int64_t count, int64_t lo, int64_t hi, int32_t *flag,
                             arvae_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Data-parallel exchange of a row-sharded training step (the reference is single-process, so
 * there is no call site to replace -- these run where the sharded step needs what the reference reads from one batch):
 *   - all-gather of the regularised latent / label columns: the attribute term averages over ALL pairs of the global batch
 *     (Trainer.reg_loss_sign, utils/trainer.py:378-403), each rank evaluates its row block against the gathered columns;
 *   - SUM all-reduce of the flat gradient arena between loss.backward() and optimizer.step() (utils/trainer.py:140-147),
 *     arvae_adam_step then applies 1 / world through grad_scale;
 *   - broadcast of rank 0's parameters / shuffling order when training starts (utils/trainer.py:39-64 builds them on one
 *     process), MAX / MIN reductions for timing and for agreeing on a fallback.
 * The collectives are RCCL calls enqueued on `stream` like any kernel of this library: nothing synchronises, nothing is
 * polled by a helper thread, and a stream capture records them into the same HIP graph as the kernels around them.
 * RCCL is resolved at run time (the librccl.so.1 already loaded in the process, else the system one): a single-GPU
 * process never loads it.  One communicator = one device; every rank of the job calls arvae_comm_init together.
 * A communicator is caller-owned state (create / destroy), not library state.
 *
 *   arvae_comm_available  -> RCCL's version code (> 0) or ARVAE_E_COMM with the reason in arvae_last_error_string()
 *   arvae_comm_unique_id  -> ARVAE_COMM_ID_BYTES host bytes; rank 0 creates them, the caller's launcher (a TCP store, MPI, a
 *                            file) hands them to every other rank
 *   arvae_comm_init       -> waits until all `world` ranks have joined, on the CURRENT HIP device, for at most timeout_ms
 *                            (0: no deadline); ARVAE_E_COMM after the deadline: RCCL cannot be called back from a half-built
 *                            communicator, so the process must exit then
 *   arvae_comm_async_error-> 0, or ARVAE_E_COMM once the communicator has failed asynchronously (a peer died)
 * dtype: ARVAE_COMM_*; `count` in elements (per rank for the all-gather: recv holds world x count, rank-major).
 * all_reduce and broadcast work in place.
 * ------------------------------------------------------------------------------------------------ */
#define ARVAE_COMM_ID_BYTES 128
#define ARVAE_COMM_F32 0
#define ARVAE_COMM_F64 1
#define ARVAE_COMM_I64 2
#define ARVAE_COMM_U8 3
#define ARVAE_COMM_SUM 0
#define ARVAE_COMM_MAX 1
#define ARVAE_COMM_MIN 2
typedef struct arvae_comm_s *arvae_comm_t;
int arvae_comm_available(void);
int arvae_comm_unique_id(void *id_out /* host */);
int arvae_comm_init(const void *id /* host */, int32_t rank, int32_t world, int32_t timeout_ms, arvae_comm_t *out /* host */);
int arvae_comm_destroy(arvae_comm_t comm);
int arvae_comm_abort(arvae_comm_t comm);
int arvae_comm_rank(arvae_comm_t comm);
int arvae_comm_world(arvae_comm_t comm);
int arvae_comm_async_error(arvae_comm_t comm);
int arvae_comm_all_gather(arvae_comm_t comm, const void *send, void *recv, int64_t count, int32_t dtype,
                          arvae_stream_t stream);
int arvae_comm_all_reduce(arvae_comm_t comm, void *buf, int64_t count, int32_t dtype, int32_t op, arvae_stream_t stream);
int arvae_comm_broadcast(arvae_comm_t comm, void *buf, int64_t count, int32_t dtype, int32_t root, arvae_stream_t stream);
/* the collectives enqueued by this thread between the two calls go out as ONE RCCL launch (ncclGroupStart / ncclGroupEnd):
 * the z and the label all-gather of a step */
int arvae_comm_group_begin(void);
int arvae_comm_group_end(void);

#ifdef __cplusplus
}
#endif
#endif /* ARVAE_HIP_H */
