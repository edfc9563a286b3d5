// Loss-side kernels of the AR-VAE step: latent head, beta-KL, all-pairs attribute regularisation,
// reconstruction terms, Adam.  All are HBM/latency-bound reductions: coalesced float4 loads,
// wavefront (64-lane) shuffle reductions, per-workgroup partials in a caller-provided workspace and a
// fixed-order finishing pass (bitwise reproducible: no float atomics on the loss values).
#include "common.h"
#include "attributes.h"
#include "regloss.h"
#include "vae_finish.h"

namespace arvae {

// =================================================================================================
// latent head: sigma = exp(log_std), z = mu + eps*sigma            (reference mnist_vae.py:65,79)
// =================================================================================================
__global__ __launch_bounds__(256) void latent_fwd_kernel(const float *__restrict__ mu, const float *__restrict__ ls,
                                                          const float *__restrict__ eps, int64_t count,
                                                          float *__restrict__ sigma, float *__restrict__ z) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        const float s = expf(ls[i]);
        sigma[i] = s;
        z[i] = fmaf(eps[i], s, mu[i]);
    }
}

__global__ __launch_bounds__(256) void latent_bwd_kernel(const float *__restrict__ gz, const float *__restrict__ gs,
                                                          const float *__restrict__ eps,
                                                          const float *__restrict__ sigma, int64_t count,
                                                          float *__restrict__ dmu, float *__restrict__ dls) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        const float g = gz != nullptr ? gz[i] : 0.f;
        dmu[i] = g;
        dls[i] = (g * eps[i] + (gs != nullptr ? gs[i] : 0.f)) * sigma[i];
    }
}

// =================================================================================================
// beta-KL: one workgroup (B*Z is a few thousand elements), fixed summation order
// =================================================================================================
// (kl_elem: vae_finish.h)

__global__ __launch_bounds__(256) void kld_fwd_kernel(const float *__restrict__ mu, const float *__restrict__ sg,
                                                       const float *__restrict__ pm, const float *__restrict__ ps,
                                                       int64_t count, float inv_batch, float beta,
                                                       const float *__restrict__ cap, float *__restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    // one workgroup: eight elements per thread in flight (one element per iteration was a memory round trip each: 13 us for the
    // MeasureVAE's 8192 latent values)
    for (int64_t i0 = threadIdx.x; i0 < count; i0 += 256 * 8) {
        float m[8], g[8], qm[8], qs[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int64_t i = i0 + 256 * u, ic = i < count ? i : 0;
            m[u] = mu[ic]; g[u] = sg[ic];
            qm[u] = pm ? pm[ic] : 0.f; qs[u] = ps ? ps[ic] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + 256 * u < count) s += kl_elem(m[u], g[u], qm[u], qs[u]);
    }
    const float tot = block_sum_256(s, red);
    if (threadIdx.x == 0) {
        const float kl = tot * inv_batch;
        out[0] = beta * fabsf(kl - (cap ? cap[0] : 0.f));
        out[1] = kl;
    }
}

__global__ __launch_bounds__(256) void kld_bwd_kernel(const float *__restrict__ g, const float *__restrict__ mu,
                                                       const float *__restrict__ sg, const float *__restrict__ pm,
                                                       const float *__restrict__ ps, int64_t count, float inv_batch,
                                                       float beta, const float *__restrict__ kl_out,
                                                       const float *__restrict__ cap, float *__restrict__ dmu,
                                                       float *__restrict__ dsg) {
    const float diff = kl_out[1] - (cap ? cap[0] : 0.f);
    const float sgn = diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f);
    const float k = g[0] * beta * sgn * inv_batch;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        const float m0 = pm ? pm[i] : 0.f, s0 = ps ? ps[i] : 1.f;
        const float s = sg[i];
        dmu[i] = k * (mu[i] - m0) / (s0 * s0);
        dsg[i] = k * (s / (s0 * s0) - 1.f / s);
    }
}

// =================================================================================================
// all-pairs attribute regularisation                       (reference utils/trainer.py:369-403)
// grid = (row blocks, R).  The column vectors z_cols[:,d] / lab_cols[:,d] are staged in LDS in chunks
// (zero N x N traffic to HBM); each wavefront owns ROWS_PER_WAVE rows, its 64 lanes stride over the
// staged columns and the two sums (|t-s| and (1-t^2) sgn(t-s)) are reduced with wave shuffles.
// =================================================================================================
__global__ __launch_bounds__(256) void reg_loss_kernel(RegArgs p) {
    __shared__ float xs[REG_CHUNK];
    __shared__ float as[REG_CHUNK];
    reg_loss_block(p, blockIdx.x, blockIdx.y, xs, as);
}
// ... and, for a data-parallel training pass that leaves its finishing step to the backward pass (ARVAE_VAE_DEFER_FINISH,
// vae_finish.h): this launch -- the pass's last -- parks that step's arguments in the workspace and poisons the scalars
__global__ __launch_bounds__(256) void reg_loss_park_kernel(RegArgs p, VaeFinishArgs fin, VaeFinishArgs *__restrict__ fin_dst) {
    __shared__ float xs[REG_CHUNK];
    __shared__ float as[REG_CHUNK];
    if (blockIdx.x == 0 && blockIdx.y == 0) {
        if (threadIdx.x == 0) *fin_dst = fin;
        if (threadIdx.x < 8 && fin.scalars != nullptr) fin.scalars[threadIdx.x] = __builtin_nanf("");
    }
    reg_loss_block(p, blockIdx.x, blockIdx.y, xs, as);
}

// fixed-order finish: loss scalar + dense dz rows.  One workgroup, so what it costs is its chain of memory round trips: every
// element's loads (row loss, the row gradient its dz entry takes) are issued before anything is summed or stored, 4096 elements
// per pass (one pass at B = 512), and dz is written once instead of zeroed, synchronised and scattered (7.8 -> ~4 us).
__global__ __launch_bounds__(1024) void reg_finish_kernel(const float *__restrict__ row_loss,
                                                           const float *__restrict__ row_grad, int64_t n_rows64, int r,
                                                           RegDims dims, int64_t ldz64, float loss_scale,
                                                           float grad_scale, float *__restrict__ loss_out,
                                                           float *__restrict__ dz) {
    __shared__ float red[16];
    constexpr int FU = 4;
    const int n_rows = (int)n_rows64, ldz = (int)ldz64, nr = n_rows * r, nz = dz != nullptr ? n_rows * ldz : 0;
    const int n_max = nr > nz ? nr : nz;
    float s = 0.f;
    for (int base = 0; base < n_max; base += FU * 1024) {
        float rl[FU], rg[FU];
        int dk[FU];
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int i = base + (int)threadIdx.x + u * 1024;
            rl[u] = row_loss[i < nr ? i : 0];
            const int ic = i < nz ? i : 0, row = ic / (ldz > 0 ? ldz : 1), c = ic - row * ldz;
            int k = -1;
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (q < r && dims.d[q] == c) k = q;
            dk[u] = k;
            rg[u] = row_grad[(k < 0 ? 0 : k) * n_rows + row];
        }
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int i = base + (int)threadIdx.x + u * 1024;
            if (i < nr) s += rl[u];
            if (i < nz) dz[i] = dk[u] < 0 ? 0.f : grad_scale * rg[u];
        }
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) tot += red[w];
        loss_out[0] = tot * loss_scale;
    }
}

// =================================================================================================
// image reconstruction term + pixel accuracy (+ d/dlogits)     (image_vae_trainer.py:623-655)
// =================================================================================================
constexpr int RECON_MAX_BLOCKS = 1024;

template <int DIST>
__global__ __launch_bounds__(256) void image_recon_kernel(const float *__restrict__ logits,
                                                           const float *__restrict__ x, int64_t count, float inv_b,
                                                           float *__restrict__ partial, float *__restrict__ dlogits) {
    __shared__ float red[4];
    float loss = 0.f, corr = 0.f;
    const int64_t n4 = count >> 2;
    const float4 *l4 = reinterpret_cast<const float4 *>(logits);
    const float4 *x4 = reinterpret_cast<const float4 *>(x);
    float4 *d4 = reinterpret_cast<float4 *>(dlogits);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 l = l4[i], xx = x4[i];
        float4 d;
        recon_elem<DIST>(l.x, xx.x, inv_b, loss, corr, d.x);
        recon_elem<DIST>(l.y, xx.y, inv_b, loss, corr, d.y);
        recon_elem<DIST>(l.z, xx.z, inv_b, loss, corr, d.z);
        recon_elem<DIST>(l.w, xx.w, inv_b, loss, corr, d.w);
        if (dlogits != nullptr) d4[i] = d;
    }
    if (blockIdx.x == 0 && threadIdx.x < (count & 3)) {       // tail
        const int64_t i = (n4 << 2) + threadIdx.x;
        float d;
        recon_elem<DIST>(logits[i], x[i], inv_b, loss, corr, d);
        if (dlogits != nullptr) dlogits[i] = d;
    }
    const float tl = block_sum_256(loss, red);
    const float tc = block_sum_256(corr, red);
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = tl;
        partial[2 * blockIdx.x + 1] = tc;
    }
}

__global__ __launch_bounds__(256) void pair_finish_kernel(const float *__restrict__ partial, int nblocks, float s0,
                                                           float s1, float *__restrict__ out) {
    __shared__ float red[4];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 256) {
        a += partial[2 * i];
        b += partial[2 * i + 1];
    }
    const float ta = block_sum_256(a, red);
    const float tb = block_sum_256(b, red);
    if (threadIdx.x == 0) {
        out[0] = ta * s0;
        out[1] = tb * s1;
    }
}

// =================================================================================================
// token reconstruction: mean cross entropy + top-1 accuracy over [rows, V]   (utils/trainer.py:247-282)
// one lane per row (V ~ 35 floats)
// =================================================================================================
constexpr int TOK_VMAX = 48;                 // widest vocabulary the staged path takes (64 rows x 48 floats of LDS)
constexpr int TOK_LPR = 4;                   // lanes per row: a 256-thread workgroup takes 64 rows, so B * 24 = 6144 rows are 96
constexpr int TOK_ROWS = 256 / TOK_LPR;      // workgroups instead of 24 (13.8 -> ~6 us at B = 256)
// (a body: the launch may carry the attribute labels' workgroups behind these -- BID / NBLK: this workgroup's index and their number)
__device__ __forceinline__ void token_recon_body(const float *__restrict__ w, const int64_t *__restrict__ tgt, int64_t rows, int vocab,
                                                 float inv_rows, float *__restrict__ partial, float *__restrict__ dw, int tk_batch,
                                                 int tk_beats, int tk_tpb, const int BID, const int NBLK) {
    // tk_beats > 0: the rows are in the tick RNN's sequence order (tick-in-beat j, beat, measure b) and tgt is the score
    // [batch][beats * tpb]: row r = (j * beats + beat) * batch + b reads tgt[b][tpb * beat + j]
    __shared__ float red[4];
    __shared__ float stage[TOK_ROWS * TOK_VMAX];
    float loss = 0.f, corr = 0.f;
    const bool staged = vocab <= TOK_VMAX;
    const int rl = threadIdx.x / TOK_LPR, part = threadIdx.x % TOK_LPR;       // row of this pass, lane's share of its columns
    for (int64_t r0 = (int64_t)BID * TOK_ROWS; r0 < rows; r0 += (int64_t)NBLK * TOK_ROWS) {
        const int64_t r = r0 + rl;
        const int nrow = (int)(rows - r0 < TOK_ROWS ? rows - r0 : TOK_ROWS);
        const float *row = w + r * vocab;
        if (staged) {
            // the rows of this pass are contiguous in memory: coalesced into LDS, then TOK_LPR lanes walk a row there (lanes
            // straight from memory would be 4-byte loads at a 140-byte stride)
            __syncthreads();
            const int total = nrow * vocab;
            for (int i = threadIdx.x; i < total; i += 256) stage[i] = w[r0 * vocab + i];
            __syncthreads();
            row = stage + rl * vocab;
        }
        const bool live = r < rows;
        // first maximum (lowest index on ties, as torch.max(1)): per lane over its columns, then across the row's lanes
        float mx = -INFINITY;
        int arg = 0x7fffffff;
        if (live)
            for (int j = part; j < vocab; j += TOK_LPR) {
                const float v = row[j];
                if (v > mx) { mx = v; arg = j; }
            }
#pragma unroll
        for (int o = 1; o < TOK_LPR; o <<= 1) {
            const float om = __shfl_xor(mx, o);
            const int oa = __shfl_xor(arg, o);
            if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }
        }
        float se = 0.f;
        if (live)
            for (int j = part; j < vocab; j += TOK_LPR) se += expf(row[j] - mx);
#pragma unroll
        for (int o = 1; o < TOK_LPR; o <<= 1) se += __shfl_xor(se, o);
        if (live) {
            int64_t ti = r;
            if (tk_beats > 0) {
                const int b = (int)(r % tk_batch), jb = (int)(r / tk_batch);
                ti = (int64_t)b * (tk_beats * tk_tpb) + (jb % tk_beats) * tk_tpb + jb / tk_beats;
            }
            const int t = (int)tgt[ti];
            if (part == 0) {
                loss += mx + logf(se) - row[t];
                corr += (arg == t) ? 1.f : 0.f;
            }
            if (dw != nullptr) {
                const float inv = 1.f / se;
                if (staged) {                                    // gradient row built in place, written out coalesced below
                    float *srow = stage + rl * vocab;
                    for (int j = part; j < vocab; j += TOK_LPR) srow[j] = (expf(srow[j] - mx) * inv - (j == t ? 1.f : 0.f)) * inv_rows;
                } else {
                    for (int j = part; j < vocab; j += TOK_LPR)
                        dw[r * vocab + j] = (expf(row[j] - mx) * inv - (j == t ? 1.f : 0.f)) * inv_rows;
                }
            }
        }
        if (staged && dw != nullptr) {
            __syncthreads();
            const int total = nrow * vocab;
            for (int i = threadIdx.x; i < total; i += 256) dw[r0 * vocab + i] = stage[i];
        }
    }
    const float tl = block_sum_256(loss, red);
    const float tc = block_sum_256(corr, red);
    if (threadIdx.x == 0) {
        partial[2 * BID] = tl;
        partial[2 * BID + 1] = tc;
    }
}

__global__ __launch_bounds__(256) void token_recon_kernel(const float *__restrict__ w, const int64_t *__restrict__ tgt,
                                                           int64_t rows, int vocab, float inv_rows,
                                                           float *__restrict__ partial, float *__restrict__ dw, int tk_batch = 0,
                                                           int tk_beats = 0, int tk_tpb = 0) {
    token_recon_body(w, tgt, rows, vocab, inv_rows, partial, dw, tk_batch, tk_beats, tk_tpb, blockIdx.x, gridDim.x);
}
// the MeasureVAE executor's cross entropy with the attribute labels riding in its grid (attributes.h: they depend on the score
// alone; a launch of their own was ~5 us of latency for 256 lanes of table look-ups): workgroups [0, nb) the token term, the rest
// one measure per lane
__global__ __launch_bounds__(256) void token_recon_attr_kernel(const float *__restrict__ w, const int64_t *__restrict__ tgt,
                                                                int64_t rows, int vocab, float inv_rows,
                                                                float *__restrict__ partial, float *__restrict__ dw, int tk_batch,
                                                                int tk_beats, int tk_tpb, AttrArgs attr, int nb) {
    if ((int)blockIdx.x >= nb) {
        measure_attributes_rows(attr, ((int)blockIdx.x - nb) * 256 + threadIdx.x, ((int)gridDim.x - nb) * 256);
        return;
    }
    token_recon_body(w, tgt, rows, vocab, inv_rows, partial, dw, tk_batch, tk_beats, tk_tpb, blockIdx.x, nb);
}

__global__ __launch_bounds__(256) void scale_by_scalar_kernel(const float *__restrict__ g, const float *__restrict__ x,
                                                               int64_t count, float *__restrict__ y) {
    const float s = g[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) y[i] = s * x[i];
}

// =================================================================================================
// Adam over the flat arena                                         (utils/trainer.py:31-34,170-174)
// =================================================================================================
// ZERO_G: the gradient is consumed here, so the arena is cleared on the way out (2 MB more of stores in a kernel that moves
// 14 MB) and the next step's zero_grad() has nothing left to do: one fill launch less per training step
template <bool ZERO_G>
__global__ __launch_bounds__(256) void adam_kernel(float *__restrict__ p, float *__restrict__ g,
                                                    float *__restrict__ m, float *__restrict__ v, int64_t count,
                                                    float step_size, float inv_bc2_sqrt, float beta1, float beta2,
                                                    float omb1, float omb2, float eps, float gscale,
                                                    unsigned *__restrict__ status) {
    const int64_t n4 = count >> 2;
    // a pass that reported a failed hand-off (arvae_image_vae_t.status) left undefined gradients: nothing reaches p, m, v;
    // the arena is still cleared for the next pass and the skipped update is counted (status[4]) for the host's step counter
    if (status != nullptr && __builtin_amdgcn_readfirstlane((int)status[0]) != 0) {
        if (ZERO_G) {
            float4 *z4 = reinterpret_cast<float4 *>(g);
            for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
                z4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (blockIdx.x == 0 && threadIdx.x < (count & 3)) g[(n4 << 2) + threadIdx.x] = 0.f;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) status[4] += 1u;
        return;
    }
    float4 *p4 = reinterpret_cast<float4 *>(p);
    float4 *g4 = reinterpret_cast<float4 *>(g);
    float4 *m4 = reinterpret_cast<float4 *>(m);
    float4 *v4 = reinterpret_cast<float4 *>(v);
    auto upd = [&](float &pp, float gg, float &mm, float &vv) {
        gg *= gscale;
        mm = beta1 * mm + omb1 * gg;
        vv = beta2 * vv + omb2 * gg * gg;
        pp -= step_size * (mm / (sqrtf(vv) * inv_bc2_sqrt + eps));
    };
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 pp = p4[i], mm = m4[i], vv = v4[i];
        const float4 gg = g4[i];
        upd(pp.x, gg.x, mm.x, vv.x);
        upd(pp.y, gg.y, mm.y, vv.y);
        upd(pp.z, gg.z, mm.z, vv.z);
        upd(pp.w, gg.w, mm.w, vv.w);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
        if (ZERO_G) g4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (blockIdx.x == 0 && threadIdx.x < (count & 3)) {
        const int64_t i = (n4 << 2) + threadIdx.x;
        upd(p[i], g[i], m[i], v[i]);
        if (ZERO_G) g[i] = 0.f;
    }
}

static inline int grid_for(int64_t count, int per_thread = 1, int max_blocks = 2048) {
    int64_t b = (count + 256ll * per_thread - 1) / (256ll * per_thread);
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int)b;
}


// =================================================================================================
// One-workgroup tail of the conv-VAE forward pass (plan.hip): reduces the reconstruction partials, computes the
// KL term against N(0, I) from (mu, sigma), reduces the regulariser's per-row partials and scatters its gradient,
// and writes the step's scalars.  Replaces four launch-latency-bound kernels; all sums in a fixed order.
// =================================================================================================
__device__ __forceinline__ float block_sum_1024(float v, float *red) {       // red: 16 floats of LDS
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w];
    return t;
}

// (VaeFinishArgs and the finishing body: vae_finish.h -- the last decoder layer's backward launch can carry it, conv_c1.hip)
__global__ __launch_bounds__(1024) void vae_finish_kernel(VaeFinishArgs p) {
    __shared__ float4 red4[16];
    vae_finish_body<1024>(p, red4);
}

int recon_partial_blocks(int64_t count) { return grid_for(count, 8, RECON_MAX_BLOCKS); }

// per-block partial sums (cross entropy, correct top-1) of the token term, rows in the tick RNN's sequence order against the
// score's (batch, tick) targets (+ d/dweights for a unit upstream gradient of the MEAN); the block count through *nb_out
int token_recon_blocks(int64_t rows) { return grid_for(rows * TOK_LPR, 1, RECON_MAX_BLOCKS); }
int token_recon_partials(const float *weights, const int64_t *score, int batch, int beats, int tpb, int32_t vocab, float *ws,
                         float *dweights, hipStream_t s, int *nb_out, const AttrArgs *attr) {
    const int64_t rows = (int64_t)batch * beats * tpb;
    const int nb = token_recon_blocks(rows);
    if (attr != nullptr && attr->out != nullptr)          // (+ the attribute labels: one measure per lane in the grid's last workgroups)
        ARVAE_LAUNCH(token_recon_attr_kernel, dim3(nb + (attr->batch + 255) / 256), dim3(256), 0, s, weights, score, rows, vocab,
                     1.f / (float)rows, ws, dweights, batch, beats, tpb, *attr, nb);
    else
        ARVAE_LAUNCH(token_recon_kernel, dim3(nb), dim3(256), 0, s, weights, score, rows, vocab, 1.f / (float)rows, ws, dweights, batch, beats, tpb);
    *nb_out = nb;
    return check_launch("token_recon");
}

// per-block partial sums of the reconstruction term (+ d/dlogits); returns the block count through *nb
int recon_partials(const float *logits, const float *x, int64_t count, int64_t batch, int32_t dist, float *ws,
                   float *dlogits, hipStream_t s, int *nb_out) {
    const int nb = grid_for(count, 8, RECON_MAX_BLOCKS);
    const float inv_b = 1.f / (float)batch;
    if (dist == ARVAE_RECON_BERNOULLI)
        ARVAE_LAUNCH(image_recon_kernel<ARVAE_RECON_BERNOULLI>, dim3(nb), dim3(256), 0, s, logits, x, count,
                           inv_b, ws, dlogits);
    else
        ARVAE_LAUNCH(image_recon_kernel<ARVAE_RECON_GAUSSIAN>, dim3(nb), dim3(256), 0, s, logits, x, count,
                           inv_b, ws, dlogits);
    *nb_out = nb;
    return check_launch("image_recon");
}

// per-row partial sums of the all-pairs regulariser into ws = [row_loss | row_grad]
int reg_partials(const float *z_rows, const float *lab_rows, int64_t n_rows, const float *z_cols, const float *lab_cols,
                 int64_t n_cols, int64_t ldz, int64_t ldl, const RegDims &rd, int32_t r, float delta, float *ws,
                 hipStream_t s, const VaeFinishArgs *park, VaeFinishArgs *park_dst) {
    const unsigned bx = (unsigned)((n_rows + REG_ROWS_PER_BLOCK - 1) / REG_ROWS_PER_BLOCK);
    RegArgs p{z_rows, lab_rows, n_rows, z_cols, lab_cols, n_cols, ldz, ldl, rd, delta, ws, ws + n_rows * r};
    if (park != nullptr && park_dst != nullptr) ARVAE_LAUNCH(reg_loss_park_kernel, dim3(bx, r), dim3(256), 0, s, p, *park, park_dst);
    else ARVAE_LAUNCH(reg_loss_kernel, dim3(bx, r), dim3(256), 0, s, p);
    return check_launch("reg_loss");
}

VaeFinishArgs vae_finish_args(const float *rec_partial, int nb, int64_t batch, int64_t pix, const float *mu, const float *sigma,
                              int64_t zdim, float beta, const float *cap, const float *reg_ws, int64_t n_cols, int64_t ldz,
                              const int32_t *dims, int32_t r, float gamma, float delta, float reg_scale, float *dz, float *rec_out,
                              float *kld_out, float *reg_out, float *scalars, int64_t rec_rows) {
    VaeFinishArgs p{};
    p.rec_partial = rec_partial; p.nb = nb; p.inv_batch = 1.f / (float)batch; p.inv_count = 1.f / (float)pix;
    p.inv_rec = rec_rows > 0 ? 1.f / (float)rec_rows : p.inv_batch;     // (a mean over rows instead of a per-sample sum: the token term)
    p.mu = mu; p.sigma = sigma; p.bz = batch * zdim; p.beta = beta; p.cap = cap;
    if (reg_ws != nullptr) {
        p.row_loss = reg_ws; p.row_grad = reg_ws + batch * r; p.n_rows = batch; p.r = r;
        for (int i = 0; i < 16; ++i) p.dims.d[i] = i < r ? dims[i] : 0;
        const double nn = (double)n_cols * (double)n_cols;
        p.ldz = ldz; p.loss_scale = (float)(gamma / nn); p.grad_scale = (float)(2.0 * gamma * delta / nn);
        p.dz = dz;
    }
    p.reg_scale = reg_scale;
    p.rec_out = rec_out; p.kld_out = kld_out; p.reg_out = reg_out; p.scalars = scalars;
    return p;
}

int vae_finish_launch(const VaeFinishArgs &p, hipStream_t s) {
    ARVAE_LAUNCH(vae_finish_kernel, dim3(1), dim3(1024), 0, s, p);
    return check_launch("image_vae_forward(finish)");
}

int vae_finish(const float *rec_partial, int nb, int64_t batch, int64_t pix, const float *mu, const float *sigma,
               int64_t zdim, float beta, const float *cap, const float *reg_ws, int64_t n_cols, int64_t ldz,
               const int32_t *dims, int32_t r, float gamma, float delta, float reg_scale, float *dz, float *rec_out,
               float *kld_out, float *reg_out, float *scalars, hipStream_t s, int64_t rec_rows) {
    return vae_finish_launch(vae_finish_args(rec_partial, nb, batch, pix, mu, sigma, zdim, beta, cap, reg_ws, n_cols, ldz, dims, r, gamma, delta,
                                             reg_scale, dz, rec_out, kld_out, reg_out, scalars, rec_rows), s);
}

}  // namespace arvae

using namespace arvae;

extern "C" int arvae_latent_fwd(const float *mu, const float *log_std, const float *eps, int64_t count, float *sigma,
                                float *z, arvae_stream_t stream) {
    ARVAE_REQUIRE(mu && log_std && eps && sigma && z && count > 0, "latent_fwd: bad argument");
    ARVAE_LAUNCH(latent_fwd_kernel, dim3(grid_for(count)), dim3(256), 0, as_stream(stream), mu, log_std, eps,
                       count, sigma, z);
    return check_launch("latent_fwd");
}

extern "C" int arvae_latent_bwd(const float *g_z, const float *g_sigma, const float *eps, const float *sigma,
                                int64_t count, float *d_mu, float *d_log_std, arvae_stream_t stream) {
    ARVAE_REQUIRE(eps && sigma && d_mu && d_log_std && count > 0, "latent_bwd: bad argument");
    ARVAE_LAUNCH(latent_bwd_kernel, dim3(grid_for(count)), dim3(256), 0, as_stream(stream), g_z, g_sigma, eps,
                       sigma, count, d_mu, d_log_std);
    return check_launch("latent_bwd");
}

extern "C" int arvae_kld_fwd(const float *mu, const float *sigma, const float *prior_mu, const float *prior_sigma,
                             int64_t batch, int64_t zdim, float beta, const float *capacity, float *out,
                             arvae_stream_t stream) {
    ARVAE_REQUIRE(mu && sigma && out && batch > 0 && zdim > 0, "kld_fwd: bad argument");
    ARVAE_LAUNCH(kld_fwd_kernel, dim3(1), dim3(256), 0, as_stream(stream), mu, sigma, prior_mu, prior_sigma,
                       batch * zdim, 1.f / (float)batch, beta, capacity, out);
    return check_launch("kld_fwd");
}

extern "C" int arvae_kld_bwd(const float *g, const float *mu, const float *sigma, const float *prior_mu,
                             const float *prior_sigma, int64_t batch, int64_t zdim, float beta, const float *kl_out,
                             const float *capacity, float *d_mu, float *d_sigma, arvae_stream_t stream) {
    ARVAE_REQUIRE(g && mu && sigma && kl_out && d_mu && d_sigma && batch > 0 && zdim > 0, "kld_bwd: bad argument");
    ARVAE_LAUNCH(kld_bwd_kernel, dim3(grid_for(batch * zdim)), dim3(256), 0, as_stream(stream), g, mu, sigma,
                       prior_mu, prior_sigma, batch * zdim, 1.f / (float)batch, beta, kl_out, capacity, d_mu, d_sigma);
    return check_launch("kld_bwd");
}

extern "C" int64_t arvae_reg_loss_ws_floats(int64_t n_rows, int32_t r) { return 2 * n_rows * (int64_t)r; }

extern "C" int arvae_reg_loss(const float *z_rows, const float *lab_rows, int64_t n_rows, const float *z_cols,
                              const float *lab_cols, int64_t n_cols, int64_t ldz, int64_t ldl, const int32_t *dims,
                              int32_t r, float gamma, float delta, float *ws, float *loss_out, float *dz,
                              arvae_stream_t stream) {
    ARVAE_REQUIRE(z_rows && lab_rows && z_cols && lab_cols && dims && ws && loss_out, "reg_loss: null pointer");
    ARVAE_REQUIRE(n_rows > 0 && n_cols > 0 && r > 0 && r <= 16, "reg_loss: need 1..16 dims and a non-empty batch");
    RegDims rd;
    for (int i = 0; i < 16; ++i) rd.d[i] = i < r ? dims[i] : 0;
    for (int i = 0; i < r; ++i)
        ARVAE_REQUIRE(dims[i] >= 0 && dims[i] < ldz && dims[i] < ldl, "reg_loss: dim %d outside z/labels", dims[i]);
    float *row_loss = ws, *row_grad = ws + n_rows * r;
    hipStream_t s = as_stream(stream);
    if (int rc = reg_partials(z_rows, lab_rows, n_rows, z_cols, lab_cols, n_cols, ldz, ldl, rd, r, delta, ws, s, nullptr, nullptr)) return rc;
    const double nn = (double)n_cols * (double)n_cols;
    ARVAE_LAUNCH(reg_finish_kernel, dim3(1), dim3(1024), 0, s, row_loss, row_grad, n_rows, r, rd, ldz,
                       (float)(gamma / nn), (float)(2.0 * gamma * delta / nn), loss_out, dz);
    return check_launch("reg_loss(finish)");
}

extern "C" int64_t arvae_recon_ws_floats(int64_t) { return 2 * RECON_MAX_BLOCKS; }

extern "C" int arvae_image_recon(const float *logits, const float *x, int64_t count, int64_t batch, int32_t dist,
                                 float *ws, float *out, float *dlogits, arvae_stream_t stream) {
    ARVAE_REQUIRE(logits && x && ws && out && count > 0 && batch > 0, "image_recon: bad argument");
    ARVAE_REQUIRE(dist == ARVAE_RECON_BERNOULLI || dist == ARVAE_RECON_GAUSSIAN, "image_recon: invalid dist");
    hipStream_t s = as_stream(stream);
    const float inv_b = 1.f / (float)batch;
    int nb = 0;
    if (int rc = recon_partials(logits, x, count, batch, dist, ws, dlogits, s, &nb)) return rc;
    ARVAE_LAUNCH(pair_finish_kernel, dim3(1), dim3(256), 0, s, ws, nb, inv_b, 1.f / (float)count, out);
    return check_launch("image_recon(finish)");
}

extern "C" int arvae_token_recon(const float *weights, const int64_t *targets, int64_t rows, int32_t vocab, float *ws,
                                 float *out, float *dweights, arvae_stream_t stream) {
    ARVAE_REQUIRE(weights && targets && ws && out && rows > 0 && vocab > 0, "token_recon: bad argument");
    const int nb = grid_for(rows * TOK_LPR, 1, RECON_MAX_BLOCKS);
    hipStream_t s = as_stream(stream);
    const float inv = 1.f / (float)rows;
    ARVAE_LAUNCH(token_recon_kernel, dim3(nb), dim3(256), 0, s, weights, targets, rows, vocab, inv, ws, dweights);
    if (int rc = check_launch("token_recon")) return rc;
    ARVAE_LAUNCH(pair_finish_kernel, dim3(1), dim3(256), 0, s, ws, nb, inv, inv, out);
    return check_launch("token_recon(finish)");
}

extern "C" int arvae_scale_by_scalar(const float *g, const float *x, int64_t count, float *y, arvae_stream_t stream) {
    ARVAE_REQUIRE(g && x && y && count > 0, "scale_by_scalar: bad argument");
    ARVAE_LAUNCH(scale_by_scalar_kernel, dim3(grid_for(count, 4)), dim3(256), 0, as_stream(stream), g, x, count,
                       y);
    return check_launch("scale_by_scalar");
}

extern "C" int arvae_adam_step(float *p, float *g, float *m, float *v, int64_t count, int64_t step, double lr,
                               double beta1, double beta2, double eps, float grad_scale, int32_t zero_grad,
                               uint32_t *status, arvae_stream_t stream) {
    ARVAE_REQUIRE(p && g && m && v && count > 0 && step >= 1, "adam_step: bad argument");
    ARVAE_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
                  "adam_step: arenas must be 16-byte aligned");
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    if (zero_grad)
        ARVAE_LAUNCH(adam_kernel<true>, dim3(grid_for(count, 4)), dim3(256), 0, as_stream(stream), p, g, m, v, count,
                           (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), (float)beta1, (float)beta2, (float)(1.0 - beta1),
                           (float)(1.0 - beta2), (float)eps, grad_scale, status);
    else
        ARVAE_LAUNCH(adam_kernel<false>, dim3(grid_for(count, 4)), dim3(256), 0, as_stream(stream), p, g, m, v, count,
                           (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), (float)beta1, (float)beta2, (float)(1.0 - beta1),
                           (float)(1.0 - beta2), (float)eps, grad_scale, status);
    return check_launch("adam_step");
}
