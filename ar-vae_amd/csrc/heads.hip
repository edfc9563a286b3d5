// Encoder heads of the conv VAEs fused with the reparameterisation (reference mnist_vae.py:59-66,79 /
// dsprites_vae.py): mu = W_mu h + b_mu, log_std = W_ls h + b_ls, sigma = exp(log_std), z = mu + eps * sigma in one
// launch, and in the backward pass d(mu, log_std) from the decoder / KL / regulariser gradients together with
// d_hidden = W_mu^T d_mu + W_ls^T d_ls (gated by the hidden layer's ReLU).  The heads are [zdim <= 16] x [h] GEMVs
// per sample: a few MFLOP per batch, so the cost is the number of launches; plain FMA loops, 8 batch rows per
// 256-thread workgroup.
#include <mutex>
#include "diag.h"
#include "common.h"
#include "rng.h"

namespace arvae {

constexpr int HEAD_ROWS = 8;
constexpr int HEAD_ZMAX = 16;
constexpr int HEAD_HMAX = 512;           // hidden width the fused kernels stage in LDS

struct HeadsFwdArgs {
    const float *hidden, *w_mu, *b_mu, *w_ls, *b_ls, *eps;
    float *mu, *log_std, *sigma, *z;
    int batch, h, zdim;
    float *eps_out;              // non-null: draw eps here (rng) and write it for the backward pass
    RngStream rng;
    // optional: the decoder's first Linear layer (z -> next_n units), out = act(W z + b), evaluated here on the rows this
    // workgroup has just produced (one launch and one memory round trip less per step)
    const float *next_w, *next_b;
    float *next_out;
    int next_n, next_act;
};
constexpr int HEAD_NEXT_MAX = 512;      // widest fused first decoder layer (two output columns per thread)

// LDS: HEAD_ROWS hidden rows | 2*zdim weight rows (stride h + 4: conflict-free float4 reads across rows) | outputs.
// Every global load is issued up front (one round of memory latency); the dot products then run from LDS.
__global__ __launch_bounds__(256) void heads_latent_fwd_kernel(HeadsFwdArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int ws = p.h + 4;
    float *hs = lds, *wl = lds + HEAD_ROWS * p.h, *outs = wl + 2 * HEAD_ZMAX * ws;
    const int row0 = blockIdx.x * HEAD_ROWS;
    const int h4 = p.h >> 2;
    for (int i = threadIdx.x; i < HEAD_ROWS * h4; i += 256) {
        const int r = i / h4, row = row0 + r;
        const int rr = row < p.batch ? row : p.batch - 1;         // clamped: unconditional load
        const float4 v = reinterpret_cast<const float4 *>(p.hidden + (int64_t)rr * p.h)[i - r * h4];
        reinterpret_cast<float4 *>(hs)[i] = v;
    }
    for (int i = threadIdx.x; i < 2 * p.zdim * h4; i += 256) {
        const int j = i / h4, k = i - j * h4;
        const float *src = j < p.zdim ? p.w_mu + (int64_t)j * p.h : p.w_ls + (int64_t)(j - p.zdim) * p.h;
        reinterpret_cast<float4 *>(wl + j * ws)[k] = reinterpret_cast<const float4 *>(src)[k];
    }
    const int r = threadIdx.x >> 5, j = threadIdx.x & 31;
    const int row = row0 + r;
    const bool lat = j < p.zdim && row < p.batch;
    const int64_t idx = lat ? (int64_t)row * p.zdim + j : 0;
    const float e = p.eps_out != nullptr ? rng_normal(p.rng, (uint64_t)idx) : p.eps[idx];
    const int jc = j < 2 * p.zdim ? j : 0;
    const float *bp = jc < p.zdim ? p.b_mu : p.b_ls;
    const float bias = bp != nullptr ? bp[jc < p.zdim ? jc : jc - p.zdim] : 0.f;
    __syncthreads();
    {
        const float4 *w = reinterpret_cast<const float4 *>(wl + jc * ws);
        const float4 *x = reinterpret_cast<const float4 *>(hs + r * p.h);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int k = 0; k < h4; ++k) {
            const float4 a = x[k], b = w[k];
            acc.x = fmaf(a.x, b.x, acc.x); acc.y = fmaf(a.y, b.y, acc.y);
            acc.z = fmaf(a.z, b.z, acc.z); acc.w = fmaf(a.w, b.w, acc.w);
        }
        outs[r * 32 + j] = (acc.x + acc.y) + (acc.z + acc.w) + bias;
    }
    __syncthreads();
    float zv = 0.f;
    if (lat) {
        const float m = outs[r * 32 + j], l = outs[r * 32 + j + p.zdim];
        const float s = expf(l);
        zv = fmaf(e, s, m);
        p.mu[idx] = m;
        p.log_std[idx] = l;
        p.sigma[idx] = s;
        p.z[idx] = zv;
        if (p.eps_out != nullptr) p.eps_out[idx] = e;
    }
    if (p.next_w != nullptr) {                                   // uniform
        // weights of this thread's two output columns: requested before the barrier
        float w0[HEAD_ZMAX], w1[HEAD_ZMAX];
        const int n0 = threadIdx.x, n1 = threadIdx.x + 256;
        const bool ok0 = n0 < p.next_n, ok1 = n1 < p.next_n;
#pragma unroll
        for (int q = 0; q < HEAD_ZMAX; ++q) {
            const int qc = q < p.zdim ? q : 0;
            w0[q] = p.next_w[(int64_t)(ok0 ? n0 : 0) * p.zdim + qc];
            w1[q] = p.next_w[(int64_t)(ok1 ? n1 : 0) * p.zdim + qc];
        }
        const float b0 = (p.next_b != nullptr && ok0) ? p.next_b[n0] : 0.f, b1 = (p.next_b != nullptr && ok1) ? p.next_b[n1] : 0.f;
        __syncthreads();                                         // everybody is done with outs[]
        if (j < HEAD_ZMAX) outs[r * HEAD_ZMAX + j] = lat ? zv : 0.f;           // z of the 8 rows
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < HEAD_ROWS; ++rr) {
            if (row0 + rr >= p.batch) break;
            float a0 = b0, a1 = b1;
#pragma unroll
            for (int q = 0; q < HEAD_ZMAX; ++q)
                if (q < p.zdim) {
                    const float zq = outs[rr * HEAD_ZMAX + q];
                    a0 = fmaf(zq, w0[q], a0);
                    a1 = fmaf(zq, w1[q], a1);
                }
            if (ok0) p.next_out[(int64_t)(row0 + rr) * p.next_n + n0] = act_fwd(a0, p.next_act);
            if (ok1) p.next_out[(int64_t)(row0 + rr) * p.next_n + n1] = act_fwd(a1, p.next_act);
        }
    }
}

struct HeadsBwdArgs {
    const float *g_z, *dz_reg, *dz_extra, *mu, *sigma, *eps, *g_loss, *kl, *cap;
    float beta, inv_batch, reg_scale;
    const float *w_mu, *w_ls, *gate;
    float *d_mu, *d_ls, *d_hidden;
    int batch, h, zdim;
    // optional (g_z null): the decoder gradient w.r.t. z computed here, g_z[row][j] = sum_n next_g[row][n] * next_w[n][j], from
    // the pre-activation gradient of the decoder's first Linear layer (the data gradient launch of that layer is dropped)
    const float *next_g, *next_w;
    int next_n;
};

__global__ __launch_bounds__(256) void heads_latent_bwd_kernel(HeadsBwdArgs p) {
    __shared__ float dm[HEAD_ROWS][HEAD_ZMAX], dl[HEAD_ROWS][HEAD_ZMAX];
    __shared__ __attribute__((aligned(16))) float ng[HEAD_ROWS * HEAD_NEXT_MAX];       // next_g rows (fused first decoder layer)
    __shared__ float gzs[HEAD_ROWS][HEAD_ZMAX];
    __shared__ float nw[HEAD_NEXT_MAX * HEAD_ZMAX];                                    // next_w, [n][zdim] as in memory
    const int row0 = blockIdx.x * HEAD_ROWS;
    if (p.next_g != nullptr) {                                   // uniform: g_z of this workgroup's rows
        const int n4 = p.next_n >> 2;
        for (int i = threadIdx.x; i < HEAD_ROWS * n4; i += 256) {
            const int r = i / n4, row = row0 + r < p.batch ? row0 + r : p.batch - 1;
            reinterpret_cast<float4 *>(ng)[i] = reinterpret_cast<const float4 *>(p.next_g + (int64_t)row * p.next_n)[i - r * n4];
        }
        for (int i = threadIdx.x; i < p.next_n * p.zdim; i += 256) nw[i] = p.next_w[i];
        // thread (r, j, part): a quarter of the n range each (4 x 32 threads per row x 8 rows would need 1024: two rows per pass)
        __syncthreads();
        const int j = threadIdx.x & 15, part = (threadIdx.x >> 4) & 3, rl = threadIdx.x >> 6;      // 4 rows per pass
#pragma unroll
        for (int pass = 0; pass < HEAD_ROWS / 4; ++pass) {
            const int r = pass * 4 + rl;
            float acc = 0.f;
            if (j < p.zdim)
                for (int n = part; n < p.next_n; n += 4) acc = fmaf(ng[r * p.next_n + n], nw[n * p.zdim + j], acc);
            acc += __shfl_xor(acc, 16, 64);
            acc += __shfl_xor(acc, 32, 64);
            if (part == 0 && j < HEAD_ZMAX) gzs[r][j] = acc;
        }
        __syncthreads();
    }
    {
        const int r = threadIdx.x >> 5, j = threadIdx.x & 31, row = row0 + r;
        const bool on = j < p.zdim && row < p.batch;
        const int64_t i = on ? (int64_t)row * p.zdim + j : 0;      // clamped: every load below is unconditional
        const float g = p.g_loss[0];
        const float diff = p.kl[0] - (p.cap != nullptr ? p.cap[0] : 0.f);
        const float k = g * p.beta * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f)) * p.inv_batch;
        float gz = p.next_g != nullptr ? gzs[r][j < HEAD_ZMAX ? j : 0] : p.g_z[i];
        if (p.dz_reg != nullptr) gz += g * p.reg_scale * p.dz_reg[i];
        if (p.dz_extra != nullptr) gz += p.dz_extra[i];
        const float s = p.sigma[i], mu = p.mu[i], e = p.eps[i];
        const float a = gz + k * mu;
        const float b = (gz * e + k * (s - 1.f / s)) * s;
        if (on) {
            p.d_mu[i] = a;
            p.d_ls[i] = b;
        }
        if (j < HEAD_ZMAX) {
            dm[r][j] = on ? a : 0.f;
            dl[r][j] = on ? b : 0.f;
        }
    }
    // weights and gate values of this thread's hidden column: issued before the barrier, used after it
    for (int k0 = 0; k0 < p.h; k0 += 256) {                       // uniform trip count (barrier inside)
        const int k = k0 + threadIdx.x;
        const bool kok = k < p.h;
        const int kc = kok ? k : 0;
        float wm[HEAD_ZMAX], wl[HEAD_ZMAX], gv[HEAD_ROWS];
#pragma unroll
        for (int j = 0; j < HEAD_ZMAX; ++j) {
            const int jc = j < p.zdim ? j : 0;
            wm[j] = p.w_mu[(int64_t)jc * p.h + kc];
            wl[j] = p.w_ls[(int64_t)jc * p.h + kc];
        }
#pragma unroll
        for (int r = 0; r < HEAD_ROWS; ++r) {
            const int rr = row0 + r < p.batch ? row0 + r : p.batch - 1;
            gv[r] = p.gate != nullptr ? p.gate[(int64_t)rr * p.h + kc] : 1.f;
        }
        if (k0 == 0) __syncthreads();
#pragma unroll
        for (int r = 0; r < HEAD_ROWS; ++r) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < HEAD_ZMAX; ++j)
                if (j < p.zdim) acc = fmaf(dm[r][j], wm[j], fmaf(dl[r][j], wl[j], acc));
            if (kok && row0 + r < p.batch) p.d_hidden[(int64_t)(row0 + r) * p.h + k] = gv[r] > 0.f ? acc : 0.f;
        }
    }
}

// both heads are plain Linear layers on the same hidden vector, small enough for the fused kernels
bool heads_fusable(const arvae_layer_t *hm, const arvae_layer_t *hl, int zdim) {
    auto plain_dense = [](const arvae_layer_t *l) {
        const arvae_link_t &k = l->link;
        return !l->is_up && k.hh == 1 && k.hw == 1 && k.lh == 1 && k.lw == 1 && k.kh == 1 && k.kw == 1 &&
               k.hi_perm_c == 0 && k.lo_perm_c == 0 && l->act == ARVAE_ACT_NONE && l->dropout == 0;
    };
    return plain_dense(hm) && plain_dense(hl) && hm->link.chi == hl->link.chi && hm->link.clo == zdim &&
           hl->link.clo == zdim && zdim <= HEAD_ZMAX && (hm->link.chi & 3) == 0 && hm->link.chi <= HEAD_HMAX;
}

// the decoder's first layer can ride in the heads kernels: a plain Linear layer on z, no dropout, at most HEAD_NEXT_MAX units
bool heads_next_fusable(const arvae_layer_t *l, int zdim) {
    // Measured at B = 512 (dSprites): the two heads kernels grow by 6.2 + 8.4 us (64 workgroups of 8 rows do the layer's work
    // behind their own dependency chain) while the two launches they replace cost 4.7 + 6.5 us: off unless ARVAE_HEADS_NEXT=1.
    static const bool on = diag_env("ARVAE_HEADS_NEXT") != nullptr;
    const arvae_link_t &k = l->link;
    return on && !l->is_up && k.hh == 1 && k.hw == 1 && k.lh == 1 && k.lw == 1 && k.kh == 1 && k.kw == 1 && k.hi_perm_c == 0 &&
           k.lo_perm_c == 0 && l->dropout == 0 && k.chi == zdim && k.clo <= HEAD_NEXT_MAX && (k.clo & 3) == 0;
}

int heads_latent_fwd(const arvae_layer_t *hm, const arvae_layer_t *hl, int batch, int zdim, const float *params,
                     const float *hidden, const float *eps, float *mu, float *log_std, float *sigma, float *z, hipStream_t s,
                     const arvae_image_vae_t *rng_model, const arvae_layer_t *next, float *next_out) {
    HeadsFwdArgs p;
    p.next_w = nullptr; p.next_b = nullptr; p.next_out = nullptr; p.next_n = 0; p.next_act = 0;
    if (next != nullptr) {
        p.next_w = params + next->w_off; p.next_b = next->b_off >= 0 ? params + next->b_off : nullptr;
        p.next_out = next_out; p.next_n = next->link.clo; p.next_act = next->act;
    }
    p.eps_out = nullptr;
    p.rng = RngStream{0, 0, nullptr, 0};
    if (rng_model != nullptr && rng_model->rng_eps) {
        p.eps_out = const_cast<float *>(eps);
        p.rng = RngStream{rng_model->rng_seed, rng_model->rng_offset, rng_model->rng_dev_step, rng_model->rng_step};
    }
    p.hidden = hidden;
    p.w_mu = params + hm->w_off; p.b_mu = hm->b_off >= 0 ? params + hm->b_off : nullptr;
    p.w_ls = params + hl->w_off; p.b_ls = hl->b_off >= 0 ? params + hl->b_off : nullptr;
    p.eps = eps; p.mu = mu; p.log_std = log_std; p.sigma = sigma; p.z = z;
    p.batch = batch; p.h = hm->link.chi; p.zdim = zdim;
    const size_t lds = (size_t)(HEAD_ROWS * p.h + 2 * HEAD_ZMAX * (p.h + 4) + HEAD_ROWS * 32) * sizeof(float);
    static std::once_flag attr;
    std::call_once(attr, [&] {
        (void)hipFuncSetAttribute((const void *)heads_latent_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (HEAD_ROWS * HEAD_HMAX + 2 * HEAD_ZMAX * (HEAD_HMAX + 4) + HEAD_ROWS * 32) * (int)sizeof(float));
    });
    ARVAE_LAUNCH(heads_latent_fwd_kernel, dim3((batch + HEAD_ROWS - 1) / HEAD_ROWS), dim3(256), lds, s, p);
    return check_launch("heads_latent_fwd");
}

int heads_latent_bwd(const arvae_layer_t *hm, const arvae_layer_t *hl, int batch, int zdim, const float *params,
                     const float *g_z, const float *dz_reg, const float *dz_extra, const float *mu, const float *sigma,
                     const float *eps, const float *g_loss, const float *kl, const float *cap, float beta, float reg_scale,
                     const float *gate, float *d_mu, float *d_ls, float *d_hidden, hipStream_t s, const arvae_layer_t *next,
                     const float *next_g) {
    HeadsBwdArgs p;
    p.next_g = nullptr; p.next_w = nullptr; p.next_n = 0;
    if (next != nullptr) { p.next_g = next_g; p.next_w = params + next->w_off; p.next_n = next->link.clo; }
    p.g_z = g_z; p.dz_reg = dz_reg; p.dz_extra = dz_extra; p.mu = mu; p.sigma = sigma; p.eps = eps;
    p.g_loss = g_loss; p.kl = kl; p.cap = cap;
    p.beta = beta; p.inv_batch = 1.f / (float)batch; p.reg_scale = reg_scale;
    p.w_mu = params + hm->w_off; p.w_ls = params + hl->w_off; p.gate = gate;
    p.d_mu = d_mu; p.d_ls = d_ls; p.d_hidden = d_hidden;
    p.batch = batch; p.h = hm->link.chi; p.zdim = zdim;
    ARVAE_LAUNCH(heads_latent_bwd_kernel, dim3((batch + HEAD_ROWS - 1) / HEAD_ROWS), dim3(256), 0, s, p);
    return check_launch("heads_latent_bwd");
}

}  // namespace arvae
