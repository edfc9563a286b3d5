// The pass's finishing step as a device function: sums the reconstruction partials, the KL term against N(0, I), the regulariser's
// row partials (+ its z-gradient scatter) and writes the pass's scalars (losses.hip vae_finish_kernel: one 1024-thread workgroup
// behind the forward pass).  Round 6: a TRAINING step may leave it to the first launch of its backward pass, where it rides as one
// more workgroup of the last decoder layer's paired launch (conv_c1.hip pair_c1_kernel: 512 threads) instead of holding the chip
// alone for ~9 us between the forward and the backward pass (ARVAE_VAE_DEFER_FINISH, include/arvae_hip.h).
#pragma once
#include "common.h"
#include "regloss.h"

namespace arvae {

__device__ __forceinline__ float kl_elem(float mu, float s, float m0, float s0) {
    const float r = s / s0, d = (mu - m0) / s0;
    const float var = r * r;
    return 0.5f * (var + d * d - 1.f - logf(var));       // torch _kl_normal_normal
}

struct VaeFinishArgs {
    const float *rec_partial; int nb; float inv_batch, inv_count, inv_rec;     // reconstruction (inv_rec scales the summed term)
    const float *mu, *sigma; int64_t bz; float beta; const float *cap;         // KL
    const float *row_loss, *row_grad; int64_t n_rows; int r; RegDims dims;     // regulariser (row_loss null: none)
    int64_t ldz; float loss_scale, grad_scale, reg_scale; float *dz;
    float *rec_out, *kld_out, *reg_out, *scalars;
};

// NT threads of ONE workgroup run it; red4: NT / 64 float4 of LDS scratch
template <int NT>
__device__ __forceinline__ void vae_finish_body(const VaeFinishArgs &p, float4 *red4) {
    // One workgroup, so the kernel is as long as its chain of dependent memory round trips: the first FU * NT (= 8192) elements of
    // every array (all of them at the sizes of this repo's models) are loaded before anything is summed -- one round trip --
    // and only what is left beyond that runs as plain loops.  Fixed summation order; 32-bit index math throughout.
    constexpr int FU = 8192 / NT;
    float a = 0.f, b = 0.f, s = 0.f, t = 0.f;
    const int bz = (int)p.bz;
    const bool reg = p.row_loss != nullptr, want_dz = reg && p.dz != nullptr;
    const int n_rows = (int)p.n_rows, ldz = (int)p.ldz, nr = reg ? n_rows * p.r : 0, nz = want_dz ? n_rows * ldz : 0;
    int n_max = p.nb > bz ? p.nb : bz;
    n_max = n_max > nr ? n_max : nr;
    n_max = n_max > nz ? n_max : nz;
    // passes of 8192 elements of every array, all loads of a pass before its sums (one pass at the dSprites sizes, two for
    // Morpho-MNIST's 1024 x 16 latent values)
    for (int base = 0; base < n_max; base += FU * NT) {
        float2 rp[FU];
        float mu[FU], sg[FU], rl[FU], rg[FU];
        int dk[FU];
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int i = base + threadIdx.x + u * NT;
            rp[u] = reinterpret_cast<const float2 *>(p.rec_partial)[i < p.nb ? i : 0];
            mu[u] = p.mu[i < bz ? i : 0];
            sg[u] = p.sigma[i < bz ? i : 0];
            rl[u] = reg ? p.row_loss[i < nr ? i : 0] : 0.f;
            // dz[row][c] = grad_scale * row_grad[k][row] for c = dims[k], else 0: index arithmetic, then ONE unconditional load
            const int ic = i < nz ? i : 0, row = ic / (ldz > 0 ? ldz : 1), c = ic - row * ldz;
            int k = -1;
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (q < p.r && p.dims.d[q] == c) k = q;
            dk[u] = k;
            rg[u] = want_dz ? p.row_grad[(k < 0 ? 0 : k) * n_rows + row] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < FU; ++u) {
            const int i = base + threadIdx.x + u * NT;
            if (i < p.nb) { a += rp[u].x; b += rp[u].y; }
            if (i < bz) s += kl_elem(mu[u], sg[u], 0.f, 1.f);
            if (i < nr) t += rl[u];
            if (i < nz) p.dz[i] = dk[u] < 0 ? 0.f : p.grad_scale * rg[u];
        }
    }
    float4 v = make_float4(wave_sum(a), wave_sum(b), wave_sum(s), wave_sum(t));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) red4[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) { tot.x += red4[w].x; tot.y += red4[w].y; tot.z += red4[w].z; tot.w += red4[w].w; }
        const float rec = tot.x * p.inv_rec, acc = tot.y * p.inv_count, kl = tot.z * p.inv_batch, reg = tot.w * p.loss_scale;
        const float dist = p.beta * fabsf(kl - (p.cap ? p.cap[0] : 0.f));
        p.rec_out[0] = rec; p.rec_out[1] = acc;
        p.kld_out[0] = dist; p.kld_out[1] = kl;
        if (p.row_loss != nullptr) p.reg_out[0] = reg;
        const float rs = p.row_loss != nullptr ? p.reg_scale * reg : 0.f;
        float *o = p.scalars;
        o[ARVAE_VAE_RECON] = rec; o[ARVAE_VAE_ACC] = acc; o[ARVAE_VAE_DIST] = dist; o[ARVAE_VAE_KL] = kl;
        o[ARVAE_VAE_REG] = rs; o[ARVAE_VAE_LOSS] = rec + dist + rs;
        o[6] = o[7] = 0.f;
    }
}

}  // namespace arvae
