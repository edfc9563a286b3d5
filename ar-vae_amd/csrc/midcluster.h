// The latent block of the dSprites-shaped conv VAE on CLUSTERS of workgroups (midcluster.hip): argument block and entry points.
#pragma once
#include "common.h"
#include "rng.h"

namespace arvae {

constexpr int MC_S = 16;        // workgroups per cluster = column slices of a partitioned layer
constexpr int MC_R = 32;        // batch rows per cluster
constexpr int MC_K0 = 512;      // conv feature width on both sides of the block (dsprites_vae.py:22,33: 32 x 4 x 4)
constexpr int MC_H = 256;       // hidden width of the Linear stacks

// Words [0, MC_TICKET_BASE) of McArgs.counters are the arrival counters (32 clusters x 32 words); the ticket heads follow,
// one 128-byte line each: up to 8 heads, the placed-clusters counter, one spare.
constexpr int MC_MAX_CLUSTERS = 32;
constexpr int MC_TICKET_BASE = MC_MAX_CLUSTERS * 32;
constexpr int MC_TICKET_DONE = MC_TICKET_BASE + 8 * 32, MC_TICKET_GLOBAL = MC_TICKET_BASE + 9 * 32;
constexpr int MC_COUNTER_WORDS = MC_TICKET_BASE + 10 * 32;

// One matrix in CLUSTER LAYOUT: the product out[rows][o] = sum_r in[rows][r] * M[r][o] with the reduce axis padded to KB
// blocks of 16 and the output axis cut into S slices of CT column tiles of 16.  Stored so that the B operand of
// v_mfma_f32_16x16x4_f32 for four consecutive steps is ONE 16-byte load per lane, lanes contiguous (1 KB per wave load):
//     w[(((slice * CT + ct) * KB + b) * 64 + lane) * 4 + j] = M[16 b + 4 (lane / 16) + j][slice * 16 CT + 16 ct + lane % 16]
// (zero where the indices pass the matrix).  midprep.h writes it.
struct McMat {
    const float *w;
    const float *bias;          // [o] in memory order, or null
};

// A 32-channel k4 / s2 / p1 conv link between the 4x4 map at the block's edge and the 8x8 map beyond it, computed by the block
// itself (McArgs.fold): its matrices in cluster layout (midprep.h, McConvPrep) and its bias in the reference's order.
struct McConv {
    const float *down, *up, *bias;
};

struct McArgs {
    int batch, zdim, clusters;
    unsigned long long wait_ticks;   // bound of a hand-off poll, in 10 ns ticks of the polling wave's own running time
    int debug_static;           // diagnostic build only (ARVAE_MIDC_STATIC): places by blockIdx instead of tickets
    int debug_drop;             // diagnostic build only (ARVAE_MIDC_DROP_ARRIVAL): one member never arrives at the first hand-off
    int heads;                  // ticket heads the places of a pass are dealt from (4, 2 or 1; divides `clusters`): midcluster.hip
    unsigned *counters;         // one arrival counter per cluster, 32 words apart; multiples of MC_S between phases; behind
                                // them (MC_TICKET_BASE) the ticket heads a pass hands its cluster places out from
    unsigned *status;           // sticky error word of the caller (arvae_image_vae_t.status) or null: a hand-off that
                                // gives up ORs ARVAE_STATUS_HANDOFF_* into it and the workgroup leaves
    // forward matrices: enc0 [K0 -> H], enc1 [H -> H], heads [H -> 2 zdim (32)], dec0 [zdim (16) -> H], dec1 [H -> H], dec2 [H -> K0]
    McMat e0f, e1f, hdf, d0f, d1f, d2f;
    // backward matrices: dec2^T [K0 -> H], dec1^T [H -> H], dec0^T [H -> zdim (16)], heads^T [2 zdim (32) -> H], enc1^T, enc0^T [H -> K0]
    McMat d2b, d1b, d0b, hdb, e1b, e0b;
    int act_e0, act_e1, act_d0, act_d1, act_d2;
    float *y_e0, *y_e1, *y_d0, *y_d1, *y_d2;       // saved outputs [batch][width]
    float *g_e0, *g_e1, *g_d0, *g_d1, *g_d2;       // backward: pre-activation gradients (for the grouped weight-gradient launch)
    const float *x0;                               // conv features [batch][K0]
    float *mu, *log_std, *sigma, *z;
    const float *eps;
    float *eps_out;
    RngStream rng;
    unsigned *amax_out;
    // backward only
    const float *g_out;
    int g_is_pre;
    const float *gate0;
    float *d_x0;
    const float *dz_reg, *dz_extra, *g_loss, *kl, *cap;
    float beta, inv_batch, reg_scale;
    float *d_mu, *d_ls;
    // ---- fold != 0 (round 5): the conv layer in front of the block (cv_e: Conv2d 8x8 -> 4x4, imagevae/dsprites_vae.py:19) and
    // the transposed one behind it (cv_d: ConvTranspose2d 4x4 -> 8x8, dsprites_vae.py:38) run inside these launches -- member m
    // of a cluster owns lo pixel (m / 4, m % 4) of its cluster's 32 images, i.e. columns [32 m, 32 m + 32) of the 512-wide
    // tensors, which is the slice it owns anyway.  Forward: x0 is WRITTEN (from hi_e), y_d2 goes on to hi_d (+ sign bits and
    // maxima: what the next conv kernel reads).  Backward: g_out is not read; the gradient arrives at hi_d's pre-activation
    // (g_hi_d), leaves at hi_e's (d_hi_e, gated by hi_e's saved output), and both layers' weight-gradient partials go to one
    // slab per workgroup (reduce.h, SLAB_C32).
    int fold;
    McConv cv_e, cv_d;
    const float *hi_e;             // [batch][8][8][32]: conv input (saved activation of the layer before)
    float *x0_out;                 // = x0, writable
    float *hi_d;                   // [batch][8][8][32]: the transposed conv's output (post-ReLU)
    unsigned char *hi_d_bits;      // relu_bits16 of hi_d (common.h), addressed by bytes
    unsigned *hi_d_amax;           // AMAX array of hi_d
    const float *g_hi_d;           // backward: gradient w.r.t. hi_d's pre-activation
    float *d_hi_e;                 // backward: gradient w.r.t. hi_e's pre-activation
    unsigned *d_hi_e_amax;
    float *slab_e, *slab_d;        // [grid][SLAB_C32_FLOATS]
};

int64_t midc_counter_words(int batch);             // uint32 words of arrival counters a batch needs (zeroed by the prep launch)
unsigned long long midc_wait_ticks();              // the default bound of a hand-off poll (10 ns ticks)
int midc_resident_capacity();                      // clustered-kernel workgroups the device holds at once (occupancy x CUs)
int midc_forward(const McArgs &a, hipStream_t s);
int midc_backward(const McArgs &a, hipStream_t s);

}  // namespace arvae
