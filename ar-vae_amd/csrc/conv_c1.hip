// The two links that touch the 1-channel 64x64 image in the dSprites stack (kernel 4x4, stride 2, pad 1,
// 32 channels on the 32x32 side): Conv2d(1->32) and ConvTranspose2d(32->1).  They are HBM-bound
// (7 FLOP/B): each 32x32x32 activation (128 KB/image) is read or written once.
//   down_c1  : lo[n,ly,lx,c]  = ep( sum_{ky,kx} img[n,2ly-1+ky,2lx-1+kx] * wt[c][0][ky][kx] )       K = 16
//              -> 8 fp32 MFMAs per 32 output pixels, image rows staged in LDS
//   up_c1    : img[n,hy,hx]   = bias + sum over the 2x2 valid taps and 32 channels of lo * wt        N = 1
//              -> vector-ALU dot products (an MFMA tile would be 31/32 empty): one lane per lo position
//                 producing its 2x2 output pixels from the 3x3 lo neighbourhood staged in LDS, weights
//                 from scalar registers
//   wgrad_c1 : dwt[c][0][ky][kx] += sum_pixels lo[pix][c] * img[pix @ tap]; bias sums ride along
#include <mutex>
#include "diag.h"
#include "common.h"
#include "reduce.h"
#include "prep32.h"
#include "wgrad_c1s.h"
#include "vae_finish.h"

namespace arvae {

constexpr int PS1 = 36;                     // LDS pixel stride (floats) of 32-channel pixels

struct Ep1 {
    const float *bias;
    const float *gate;          // saved activation of the output location (float), or
    const uint16_t *gate_bits;  // its sign bits (common.h: relu_bits16), or neither
    uint16_t *bits_out;         // with relu: sign bits of the result for a later gated kernel, may be null
    float *out;
    int relu;
    unsigned *amax_out = nullptr;   // AMAX array of `out` (conv32_common.h) for the 32-channel kernel that reads it next, or null
};

// ================================================================================================
// down_c1: the weight is the MFMA's A operand (row = channel) and the image value its B operand (column = lo pixel), so a lane
// ends up with 4 x 4 consecutive channels of ONE pixel and the ReLU sign bits of a pixel-half fit one uint16 (relu_bits16).
// Streaming form (a tiled, barrier-per-tile first generation was removed in round 3).  The kernel writes 128 KB per image and reads 16 KB: what it must do is keep the
// store path busy (tools/probes/store_probe.hip: a bare 67 MB fill takes 13.7 us on this chip; 16-byte pieces at a 128-byte
// stride, the MFMA accumulator layout stored as it is, 16 us; eight dword image loads per lane per row in the same
// memory queue as the stores, 19-21 us).  So:
//   * a WAVE owns one output row (32 pixels x 32 channels = 4 KB) at a time and shares nothing: no barrier;
//   * the reduction index is ordered k = (ky, kx) with ky = 2 (s >> 2) + half, kx = s & 3, so that a lane's eight B values
//     are the four consecutive pixels 2 rc - 1 .. 2 rc + 2 of two image rows: ONE aligned float2 load per image row (two per
//     output row instead of eight) and the two neighbours from the adjacent lanes by ds_bpermute;
//   * the finished row goes through a padded per-wave LDS tile and leaves as four 1 KB contiguous stores.
// GATE: 0 none, 1 sign bits, 2 saved float activation
// (body + wrappers: the pair kernel below runs it next to the weight gradient in one grid)
template <int GATE, int WAVES>
__device__ __forceinline__ void down_c1s_body(Operand img, const float *__restrict__ wt, Ep1 ep, int n_rows, const int BID,
                                              const int NBLK, float *stage) {
    const int lane = threadIdx.x & 63, half = lane >> 5, rc = lane & 31;
    float *tile = stage + (threadIdx.x >> 6) * (LO1 * PS1);
    float w8[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) w8[s] = wt[rc * 16 + 4 * (2 * (s >> 2) + half) + (s & 3)];
    float4 b4[4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
        b4[g] = ep.bias != nullptr ? *reinterpret_cast<const float4 *>(ep.bias + 8 * g + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float gs = img.scale != nullptr ? img.scale[0] : 1.f;
    const int wave0 = BID * WAVES + (threadIdx.x >> 6), n_waves = NBLK * WAVES;
    // image rows 2 r - 1 + half and 2 r + 1 + half, pixels 2 rc and 2 rc + 1
    auto fetch = [&](int row, float2 (&v)[2]) __attribute__((always_inline)) {
        const int n = row >> 5, r = row & 31;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int gy = 2 * r - 1 + 2 * q + half;
            const bool ok = row < n_rows && (unsigned)gy < (unsigned)HI1;
            const int64_t at = ok ? ((int64_t)n * HI1 + gy) * HI1 + 2 * rc : 0;        // unconditional load, clamped index
            const float2 xy = img.at2(at);
            v[q] = make_float2(ok ? gs * xy.x : 0.f, ok ? gs * xy.y : 0.f);
        }
    };
    float2 v[2], vn[2];
    float amax_run = 0.f;
    fetch(wave0, v);
    for (int row = wave0; row < n_rows; row += n_waves) {
        fetch(row + n_waves, vn);                               // next row's pixels fly during this row's MFMAs and stores
        const int pix = row * LO1 + rc;
        float4 gv[4];
        unsigned gb = 0xffffu;
        if (GATE == 1) gb = ep.gate_bits[pix * 2 + half];
        else if (GATE == 2) {
#pragma unroll
            for (int g = 0; g < 4; ++g) gv[g] = *reinterpret_cast<const float4 *>(ep.gate + (int64_t)pix * CC + 8 * g + 4 * half);
        }
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const float left = __shfl_up(v[q].y, 1, 32), right = __shfl_down(v[q].x, 1, 32);
            const float a4[4] = {rc == 0 ? 0.f : left, v[q].x, v[q].y, rc == 31 ? 0.f : right};
#pragma unroll
            for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w8[4 * q + t], a4[t], acc, 0, 0, 0);
        }
        unsigned bits = 0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float o[4] = {acc[4 * g] + b4[g].x, acc[4 * g + 1] + b4[g].y, acc[4 * g + 2] + b4[g].z, acc[4 * g + 3] + b4[g].w};
            const float gf[4] = {gv[g].x, gv[g].y, gv[g].z, gv[g].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (ep.relu) o[j] = fmaxf(o[j], 0.f);
                if (GATE == 1) o[j] = ((gb >> (4 * g + j)) & 1u) ? o[j] : 0.f;
                else if (GATE == 2) o[j] = gf[j] > 0.f ? o[j] : 0.f;
                bits |= (o[j] > 0.f ? 1u : 0u) << (4 * g + j);
            }
            const float4 ov = make_float4(o[0], o[1], o[2], o[3]);
            amax_run = fmaxf(amax_run, amax4(ov));
            *reinterpret_cast<float4 *>(tile + rc * PS1 + 8 * g + 4 * half) = ov;
        }
        if (ep.bits_out != nullptr) ep.bits_out[pix * 2 + half] = (uint16_t)bits;
        // the wave's LDS operations execute in order: no barrier, only keep the compiler from moving them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float4 *dst = reinterpret_cast<float4 *>(ep.out + (int64_t)row * LO1 * CC);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            dst[64 * i + lane] = *reinterpret_cast<const float4 *>(tile + (8 * i + (lane >> 3)) * PS1 + 4 * (lane & 7));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        v[0] = vn[0]; v[1] = vn[1];
    }
    if (ep.amax_out != nullptr) {                               // one writer unit per workgroup
        __shared__ float wmax[WAVES];
        amax_run = wave_max(amax_run);
        if (lane == 0) wmax[threadIdx.x >> 6] = amax_run;
        __syncthreads();
        if (threadIdx.x < 64) {
            float m = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES; ++w) m = fmaxf(m, wmax[w]);
            amax_publish(ep.amax_out, BID, NBLK, m);
        }
    }
}
template <int GATE>
__global__ __launch_bounds__(256) void down_c1s_kernel(Operand img, const float *__restrict__ wt, Ep1 ep, int n_rows) {
    __shared__ __attribute__((aligned(16))) float stage[4 * LO1 * PS1];
    down_c1s_body<GATE, 4>(img, wt, ep, n_rows, blockIdx.x, gridDim.x, stage);
}

// The first encoder layer and the step's weight preparation in ONE grid (horizontal pair, as pair_c1_kernel): the prep's few
// hundred short workgroups (prep32.h: nothing this layer reads) are dispatched first and finish under the convolution's
// 67 MB output stream; what they write is first read by the NEXT launch.  One launch and ~8 us less per training step.
__global__ __launch_bounds__(256) void down_c1s_prep_kernel(Operand img, const float *__restrict__ wt, Ep1 ep, int n_rows, PrepArgs prep,
                                                             MidPrepArgs mid, int conv_blocks, int prep_blocks) {
    __shared__ __attribute__((aligned(16))) float stage[4 * LO1 * PS1];
    if ((int)blockIdx.x < prep_blocks) {
        prep_all_block(prep, mid, conv_blocks, blockIdx.x);
        return;
    }
    down_c1s_body<0, 4>(img, wt, ep, n_rows, blockIdx.x - prep_blocks, gridDim.x - prep_blocks, stage);
}


// ================================================================================================
// up_c1: img[n,hy,hx] = bias + sum over the 2x2 valid taps and 32 channels of lo * wt, in two steps:
//   T[pos][tap] = sum_c lo[pos][c] * wt[c][tap]        a dense [positions x 32] x [32 x 16] product on the 16x16x4
//                                                        MFMA (all 16 columns useful, operands straight from HBM in
//                                                        the MFMA's own lane layout: no LDS staging of lo)
//   img(2y+py, 2x+px) = bias + sum_{ty,tx} T[(y+py-ty, x+px-tx)][(1-py+2ty)*4 + 1-px+2tx]        4 LDS reads / pixel
// A tile is 16 lo rows of one image (+1 halo row each side, zero outside the image) = 36 groups of 16 positions,
// 9 per wave; T is double-buffered in LDS so a tile needs one barrier.  With LOSS the reconstruction term of the
// trainer (sum of the per-pixel loss, correct-pixel count, d/dlogits) is computed on the pixels as they are
// produced: per-workgroup partial sums go to partial[2*blockIdx.x ..], the logits are still written.
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int TRU = 16;
constexpr int UG_PER_WAVE = (TRU + 2) * 2 / 4;          // 9
constexpr int TS1 = 17;                                 // floats per position in the LDS T tile
constexpr int T_FLOATS = (TRU + 2) * LO1 * TS1;

template <int DIST, bool LOSS>
__global__ __launch_bounds__(256, 2) void up_c1_kernel(const float *__restrict__ lo, const float *__restrict__ wt,
                                                        const float *__restrict__ bias_p, float *__restrict__ out,
                                                        const float *__restrict__ x, float inv_b,
                                                        float *__restrict__ partial, float *__restrict__ dlogits, int n_img,
                                                        int n_tiles, VaeFinishArgs fin = VaeFinishArgs{},
                                                        VaeFinishArgs *__restrict__ fin_dst = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float lds[];                // 2 x T_FLOATS
    __shared__ float red[4];
    // a training pass that leaves its finishing step to the backward pass (ARVAE_VAE_DEFER_FINISH, vae_finish.h): the step's
    // arguments are parked in the workspace for the launch that will run it, and until it has run the pass's eight scalars read
    // as NaN -- a caller that looks too early sees it
    if (fin_dst != nullptr && blockIdx.x == 0) {
        if (threadIdx.x == 0) *fin_dst = fin;
        if (threadIdx.x < 8 && fin.scalars != nullptr) fin.scalars[threadIdx.x] = __builtin_nanf("");
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
    float wreg[2][4];                                   // wt[c = 16s + 4g + j][tap = li]
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) wreg[s][j] = wt[(16 * s + 4 * g + j) * 16 + li];
    const float bias = bias_p != nullptr ? bias_p[0] : 0.f;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(lo), 0, n_img * LO1 * LO1 * CC * 4, 0x00020000);
    auto load4 = [&](unsigned off) {
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0));
        return v;
    };
    f32x4 v[UG_PER_WAVE][2];
    // group i of this wave in tile `tile`: T-region row prow, columns col0 .. col0+15; lane = (position li, channel quad g)
    auto issue = [&](int tile, int i) {
        const int gidx = wave + 4 * i, prow = gidx >> 1, col0 = (gidx & 1) * 16;
        const int n = tile >> 1, gy = (tile & 1) * TRU - 1 + prow;
        const bool ok = tile < n_tiles && (unsigned)gy < (unsigned)LO1;
        const unsigned off = ok ? (unsigned)((((n * LO1 + gy) * LO1 + col0 + li) * CC + g * 4) * 4) : 0x7fffffffu;
        v[i][0] = load4(off);
        v[i][1] = load4(off + 64);
    };
#pragma unroll
    for (int i = 0; i < UG_PER_WAVE; ++i) issue(blockIdx.x, i);

    float loss = 0.f, corr = 0.f;
    int buf = 0;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x, buf ^= 1) {
        float *T = lds + buf * T_FLOATS;
        const int n = tile >> 1, r0 = (tile & 1) * TRU;
        // the image pixels this thread will score: fetched now, used after the barrier
        float xv[8];
        if (LOSS) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                xv[i] = x[((int64_t)n * HI1 + 2 * r0 + (threadIdx.x >> 6) + 4 * i) * HI1 + (threadIdx.x & 63)];
        }
#pragma unroll
        for (int i = 0; i < UG_PER_WAVE; ++i) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(v[i][s][j], wreg[s][j], acc, 0, 0, 0);
            issue(tile + gridDim.x, i);                  // refill this group's registers for the next tile
            const int gidx = wave + 4 * i, prow = gidx >> 1, col0 = (gidx & 1) * 16;
#pragma unroll
            for (int r = 0; r < 4; ++r) T[(prow * LO1 + col0 + 4 * g + r) * TS1 + li] = acc[r];   // D row = 4g + r, col = li
        }
        __syncthreads();
        const int hx = threadIdx.x & 63, px = hx & 1, xq = hx >> 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int hy = (threadIdx.x >> 6) + 4 * i, py = hy & 1, y = hy >> 1;       // tile-relative hi row / lo row
            float sum = bias;
#pragma unroll
            for (int ty = 0; ty < 2; ++ty)
#pragma unroll
                for (int tx = 0; tx < 2; ++tx) {
                    const int col = xq + px - tx;
                    const bool ok = (unsigned)col < (unsigned)LO1;
                    const int tap = (1 - py + 2 * ty) * 4 + 1 - px + 2 * tx;
                    const float t = T[((y + py - ty + 1) * LO1 + (ok ? col : 0)) * TS1 + tap];
                    sum += ok ? t : 0.f;
                }
            const int64_t o = ((int64_t)n * HI1 + 2 * r0 + hy) * HI1 + hx;
            out[o] = sum;
            if (LOSS) {
                float d;
                recon_elem<DIST>(sum, xv[i], inv_b, loss, corr, d);
                if (dlogits != nullptr) dlogits[o] = d;
            }
        }
    }
    if (LOSS) {
        const float tl = block_sum_256(loss, red);
        const float tc = block_sum_256(corr, red);
        if (threadIdx.x == 0) {
            partial[2 * blockIdx.x] = tl;
            partial[2 * blockIdx.x + 1] = tc;
        }
    }
}

__global__ __launch_bounds__(64 * WGS_WAVES) void wgrad_c1s_kernel(Operand lo, Operand img, float *__restrict__ slab, int n_rows) {
    __shared__ __attribute__((aligned(16))) float lo_s[WGS_WAVES * LO1 * WS1];     // reused as the reduce buffer
    __shared__ __attribute__((aligned(16))) float im_s[WGS_WAVES * 4 * IMS];
    wgrad_c1s_body(lo, img, slab, n_rows, blockIdx.x, gridDim.x, lo_s, im_s);
}

// ---- the last decoder layer's backward in ONE grid: its data gradient (down_c1s: writes 67 MB) next to its weight gradient
// (wgrad_c1s: reads 75 MB).  Workgroups [0, grid_a) run the first body, the rest the second; both are 8-wave workgroups sharing
// one LDS allocation (58 KB: two per CU, so every CU holds one of each): a write stream and a read stream side by side move
// more bytes than either alone, and a launch ramp and a tail are saved.
template <int GATE>
__global__ __launch_bounds__(64 * WGS_WAVES) void pair_c1_kernel(Operand d_img, const float *__restrict__ wt, Ep1 ep, int n_rows_d,
                                                                 Operand w_lo, Operand w_img, float *__restrict__ slab, int n_rows_w, int grid_a,
                                                                 int rider, const VaeFinishArgs *__restrict__ fin) {
    __shared__ __attribute__((aligned(16))) float raw[WGS_WAVES * LO1 * WS1 + WGS_WAVES * 4 * IMS];
    // round 6: the forward pass's finishing step, left to this launch by a training step (vae_finish.h), as workgroup 0 of the
    // grid (rider = 1: the pair's own workgroups follow; the weight-gradient role gives up one workgroup so that the whole grid is
    // still resident at once -- dispatched LAST the rider waited for a slot until the pair was done: no gain).  Nothing in this
    // launch reads what it writes.
    if (rider && blockIdx.x == 0) {
        static_assert(64 * WGS_WAVES == 512, "the finishing body is instantiated for this kernel's 512 threads");
        const VaeFinishArgs args = *fin;                        // (parked in the workspace by the forward pass's last launch)
        vae_finish_body<512>(args, reinterpret_cast<float4 *>(raw));
        return;
    }
    const int bid = (int)blockIdx.x - rider, nblk = (int)gridDim.x - rider;
    if (bid < grid_a) down_c1s_body<GATE, WGS_WAVES>(d_img, wt, ep, n_rows_d, bid, grid_a, raw);
    else wgrad_c1s_body(w_lo, w_img, slab, n_rows_w, bid - grid_a, nblk - grid_a, raw, raw + WGS_WAVES * LO1 * WS1);
}
static_assert(WGS_WAVES * LO1 * PS1 <= WGS_WAVES * LO1 * WS1, "pair_c1_kernel: the down body's row tiles fit the shared allocation");

static_assert(WGS_WAVES == 8, "the reduce step gives one of the 8 accumulator registers to each wave");

// ================================================================================================
// wgrad_c1w: the same per-wave streaming weight gradient for the STRIDE-1 single-channel links with 64 channels on the other
// side (Morpho-MNIST Conv2d(1, 64, 4) and ConvTranspose2d(64, 1, 4), imagevae/mnist_vae.py:16-47):
//     dwt[c][ky][kx] += sum over (n, r, col) of lo[n][r][col][c] * img[n][r + ky][col + kx]
// The generic gather-GEMM took 104 us + a 17 us slice reduce + 33 us of bias sums per launch for 164 MB of operands.  A wave
// owns one lo row (lw <= 28 positions x 64 channels) and the four image rows it meets; 16x16x4 fp32 MFMAs with N = the 16 taps
// and four channel tiles; positions beyond lw are zero slots of the row buffer.  Bias sums ride along (slab [64][16] + 64 + 1).
constexpr int W1_CH = 64, W1_SLOTS = 28;        // channels; position slots per row (seven quads)
constexpr int W1_PS = 80;                       // LDS position stride: the A reads of the four position slots (g * 80 + li) fall on
                                                // four disjoint bank groups
constexpr int W1_IMS = 32;                      // LDS pitch of an image row (hw <= 31)
constexpr int W1_WAVES = 8;
constexpr int W1_LO_LOADS = (W1_SLOTS * W1_CH / 4 + 63) / 64;       // 16-byte loads per lane and lo row (7)

__global__ __launch_bounds__(64 * W1_WAVES) void wgrad_c1w_kernel(Operand lo, Operand img, float *__restrict__ slab, int n_img, int lh,
                                                                 int lw, int hw) {
    __shared__ __attribute__((aligned(16))) float lo_s[W1_WAVES][W1_SLOTS * W1_PS];     // reused as the reduce buffer
    __shared__ float im_s[W1_WAVES][4 * W1_IMS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4, li = lane & 15;
    const int ky = li >> 2, kx = li & 3;
    float *lw_s = lo_s[wave], *iw = im_s[wave];
    for (int i = lane; i < W1_SLOTS * W1_PS; i += 64) lw_s[i] = 0.f;          // slots >= lw stay zero for the whole launch
    for (int i = lane; i < 4 * W1_IMS; i += 64) iw[i] = 0.f;
    f32x4 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float lo_sum[4] = {0.f, 0.f, 0.f, 0.f}, img_sum = 0.f;
    const float gs = img.scale != nullptr ? img.scale[0] : 1.f;
    const int n_rows = n_img * lh, hh = lh + 3;
    const int wave0 = blockIdx.x * W1_WAVES + wave, n_waves = gridDim.x * W1_WAVES;
    const int row_f4 = lw * (W1_CH / 4);                                     // 16-byte pieces of a lo row
    float4 lr[W1_LO_LOADS];
    float ir[2];
    auto issue = [&](int row) __attribute__((always_inline)) {
        const bool in = row < n_rows;
        const int n = row / lh, r = row - n * lh;
#pragma unroll
        for (int i = 0; i < W1_LO_LOADS; ++i) {
            const int e = 64 * i + lane;
            const bool ok = in && e < row_f4;
            const float4 v = lo.at4(ok ? (int64_t)row * lw * W1_CH + 4 * e : 0);
            lr[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {                            // image rows r .. r + 3, hw pixels each
            const int e = 64 * i + lane, pr = e / hw, px = e - pr * hw;
            const bool ok = in && pr < 4;
            const float v = img.at(ok ? ((int64_t)n * hh + r + pr) * hw + px : 0);
            ir[i] = ok ? gs * v : 0.f;
        }
    };
    issue(wave0);
    for (int row = wave0; row < n_rows; row += n_waves) {
        const int r = row % lh;
#pragma unroll
        for (int i = 0; i < W1_LO_LOADS; ++i) {
            const int e = 64 * i + lane;
            if (e < row_f4) *reinterpret_cast<float4 *>(lw_s + (e >> 4) * W1_PS + 4 * (e & 15)) = lr[i];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int e = 64 * i + lane, pr = e / hw, px = e - pr * hw;
            if (pr < 4) {
                iw[pr * W1_IMS + px] = ir[i];
                if (pr == 0 || r == lh - 1) img_sum += ir[i];    // every image pixel once: row r, and the last lo row's three extra rows
            }
        }
        // the wave's LDS operations execute in order: no barrier, only keep the compiler from moving them
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        issue(row + n_waves);
#pragma unroll
        for (int q = 0; q < W1_SLOTS / 4; ++q) {                 // positions 4 q .. 4 q + 3, this lane's slot: c = 4 q + g
            const int c = 4 * q + g;
            const float b = iw[ky * W1_IMS + c + kx];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const float a = lw_s[c * W1_PS + 16 * ct + li];
                lo_sum[ct] += a;
                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[ct], 0, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    // reduce the waves' tiles: red[wave][ct*4 + r][lane]; D row = 4g + r -> channel 16 ct + 4g + r, column = tap li
    float *red = &lo_s[0][0];
    __syncthreads();
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(wave * 16 + ct * 4 + r) * 64 + lane] = acc[ct][r];
    __syncthreads();
    float *out = slab + (int64_t)blockIdx.x * SLAB_C1W_FLOATS;
#pragma unroll
    for (int e = 0; e < 2; ++e) {                                // wave w finishes registers 2w, 2w + 1 of the 16
        const int reg = 2 * wave + e, ct = reg >> 2, r = reg & 3;
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < W1_WAVES; ++w) tot += red[(w * 16 + reg) * 64 + lane];
        out[(16 * ct + 4 * g + r) * 16 + li] = tot;
    }
    // bias sums: lo per channel (lane li of tile ct, over the 4 position slots and the waves), image total
    __syncthreads();
    constexpr int T = 64 * W1_WAVES;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) red[ct * T + threadIdx.x] = lo_sum[ct];
    red[4 * T + threadIdx.x] = img_sum;
    __syncthreads();
    if (threadIdx.x < W1_CH) {
        const int ct = threadIdx.x >> 4, i = threadIdx.x & 15;
        float tot = 0.f;
        for (int j = 0; j < T / 16; ++j) tot += red[ct * T + j * 16 + i];
        out[W1_CH * 16 + threadIdx.x] = tot;
    } else if (threadIdx.x < 128) {
        float tot = 0.f;
        for (int j = lane; j < T; j += 64) tot += red[4 * T + j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
        if (lane == 0) out[W1_CH * 16 + W1_CH] = tot;
    }
}
static_assert(W1_WAVES == 8 && W1_WAVES * W1_SLOTS * W1_PS >= W1_WAVES * 16 * 64, "wgrad_c1w_kernel: the row buffers hold the reduce tiles");

// ------------------------------------------------------------------------------------------------
bool conv_c1_fits(const arvae_link_t *l) {
    return l->chi == 1 && l->clo == CC && l->kh == 4 && l->kw == 4 && l->stride == 2 && l->pad == 1 && l->hh == HI1 &&
           l->hw == HI1 && l->lh == LO1 && l->lw == LO1 && l->hi_perm_c == 0 && l->lo_perm_c == 0;
}

// conv_c1_down (plain input, no gate) with the step's weight preparation riding in the same grid (down_c1s_prep_kernel)
int conv_c1_down_with_prep(const arvae_link_t *l, const Operand &img, const float *wt, const float *bias, int relu, uint16_t *bits_out,
                           float *out, const float *const *prep_wts, float *const *preps, int n_prep, const MidPrepArgs &mid,
                           hipStream_t s, unsigned *amax_out) {
    ARVAE_REQUIRE(n_prep > 0 && n_prep <= PREP_MAX_LAYERS && mid.count > 0, "conv_c1_down_with_prep: nothing to prepare");
    Ep1 ep{bias, nullptr, nullptr, bits_out, out, relu, amax_out};
    PrepArgs p{};
    for (int i = 0; i < n_prep; ++i) {
        p.wt[i] = prep_wts[i];
        p.out[i] = reinterpret_cast<uint4 *>(preps[i]);
    }
    const int conv_blocks = 16 * n_prep, prep_blocks = conv_blocks + mid_prep_blocks(mid);
    const int n_rows = l->n * LO1;
    int grid = 256 * 16 / 4;                                     // as conv_c1_down: workgroups of four independent waves
    if (grid > (n_rows + 3) / 4) grid = (n_rows + 3) / 4;
    if (grid > AMAX_N) grid = AMAX_N;                            // one AMAX writer unit per workgroup
    ARVAE_LAUNCH(down_c1s_prep_kernel, dim3(prep_blocks + grid), dim3(256), 0, s, img, wt, ep, n_rows, p, mid, conv_blocks, prep_blocks);
    return check_launch("down_c1_kernel(+ weight prep)");
}

int conv_c1_down(const arvae_link_t *l, const Operand &img, const float *wt, const float *bias, int relu,
                 const float *gate, const uint16_t *gate_bits, uint16_t *bits_out, float *out, hipStream_t s, unsigned *amax_out) {
    Ep1 ep{bias, gate, gate_bits, bits_out, out, relu, amax_out};
    static const int waves_per_cu = diag_env("ARVAE_C1_DOWN_WAVES") ? atoi(diag_env("ARVAE_C1_DOWN_WAVES")) : 16;
    const int n_rows = l->n * LO1;
    int grid = 256 * waves_per_cu / 4;                           // workgroups of four independent waves
    if (grid > (n_rows + 3) / 4) grid = (n_rows + 3) / 4;
    if (amax_out != nullptr && grid > AMAX_N) grid = AMAX_N;     // one AMAX writer unit per workgroup
    if (gate_bits != nullptr) ARVAE_LAUNCH(down_c1s_kernel<1>, dim3(grid), dim3(256), 0, s, img, wt, ep, n_rows);
    else if (gate != nullptr) ARVAE_LAUNCH(down_c1s_kernel<2>, dim3(grid), dim3(256), 0, s, img, wt, ep, n_rows);
    else ARVAE_LAUNCH(down_c1s_kernel<0>, dim3(grid), dim3(256), 0, s, img, wt, ep, n_rows);
    return check_launch("down_c1_kernel");
}

static int up_c1_grid(int tiles) {
    int dev = 0, cus = 256;
    hipDeviceProp_t prop;
    static int cached = 0;
    if (cached == 0) {
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        cached = cus > 0 ? cus : 256;
    }
    static const int cap = diag_env("ARVAE_C1_UP_GRID") ? atoi(diag_env("ARVAE_C1_UP_GRID")) : 0;
    if (cap > 0) return tiles < cap ? tiles : cap;
    return tiles < 2 * cached ? tiles : 2 * cached;
}

template <class K> static void up_c1_lds(K kernel) {
    (void)hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * T_FLOATS * 4);
}

int conv_c1_up(const arvae_link_t *l, const float *lo, const float *wt, const float *bias, float *out, hipStream_t s) {
    const int tiles = l->n * (LO1 / TRU);
    static std::once_flag attr;
    std::call_once(attr, [&] { up_c1_lds(up_c1_kernel<ARVAE_RECON_BERNOULLI, false>); });
    ARVAE_LAUNCH((up_c1_kernel<ARVAE_RECON_BERNOULLI, false>), dim3(up_c1_grid(tiles)), dim3(256), 2 * T_FLOATS * 4, s, lo,
                       wt, bias, out, nullptr, 0.f, nullptr, nullptr, l->n, tiles);
    return check_launch("up_c1_kernel");
}

// the same link with the trainer's reconstruction term fused in: logits -> out, per-workgroup (loss, correct) partial
// sums -> partial[2 * nb], d loss / d logits -> dlogits (may be null); *nb_out = number of partial pairs
// workgroups (= reconstruction partial pairs) of conv_c1_up_recon for this batch
int conv_c1_up_recon_blocks(const arvae_link_t *l) { return up_c1_grid(l->n * (LO1 / TRU)); }

int conv_c1_up_recon(const arvae_link_t *l, const float *lo, const float *wt, const float *bias, float *out, const float *x,
                     int dist, float *partial, float *dlogits, hipStream_t s, int *nb_out, const VaeFinishArgs *fin, VaeFinishArgs *fin_dst) {
    const int tiles = l->n * (LO1 / TRU), grid = up_c1_grid(tiles);
    const float inv_b = 1.f / (float)l->n;
    const VaeFinishArgs fa = fin != nullptr ? *fin : VaeFinishArgs{};
    if (fin == nullptr) fin_dst = nullptr;
    static std::once_flag attr;
    std::call_once(attr, [&] {
        up_c1_lds(up_c1_kernel<ARVAE_RECON_BERNOULLI, true>);
        up_c1_lds(up_c1_kernel<ARVAE_RECON_GAUSSIAN, true>);
    });
    if (dist == ARVAE_RECON_BERNOULLI)
        ARVAE_LAUNCH((up_c1_kernel<ARVAE_RECON_BERNOULLI, true>), dim3(grid), dim3(256), 2 * T_FLOATS * 4, s, lo, wt, bias,
                           out, x, inv_b, partial, dlogits, l->n, tiles, fa, fin_dst);
    else
        ARVAE_LAUNCH((up_c1_kernel<ARVAE_RECON_GAUSSIAN, true>), dim3(grid), dim3(256), 2 * T_FLOATS * 4, s, lo, wt, bias,
                           out, x, inv_b, partial, dlogits, l->n, tiles, fa, fin_dst);
    *nb_out = grid;
    return check_launch("up_c1_kernel(recon)");
}

int wgrad_c1_groups(const arvae_link_t *l) {
    static const int cap = diag_env("ARVAE_C1_WGRAD_GRID") ? atoi(diag_env("ARVAE_C1_WGRAD_GRID")) : 256;       // streaming form: one 8-wave workgroup per CU (256 / 512 / 1024 measured: 19.3 / 19.0 / 21.8 us, and the slab reduce grows with it)
    const int units = (l->n * LO1 + WGS_WAVES - 1) / WGS_WAVES;
    return units < cap ? units : cap;
}

int64_t conv_c1_wgrad_ws_floats(const arvae_link_t *l) { return (int64_t)wgrad_c1_groups(l) * WG1_SLAB; }

int conv_c1_wgrad_partial(const arvae_link_t *l, const Operand &lo, const Operand &img, float *dwt, float *dbias,
                          int bias_mode, float *slab, hipStream_t s, SlabJob *job) {
    const int grid = wgrad_c1_groups(l);
    ARVAE_LAUNCH(wgrad_c1s_kernel, dim3(grid), dim3(64 * WGS_WAVES), 0, s, lo, img, slab, l->n * LO1);
    *job = SlabJob{slab, dwt, dbias, grid, SLAB_C1, bias_mode};
    return check_launch("wgrad_c1_kernel");
}

// the deferred finishing step by itself (a backward pass whose first launch is not the paired one above)
__global__ __launch_bounds__(512) void vae_finish_deferred_kernel(const VaeFinishArgs *__restrict__ fin) {
    __shared__ float4 red4[8];
    const VaeFinishArgs args = *fin;
    vae_finish_body<512>(args, red4);
}
int vae_finish_deferred(const VaeFinishArgs *fin_dev, hipStream_t s) {
    ARVAE_LAUNCH(vae_finish_deferred_kernel, dim3(1), dim3(512), 0, s, fin_dev);
    return check_launch("image_vae_backward(finish)");
}

// gated data gradient of the forward-UP single-channel link + its weight-gradient partials in one launch (pair_c1_kernel)
bool conv_c1_pair_fits(const arvae_link_t *l) {
    static const bool off = diag_env("ARVAE_NO_PAIR_C1") != nullptr;
    return !off && conv_c1_fits(l) && l->n * LO1 >= 8 * 256 && wgrad_c1_groups(l) == 256;
}
int conv_c1_pair(const arvae_link_t *l, const Operand &g_img, const float *wt, const float *gate, const uint16_t *gate_bits, float *d_lo,
                 const Operand &w_lo, float *dwt, float *dbias, int bias_mode, float *slab, hipStream_t s, SlabJob *job,
                 unsigned *amax_out, const VaeFinishArgs *finish) {
    Ep1 ep{nullptr, gate, gate_bits, nullptr, d_lo, 0, amax_out};
    const int rider = finish != nullptr ? 1 : 0;                // (+ the deferred finishing step's workgroup; finish: DEVICE pointer)
    const int n_rows = l->n * LO1, grid_a = 256, grid_b = wgrad_c1_groups(l) - rider;
    const dim3 grid(rider + grid_a + grid_b);
    if (gate_bits != nullptr) ARVAE_LAUNCH(pair_c1_kernel<1>, grid, dim3(64 * WGS_WAVES), 0, s, g_img, wt, ep, n_rows, w_lo, g_img, slab, n_rows, grid_a, rider, finish);
    else if (gate != nullptr) ARVAE_LAUNCH(pair_c1_kernel<2>, grid, dim3(64 * WGS_WAVES), 0, s, g_img, wt, ep, n_rows, w_lo, g_img, slab, n_rows, grid_a, rider, finish);
    else ARVAE_LAUNCH(pair_c1_kernel<0>, grid, dim3(64 * WGS_WAVES), 0, s, g_img, wt, ep, n_rows, w_lo, g_img, slab, n_rows, grid_a, rider, finish);
    *job = SlabJob{slab, dwt, dbias, grid_b, SLAB_C1, bias_mode};
    return check_launch("pair_c1(down_c1 + wgrad_c1)");
}

int conv_c1_wgrad(const arvae_link_t *l, const Operand &lo, const Operand &img, float *dwt, float *dbias, int bias_mode,
                  float *slab, hipStream_t s) {
    SlabJob job;
    if (int rc = conv_c1_wgrad_partial(l, lo, img, dwt, dbias, bias_mode, slab, s, &job)) return rc;
    return slab_reduce(job, s);
}

// ---- the stride-1 64-channel single-channel links (Morpho-MNIST) ----------------------------------------------------
bool conv_c1w_fits(const arvae_link_t *l) {
    static const bool off = diag_env("ARVAE_C1W_GENERIC") != nullptr;            // diagnostic: the generic gather-GEMM instead
    return !off && l->chi == 1 && l->clo == W1_CH && l->kh == 4 && l->kw == 4 && l->stride == 1 && l->pad == 0 && l->lw <= W1_SLOTS &&
           l->hw == l->lw + 3 && l->hh == l->lh + 3 && l->hw < W1_IMS && 4 * l->hw <= 128 && l->hi_perm_c == 0 && l->lo_perm_c == 0;
}
static int wgrad_c1w_groups(const arvae_link_t *l) {
    const int units = (l->n * l->lh + W1_WAVES - 1) / W1_WAVES;
    return units < 256 ? units : 256;
}
int64_t conv_c1w_wgrad_ws_floats(const arvae_link_t *l) { return (int64_t)wgrad_c1w_groups(l) * SLAB_C1W_FLOATS; }

int conv_c1w_wgrad(const arvae_link_t *l, const Operand &lo, const Operand &img, float *dwt, float *dbias, int bias_mode,
                   float *slab, hipStream_t s) {
    const int grid = wgrad_c1w_groups(l);
    ARVAE_LAUNCH(wgrad_c1w_kernel, dim3(grid), dim3(64 * W1_WAVES), 0, s, lo, img, slab, l->n, l->lh, l->lw, l->hw);
    if (int rc = check_launch("wgrad_c1w_kernel")) return rc;
    return slab_reduce(SlabJob{slab, dwt, dbias, grid, SLAB_C1W, bias_mode}, s);
}

}  // namespace arvae
