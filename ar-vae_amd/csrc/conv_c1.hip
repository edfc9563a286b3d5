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
#include "common.h"

namespace arvae {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HI1 = 64, LO1 = 32, CC = 32;
constexpr int TR1 = 8;                      // lo rows per tile (x 32 cols = 256 lo positions)
constexpr int IPR = 2 * TR1 + 2, IPC = 68;  // image patch: 18 rows x 66 cols (+2 pad)
constexpr int PS1 = 36;                     // LDS pixel stride (floats) of 32-channel pixels

struct Ep1 {
    const float *bias;
    const float *gate;
    float *out;
    int relu;
};

// image rows [2*r0-1, 2*r0-1+IPR) x cols [-1, 65) of image n -> LDS; returns the sum of the pixels this
// tile owns (rows 1..16, cols 1..64 of the patch) for the transposed-conv bias gradient
__device__ __forceinline__ float load_img_patch(float *patch, const Operand &img, int n, int r0) {
    float own = 0.f;
    for (int idx = threadIdx.x; idx < IPR * 66; idx += 256) {
        const int pr = idx / 66, pc = idx - pr * 66;
        const int gy = 2 * r0 - 1 + pr, gx = pc - 1;
        float v = 0.f;
        if ((unsigned)gy < (unsigned)HI1 && (unsigned)gx < (unsigned)HI1) {
            v = img.at(((int64_t)n * HI1 + gy) * HI1 + gx);
            if (pr >= 1 && pr <= 2 * TR1) own += v;
        }
        patch[pr * IPC + pc] = v;
    }
    return own;
}

// ================================================================================================
__global__ __launch_bounds__(256) void down_c1_kernel(Operand img, const float *__restrict__ wt, Ep1 ep, int n_tiles) {
    __shared__ float patch[IPR * IPC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, rc = lane & 31;
    float w8[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) w8[s] = wt[rc * 16 + 2 * s + half];
    const float bias = ep.bias != nullptr ? ep.bias[rc] : 0.f;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int n = tile / (LO1 / TR1), r0 = (tile % (LO1 / TR1)) * TR1;
        __syncthreads();
        load_img_patch(patch, img, n, r0);
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int r = 2 * wave + mt;              // lo row inside the tile; lane rc = lo column
            f32x16 acc;
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
            for (int s = 0; s < 8; ++s) {             // k = 2s + half -> (ky, kx) = (s >> 1, 2 (s & 1) + half)
                const float a = patch[(2 * r + (s >> 1)) * IPC + 2 * rc + 2 * (s & 1) + half];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w8[s], acc, 0, 0, 0);
            }
            float gv[16];
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) gv[reg] = 1.f;
            if (ep.gate != nullptr) {
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const int ox = (reg & 3) + 8 * (reg >> 2) + 4 * half;
                    gv[reg] = ep.gate[(((n * LO1) + r0 + r) * LO1 + ox) * CC + rc];
                }
            }
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int ox = (reg & 3) + 8 * (reg >> 2) + 4 * half;
                const int idx = (((n * LO1) + r0 + r) * LO1 + ox) * CC + rc;
                float v = acc[reg] + bias;
                if (ep.relu) v = fmaxf(v, 0.f);
                ep.out[idx] = gv[reg] > 0.f ? v : 0.f;
            }
        }
    }
}

// ================================================================================================
// one lane per lo position (yy, xx): outputs (2yy+py, 2xx+px); tap (ty,tx) of class (py,px) reads lo
// (yy + py - ty, xx + px - tx) with weight wt[c][0][1 - py + 2 ty][1 - px + 2 tx]
__global__ __launch_bounds__(256) void up_c1_kernel(const float *__restrict__ lo, const float *__restrict__ wt,
                                                     const float *__restrict__ bias_p, float *__restrict__ out,
                                                     int n_tiles) {
    constexpr int PR = TR1 + 2, PC = LO1 + 2;
    extern __shared__ __attribute__((aligned(16))) float patch[];          // PR*PC pixels x PS1
    const int t = threadIdx.x;
    const int yy = t >> 5, xx = t & 31;
    const float bias = bias_p != nullptr ? bias_p[0] : 0.f;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int n = tile / (LO1 / TR1), r0 = (tile % (LO1 / TR1)) * TR1;
        __syncthreads();
        for (int idx = t; idx < PR * PC * 8; idx += 256) {
            const int q = idx & 7, pix = idx >> 3;
            const int pc = pix % PC, pr = pix / PC;
            const int gy = r0 - 1 + pr, gx = pc - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)gy < (unsigned)LO1 && (unsigned)gx < (unsigned)LO1)
                v = *reinterpret_cast<const float4 *>(lo + (((int64_t)n * LO1 + gy) * LO1 + gx) * CC + q * 4);
            *reinterpret_cast<float4 *>(patch + pix * PS1 + q * 4) = v;
        }
        __syncthreads();
        float acc[2][2] = {{bias, bias}, {bias, bias}};
        // channel chunk outermost and NOT unrolled: only its 4 x 16 weights are live in scalar registers
        // (all 512 at once spill through v_readlane)
#pragma unroll 1
        for (int q = 0; q < 8; ++q) {
            const float *wq = wt + q * 64;                                  // wt[(q*4 + e)*16 + tap]
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const float4 a = *reinterpret_cast<const float4 *>(patch + ((yy + 1 + dy) * PC + xx + 1 + dx) * PS1 + q * 4);
                    const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
                    for (int py = 0; py < 2; ++py) {
                        const int ty = py - dy;                       // compile-time after unrolling
                        if (ty < 0 || ty > 1) continue;
#pragma unroll
                        for (int px = 0; px < 2; ++px) {
                            const int tx = px - dx;
                            if (tx < 0 || tx > 1) continue;
                            const int tap = (1 - py + 2 * ty) * 4 + (1 - px + 2 * tx);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[py][px] = fmaf(av[e], wq[e * 16 + tap], acc[py][px]);
                        }
                    }
                }
        }
        const int64_t o = (((int64_t)n * HI1) + 2 * (r0 + yy)) * HI1 + 2 * xx;
        *reinterpret_cast<float2 *>(out + o) = make_float2(acc[0][0], acc[0][1]);
        *reinterpret_cast<float2 *>(out + o + HI1) = make_float2(acc[1][0], acc[1][1]);
    }
}

// ================================================================================================
// slab per workgroup: [32 clo][16 taps] + 32 lo sums + 1 image sum
constexpr int WG1_SLAB = CC * 16 + CC + 1;

__global__ __launch_bounds__(256) void wgrad_c1_kernel(Operand lo, Operand img, float *__restrict__ slab, int n_tiles) {
    __shared__ float patch[IPR * IPC];
    __shared__ __attribute__((aligned(16))) float lo_t[256 * PS1];     // reused as the reduce buffer (needs 4096 floats)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, rc = lane & 31;
    const int ky = (rc >> 2) & 3, kx = rc & 3;        // lanes with rc >= 16 carry zeros in the B operand
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float lo_sum = 0.f, img_sum = 0.f;
    // register-staged loads (issued for tile t+1 before the MFMAs of tile t)
    constexpr int IMG_ITERS = (IPR * 66 + 255) / 256;
    float4 lr[8];
    float ir[IMG_ITERS];
    auto issue = [&](int tile) {
        const int n = tile / (LO1 / TR1), r0 = (tile % (LO1 / TR1)) * TR1;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = threadIdx.x + it * 256;
            const int64_t g = (((int64_t)n * LO1 + r0) * LO1 + (idx >> 3)) * CC + (idx & 7) * 4;
            float4 v = *reinterpret_cast<const float4 *>(lo.v + g);
            if (lo.y != nullptr) {
                const float4 y = *reinterpret_cast<const float4 *>(lo.y + g);
                v.x *= act_bwd_from_out(y.x, lo.act); v.y *= act_bwd_from_out(y.y, lo.act);
                v.z *= act_bwd_from_out(y.z, lo.act); v.w *= act_bwd_from_out(y.w, lo.act);
            }
            lr[it] = v;
        }
#pragma unroll
        for (int it = 0; it < IMG_ITERS; ++it) {
            const int idx = threadIdx.x + it * 256;
            const int pr = idx / 66, pc = idx - pr * 66;
            const int gy = 2 * r0 - 1 + pr, gx = pc - 1;
            ir[it] = 0.f;
            if (idx < IPR * 66 && (unsigned)gy < (unsigned)HI1 && (unsigned)gx < (unsigned)HI1)
                ir[it] = img.at(((int64_t)n * HI1 + gy) * HI1 + gx);
        }
    };
    if (blockIdx.x < n_tiles) issue(blockIdx.x);
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int idx = threadIdx.x + it * 256;
            *reinterpret_cast<float4 *>(lo_t + (idx >> 3) * PS1 + (idx & 7) * 4) = lr[it];
        }
#pragma unroll
        for (int it = 0; it < IMG_ITERS; ++it) {
            const int idx = threadIdx.x + it * 256;
            if (idx < IPR * 66) {
                const int pr = idx / 66, pc = idx - pr * 66;
                patch[pr * IPC + pc] = ir[it];
                if (pr >= 1 && pr <= 2 * TR1 && pc >= 1 && pc <= HI1) img_sum += ir[it];
            }
        }
        __syncthreads();
        if (tile + gridDim.x < n_tiles) issue(tile + gridDim.x);
#pragma unroll
        for (int s = 0; s < 32; ++s) {                  // wave's 64 positions: rows 2w, 2w+1; pair (2s, 2s+1)
            const int r = 2 * wave + (s >> 4), c = (2 * s) & 31;
            const float a = lo_t[(r * 32 + c + half) * PS1 + rc];
            lo_sum += a;
            float b = patch[(2 * r + ky) * IPC + 2 * (c + half) + kx];
            b = rc < 16 ? b : 0.f;
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        }
    }
    // reduce the 4 waves' tiles: red[wave][reg][lane]
    __syncthreads();
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) lo_t[(wave * 16 + reg) * 64 + lane] = acc[reg];
    __syncthreads();
    float *out = slab + (int64_t)blockIdx.x * WG1_SLAB;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int reg = 4 * wave + e;
        const float tot = (lo_t[(0 * 16 + reg) * 64 + lane] + lo_t[(1 * 16 + reg) * 64 + lane]) +
                          (lo_t[(2 * 16 + reg) * 64 + lane] + lo_t[(3 * 16 + reg) * 64 + lane]);
        const int clo = (reg & 3) + 8 * (reg >> 2) + 4 * half;
        if (rc < 16) out[clo * 16 + rc] = tot;
    }
    // bias sums
    __syncthreads();
    lo_t[threadIdx.x] = lo_sum;
    lo_t[256 + threadIdx.x] = img_sum;
    __syncthreads();
    if (threadIdx.x < CC) {
        float tot = 0.f;
        for (int j = 0; j < 8; ++j) tot += lo_t[j * 32 + threadIdx.x];      // 4 waves x 2 halves
        out[CC * 16 + threadIdx.x] = tot;
    } else if (threadIdx.x == 64) {
        float tot = 0.f;
        for (int j = 0; j < 256; ++j) tot += lo_t[256 + j];
        out[CC * 16 + CC] = tot;
    }
}

__global__ __launch_bounds__(256) void wgrad_c1_reduce_kernel(const float *__restrict__ slab, int n_wg,
                                                               float *__restrict__ dwt, float *__restrict__ dbias,
                                                               int bias_mode) {
    __shared__ float red[16][17];
    const int il = threadIdx.x & 15, zg = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + il;
    float s = 0.f;
    if (i < WG1_SLAB)
        for (int z = zg; z < n_wg; z += 16) s += slab[(int64_t)z * WG1_SLAB + i];
    red[zg][il] = s;
    __syncthreads();
    if (zg == 0 && i < WG1_SLAB) {
        float tot = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) tot += red[j][il];
        if (i < CC * 16)
            dwt[i] += tot;                                   // wt[clo][0][ky][kx] is exactly [clo][tap]
        else if (i < CC * 16 + CC) {
            if (bias_mode == 1) dbias[i - CC * 16] += tot;
        } else if (bias_mode == 2)
            dbias[0] += tot;
    }
}

// ------------------------------------------------------------------------------------------------
bool conv_c1_fits(const arvae_link_t *l) {
    return l->chi == 1 && l->clo == CC && l->kh == 4 && l->kw == 4 && l->stride == 2 && l->pad == 1 && l->hh == HI1 &&
           l->hw == HI1 && l->lh == LO1 && l->lw == LO1 && l->hi_perm_c == 0 && l->lo_perm_c == 0;
}

int conv_c1_down(const arvae_link_t *l, const Operand &img, const float *wt, const float *bias, int relu,
                 const float *gate, float *out, hipStream_t s) {
    const int tiles = l->n * (LO1 / TR1);
    Ep1 ep{bias, gate, out, relu};
    hipLaunchKernelGGL(down_c1_kernel, dim3(tiles < 2048 ? tiles : 2048), dim3(256), 0, s, img, wt, ep, tiles);
    return check_launch("down_c1_kernel");
}

int conv_c1_up(const arvae_link_t *l, const float *lo, const float *wt, const float *bias, float *out, hipStream_t s) {
    const int tiles = l->n * (LO1 / TR1);
    constexpr int LDS = (TR1 + 2) * (LO1 + 2) * PS1 * 4;
    hipLaunchKernelGGL(up_c1_kernel, dim3(tiles < 768 ? tiles : 768), dim3(256), LDS, s, lo, wt, bias, out, tiles);
    return check_launch("up_c1_kernel");
}

static int wgrad_c1_groups(const arvae_link_t *l) {
    const int tiles = l->n * (LO1 / TR1);
    return tiles < 512 ? tiles : 512;
}

int64_t conv_c1_wgrad_ws_floats(const arvae_link_t *l) { return (int64_t)wgrad_c1_groups(l) * WG1_SLAB; }

int conv_c1_wgrad(const arvae_link_t *l, const Operand &lo, const Operand &img, float *dwt, float *dbias, int bias_mode,
                  float *slab, hipStream_t s) {
    const int tiles = l->n * (LO1 / TR1), grid = wgrad_c1_groups(l);
    hipLaunchKernelGGL(wgrad_c1_kernel, dim3(grid), dim3(256), 0, s, lo, img, slab, tiles);
    if (int rc = check_launch("wgrad_c1_kernel")) return rc;
    hipLaunchKernelGGL(wgrad_c1_reduce_kernel, dim3((WG1_SLAB + 15) / 16), dim3(256), 0, s, slab, grid, dwt, dbias,
                       bias_mode);
    return check_launch("wgrad_c1_reduce_kernel");
}

}  // namespace arvae
