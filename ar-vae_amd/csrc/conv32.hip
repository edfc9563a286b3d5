// Specialised gfx950 kernels for the dSprites stack's 32-channel links (kernel 4x4, stride 2, pad 1):
// the six Conv2d / ConvTranspose2d layers between the 32x32, 16x16, 8x8 and 4x4 feature maps, which
// carry 85 % of the training step's FLOPs.  Same math and C-ABI as the
// generic gather-GEMM (link_gemm.hip); arvae_link_down/up/wgrad dispatch here when the geometry fits.
//
// Common structure (one workgroup = 4 wavefronts of 64 lanes, persistent over tiles):
//   * the activation patch a tile needs (with its halo, zero-filled outside the image) is staged
//     global -> LDS once with coalesced 16-byte loads; pixel stride in LDS is 36 floats (144 B), which
//     keeps ds_read_b128 of stride-1 / stride-2 pixel walks at <= 2-way bank conflicts;
//   * the 32x32 weight slices a wave needs live in 64 VGPRs for the whole kernel, in the K-order
//     the MFMA wants: v_mfma_f32_32x32x2_f32 lane (row|col = lane&31, half = lane>>5) supplies
//     k = 2*s + half, and we choose K = (tap, 8-channel chunk, half, t) so that ONE ds_read_b128 of
//     4 consecutive channels feeds 4 MFMAs;
//   * Down : wave w owns kernel row ky = w (K split 4 ways), partial 32x32 tiles are summed through LDS;
//     Up   : wave w owns one of the 4 stride-parity classes of output pixels (no reduction needed);
//     Wgrad: wave w owns kernel row ky = w, 4 accumulator tiles (kx) per wave, pixels are the K axis;
//            per-workgroup partial sums go to a slab that wgrad32_reduce_kernel adds up in a fixed order.
#include "common.h"

namespace arvae {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int C32 = 32;
constexpr int PS = 36;   // LDS pixel stride in floats

// tile geometry per lo-resolution size LO (hi = 2*LO): 64 lo pixels = two 32-row MFMA tiles
template <int LO> struct Tile;
template <> struct Tile<16> { static constexpr int TI = 1, TR = 4, TC = 16; };   // 4 rows x 16 cols of one image
template <> struct Tile<8>  { static constexpr int TI = 1, TR = 8, TC = 8;  };   // one whole 8x8 image
template <> struct Tile<4>  { static constexpr int TI = 4, TR = 4, TC = 4;  };   // four whole 4x4 images

// lo pixel p (0..63) of a tile -> (image, row, col) inside the tile
template <int LO> __device__ __forceinline__ void tile_pixel(int p, int &img, int &r, int &c) {
    using T = Tile<LO>;
    c = p % T::TC;
    r = (p / T::TC) % T::TR;
    img = p / (T::TC * T::TR);
}

struct Ep32 {
    const float *bias;   // per output channel or null
    const float *gate;   // null, or saved activation of the OUTPUT location: result *= (gate > 0)
    float *out;
    int relu;            // apply ReLU after bias
};

__device__ __forceinline__ float ep_apply(const Ep32 &ep, float acc, float bias, int idx) {
    float v = acc + bias;
    if (ep.relu) v = fmaxf(v, 0.f);
    if (ep.gate != nullptr) v = ep.gate[idx] > 0.f ? v : 0.f;
    return v;
}

__device__ __forceinline__ float4 operand_load4(const Operand &op, int64_t idx) {
    float4 v = *reinterpret_cast<const float4 *>(op.v + idx);
    if (op.y != nullptr) {
        const float4 y = *reinterpret_cast<const float4 *>(op.y + idx);
        v.x *= act_bwd_from_out(y.x, op.act);
        v.y *= act_bwd_from_out(y.y, op.act);
        v.z *= act_bwd_from_out(y.z, op.act);
        v.w *= act_bwd_from_out(y.w, op.act);
    }
    return v;
}

// ------------------------------------------------------------------------------------------------
// Register-staged patch loader.  issue() starts every 16-byte global load of a tile's patch (all of them
// in flight at once; called for tile t+1 before the MFMAs of tile t so HBM/L2 latency hides under them),
// commit() writes the registers to LDS once the previous tile's readers have passed the barrier.
//   STRIDE 2: hi patch of a lo tile: rows [2*r0-1, 2*r0-1+PR), cols [-1, PC-1), PR = 2*TR+2, PC = 2*TC+2
//   STRIDE 1: lo patch with a 1-pixel halo: rows [r0-1, r0-1+PR), cols [-1, PC-1), PR = TR+2, PC = TC+2
// ------------------------------------------------------------------------------------------------
template <int LO, int STRIDE>
struct PatchLoader {
    using T = Tile<LO>;
    static constexpr int SZ = STRIDE * LO;                                  // spatial size of the source tensor
    static constexpr int PR = STRIDE * T::TR + 2, PC = STRIDE * T::TC + 2;
    static constexpr int SLOTS = T::TI * PR * PC * 8;
    static constexpr int ITERS = (SLOTS + 255) / 256;
    float4 r[ITERS];

    __device__ __forceinline__ void issue(const Operand &src, int img0, int r0, int n_img) {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int idx = threadIdx.x + it * 256;
            const int q = idx & 7, pix = idx >> 3;
            const int pc = pix % PC, pr = (pix / PC) % PR, im = pix / (PC * PR);
            const int gy = STRIDE * r0 - 1 + pr, gx = pc - 1, n = img0 + im;
            r[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < SLOTS && n < n_img && (unsigned)gy < (unsigned)SZ && (unsigned)gx < (unsigned)SZ)
                r[it] = operand_load4(src, ((int64_t)(n * SZ + gy) * SZ + gx) * C32 + q * 4);
        }
    }
    // BIAS_SUM: also accumulate the pixels this tile owns (not the halo) per channel chunk q = threadIdx.x & 7
    template <bool BIAS_SUM>
    __device__ __forceinline__ void commit(float *patch, float4 &bsum) const {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int idx = threadIdx.x + it * 256;
            if (idx < SLOTS) {
                const int q = idx & 7, pix = idx >> 3;
                *reinterpret_cast<float4 *>(patch + pix * PS + q * 4) = r[it];
                if (BIAS_SUM) {
                    const int pc = pix % PC, pr = (pix / PC) % PR;
                    if (pr >= 1 && pr <= PR - 2 && pc >= 1 && pc <= PC - 2) {
                        bsum.x += r[it].x; bsum.y += r[it].y; bsum.z += r[it].z; bsum.w += r[it].w;
                    }
                }
            }
        }
    }
};

template <int LO> __device__ __forceinline__ void tile_origin(int tile, int &img0, int &r0) {
    using T = Tile<LO>;
    constexpr int TILES_PER_IMG = LO / T::TR;
    img0 = (T::TI == 1) ? tile / TILES_PER_IMG : tile * T::TI;
    r0 = (T::TI == 1) ? (tile % TILES_PER_IMG) * T::TR : 0;
}

// ================================================================================================
// Down: lo[n,ly,lx,clo] = ep( sum_{ky,kx,chi} hi[n,2ly-1+ky,2lx-1+kx,chi] * wt[clo][chi][ky][kx] )
// ================================================================================================
template <int LO>
__global__ __launch_bounds__(256, 2) void down32_kernel(Operand hi, const float *__restrict__ wt, Ep32 ep, int n_img,
                                                         int n_tiles) {
    using T = Tile<LO>;
    constexpr int PR = 2 * T::TR + 2, PC = 2 * T::TC + 2;
    // LDS: max(patch, 8192) floats -- the patch, then reused as the reduce buffer [wave][mtile][reg/4][lane][4]
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, rc = lane & 31;

    // weights of kernel row ky = wave: w[kx][chunk][t] = wt[clo=rc][chi = chunk*8 + half*4 + t][ky][kx]
    float w[4][4][4];
#pragma unroll
    for (int kx = 0; kx < 4; ++kx)
#pragma unroll
        for (int ch = 0; ch < 4; ++ch)
#pragma unroll
            for (int t = 0; t < 4; ++t) w[kx][ch][t] = wt[((rc * C32) + ch * 8 + half * 4 + t) * 16 + wave * 4 + kx];

    // lane's two output pixels (one per MFMA tile) -> patch offsets
    int aoff[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        int img, r, c;
        tile_pixel<LO>(mt * 32 + rc, img, r, c);
        aoff[mt] = ((img * PR + 2 * r + wave) * PC + 2 * c) * PS + half * 4;
    }
    const float bias = ep.bias != nullptr ? ep.bias[rc] : 0.f;
    float4 dummy;
    PatchLoader<LO, 2> pl;
    int img0, r0;
    if (blockIdx.x < n_tiles) {
        tile_origin<LO>(blockIdx.x, img0, r0);
        pl.issue(hi, img0, r0, n_img);
    }

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        tile_origin<LO>(tile, img0, r0);
        __syncthreads();                                         // previous tile's reduce reads are done
        pl.template commit<false>(lds, dummy);
        __syncthreads();
        if (tile + gridDim.x < n_tiles) {                        // next tile's loads fly under this tile's MFMAs
            int ni, nr;
            tile_origin<LO>(tile + gridDim.x, ni, nr);
            pl.issue(hi, ni, nr, n_img);
        }

        f32x16 acc[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
#pragma unroll
        for (int kx = 0; kx < 4; ++kx)
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) {
                const float4 a0 = *reinterpret_cast<const float4 *>(lds + aoff[0] + kx * PS + ch * 8);
                const float4 a1 = *reinterpret_cast<const float4 *>(lds + aoff[1] + kx * PS + ch * 8);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, w[kx][ch][0], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, w[kx][ch][0], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, w[kx][ch][1], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, w[kx][ch][1], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, w[kx][ch][2], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, w[kx][ch][2], acc[1], 0, 0, 0);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, w[kx][ch][3], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, w[kx][ch][3], acc[1], 0, 0, 0);
            }

        // sum the four ky-partials through LDS (patch memory is reused)
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<float4 *>(lds + (((wave * 2 + mt) * 4 + q) * 64 + lane) * 4) =
                    make_float4(acc[mt][4 * q], acc[mt][4 * q + 1], acc[mt][4 * q + 2], acc[mt][4 * q + 3]);
        __syncthreads();
        // wave w finishes registers [8*(w&1), +8) of MFMA tile (w>>1); gate loads first, then the stores
        const int mt = wave >> 1, q0 = (wave & 1) * 2;
        int oidx[8];
        float gv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int reg = 4 * q0 + j;
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * half;      // pixel inside the MFMA tile
            int img, r, c;
            tile_pixel<LO>(mt * 32 + row, img, r, c);
            const int n = img0 + img;
            oidx[j] = n < n_img ? ((n * LO + r0 + r) * LO + c) * C32 + rc : -1;
            gv[j] = 1.f;
        }
        if (ep.gate != nullptr) {
#pragma unroll
            for (int j = 0; j < 8; ++j) gv[j] = ep.gate[oidx[j] < 0 ? 0 : oidx[j]];
        }
#pragma unroll
        for (int q = q0; q < q0 + 2; ++q) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int ws = 0; ws < 4; ++ws) {
                const float4 v = *reinterpret_cast<const float4 *>(lds + (((ws * 2 + mt) * 4 + q) * 64 + lane) * 4);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = 4 * (q - q0) + e;
                if (oidx[j] >= 0) {
                    float v = sv[e] + bias;
                    if (ep.relu) v = fmaxf(v, 0.f);
                    ep.out[oidx[j]] = gv[j] > 0.f ? v : 0.f;
                }
            }
        }
    }
}

// ================================================================================================
// Up: hi[n,hy,hx,chi] = ep( sum over the 2x2 taps valid for (hy,hx)'s parity and clo of lo * wt )
// wave w = parity class (py, px) = (w>>1, w&1); a tile is 64 lo positions -> 256 hi pixels
// ================================================================================================
template <int LO>
__global__ __launch_bounds__(256, 2) void up32_kernel(Operand lo, const float *__restrict__ wt, Ep32 ep, int n_img,
                                                       int n_tiles) {
    using T = Tile<LO>;
    constexpr int HI = 2 * LO, PR = T::TR + 2, PC = T::TC + 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // TI*PR*PC*PS floats
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, rc = lane & 31;
    const int py = wave >> 1, px = wave & 1;
    const int ky0 = 1 - py, kx0 = 1 - px;

    // w[ty][tx][chunk][t] = wt[clo = chunk*8 + half*4 + t][chi = rc][ky0 + 2ty][kx0 + 2tx]
    float w[2][2][4][4];
#pragma unroll
    for (int ty = 0; ty < 2; ++ty)
#pragma unroll
        for (int tx = 0; tx < 2; ++tx)
#pragma unroll
            for (int ch = 0; ch < 4; ++ch)
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    w[ty][tx][ch][t] = wt[((ch * 8 + half * 4 + t) * C32 + rc) * 16 + (ky0 + 2 * ty) * 4 + kx0 + 2 * tx];

    int aoff[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        int img, r, c;
        tile_pixel<LO>(mt * 32 + rc, img, r, c);
        // patch origin is lo (r0-1, -1); tap (ty,tx) reads lo (r + py - ty, c + px - tx)
        aoff[mt] = ((img * PR + r + 1 + py) * PC + c + 1 + px) * PS + half * 4;
    }
    const float bias = ep.bias != nullptr ? ep.bias[rc] : 0.f;
    float4 dummy;
    PatchLoader<LO, 1> pl;
    int img0, r0;
    if (blockIdx.x < n_tiles) {
        tile_origin<LO>(blockIdx.x, img0, r0);
        pl.issue(lo, img0, r0, n_img);
    }

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        tile_origin<LO>(tile, img0, r0);
        __syncthreads();
        pl.template commit<false>(lds, dummy);
        __syncthreads();
        if (tile + gridDim.x < n_tiles) {
            int ni, nr;
            tile_origin<LO>(tile + gridDim.x, ni, nr);
            pl.issue(lo, ni, nr, n_img);
        }

        f32x16 acc[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
#pragma unroll
        for (int ty = 0; ty < 2; ++ty)
#pragma unroll
            for (int tx = 0; tx < 2; ++tx)
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) {
                    const int toff = -(ty * PC + tx) * PS + ch * 8;
                    const float4 a0 = *reinterpret_cast<const float4 *>(lds + aoff[0] + toff);
                    const float4 a1 = *reinterpret_cast<const float4 *>(lds + aoff[1] + toff);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, w[ty][tx][ch][0], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, w[ty][tx][ch][0], acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, w[ty][tx][ch][1], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, w[ty][tx][ch][1], acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.z, w[ty][tx][ch][2], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.z, w[ty][tx][ch][2], acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.w, w[ty][tx][ch][3], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.w, w[ty][tx][ch][3], acc[1], 0, 0, 0);
                }

        // epilogue in two passes: all gate loads first (they must not queue behind the stores)
        int oidx[2][16];
        float gv[2][16];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int row = (reg & 3) + 8 * (reg >> 2) + 4 * half;
                int img, r, c;
                tile_pixel<LO>(mt * 32 + row, img, r, c);
                const int n = img0 + img;
                oidx[mt][reg] = n < n_img ? ((n * HI + 2 * (r0 + r) + py) * HI + 2 * c + px) * C32 + rc : -1;
                gv[mt][reg] = 1.f;
            }
        if (ep.gate != nullptr) {                       // one uniform branch, then 32 independent loads in flight
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) gv[mt][reg] = ep.gate[oidx[mt][reg] < 0 ? 0 : oidx[mt][reg]];
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                if (oidx[mt][reg] >= 0) {
                    float v = acc[mt][reg] + bias;
                    if (ep.relu) v = fmaxf(v, 0.f);
                    ep.out[oidx[mt][reg]] = gv[mt][reg] > 0.f ? v : 0.f;
                }
            }
    }
}

// ================================================================================================
// Wgrad: dwt[clo][chi][ky][kx] += sum_{n,ly,lx} lo[n,ly,lx,clo] * hi[n,2ly-1+ky,2lx-1+kx,chi]
// wave w = ky; acc[kx] = 32(clo) x 32(chi); pixels are the MFMA K axis (2 per instruction).
// slab layout per workgroup: [ky][kx][clo][chi] (16384 floats) + 32 bias sums.
// BIAS: 0 none, 1 = sum of the lo operand per clo, 2 = sum of the hi operand per chi (interior pixels)
// ================================================================================================
constexpr int WG32_SLAB = 16 * C32 * C32 + C32;

template <int LO, int BIAS>
__global__ __launch_bounds__(256, 2) void wgrad32_kernel(Operand lo, Operand hi, float *__restrict__ slab, int n_img,
                                                          int n_tiles) {
    using T = Tile<LO>;
    constexpr int PR = 2 * T::TR + 2, PC = 2 * T::TC + 2;
    constexpr int PATCH = T::TI * PR * PC * PS;
    extern __shared__ __attribute__((aligned(16))) float lds[];  // hi patch | lo tile [64][PS]
    float *lo_t = lds + PATCH;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, rc = lane & 31;

    f32x16 acc[4];
#pragma unroll
    for (int kx = 0; kx < 4; ++kx)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[kx][i] = 0.f;
    float lo_sum = 0.f;
    float4 hi_sum = make_float4(0.f, 0.f, 0.f, 0.f);

    // register-staged loads: hi patch + the 64-pixel lo tile (2 float4 per thread)
    PatchLoader<LO, 2> pl;
    float4 lr[2];
    auto issue_lo = [&](int i0, int rr0) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = threadIdx.x + it * 256;
            const int q = idx & 7, p = idx >> 3;
            int img, r, c;
            tile_pixel<LO>(p, img, r, c);
            const int n = i0 + img;
            lr[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (n < n_img) lr[it] = operand_load4(lo, ((int64_t)(n * LO + rr0 + r) * LO + c) * C32 + q * 4);
        }
    };
    int img0, r0;
    if (blockIdx.x < n_tiles) {
        tile_origin<LO>(blockIdx.x, img0, r0);
        pl.issue(hi, img0, r0, n_img);
        issue_lo(img0, r0);
    }

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        tile_origin<LO>(tile, img0, r0);
        __syncthreads();
        pl.template commit<BIAS == 2>(lds, hi_sum);
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = threadIdx.x + it * 256;
            *reinterpret_cast<float4 *>(lo_t + (idx >> 3) * PS + (idx & 7) * 4) = lr[it];
        }
        __syncthreads();
        if (tile + gridDim.x < n_tiles) {
            int ni, nr;
            tile_origin<LO>(tile + gridDim.x, ni, nr);
            pl.issue(hi, ni, nr, n_img);
            issue_lo(ni, nr);
        }

#pragma unroll
        for (int s = 0; s < 32; ++s) {
            // k-pair s covers lo pixels 2s and 2s+1 (adjacent columns of one row); this lane takes 2s+half
            int img, r, c;
            tile_pixel<LO>(2 * s, img, r, c);            // compile-time after unrolling
            const float a = lo_t[(2 * s + half) * PS + rc];
            if (BIAS == 1) lo_sum += a;
            const int boff = ((img * PR + 2 * r + wave) * PC + 2 * (c + half)) * PS + rc;
#pragma unroll
            for (int kx = 0; kx < 4; ++kx)
                acc[kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, lds[boff + kx * PS], acc[kx], 0, 0, 0);
        }
    }

    // partial results -> slab[blockIdx][ky][kx][clo][chi]
    float *out = slab + (int64_t)blockIdx.x * WG32_SLAB;
#pragma unroll
    for (int kx = 0; kx < 4; ++kx)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int row = (reg & 3) + 8 * (reg >> 2) + 4 * half;      // clo
            out[((wave * 4 + kx) * C32 + row) * C32 + rc] = acc[kx][reg];
        }
    if (BIAS == 1) {
        if (wave == 0) {
            const float tot = lo_sum + __shfl_xor(lo_sum, 32, 64);
            if (half == 0) out[16 * C32 * C32 + rc] = tot;
        }
    } else if (BIAS == 2) {
        __syncthreads();
        // every thread summed channel chunk q = threadIdx.x & 7 (SLOTS stride 256 keeps q fixed)
        *reinterpret_cast<float4 *>(lds + threadIdx.x * 4) = hi_sum;
        __syncthreads();
        if (threadIdx.x < C32) {
            const int q = threadIdx.x >> 2, e = threadIdx.x & 3;
            float tot = 0.f;
            for (int j = 0; j < 32; ++j) tot += lds[(j * 8 + q) * 4 + e];
            out[16 * C32 * C32 + threadIdx.x] = tot;
        }
    }
}

// dwt[clo][chi][ky][kx] += sum_wg slab[wg][ky][kx][clo][chi];  dbias[c] += sum_wg slab[wg][16384 + c]
// 16 outputs x 16 slab groups per workgroup: ~1000 workgroups keep enough loads in flight to stream the slab
__global__ __launch_bounds__(256) void wgrad32_reduce_kernel(const float *__restrict__ slab, int n_wg,
                                                              float *__restrict__ dwt, float *__restrict__ dbias) {
    __shared__ float red[16][17];
    const int il = threadIdx.x & 15, zg = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + il;                    // 0 .. 16384+32
    float s = 0.f;
    if (i < WG32_SLAB)
        for (int z = zg; z < n_wg; z += 16) s += slab[(int64_t)z * WG32_SLAB + i];
    red[zg][il] = s;
    __syncthreads();
    if (zg == 0 && i < WG32_SLAB) {
        float tot = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) tot += red[j][il];
        if (i < 16 * C32 * C32) {
            const int chi = i & 31, clo = (i >> 5) & 31, tap = i >> 10;
            dwt[(clo * C32 + chi) * 16 + tap] += tot;
        } else if (dbias != nullptr) {
            dbias[i - 16 * C32 * C32] += tot;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            n = prop.multiProcessorCount;
        if (n <= 0) n = 256;
    }
    return n;
}

template <int LO> static constexpr int tiles_for(int n) { return Tile<LO>::TI == 1 ? n * (LO / Tile<LO>::TR) : (n + Tile<LO>::TI - 1) / Tile<LO>::TI; }

bool conv32_fits(const arvae_link_t *l) {
    return l->chi == 32 && l->clo == 32 && l->kh == 4 && l->kw == 4 && l->stride == 2 && l->pad == 1 &&
           l->hh == l->hw && l->lh == l->lw && (l->lh == 16 || l->lh == 8 || l->lh == 4) && l->hi_perm_c == 0 &&
           l->lo_perm_c == 0;
}

template <int LO> static int launch_down(const arvae_link_t *l, const Operand &hi, const float *wt, const Ep32 &ep, hipStream_t s) {
    using T = Tile<LO>;
    constexpr int PATCH = T::TI * (2 * T::TR + 2) * (2 * T::TC + 2) * PS;
    constexpr int LDS = (PATCH > 8192 ? PATCH : 8192) * 4;
    const int tiles = tiles_for<LO>(l->n);
    const int grid = tiles < 2 * cu_count() ? tiles : 2 * cu_count();
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void *)down32_kernel<LO>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS); attr = true; }
    hipLaunchKernelGGL(down32_kernel<LO>, dim3(grid), dim3(256), LDS, s, hi, wt, ep, l->n, tiles);
    return check_launch(LO == 16 ? "down32_kernel<16>" : LO == 8 ? "down32_kernel<8>" : "down32_kernel<4>");
}

int conv32_down(const arvae_link_t *l, const Operand &hi, const float *wt, const float *bias, int relu, const float *gate,
                float *out, hipStream_t s) {
    Ep32 ep{bias, gate, out, relu};
    switch (l->lh) {
        case 16: return launch_down<16>(l, hi, wt, ep, s);
        case 8: return launch_down<8>(l, hi, wt, ep, s);
        default: return launch_down<4>(l, hi, wt, ep, s);
    }
}

template <int LO> static int launch_up(const arvae_link_t *l, const Operand &lo, const float *wt, const Ep32 &ep, hipStream_t s) {
    using T = Tile<LO>;
    constexpr int LDS = T::TI * (T::TR + 2) * (T::TC + 2) * PS * 4;
    const int tiles = tiles_for<LO>(l->n);
    const int grid = tiles < 2 * cu_count() ? tiles : 2 * cu_count();
    hipLaunchKernelGGL(up32_kernel<LO>, dim3(grid), dim3(256), LDS, s, lo, wt, ep, l->n, tiles);
    return check_launch(LO == 16 ? "up32_kernel<16>" : LO == 8 ? "up32_kernel<8>" : "up32_kernel<4>");
}

int conv32_up(const arvae_link_t *l, const Operand &lo, const float *wt, const float *bias, int relu, const float *gate,
              float *out, hipStream_t s) {
    Ep32 ep{bias, gate, out, relu};
    switch (l->lh) {
        case 16: return launch_up<16>(l, lo, wt, ep, s);
        case 8: return launch_up<8>(l, lo, wt, ep, s);
        default: return launch_up<4>(l, lo, wt, ep, s);
    }
}

int conv32_wgrad_groups(const arvae_link_t *l) {
    int tiles;
    switch (l->lh) {
        case 16: tiles = tiles_for<16>(l->n); break;
        case 8: tiles = tiles_for<8>(l->n); break;
        default: tiles = tiles_for<4>(l->n); break;
    }
    // one workgroup per CU for the big maps (>= 4 tiles each amortise the 64 KB partial it writes);
    // small maps keep every CU busy with one tile per workgroup
    int g = tiles;
    if (g > cu_count()) g = cu_count();
    if (g < 1) g = 1;
    return g;
}

int64_t conv32_wgrad_ws_floats(const arvae_link_t *l) { return (int64_t)conv32_wgrad_groups(l) * WG32_SLAB; }

template <int LO> static int launch_wgrad(const arvae_link_t *l, const Operand &lo, const Operand &hi, float *slab, int bias_mode,
                                          int grid, hipStream_t s) {
    using T = Tile<LO>;
    constexpr int LDS = (T::TI * (2 * T::TR + 2) * (2 * T::TC + 2) * PS + 64 * PS) * 4;
    const int tiles = tiles_for<LO>(l->n);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void *)wgrad32_kernel<LO, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void *)wgrad32_kernel<LO, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        (void)hipFuncSetAttribute((const void *)wgrad32_kernel<LO, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr = true;
    }
    if (bias_mode == 1)
        hipLaunchKernelGGL((wgrad32_kernel<LO, 1>), dim3(grid), dim3(256), LDS, s, lo, hi, slab, l->n, tiles);
    else if (bias_mode == 2)
        hipLaunchKernelGGL((wgrad32_kernel<LO, 2>), dim3(grid), dim3(256), LDS, s, lo, hi, slab, l->n, tiles);
    else
        hipLaunchKernelGGL((wgrad32_kernel<LO, 0>), dim3(grid), dim3(256), LDS, s, lo, hi, slab, l->n, tiles);
    return check_launch(LO == 16 ? "wgrad32_kernel<16>" : LO == 8 ? "wgrad32_kernel<8>" : "wgrad32_kernel<4>");
}

// bias_mode: 0 none, 1 dbias[clo] += sum lo, 2 dbias[chi] += sum hi
int conv32_wgrad(const arvae_link_t *l, const Operand &lo, const Operand &hi, float *dwt, float *dbias, int bias_mode,
                 float *slab, hipStream_t s) {
    const int grid = conv32_wgrad_groups(l);
    int rc;
    switch (l->lh) {
        case 16: rc = launch_wgrad<16>(l, lo, hi, slab, bias_mode, grid, s); break;
        case 8: rc = launch_wgrad<8>(l, lo, hi, slab, bias_mode, grid, s); break;
        default: rc = launch_wgrad<4>(l, lo, hi, slab, bias_mode, grid, s); break;
    }
    if (rc) return rc;
    hipLaunchKernelGGL(wgrad32_reduce_kernel, dim3((WG32_SLAB + 15) / 16), dim3(256), 0, s, slab, grid, dwt,
                       bias_mode ? dbias : nullptr);
    return check_launch("wgrad32_reduce_kernel");
}

}  // namespace arvae
