// Stride-1 convolutions between wide layers (the Morpho-MNIST 64 <-> 64 channel k4 layers, imagevae/mnist_vae.py) as
// implicit GEMMs on the split-bf16 MFMA (x3tile.h): rows = output pixels, columns = output channels, reduction =
// (tap, source channel).  The A tile is GATHERED: a thread owns two (pixel, 4-channel) slots, works out its pixel's
// coordinates once, and per 32-channel chunk adds the tap's offset -- a coalesced 16-byte load per slot, zero where the
// tap falls outside the source (padding).  The weights are re-ordered to [out channel][tap][source channel] by a small
// kernel first (in memory order the taps are innermost, and gathering them cost more than the MFMAs).  The same kernel serves
//   Conv2d forward and ConvTranspose2d data gradient   (source pixel = output pixel + tap - pad), and
//   ConvTranspose2d forward and Conv2d data gradient   (source pixel = output pixel + pad - tap),
// with the activation derivative / dropout keep-mask of a gradient operand folded into the gather, and bias, activation
// and keep-mask in the epilogue.  The weight gradient is the same tile machinery with both operands "K x rows" (pixels
// are the reduction axis), sliced over pixels into a workspace and summed in fixed order.
#include <mutex>
#include "diag.h"
#include "common.h"
#include "x3tile.h"
#include "conv32_common.h"

namespace arvae {

constexpr int C64_TP = 64, C64_TQ = 64;
typedef float f32x16c __attribute__((ext_vector_type(16)));

struct ConvRows {
    Operand src;                 // [n][sh][sw][cs] channels-last
    int n, sh, sw, cs;
    int oh, ow, q;               // output grid and channels
    int kh, kw, sgn, off;        // source coordinate = output coordinate + sgn * k + off
    const float *wt;             // re-ordered weights [q][tap][c] (conv64_weight_prep_kernel): reduction index contiguous
    const float *bias;
    const uint8_t *mask;
    int act;
    float *out;                  // [n][oh][ow][q]
    GateOp gate;                 // data-gradient launches: result *= act'(gate.y) * 2 gate.mask at the output location
    unsigned *amax_out;          // AMAX array of the output (conv32_common.h) or null: zeroed by the weight-prep launch in front, every
                                 // wave folds its maximum into entry (workgroup % AMAX_N) with an integer atomic max (bit patterns of
                                 // non-negative floats order like the floats: exact, order-independent)
    const unsigned *amax_in;     // conv_rows_h2_kernel: AMAX array of the (plain) source
    const unsigned *w_amax;      // conv_rows_h2_kernel: bit pattern of max |wt| (written by the weight-prep launch behind the packed weights)
};

// 16 bytes through a buffer resource at per-lane offset + scalar offset (the scalar part is not range-checked)
__device__ __forceinline__ float4 buf_load4s(__amdgpu_buffer_rsrc_t r, unsigned off, int soff) {
    const f32x4v v = __builtin_bit_cast(f32x4v, __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, soff, 0));
    return make_float4(v.x, v.y, v.z, v.w);
}
// value of a gradient operand (activation derivative of the saved output, keep-mask) for 4 consecutive channels
template <bool PLAIN>
__device__ __forceinline__ float4 load_src4(const Operand &o, int64_t at, bool ok) {
    const int64_t a = ok ? at : 0;
    float4 v = *reinterpret_cast<const float4 *>(o.v + a);
    if (!PLAIN) {
        float4 y = *reinterpret_cast<const float4 *>(o.y + a);
        if (o.mask != nullptr) {
            const uchar4 m = *reinterpret_cast<const uchar4 *>(o.mask + a);
            v.x *= 2.f * (float)m.x; v.y *= 2.f * (float)m.y; v.z *= 2.f * (float)m.z; v.w *= 2.f * (float)m.w;
            y.x *= 0.5f; y.y *= 0.5f; y.z *= 0.5f; y.w *= 0.5f;
        }
        v.x *= act_bwd_from_out(y.x, o.act); v.y *= act_bwd_from_out(y.y, o.act);
        v.z *= act_bwd_from_out(y.z, o.act); v.w *= act_bwd_from_out(y.w, o.act);
    }
    return ok ? v : float4{0.f, 0.f, 0.f, 0.f};
}

template <bool PLAIN>
__global__ __launch_bounds__(256) void conv_rows_x3_kernel(ConvRows g) {
    typedef X3Plane<RG_ROWSK, C64_TP> PlaneA;
    typedef X3Plane<RG_ROWSK, C64_TQ> PlaneB;
    __shared__ __attribute__((aligned(16))) unsigned short As[3 * PlaneA::PLANE];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[3 * PlaneB::PLANE];
    const int lane = threadIdx.x & 63, half = lane >> 5, rc = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wp = wave & 1, wq = wave >> 1;
    const int p0 = blockIdx.x * C64_TP, q0 = blockIdx.y * C64_TQ;
    const int M = g.n * g.oh * g.ow;
    // reduction index kk = tap * cs + c, walked in chunks of RG_R; a chunk lies inside one tap when 32 | cs, and spans
    // several taps for narrow sources (cs = 8: four taps per chunk), so the tap is worked out per gather slot
    const int chunks = g.kh * g.kw * g.cs / RG_R;

    // this thread's two gather slots: (pixel, 4 channels)
    int oy[2], ox[2], c4[2];
    int64_t img_base[2];
    bool pok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int p = p0 + idx / (RG_R / 4);
        c4[i] = 4 * (idx % (RG_R / 4));
        pok[i] = p < M;
        const int pc = pok[i] ? p : 0;
        const int img = pc / (g.oh * g.ow), rem = pc - img * g.oh * g.ow;
        oy[i] = rem / g.ow; ox[i] = rem - oy[i] * g.ow;
        img_base[i] = (int64_t)img * g.sh * g.sw;
    }
    // weight slots: (output channel, 4 source channels)
    int wq_row[2], wc4[2];
    bool qok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = threadIdx.x + 256 * i;
        wq_row[i] = q0 + idx / (RG_R / 4);
        wc4[i] = 4 * (idx % (RG_R / 4));
        qok[i] = wq_row[i] < g.q;
        if (!qok[i]) wq_row[i] = 0;
    }
    float4 va[2], vb[2];
    const int kwidth = g.kh * g.kw * g.cs;
    auto load = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int kk = chunk * RG_R + c4[i];                 // c4[i] == wc4[i]: the slot's offset inside the chunk
            const int tap = kk / g.cs, c = kk - tap * g.cs;
            const int ky = tap / g.kw, kx = tap - ky * g.kw;
            const int sy = oy[i] + g.sgn * ky + g.off, sx = ox[i] + g.sgn * kx + g.off;
            const bool ok = pok[i] && sy >= 0 && sy < g.sh && sx >= 0 && sx < g.sw;
            va[i] = load_src4<PLAIN>(g.src, ((img_base[i] + (int64_t)sy * g.sw + sx) * g.cs + c), ok);
            const float4 w = *reinterpret_cast<const float4 *>(g.wt + (int64_t)wq_row[i] * kwidth + kk);
            vb[i] = qok[i] ? w : float4{0.f, 0.f, 0.f, 0.f};
        }
    };
    f32x16c acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    load(0);
    const int abase = PlaneA::lane_base(wp), bbase = PlaneB::lane_base(wq);
    for (int chunk = 0; chunk < chunks; ++chunk) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            PlaneA::commit(As, threadIdx.x + 256 * i, va[i]);
            PlaneB::commit(Bs, threadIdx.x + 256 * i, vb[i]);
        }
        __syncthreads();
        if (chunk + 1 < chunks) load(chunk + 1);
#pragma unroll
        for (int s = 0; s < RG_R / 16; ++s) {
            const rg_bf16x8 ah = PlaneA::operand(As, abase, 0, s), am = PlaneA::operand(As, abase, 1, s), al = PlaneA::operand(As, abase, 2, s);
            const rg_bf16x8 bh = PlaneB::operand(Bs, bbase, 0, s), bm = PlaneB::operand(Bs, bbase, 1, s), bl = PlaneB::operand(Bs, bbase, 2, s);
            X3_MFMA6(acc, ah, am, al, bh, bm, bl);
        }
    }
    const int q = q0 + 32 * wq + rc;
    const float bias = (g.bias != nullptr && q < g.q) ? g.bias[q] : 0.f;
    // keep-mask bytes / gate values of this lane's 16 outputs: all requested first (one load -> multiply -> store chain per
    // output made the gated launch 184 us instead of 78)
    float gy[16];
    unsigned char gm[16];
    const uint8_t *mp = g.gate.y != nullptr ? g.gate.mask : g.mask;
    if (mp != nullptr || g.gate.y != nullptr) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p = p0 + 32 * wp + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int64_t o = (p < M && q < g.q) ? (int64_t)p * g.q + q : 0;
            gm[r] = mp != nullptr ? mp[o] : (unsigned char)1;
            gy[r] = g.gate.y != nullptr ? g.gate.y[o] : 0.f;
        }
    }
    float vmax = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int p = p0 + 32 * wp + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (p < M && q < g.q) {
            const int64_t o = (int64_t)p * g.q + q;
            float v = act_fwd(acc[r] + bias, g.act);
            if (g.gate.y != nullptr) {
                // (the saved output of a dropout layer is the kept activation times two: Operand::apply)
                const float ys = g.gate.mask != nullptr ? 0.5f : 1.f, k2 = g.gate.mask != nullptr ? 2.f : 1.f;
                v *= act_bwd_from_out_sel(ys * gy[r], g.gate.act) * k2 * (float)gm[r];
            } else if (g.mask != nullptr) {
                v *= 2.f * (float)gm[r];
            }
            g.out[o] = v;
            vmax = fmaxf(vmax, fabsf(v));
        }
    }
    if (g.amax_out != nullptr) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
        float *wmax = reinterpret_cast<float *>(As);
        __syncthreads();
        if (lane == 0) wmax[wave] = vmax;
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_fetch_max(g.amax_out + ((blockIdx.y * gridDim.x + blockIdx.x) & (AMAX_N - 1)),
                                   __builtin_bit_cast(unsigned, fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The same product on the fp16 MFMA with scaled two-term operands (conv32_common.h: three partial products, two LDS planes per
// operand, the scales from the source's AMAX array and the weights' maximum): for PLAIN sources that come with their maxima -- the
// 8 -> 64 products of the Morpho-MNIST step, which ran six bf16 products per multiply-add here through most of round 4.
template <int TP> struct H2PlaneRows {          // [TP][RG_XP] fp16, reduction index contiguous; planes h | l
    static constexpr int PLANE = TP * RG_XP;
    __device__ static __forceinline__ void commit(unsigned short *lds, int idx, const float4 &v, float sc) {
        unsigned h0, l0, h1, l1;
        split_pair_h2(v.x, v.y, sc, h0, l0);
        split_pair_h2(v.z, v.w, sc, h1, l1);
        unsigned short *d = lds + (idx / (RG_R / 4)) * RG_XP + 4 * (idx % (RG_R / 4));
        *reinterpret_cast<uint2 *>(d) = uint2{h0, h1};
        *reinterpret_cast<uint2 *>(d + PLANE) = uint2{l0, l1};
    }
    __device__ static __forceinline__ int lane_base(int w) {
        const int lane = threadIdx.x & 63;
        return (32 * w + (lane & 31)) * RG_XP + 8 * (lane >> 5);
    }
    __device__ static __forceinline__ f16x8 operand(const unsigned short *lds, int base, int t, int s) {
        return __builtin_bit_cast(f16x8, *reinterpret_cast<const rg_i32x4 *>(lds + t * PLANE + base + 16 * s));
    }
};

__global__ __launch_bounds__(256) void conv_rows_h2_kernel(ConvRows g) {
    typedef H2PlaneRows<C64_TP> PlaneA;
    typedef H2PlaneRows<C64_TQ> PlaneB;
    __shared__ __attribute__((aligned(16))) unsigned short As[2 * PlaneA::PLANE];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2 * PlaneB::PLANE];
    const AmaxLoad al = amax_issue(g.amax_in);
    const unsigned wbits = *g.w_amax;
    const int lane = threadIdx.x & 63, half = lane >> 5, rc = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wp = wave & 1, wq = wave >> 1;
    const int p0 = blockIdx.x * C64_TP, q0 = blockIdx.y * C64_TQ;
    const int M = g.n * g.oh * g.ow;
    const int chunks = g.kh * g.kw * g.cs / RG_R;
    int oy[2], ox[2], c4[2];
    int64_t img_base[2];
    bool pok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int p = p0 + idx / (RG_R / 4);
        c4[i] = 4 * (idx % (RG_R / 4));
        pok[i] = p < M;
        const int pc = pok[i] ? p : 0;
        const int img = pc / (g.oh * g.ow), rem = pc - img * g.oh * g.ow;
        oy[i] = rem / g.ow; ox[i] = rem - oy[i] * g.ow;
        img_base[i] = (int64_t)img * g.sh * g.sw;
    }
    int wq_row[2];
    bool qok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = threadIdx.x + 256 * i;
        wq_row[i] = q0 + idx / (RG_R / 4);
        qok[i] = wq_row[i] < g.q;
        if (!qok[i]) wq_row[i] = 0;
    }
    float4 va[2], vb[2];
    const int kwidth = g.kh * g.kw * g.cs;
    auto load = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int kk = chunk * RG_R + c4[i];
            const int tap = kk / g.cs, c = kk - tap * g.cs;
            const int ky = tap / g.kw, kx = tap - ky * g.kw;
            const int sy = oy[i] + g.sgn * ky + g.off, sx = ox[i] + g.sgn * kx + g.off;
            const bool ok = pok[i] && sy >= 0 && sy < g.sh && sx >= 0 && sx < g.sw;
            va[i] = load_src4<true>(g.src, ((img_base[i] + (int64_t)sy * g.sw + sx) * g.cs + c), ok);
            const float4 w = *reinterpret_cast<const float4 *>(g.wt + (int64_t)wq_row[i] * kwidth + kk);
            vb[i] = qok[i] ? w : float4{0.f, 0.f, 0.f, 0.f};
        }
    };
    f32x16c acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    load(0);
    const Pow2 sc_a = amax_scale(al), sc_w = pow2_for(wbits);
    const int abase = PlaneA::lane_base(wp), bbase = PlaneB::lane_base(wq);
    for (int chunk = 0; chunk < chunks; ++chunk) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            PlaneA::commit(As, threadIdx.x + 256 * i, va[i], sc_a.s);
            PlaneB::commit(Bs, threadIdx.x + 256 * i, vb[i], sc_w.s);
        }
        __syncthreads();
        if (chunk + 1 < chunks) load(chunk + 1);
#pragma unroll
        for (int s = 0; s < RG_R / 16; ++s) {
            const f16x8 ah = PlaneA::operand(As, abase, 0, s), al2 = PlaneA::operand(As, abase, 1, s);
            const f16x8 bh = PlaneB::operand(Bs, bbase, 0, s), bl = PlaneB::operand(Bs, bbase, 1, s);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al2, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
        }
    }
    const float inv = sc_a.inv * sc_w.inv;
    const int q = q0 + 32 * wq + rc;
    const float bias = (g.bias != nullptr && q < g.q) ? g.bias[q] : 0.f;
    float gy[16];
    unsigned char gm[16];
    const uint8_t *mp = g.gate.y != nullptr ? g.gate.mask : g.mask;
    if (mp != nullptr || g.gate.y != nullptr) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int p = p0 + 32 * wp + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int64_t o = (p < M && q < g.q) ? (int64_t)p * g.q + q : 0;
            gm[r] = mp != nullptr ? mp[o] : (unsigned char)1;
            gy[r] = g.gate.y != nullptr ? g.gate.y[o] : 0.f;
        }
    }
    float vmax = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int p = p0 + 32 * wp + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (p < M && q < g.q) {
            const int64_t o = (int64_t)p * g.q + q;
            float v = act_fwd(fmaf(acc[r], inv, bias), g.act);
            if (g.gate.y != nullptr) {
                const float ys = g.gate.mask != nullptr ? 0.5f : 1.f, k2 = g.gate.mask != nullptr ? 2.f : 1.f;
                v *= act_bwd_from_out_sel(ys * gy[r], g.gate.act) * k2 * (float)gm[r];
            } else if (g.mask != nullptr) {
                v *= 2.f * (float)gm[r];
            }
            g.out[o] = v;
            vmax = fmaxf(vmax, fabsf(v));
        }
    }
    if (g.amax_out != nullptr) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
        float *wmax = reinterpret_cast<float *>(As);
        __syncthreads();
        if (lane == 0) wmax[wave] = vmax;
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_fetch_max(g.amax_out + ((blockIdx.y * gridDim.x + blockIdx.x) & (AMAX_N - 1)),
                                   __builtin_bit_cast(unsigned, fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- 8 source channels, 4 x 4 taps: the whole reduction (K = 128) is eight MFMA k-steps --------------------------------------------
// The Morpho-MNIST 8 -> 64 products (ConvTranspose2d(8, 64) forward, Conv2d(64, 8) data gradient) on the gathering kernels above
// were bound by their gather: every 64 x 64 tile re-read its 64 x 128 operand from L2, four chunks with two barriers each.  Here
//   * a lane's MFMA A operand for k-step s is the 8 channels of ONE tap (2 s + lane half) of its pixel: 32 contiguous bytes of the
//     source -- so the source rows a tile touches are staged ONCE in LDS as two fp16 terms (16 + 16 bytes per pixel, <= 7 rows) and
//     every operand is a single 16-byte LDS read per term at a per-lane pixel address (out-of-image taps read a zero pixel);
//   * the wave's 32 x 128 weight slice lives in registers (two terms, 64 VGPRs) for the whole launch;
//   * a tile is 64 consecutive output pixels of an image in row-major order (94 % of the MFMA rows at 22 x 22), workgroups walk
//     tiles persistently, four workgroups per CU hide each other's staging round trip.
// Arithmetic: scaled two-term fp16, three products (conv32_common.h); plain sources that come with their maxima.
#ifdef S8_STAMPS
__device__ unsigned long long g_s8_stamps[8];
#define S8STAMP(k) { const unsigned long long now_ = __builtin_readcyclecounter(); s8ph[k] += now_ - s8tc; s8tc = now_; }
#else
#define S8STAMP(k)
#endif
constexpr int S8_PIX = 255;                                      // staged source pixels per tile (one per thread; 255 = the zero pixel)
__device__ __forceinline__ void s8_split8(const float (&x)[8], float sc, f16x8 &hi, f16x8 &lo) {
    uint4 h, l;
    split_pair_h2(x[0], x[1], sc, h.x, l.x);
    split_pair_h2(x[2], x[3], sc, h.y, l.y);
    split_pair_h2(x[4], x[5], sc, h.z, l.z);
    split_pair_h2(x[6], x[7], sc, h.w, l.w);
    hi = __builtin_bit_cast(f16x8, h); lo = __builtin_bit_cast(f16x8, l);
}
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void conv_s8_h2_kernel(ConvRows g, int tiles_per_img, int n_tiles) {
    __shared__ uint4 src_h[S8_PIX + 1], src_l[S8_PIX + 1];
    __shared__ __attribute__((aligned(16))) float otile[64][68];   // the tile's results, so that the epilogue is 16-byte loads and stores
    __shared__ float wmax[4];
    const AmaxLoad al = amax_issue(g.amax_in);
    const unsigned wbits = *g.w_amax;
    const int lane = threadIdx.x & 63, half = lane >> 5, rc = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wp = wave & 1, wq = wave >> 1;
    const int q = 32 * wq + rc;
    const Pow2 sc_w = pow2_for(wbits);
    f16x8 wh[8], wl[8];
#pragma unroll
    for (int s8 = 0; s8 < 8; ++s8) {
        const float *wsrc = g.wt + ((int64_t)(q < g.q ? q : 0) * 16 + 2 * s8 + half) * 8;
        const float4 a = *reinterpret_cast<const float4 *>(wsrc), b = *reinterpret_cast<const float4 *>(wsrc + 4);
        const float z = q < g.q ? 1.f : 0.f;
        const float x[8] = {a.x * z, a.y * z, a.z * z, a.w * z, b.x * z, b.y * z, b.z * z, b.w * z};
        s8_split8(x, sc_w.s, wh[s8], wl[s8]);
    }
    // epilogue slot of this thread: pixel (threadIdx.x / 16) + 16 k of the tile, channels 4 (threadIdx.x % 16) ..
    const int e_px = threadIdx.x >> 4, e_c = 4 * (threadIdx.x & 15);
    const bool e_ok = e_c < g.q;
    const float4 bias4 = (g.bias != nullptr && e_ok) ? *reinterpret_cast<const float4 *>(g.bias + e_c) : make_float4(0.f, 0.f, 0.f, 0.f);
    const Pow2 sc_a = amax_scale(al);
    const float inv = sc_a.inv * sc_w.inv;
    (void)q;
    if (threadIdx.x == 255) { src_h[S8_PIX] = make_uint4(0u, 0u, 0u, 0u); src_l[S8_PIX] = make_uint4(0u, 0u, 0u, 0u); }
    const int opix = g.oh * g.ow;
    const uint8_t *mp = g.gate.y != nullptr ? g.gate.mask : g.mask;
    const ActCoef ac = act_coef(g.act);
    const GateCoef gc = epilogue_coef(g.gate.y, g.gate.act, g.gate.mask, g.mask);
    float vmax = 0.f;
    // A tile's global round trips are requested a phase ahead (round 5): the NEXT tile's source pixels while this tile is multiplied,
    // this tile's keep bytes / gate values before its MFMAs instead of behind them.  As a chain -- load the source, split, multiply,
    // then load the epilogue's operands -- a tile cost ~6.5 us per workgroup and the launch wrote its 127 MB at 1.8 TB/s.
    auto geom = [&](int tile, int &img, int &P0, int &sy0, int &nrows) {
        img = tile / tiles_per_img;
        P0 = (tile - img * tiles_per_img) * 64;
        const int y_first = P0 / g.ow, y_last = min(P0 + 63, opix - 1) / g.ow;
        sy0 = g.sgn < 0 ? y_first - 3 + g.off : y_first + g.off;
        nrows = y_last - y_first + 4;
    };
    float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa;           // this thread's staged pixel of the coming tile (zeros outside)
    // (raw buffer loads and stores with an out-of-range offset for "none": a load under a run-time branch makes the compiler drain the
    // wave's memory queue at the join -- stamped here: half of a tile's 8600 cycles sat in the phase that only REQUESTS operands)
    const int64_t out_bytes = (int64_t)g.n * opix * g.q * 4;
    const __amdgpu_buffer_rsrc_t rs_src = make_rsrc(g.src.v, (int64_t)g.n * g.sh * g.sw * 32), rs_out = make_rsrc(g.out, out_bytes);
    const __amdgpu_buffer_rsrc_t rs_mask = make_rsrc(mp != nullptr ? (const void *)mp : (const void *)g.out, mp != nullptr ? out_bytes / 4 : 0);
    const __amdgpu_buffer_rsrc_t rs_gate = make_rsrc(g.gate.y != nullptr ? (const void *)g.gate.y : (const void *)g.out, g.gate.y != nullptr ? out_bytes : 0);
    const bool has_mask = mp != nullptr;
    auto fetch_src = [&](int tile) {
        int img, P0, sy0, nrows;
        geom(tile < n_tiles ? tile : 0, img, P0, sy0, nrows);
        const int r = threadIdx.x / g.sw, x = threadIdx.x - r * g.sw, sy = sy0 + r;
        const bool ok = tile < n_tiles && (int)threadIdx.x < nrows * g.sw && sy >= 0 && sy < g.sh;
        const unsigned off = ok ? (unsigned)((((img * g.sh + sy) * g.sw) + x) * 32) : OOB;
        sa = buf_load4(rs_src, off);
        sb = buf_load4(rs_src, ok ? off + 16u : OOB);
    };
    fetch_src(blockIdx.x);
#ifdef S8_STAMPS
    unsigned long long s8ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s8tc = __builtin_readcyclecounter();
#endif
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        int img, P0, sy0, nrows;
        geom(tile, img, P0, sy0, nrows);
        S8STAMP(0);
        // (LDS-only barriers, common.h: __syncthreads() would also wait for the previous tile's STORES and for the loads just
        // requested ahead)
        lds_barrier();                                           // the previous tile's operand reads are done
        if ((int)threadIdx.x < nrows * g.sw) {
            const float v[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
            f16x8 h, l;
            s8_split8(v, sc_a.s, h, l);
            src_h[threadIdx.x] = __builtin_bit_cast(uint4, h);
            src_l[threadIdx.x] = __builtin_bit_cast(uint4, l);
        }
        S8STAMP(1);
        lds_barrier();
        S8STAMP(2);
        // this tile's epilogue operands and the next tile's source pixels: in flight under the MFMAs
        const int64_t obase = (int64_t)img * opix;
        unsigned gm[4], oo[4];                                    // (oo: the slot's element offset in the output, OOB / 4 for none)
        float4 gy[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int pr = P0 + e_px + 16 * u;
            oo[u] = (pr < opix && e_ok) ? (unsigned)((obase + pr) * g.q + e_c) : OOB / 4;
            const unsigned mk = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs_mask, (int)oo[u], 0, 0);
            gm[u] = has_mask ? mk : 0x01010101u;
            gy[u] = buf_load4(rs_gate, oo[u] == OOB / 4 ? OOB : oo[u] * 4u);
        }
        fetch_src(tile + gridDim.x);
        S8STAMP(3);
        const int P = P0 + 32 * wp + rc;
        const bool pok = P < opix;
        const int y = P / g.ow, x = P - y * g.ow;
        f32x16c acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        // a k-step's two operand reads are requested one step ahead of its MFMAs (pinned: left to the scheduler every read was
        // followed by a wait for it, sixteen exposed LDS round trips per tile)
        auto operand = [&](int s8, f16x8 &ah, f16x8 &al2) __attribute__((always_inline)) {
            const int tap = 2 * s8 + half, ky = tap >> 2, kx = tap & 3;
            const int sy = y + g.sgn * ky + g.off, sx = x + g.sgn * kx + g.off;
            const bool ok = pok && (unsigned)sy < (unsigned)g.sh && (unsigned)sx < (unsigned)g.sw;
            const int idx = ok ? (sy - sy0) * g.sw + sx : S8_PIX;
            ah = __builtin_bit_cast(f16x8, src_h[idx]);
            al2 = __builtin_bit_cast(f16x8, src_l[idx]);
        };
        f16x8 ah_c, al_c, ah_n, al_n;
        operand(0, ah_c, al_c);
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) {
            if (s8 + 1 < 8) operand(s8 + 1, ah_n, al_n);
            __builtin_amdgcn_sched_barrier(0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al_c, wh[s8], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_c, wl[s8], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah_c, wh[s8], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            ah_c = ah_n;
            al_c = al_n;
        }
        // the accumulators meet in LDS ([pixel][channel]); then every thread finishes four (pixel, 4 channels) slots: bias, activation,
        // gate, one 16-byte store (the 16 dword stores per lane of the gathering kernels moved 256 bytes per instruction: the launch
        // ran at 1.7 TB/s of output)
#pragma unroll
        for (int r = 0; r < 16; ++r) otile[32 * wp + (r & 3) + 8 * (r >> 2) + 4 * half][q] = acc[r];
        S8STAMP(4);
        lds_barrier();
        S8STAMP(5);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            {
                const float4 a4 = *reinterpret_cast<const float4 *>(&otile[e_px + 16 * u][e_ok ? e_c : 0]);
                // (common.h's coefficient form: result = act(..) * d(y) * keep byte)
                float4 v = make_float4(act_fwd_coef(fmaf(a4.x, inv, bias4.x), ac), act_fwd_coef(fmaf(a4.y, inv, bias4.y), ac),
                                       act_fwd_coef(fmaf(a4.z, inv, bias4.z), ac), act_fwd_coef(fmaf(a4.w, inv, bias4.w), ac));
                const unsigned m = gm[u];
                v.x *= gate_deriv(gy[u].x, gc) * (float)(m & 255u);
                v.y *= gate_deriv(gy[u].y, gc) * (float)((m >> 8) & 255u);
                v.z *= gate_deriv(gy[u].z, gc) * (float)((m >> 16) & 255u);
                v.w *= gate_deriv(gy[u].w, gc) * (float)(m >> 24);
                const bool live = oo[u] != OOB / 4;
                buf_store4(v, rs_out, live ? oo[u] * 4u : OOB);
                vmax = live ? fmaxf(vmax, amax4(v)) : vmax;
            }
        }
        S8STAMP(6);
    }
#ifdef S8_STAMPS
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int k = 0; k < 7; ++k) g_s8_stamps[k] = s8ph[k];
        g_s8_stamps[7] = (n_tiles - 1) / gridDim.x + 1;
    }
#endif
    if (g.amax_out != nullptr) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
        if (lane == 0) wmax[wave] = vmax;
        __syncthreads();
        if (threadIdx.x == 0)
            __hip_atomic_fetch_max(g.amax_out + (blockIdx.x & (AMAX_N - 1)),
                                   __builtin_bit_cast(unsigned, fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
    }
}

// nn.Conv2d / nn.ConvTranspose2d weights [a][b][taps] -> [q][tap][c]; (q, c) = (a, b) for the Conv2d-forward direction,
// (b, a) for the transposed one
__global__ __launch_bounds__(256) void conv64_weight_prep_kernel(const float *__restrict__ wt, float *__restrict__ out, int q_count,
                                                                  int c_count, int taps, int transposed, unsigned *__restrict__ amax_zero,
                                                                  unsigned *__restrict__ w_amax) {
    const int i = blockIdx.x * 256 + threadIdx.x;                      // output index (q, tap, c)
    if (amax_zero != nullptr && i < AMAX_N) amax_zero[i] = 0u;          // (the convolution behind this launch folds its maxima in)
    if (w_amax != nullptr && blockIdx.x == 0) {                         // max |wt| for the fp16 kernel's weight scale: workgroup 0 walks the tensor
        __shared__ float wm[4];
        float m = 0.f;
        for (int e = threadIdx.x; e < q_count * taps * c_count; e += 256) m = fmaxf(m, fabsf(wt[e]));
        m = wave_max(m);
        if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) *w_amax = __builtin_bit_cast(unsigned, fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])));
    }
    if (i >= q_count * taps * c_count) return;
    const int c = i % c_count, tap = (i / c_count) % taps, q = i / (c_count * taps);
    out[i] = transposed ? wt[((int64_t)c * q_count + q) * taps + tap] : wt[((int64_t)q * c_count + c) * taps + tap];
}

// The re-ordered copy lives in the CALLER's workspace (arvae_link_ws_floats): every launch re-creates it on the caller's
// stream right before the convolution.
// (conv64s.hip: the row-staged kernel of the 64 <-> 64 channel layers and its split weights)
int64_t conv64s_ws_floats();
bool conv64s_fits(const arvae_link_t *l, bool up);
int conv64s_run(const Operand &src, int n, int sh, int sw, int oh, int ow, int q, int sgn, int off, const float *wt, bool transposed,
                const float *bias, int act, const uint8_t *mask, float *out, float *ws, hipStream_t s, const char *what, const GateOp *gate,
                const unsigned *amax_in, unsigned *amax_out, bool prepped);

int64_t conv64_ws_floats(const arvae_link_t *l) {
    const int64_t packed = ((int64_t)l->kh * l->kw * l->chi * l->clo + 3) / 4 * 4 + 4;     // (+ the weights' maximum, conv_rows_h2_kernel)
    const bool staged = conv64s_fits(l, false) || conv64s_fits(l, true);
    return staged && conv64s_ws_floats() > packed ? conv64s_ws_floats() : packed;
}

static bool plain_op(const Operand &o) { return o.y == nullptr || (o.act == ARVAE_ACT_NONE && o.mask == nullptr); }

// links this file serves: k x k (<= 16 taps), stride 1, 32 | channels on the reduction side, channels-last, no permutation
bool conv64_fits(const arvae_link_t *l, bool up) {
    static const bool off = diag_env("ARVAE_CONV64_GENERIC") != nullptr;
    const int red = up ? l->clo : l->chi, outc = up ? l->chi : l->clo;
    // narrow outputs (the 64 -> 8 layers) waste MFMA columns but these products are bound by the gather, not the MFMA
    return !off && l->stride == 1 && l->kh * l->kw <= 16 && l->kh * l->kw > 1 && red % 4 == 0 && red >= 8 &&
           (l->kh * l->kw * red) % RG_R == 0 && outc >= 4 && red <= 128 && outc <= 128 && l->hi_perm_c == 0 && l->lo_perm_c == 0;
}

static int launch_conv_rows(ConvRows g, const float *wt, bool transposed, float *packed, hipStream_t s, const char *what,
                            unsigned *amax_out = nullptr, const unsigned *amax_in = nullptr) {
    const int taps = g.kh * g.kw, wcount = g.q * taps * g.cs;
    g.amax_out = amax_out;
    static const bool no_h2 = diag_env("ARVAE_CONV_ROWS_X3") != nullptr;        // A/B: the three-term bf16 kernel
    const bool h2 = !no_h2 && amax_in != nullptr && plain_op(g.src);            // (one float behind the packed weights holds max |wt|)
    g.amax_in = amax_in;
    g.w_amax = reinterpret_cast<const unsigned *>(packed + wcount);
    if (packed == nullptr || (reinterpret_cast<uintptr_t>(packed) & 15) != 0)
        return fail(ARVAE_E_INVALID, "%s: needs arvae_link_ws_floats() floats of 16-byte aligned workspace for the re-ordered weights", what);
    ARVAE_LAUNCH(conv64_weight_prep_kernel, dim3(((wcount > AMAX_N ? wcount : AMAX_N) + 255) / 256), dim3(256), 0, s, wt, packed, g.q, g.cs, taps,
                 transposed ? 1 : 0, amax_out, h2 ? reinterpret_cast<unsigned *>(packed + wcount) : nullptr);
    g.wt = packed;
    const int M = g.n * g.oh * g.ow;
    const dim3 grid((M + C64_TP - 1) / C64_TP, (g.q + C64_TQ - 1) / C64_TQ);
    static const bool no_s8 = diag_env("ARVAE_CONV_S8_GATHER") != nullptr;      // A/B: the gathering kernel for 8-channel sources too
    const int span_rows = 63 / g.ow + 2 + 3;                                    // source rows a 64-pixel tile can touch
    if (h2 && !no_s8 && g.cs == 8 && g.kh == 4 && g.kw == 4 && g.q <= 64 && (g.q & 3) == 0 && span_rows * g.sw <= S8_PIX &&
        (reinterpret_cast<uintptr_t>(g.out) & 15) == 0 && (int64_t)M * g.q * 4 < 0x7fff0000ll) {       // (32-bit byte offsets into the output)
        ConvRows p = g;
        p.src.y = nullptr;
        const int tiles_per_img = (g.oh * g.ow + 63) / 64, n_tiles = g.n * tiles_per_img;
        // as many persistent workgroups as the chip holds AT ONCE (asked of the runtime: a grid of three per CU ran as two rounds when
        // only two were resident -- workgroup 0 was done after 33 of the launch's 56 us)
        static const int per_cu = [] {
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)conv_s8_h2_kernel, 256, 0) != hipSuccess || n < 1) {
                (void)hipGetLastError();
                n = 2;
            }
            if (const char *e = diag_env("ARVAE_S8_PER_CU")) n = atoi(e) > 0 ? atoi(e) : n;
            return n;
        }();
        const int slots = per_cu * device_cu_count();
        ARVAE_LAUNCH(conv_s8_h2_kernel, dim3(n_tiles < slots ? n_tiles : slots), dim3(256), 0, s, p, tiles_per_img, n_tiles);
    } else if (h2) {
        ConvRows p = g;
        p.src.y = nullptr;
        ARVAE_LAUNCH(conv_rows_h2_kernel, grid, dim3(256), 0, s, p);
    } else if (plain_op(g.src)) {
        ConvRows p = g;
        p.src.y = nullptr;
        ARVAE_LAUNCH(conv_rows_x3_kernel<true>, grid, dim3(256), 0, s, p);
    } else {
        ARVAE_LAUNCH(conv_rows_x3_kernel<false>, grid, dim3(256), 0, s, g);
    }
    return check_launch(what);
}

// lo[n][lh][lw][clo] = act(conv(hi) + bias) * mask     (Conv2d forward / ConvTranspose2d data gradient)
int conv64_down(const arvae_link_t *l, const Operand &hi, const float *wt, const float *bias, int act, const uint8_t *mask,
                float *lo, float *ws, hipStream_t s, const GateOp *gate, const unsigned *amax_in, unsigned *amax_out, float *prepped) {
    // timeline labels tell the 64 -> 64 launches from the ones with a narrow side (64 -> 8): different kernels, 3x apart
    const char *what = (l->chi >= 64 && l->clo >= 64) ? "conv64_down(wide)" : "conv64_down(narrow)";
    if (conv64s_fits(l, false))
        return conv64s_run(hi, l->n, l->hh, l->hw, l->lh, l->lw, l->clo, 1, -l->pad, wt, false, bias, act, mask, lo, prepped ? prepped : ws, s, what, gate,
                           amax_in, amax_out, prepped != nullptr);
    ConvRows g{};
    if (gate != nullptr) g.gate = *gate;
    g.src = hi; g.n = l->n; g.sh = l->hh; g.sw = l->hw; g.cs = l->chi;
    g.oh = l->lh; g.ow = l->lw; g.q = l->clo;
    g.kh = l->kh; g.kw = l->kw; g.sgn = 1; g.off = -l->pad;
    g.bias = bias; g.mask = mask; g.act = act; g.out = lo;
    return launch_conv_rows(g, wt, false, ws, s, what, amax_out, amax_in);          // wt[clo][chi][ky][kx]: q = clo, c = chi
}

// hi[n][hh][hw][chi] = act(convT(lo) + bias) * mask    (ConvTranspose2d forward / Conv2d data gradient)
int conv64_up(const arvae_link_t *l, const Operand &lo, const float *wt, const float *bias, int act, const uint8_t *mask,
              float *hi, float *ws, hipStream_t s, const GateOp *gate, const unsigned *amax_in, unsigned *amax_out, float *prepped) {
    const char *what = (l->chi >= 64 && l->clo >= 64) ? "conv64_up(wide)" : "conv64_up(narrow)";
    if (conv64s_fits(l, true))
        return conv64s_run(lo, l->n, l->lh, l->lw, l->hh, l->hw, l->chi, -1, l->pad, wt, true, bias, act, mask, hi, prepped ? prepped : ws, s, what, gate,
                           amax_in, amax_out, prepped != nullptr);
    ConvRows g{};
    if (gate != nullptr) g.gate = *gate;
    g.src = lo; g.n = l->n; g.sh = l->lh; g.sw = l->lw; g.cs = l->clo;
    g.oh = l->hh; g.ow = l->hw; g.q = l->chi;
    g.kh = l->kh; g.kw = l->kw; g.sgn = -1; g.off = l->pad;
    g.bias = bias; g.mask = mask; g.act = act; g.out = hi;
    return launch_conv_rows(g, wt, true, ws, s, what, amax_out, amax_in);             // wt[clo][chi][ky][kx]: q = chi, c = clo
}

// ---- weight gradient ----------------------------------------------------------------------------------------------
// dW[clo][chi][tap] = sum over lo pixels p of LO[p][clo] * HI[p + tap - pad][chi]  (LO / HI = the link's two operands,
// either of which may be a gradient operand).  Per tap a [clo x chi] product with the pixels as the reduction axis: both
// operands are "K x rows" tiles (32 pixels x 64 channels, coalesced 16-byte loads, HI rows gathered at the tap's
// offset) read through the transposing LDS read.  blockIdx.x = tap, blockIdx.y = a slice of C64_WG_SLICE pixels whose
// partial [taps][clo][chi] goes to the workspace; conv64_wgrad_reduce_kernel adds the slices in order.
constexpr int C64_WG_SLICE = 4096;

struct ConvWgrad {
    Operand lo, hi;
    int n, lh, lw, clo, hh, hw, chi, kh, kw, pad;
    float *ws;                   // [slices][taps][clo][chi]
    // the paired-rows kernel only: bias gradient riding along.  bias_side 1: column sums of lo (the workgroups of chi half 0),
    // 2: of hi (each chi half its own 32 channels); partials at bias_ws[slice][64], added to dbias by the reduction launch
    float *bias_ws;
    int bias_side;
};

template <bool PLAIN_LO, bool PLAIN_HI>
__global__ __launch_bounds__(256) void conv_wgrad_x3_kernel(ConvWgrad g) {
    typedef X3Plane<RG_KROWS, 64> Plane;
    __shared__ __attribute__((aligned(16))) unsigned short As[3 * Plane::PLANE];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[3 * Plane::PLANE];
    const int lane = threadIdx.x & 63, half = lane >> 5, rc = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wp = wave & 1, wq = wave >> 1;
    const int tap = blockIdx.x, ky = tap / g.kw, kx = tap - ky * g.kw;
    const int M = g.n * g.lh * g.lw;
    const int pbeg = blockIdx.y * C64_WG_SLICE, pend = min(M, pbeg + C64_WG_SLICE);
    // slot i: pixel row r = idx / 16 of the chunk, channels 4 (idx % 16) ..
    int r_of[2], c4[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = threadIdx.x + 256 * i;
        r_of[i] = idx / 16; c4[i] = 4 * (idx % 16);
    }
    float4 va[2], vb[2];
    auto load = [&](int pix0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int p = pix0 + r_of[i];
            const bool pok = p < pend;
            const int pc = pok ? p : pbeg;
            va[i] = load_src4<PLAIN_LO>(g.lo, (int64_t)pc * g.clo + c4[i], pok && c4[i] < g.clo);
            const int img = pc / (g.lh * g.lw), rem = pc - img * g.lh * g.lw;
            const int ly = rem / g.lw, lx = rem - ly * g.lw;
            const int hy = ly - g.pad + ky, hx = lx - g.pad + kx;
            const bool ok = pok && hy >= 0 && hy < g.hh && hx >= 0 && hx < g.hw && c4[i] < g.chi;
            vb[i] = load_src4<PLAIN_HI>(g.hi, (((int64_t)img * g.hh + hy) * g.hw + hx) * g.chi + c4[i], ok);
        }
    };
    f32x16c acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    load(pbeg);
    const int abase = Plane::lane_base(wp), bbase = Plane::lane_base(wq);
    for (int pix0 = pbeg; pix0 < pend; pix0 += RG_R) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            Plane::commit(As, threadIdx.x + 256 * i, va[i]);
            Plane::commit(Bs, threadIdx.x + 256 * i, vb[i]);
        }
        __syncthreads();
        if (pix0 + RG_R < pend) load(pix0 + RG_R);
#pragma unroll
        for (int s = 0; s < RG_R / 16; ++s) {
            const rg_bf16x8 ah = Plane::operand(As, abase, 0, s), am = Plane::operand(As, abase, 1, s), al = Plane::operand(As, abase, 2, s);
            const rg_bf16x8 bh = Plane::operand(Bs, bbase, 0, s), bm = Plane::operand(Bs, bbase, 1, s), bl = Plane::operand(Bs, bbase, 2, s);
            X3_MFMA6(acc, ah, am, al, bh, bm, bl);
        }
    }
    float *out = g.ws + ((int64_t)blockIdx.y * g.kh * g.kw + tap) * g.clo * g.chi;
    const int q = 32 * wq + rc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int p = 32 * wp + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (p < g.clo && q < g.chi) out[p * g.chi + q] = acc[r];
    }
}

// dwt[clo][chi][tap] += sum_slices ws[slice][tap][clo][chi]; 4 lanes per element, fixed order
__global__ __launch_bounds__(256) void conv64_wgrad_reduce_kernel(const float *__restrict__ ws, int slices, int taps, int clo, int chi,
                                                                   float *__restrict__ dwt, const float *__restrict__ bias_ws = nullptr,
                                                                   int nbias = 0, float *__restrict__ dbias = nullptr) {
    // elements [count, count + nbias): the bias partials the paired-rows kernel left at bias_ws[slice][64]
    const int count = taps * clo * chi, total = count + nbias;
    const int i = (blockIdx.x * 256 + threadIdx.x) >> 2, q4 = threadIdx.x & 3;
    const bool is_bias = i >= count;
    const int ic = i < total ? (is_bias ? i - count : i) : 0;
    const float *src = is_bias ? bias_ws : ws;
    const int64_t pitch = is_bias ? 64 : count;
    float s = 0.f;
    for (int z0 = q4; z0 < slices; z0 += 32) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int z = z0 + 4 * u;
            const float v = src[(int64_t)(z < slices ? z : 0) * pitch + ic];
            t[u] = z < slices ? v : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += t[u];
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (q4 != 0 || i >= total) return;
    if (is_bias) {
        dbias[ic] += s;
        return;
    }
    const int c = i % chi, a = (i / chi) % clo, tap = i / (chi * clo);       // ws index (tap, clo, chi)
    dwt[((int64_t)a * chi + c) * taps + tap] += s;
}

// ---- weight gradient, row-staged ------------------------------------------------------------------------------------
// The kernel above fetches every operand value once per tap.  Here a workgroup walks whole images row by row: the
// reduction chunk is ONE lo row (lw <= 32 pixels, zero padded), staged in LDS as the A image, and the hi rows it meets
// (ly - pad + ky, ky < kh) live in a five-slot ring of row images that gains one row per step -- every operand value
// is fetched and split once.  Wave = ky; the kx taps are the same hi row image read kx rows further down (the
// transposing read takes any row offset), so a wave holds 2 x kw accumulator tiles: [clo half][kx] of 32 x 32, for one
// 32-channel half of chi (blockIdx.x).  Next row's operands are fetched into registers while this row's MFMAs run; one
// barrier per row.
constexpr int WR_HROWS = 36;                                   // hi row image: 32 + kw - 1 pixel rows, padded
constexpr int WR_RING = 5;
constexpr int WR_APLANE = RG_R * RG_TRP, WR_HPLANE = WR_HROWS * RG_TRP;

__device__ __forceinline__ rg_bf16x8 wr_tr_operand(const unsigned short *p) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(3))) s16x4 *lds_ptr;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + 4 * RG_TRP));
    return __builtin_bit_cast(rg_bf16x8, (s16x8)__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}
__device__ __forceinline__ void wr_commit(unsigned short *d, int plane, const float4 &v) {
    unsigned h0, m0, l0, h1, m1, l1;
    rg_split3(v.x, v.y, h0, m0, l0);
    rg_split3(v.z, v.w, h1, m1, l1);
    *reinterpret_cast<uint2 *>(d) = uint2{h0, h1};
    *reinterpret_cast<uint2 *>(d + plane) = uint2{m0, m1};
    *reinterpret_cast<uint2 *>(d + 2 * plane) = uint2{l0, l1};
}

template <bool PLAIN_LO, bool PLAIN_HI, int KW>
__global__ __launch_bounds__(256) void conv_wgrad_rows_x3_kernel(ConvWgrad g, int img_per_wg) {
    extern __shared__ __attribute__((aligned(16))) unsigned short wr_lds[];
    unsigned short *lo_img = wr_lds;                            // [2][3][32][RG_TRP]
    unsigned short *hi_ring = wr_lds + 2 * 3 * WR_APLANE;       // [WR_RING][3][WR_HROWS][RG_TRP]
    const int lane = threadIdx.x & 63, half = lane >> 5, rc = lane & 31;
    const int ky = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ch0 = 32 * blockIdx.x;                            // this workgroup's chi half
    const int n0 = blockIdx.y * img_per_wg, n1 = min(g.n, n0 + img_per_wg);
    const int na = g.clo > 32 ? 2 : 1;                          // clo halves in use

    // gather slots.  lo row: pixel r = idx / 16, channels 4 (idx % 16); hi row: pixel row r = idx / 8, channels ch0 + 4 (idx % 8)
    const int lo_r[2] = {(int)threadIdx.x / 16, (int)threadIdx.x / 16 + 16};
    const int lo_c = 4 * (threadIdx.x % 16);
    const int hi_r[2] = {(int)threadIdx.x / 8, (int)threadIdx.x / 8 + 32};
    const int hi_c = 4 * (threadIdx.x % 8);
    float4 vlo[2], vhi[2];
    auto fetch_lo = [&](int n, int ly) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool ok = lo_r[i] < g.lw && lo_c < g.clo;
            vlo[i] = load_src4<PLAIN_LO>(g.lo, (((int64_t)n * g.lh + ly) * g.lw + lo_r[i]) * g.clo + lo_c, ok);
        }
    };
    auto fetch_hi = [&](int n, int hy) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hx = hi_r[i] - g.pad;
            const bool ok = hi_r[i] < WR_HROWS && hy >= 0 && hy < g.hh && hx >= 0 && hx < g.hw && ch0 + hi_c < g.chi;
            vhi[i] = load_src4<PLAIN_HI>(g.hi, (((int64_t)n * g.hh + hy) * g.hw + hx) * g.chi + ch0 + hi_c, ok);
        }
    };
    auto commit_lo = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) wr_commit(lo_img + buf * 3 * WR_APLANE + lo_r[i] * RG_TRP + lo_c, WR_APLANE, vlo[i]);
    };
    auto commit_hi = [&](int hy) {
        unsigned short *img = hi_ring + ((hy + 2 * WR_RING) % WR_RING) * 3 * WR_HPLANE;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (hi_r[i] < WR_HROWS) wr_commit(img + hi_r[i] * RG_TRP + hi_c, WR_HPLANE, vhi[i]);
    };

    f32x16c acc[2][KW];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int k = 0; k < KW; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][k][i] = 0.f;
    // transposed-read lane offsets inside an image (x3tile.h X3Plane<RG_KROWS>::lane_base with w = 0)
    const int g16 = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int tr_base = (8 * (g16 >> 1) + tq) * RG_TRP + 16 * (g16 & 1) + 4 * tp;

    for (int n = n0; n < n1; ++n) {
        __syncthreads();                                        // the previous image's last row is done with the ring
        for (int k = 0; k < g.kh - 1; ++k) {                    // rows ky = 0 .. kh-2 of the first lo row, blocking
            fetch_hi(n, -g.pad + k);
            commit_hi(-g.pad + k);
        }
        fetch_lo(n, 0);
        fetch_hi(n, -g.pad + g.kh - 1);
        for (int ly = 0; ly < g.lh; ++ly) {
            commit_lo(ly & 1);
            commit_hi(ly - g.pad + g.kh - 1);
            __syncthreads();
            if (ly + 1 < g.lh) {
                fetch_lo(n, ly + 1);
                fetch_hi(n, ly + 1 - g.pad + g.kh - 1);
            }
            if (ky < g.kh) {
                const unsigned short *ab = lo_img + (ly & 1) * 3 * WR_APLANE + tr_base;
                const int hy = ly - g.pad + ky;
                const unsigned short *hb = hi_ring + ((hy + 2 * WR_RING) % WR_RING) * 3 * WR_HPLANE + tr_base;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    rg_bf16x8 a3[2][3];
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        a3[0][t] = wr_tr_operand(ab + t * WR_APLANE + 16 * s * RG_TRP);
                        a3[1][t] = wr_tr_operand(ab + t * WR_APLANE + 16 * s * RG_TRP + 32);
                    }
#pragma unroll
                    for (int kx = 0; kx < KW; ++kx) {
                        rg_bf16x8 b3[3];
#pragma unroll
                        for (int t = 0; t < 3; ++t) b3[t] = wr_tr_operand(hb + t * WR_HPLANE + (16 * s + kx) * RG_TRP);
                        X3_MFMA6(acc[0][kx], a3[0][0], a3[0][1], a3[0][2], b3[0], b3[1], b3[2]);
                        if (na > 1) { X3_MFMA6(acc[1][kx], a3[1][0], a3[1][1], a3[1][2], b3[0], b3[1], b3[2]); }
                    }
                }
            }
        }
    }
    if (ky < g.kh) {
        const int q = ch0 + rc;
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
            if (kx >= g.kw) break;
            float *out = g.ws + ((int64_t)blockIdx.y * g.kh * g.kw + ky * g.kw + kx) * g.clo * g.chi;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (p < g.clo && q < g.chi) out[p * g.chi + q] = acc[a][kx][r];
                }
        }
    }
}

// ---- weight gradient, row-staged, TWO lo rows per step --------------------------------------------------------------
// conv_wgrad_rows_x3_kernel's reduction chunk is one lo row padded to 32 pixels: at lw = 22 (Morpho-MNIST) 31 % of its MFMAs
// multiply zeros.  Here a step is a PAIR of lo rows laid side by side in one 48-slot image (slots [0, lw) and [lw, 2 lw), the
// rest zero: three 16-pixel k-steps instead of four), for lw <= 24.  The hi operand of slot p is pixel p + kx of hi row
// ly - pad + ky for the first row and pixel p - lw + kx of the next hi row for the second: the transposing read takes a
// per-lane address, so a lane picks its ring image by slot.  The ring holds 8 hi row images of 32 channels (96-byte pitch:
// 83 KB); two lo rows and two hi rows are fetched into registers while a pair's 144 MFMAs per wave run; one barrier per pair;
// the three hi rows an image starts with arrive in ONE round trip.
constexpr int WP_SLOTS = 48, WP_RING = 8, WP_HTRP = 48;
constexpr int WP_APLANE = WP_SLOTS * RG_TRP, WP_HPLANE = WR_HROWS * WP_HTRP;

__device__ __forceinline__ rg_bf16x8 wp_tr_operand(const unsigned short *p0, const unsigned short *p1) {
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(3))) s16x4 *lds_ptr;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p0);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)p1);
    return __builtin_bit_cast(rg_bf16x8, (s16x8)__builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

// (on the fp16 MFMA with scaled two-term operands, conv32_common.h: three partial products, the operands' maxima in AMAX
// arrays; 339 / 217 us per launch with six bf16 products through round 3)
__device__ __forceinline__ void wp_commit_h2(unsigned short *d, int plane, const float4 &v, float sc) {
    unsigned h0, l0, h1, l1;
    split_pair_h2(v.x, v.y, sc, h0, l0);
    split_pair_h2(v.z, v.w, sc, h1, l1);
    *reinterpret_cast<uint2 *>(d) = uint2{h0, h1};
    *reinterpret_cast<uint2 *>(d + plane) = uint2{l0, l1};
}
// acc += A . B for one 16-deep k-step: (l, h), (h, l), (h, h)
#define H2_MFMA3(ACC, AH, AL, BH, BL)                                              \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AL, BH, ACC, 0, 0, 0);            \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BL, ACC, 0, 0, 0);            \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AH, BH, ACC, 0, 0, 0)

// TWO workgroups per CU (72 KB of LDS and <= 128 + 128 registers each): with one, a SIMD held a single wave and every transposed
// operand read stalled its MFMAs -- the launch ran at 13 % of the fp16 MFMA rate.  The lo pair image is single-buffered for that
// (a second barrier per pair keeps the next pair's commit off the image still being read).
template <bool PLAIN_LO, bool PLAIN_HI>
__global__ __launch_bounds__(256, 2) void conv_wgrad_pairs_h2_kernel(ConvWgrad g, int img_per_wg, const unsigned *amax_lo, const unsigned *amax_hi) {
    extern __shared__ __attribute__((aligned(16))) unsigned short wp_lds[];
    unsigned short *lo_img = wp_lds;                            // [2][WP_SLOTS][RG_TRP]
    unsigned short *hi_ring = wp_lds + 2 * WP_APLANE;           // [WP_RING][2][WR_HROWS][WP_HTRP]
    const AmaxLoad al_l = amax_issue(amax_lo), al_h = amax_issue(amax_hi);
    const int lane = threadIdx.x & 63, half = lane >> 5, rc = lane & 31;
    const int ky = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ch0 = 32 * blockIdx.x;                            // this workgroup's chi half
    const int n0 = blockIdx.y * img_per_wg, n1 = min(g.n, n0 + img_per_wg);
    const int na = g.clo > 32 ? 2 : 1;                          // clo halves in use
    const int pairs = (g.lh + 1) >> 1;

    // zero both lo pair images once: slots >= 2 lw are never written again
    for (int i = threadIdx.x; i < 2 * WP_APLANE / 2; i += 256) reinterpret_cast<unsigned *>(lo_img)[i] = 0u;
    const Pow2 sc_l = amax_scale(al_l), sc_h = amax_scale(al_h);

    // gather slots.  lo row: pixel idx / 16, channels 4 (idx % 16); hi row: pixel row idx / 8, channels ch0 + 4 (idx % 8)
    const int lo_r[2] = {(int)threadIdx.x / 16, (int)threadIdx.x / 16 + 16};
    const int lo_c = 4 * (threadIdx.x % 16);
    const int hi_r[2] = {(int)threadIdx.x / 8, (int)threadIdx.x / 8 + 32};
    const int hi_c = 4 * (threadIdx.x % 8);
    float4 vlo[2][2], vhi[2][2];
    // A plain operand's rows come through raw buffer loads: the row (image, y) is a scalar byte offset -- an absent row selects the
    // empty range --, the thread's place in the row one per-lane offset fixed for the launch (beyond the range for a slot outside
    // the row): no 64-bit address and no lane test per load (conv64_wgrad bounds the tensors at 2 GB).
    constexpr unsigned OOBV = 0xfffffff0u;
    const __amdgpu_buffer_rsrc_t rs_none = make_rsrc(g.lo.v, 0);
    const __amdgpu_buffer_rsrc_t rs_lo = make_rsrc(g.lo.v, (int64_t)g.n * g.lh * g.lw * g.clo * 4);
    const __amdgpu_buffer_rsrc_t rs_hi = make_rsrc(g.hi.v, (int64_t)g.n * g.hh * g.hw * g.chi * 4);
    unsigned lo_off[2], hi_off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        lo_off[i] = (lo_r[i] < g.lw && lo_c < g.clo) ? (unsigned)((lo_r[i] * g.clo + lo_c) * 4) : OOBV;
        const int hx = hi_r[i] - g.pad;
        hi_off[i] = (hi_r[i] < WR_HROWS && hx >= 0 && hx < g.hw && ch0 + hi_c < g.chi) ? (unsigned)((hx * g.chi + ch0 + hi_c) * 4) : OOBV;
    }
    auto fetch_lo = [&](int which, int n, int ly) __attribute__((always_inline)) {
        if (PLAIN_LO) {
            const __amdgpu_buffer_rsrc_t rs = ly < g.lh ? rs_lo : rs_none;
            const int so = ((n * g.lh + ly) * g.lw * g.clo) * 4;
#pragma unroll
            for (int i = 0; i < 2; ++i) vlo[which][i] = buf_load4s(rs, lo_off[i], so);
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool ok = ly < g.lh && lo_r[i] < g.lw && lo_c < g.clo;
            vlo[which][i] = load_src4<PLAIN_LO>(g.lo, (((int64_t)n * g.lh + ly) * g.lw + lo_r[i]) * g.clo + lo_c, ok);
        }
    };
    auto fetch_hi = [&](int which, int n, int hy) __attribute__((always_inline)) {
        if (PLAIN_HI) {
            const bool row = hy >= 0 && hy < g.hh;
            const __amdgpu_buffer_rsrc_t rs = row ? rs_hi : rs_none;
            const int so = row ? ((n * g.hh + hy) * g.hw * g.chi) * 4 : 0;
#pragma unroll
            for (int i = 0; i < 2; ++i) vhi[which][i] = buf_load4s(rs, hi_off[i], so);
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hx = hi_r[i] - g.pad;
            const bool ok = hi_r[i] < WR_HROWS && hy >= 0 && hy < g.hh && hx >= 0 && hx < g.hw && ch0 + hi_c < g.chi;
            vhi[which][i] = load_src4<PLAIN_HI>(g.hi, (((int64_t)n * g.hh + hy) * g.hw + hx) * g.chi + ch0 + hi_c, ok);
        }
    };
    // bias gradient riding along: every lo / hi row passes through a commit exactly once (rows outside the tensor as zeros)
    float4 bsum[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const bool bias_lo = g.bias_side == 1 && blockIdx.x == 0, bias_hi = g.bias_side == 2;
    auto commit_lo = [&](int which, int buf) __attribute__((always_inline)) {
        if (bias_lo) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bsum[i].x += vlo[which][i].x; bsum[i].y += vlo[which][i].y; bsum[i].z += vlo[which][i].z; bsum[i].w += vlo[which][i].w;
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (lo_r[i] < g.lw)
                wp_commit_h2(lo_img + (which * g.lw + lo_r[i]) * RG_TRP + lo_c, WP_APLANE, vlo[which][i], sc_l.s);
    };
    auto ring_of = [&](int hy) { return hi_ring + ((hy + 4 * WP_RING) & (WP_RING - 1)) * 2 * WP_HPLANE; };
    auto commit_hi = [&](int which, int hy) __attribute__((always_inline)) {
        unsigned short *img = ring_of(hy);
        if (bias_hi) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bsum[i].x += vhi[which][i].x; bsum[i].y += vhi[which][i].y; bsum[i].z += vhi[which][i].z; bsum[i].w += vhi[which][i].w;
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (hi_r[i] < WR_HROWS) wp_commit_h2(img + hi_r[i] * WP_HTRP + hi_c, WP_HPLANE, vhi[which][i], sc_h.s);
    };

    f32x16c acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[a][k][i] = 0.f;
    // transposed-read lane geometry: a lane addresses pixel slot 16 s + 8 (g16 >> 1) + tq (+ 4 for the second read) and
    // columns 16 (g16 & 1) + 4 tp
    const int g16 = lane >> 4, tq = (lane & 15) >> 2, tp = lane & 3;
    const int a_base = (8 * (g16 >> 1) + tq) * RG_TRP + 16 * (g16 & 1) + 4 * tp;
    int h_off[3][2];                                            // bf16 offset of the slot's pixel inside ITS hi row image (+ kx rows)
    unsigned h_sel = 0;                                         // bit 2 s + r: the slot belongs to the pair's second row
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int slot = 16 * s + 8 * (g16 >> 1) + tq + 4 * r;
            const bool second = slot >= g.lw && slot < 2 * g.lw;
            const int px = slot < g.lw ? slot : (second ? slot - g.lw : 0);             // slots >= 2 lw: lo is zero, any finite pixel will do
            h_off[s][r] = px * WP_HTRP + 16 * (g16 & 1) + 4 * tp;
            h_sel |= (second ? 1u : 0u) << (2 * s + r);
        }

    for (int n = n0; n < n1; ++n) {
        __syncthreads();                                        // the previous image's last pair is done with the ring and the lo buffers
        // hi rows -pad .. -pad + 2 of the image: one round trip, then the first pair's own two rows + its lo rows in flight
        {
            float4 v3[3][2];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                fetch_hi(0, n, -g.pad + k);
                v3[k][0] = vhi[0][0]; v3[k][1] = vhi[0][1];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                vhi[0][0] = v3[k][0]; vhi[0][1] = v3[k][1];
                commit_hi(0, -g.pad + k);
            }
        }
        fetch_lo(0, n, 0);
        fetch_lo(1, n, 1);
        fetch_hi(0, n, -g.pad + 3);
        fetch_hi(1, n, -g.pad + 4);
        for (int pi = 0; pi < pairs; ++pi) {
            const int ly = 2 * pi;
            if (pi > 0) __syncthreads();                        // the previous pair's MFMAs are done with the lo image
            commit_lo(0, 0);
            commit_lo(1, 0);
            commit_hi(0, ly - g.pad + 3);
            commit_hi(1, ly - g.pad + 4);
            __syncthreads();
            if (pi + 1 < pairs) {
                fetch_lo(0, n, ly + 2);
                fetch_lo(1, n, ly + 3);
                fetch_hi(0, n, ly + 2 - g.pad + 3);
                fetch_hi(1, n, ly + 2 - g.pad + 4);
            }
            if (ky < g.kh) {
                const unsigned short *ab = lo_img + a_base;
                const unsigned short *hbA = ring_of(ly - g.pad + ky), *hbB = ring_of(ly + 1 - g.pad + ky);
                // the hi operand of (s, kx) is read ONE step ahead of its MFMAs (the compiler placed every group of transposed reads
                // right in front of the MFMAs that use it, with a wait: twelve exposed LDS round trips per pair), the lo operand of
                // s + 1 behind the last MFMAs of s
                f16x8 a2[2][2], b2[2][2];
                auto load_a = [&](int s) __attribute__((always_inline)) {
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const unsigned short *ap = ab + t * WP_APLANE + 16 * s * RG_TRP;
                        a2[0][t] = __builtin_bit_cast(f16x8, wp_tr_operand(ap, ap + 4 * RG_TRP));
                        a2[1][t] = __builtin_bit_cast(f16x8, wp_tr_operand(ap + 32, ap + 32 + 4 * RG_TRP));
                    }
                };
                auto load_b = [&](int buf, int s, int kx) __attribute__((always_inline)) {
                    const unsigned short *h0 = ((h_sel >> (2 * s)) & 1u) ? hbB : hbA, *h1 = ((h_sel >> (2 * s + 1)) & 1u) ? hbB : hbA;
#pragma unroll
                    for (int t = 0; t < 2; ++t)
                        b2[buf][t] = __builtin_bit_cast(f16x8, wp_tr_operand(h0 + t * WP_HPLANE + h_off[s][0] + kx * WP_HTRP, h1 + t * WP_HPLANE + h_off[s][1] + kx * WP_HTRP));
                };
                load_a(0);
                load_b(0, 0, 0);
#pragma unroll
                for (int step = 0; step < 12; ++step) {
                    const int s = step >> 2, kx = step & 3, cur = step & 1;
                    if (step + 1 < 12) load_b(cur ^ 1, (step + 1) >> 2, (step + 1) & 3);
                    __builtin_amdgcn_sched_barrier(0);
                    H2_MFMA3(acc[0][kx], a2[0][0], a2[0][1], b2[cur][0], b2[cur][1]);
                    if (na > 1) { H2_MFMA3(acc[1][kx], a2[1][0], a2[1][1], b2[cur][0], b2[cur][1]); }
                    __builtin_amdgcn_sched_barrier(0);
                    if (kx == 3 && s + 1 < 3) load_a(s + 1);
                }
            }
        }
    }
    if (ky < g.kh) {
        const int q = ch0 + rc;
        const float inv = sc_l.inv * sc_h.inv;                  // accumulators -> fp32 partial sums (exact)
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
            if (kx >= g.kw) break;
            float *out = g.ws + ((int64_t)blockIdx.y * g.kh * g.kw + ky * g.kw + kx) * g.clo * g.chi;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int p = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (p < g.clo && q < g.chi) out[p * g.chi + q] = acc[a][kx][r] * inv;
                }
        }
    }
    if (bias_lo || bias_hi) {
        // a thread holds 4 channels of its pixel rows: lo 16 channel groups x 16 row groups, hi 8 x 32; summed in a fixed order
        __syncthreads();
        float4 *red = reinterpret_cast<float4 *>(wp_lds);
        red[threadIdx.x] = float4{bsum[0].x + bsum[1].x, bsum[0].y + bsum[1].y, bsum[0].z + bsum[1].z, bsum[0].w + bsum[1].w};
        __syncthreads();
        const int groups = bias_lo ? 16 : 8, rows = 256 / groups;
        if ((int)threadIdx.x < groups) {
            float4 tot = red[threadIdx.x];
            for (int j = 1; j < rows; ++j) {
                const float4 v = red[j * groups + threadIdx.x];
                tot.x += v.x; tot.y += v.y; tot.z += v.z; tot.w += v.w;
            }
            const int c = (bias_lo ? 0 : ch0) + 4 * threadIdx.x;
            const int cmax = bias_lo ? g.clo : g.chi;
            float *dst = g.bias_ws + (int64_t)blockIdx.y * 64 + c;
            if (c + 3 < cmax) *reinterpret_cast<float4 *>(dst) = tot;
            else { if (c < cmax) dst[0] = tot.x; if (c + 1 < cmax) dst[1] = tot.y; if (c + 2 < cmax) dst[2] = tot.z; }
        }
    }
}

// images per workgroup: one workgroup per CU (256) when the batch allows it
static int wr_img_per_wg(const arvae_link_t *l) {
    const int halves = (l->chi + 31) / 32;
    const int per = l->n * halves / 256;
    return per < 1 ? 1 : per;
}

static bool conv64_wgrad_rows_fits(const arvae_link_t *l) {
    static const bool off = diag_env("ARVAE_CONV64_WGRAD_TAPS") != nullptr;      // A/B: the per-tap kernel
    return !off && l->kh <= 4 && l->kw == 4 && l->lw <= 32 && l->lw + l->kw - 1 <= WR_HROWS && l->pad <= 4;
}

bool conv64_wgrad_fits(const arvae_link_t *l) {
    static const bool off = diag_env("ARVAE_CONV64_GENERIC") != nullptr;
    return !off && l->stride == 1 && l->kh * l->kw <= 16 && l->kh * l->kw > 1 && l->clo % 4 == 0 && l->chi % 4 == 0 &&
           l->clo >= 4 && l->chi >= 4 && l->clo <= 64 && l->chi <= 64 && (l->clo >= 32 || l->chi >= 32) &&
           l->hi_perm_c == 0 && l->lo_perm_c == 0 &&
           // (the paired-rows kernel addresses plain operands with 32-bit byte offsets)
           (int64_t)l->n * l->lh * l->lw * l->clo * 4 < ((int64_t)1 << 31) && (int64_t)l->n * l->hh * l->hw * l->chi * 4 < ((int64_t)1 << 31);
}

int64_t conv64_wgrad_ws_floats(const arvae_link_t *l) {
    const int64_t M = (int64_t)l->n * l->lh * l->lw;
    const int64_t per = (int64_t)l->kh * l->kw * l->clo * l->chi;
    const int64_t taps_kernel = ((M + C64_WG_SLICE - 1) / C64_WG_SLICE) * per;
    const int ipw_pairs = (wr_img_per_wg(l) + 1) / 2;                                        // (the paired-rows kernel: two workgroups per CU)
    const int64_t rows_kernel = ((l->n + ipw_pairs - 1) / ipw_pairs) * per;
    const int64_t bias_partials = ((l->n + ipw_pairs - 1) / ipw_pairs) * 64;                  // (its bias sums riding along)
    return (taps_kernel > rows_kernel ? taps_kernel : rows_kernel) + bias_partials + 2 * AMAX_N;     // + the two operands' AMAX arrays
}

int conv64_operand_amax(const Operand &x, int64_t count, unsigned *out, hipStream_t s);      // conv64s.hip

int conv64_wgrad(const arvae_link_t *l, const Operand &lo, const Operand &hi, float *dwt, float *ws, hipStream_t s,
                 const unsigned *amax_lo, const unsigned *amax_hi, float *dbias, int bias_side, bool *bias_done) {
    if (bias_done != nullptr) *bias_done = false;
    ConvWgrad g{};
    g.lo = lo; g.hi = hi; g.n = l->n; g.lh = l->lh; g.lw = l->lw; g.clo = l->clo; g.hh = l->hh; g.hw = l->hw; g.chi = l->chi;
    g.kh = l->kh; g.kw = l->kw; g.pad = l->pad; g.ws = ws;
    const int taps = l->kh * l->kw;
    const bool pl = plain_op(lo), ph = plain_op(hi);
    if (pl) g.lo.y = nullptr;
    if (ph) g.hi.y = nullptr;
    static const bool no_pairs = diag_env("ARVAE_CONV64_WGRAD_ROWS") != nullptr;     // diagnostic: one lo row per step
    if (conv64_wgrad_rows_fits(l) && !no_pairs && l->lw <= 24 && l->kh == 4) {
        const int ipw = (wr_img_per_wg(l) + 1) / 2, slices = (l->n + ipw - 1) / ipw;      // two workgroups per CU
        const dim3 grid((l->chi + 31) / 32, slices);
        const size_t lds = (2 * WP_APLANE + WP_RING * 2 * WP_HPLANE) * sizeof(unsigned short);
        static std::once_flag attr;
        std::call_once(attr, [&] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wgrad_pairs_h2_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wgrad_pairs_h2_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wgrad_pairs_h2_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wgrad_pairs_h2_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        });
        // maxima of the operands AS MULTIPLIED: the caller's (plain tensors only) or taken here, behind the partial sums in ws
        unsigned *am = reinterpret_cast<unsigned *>(ws + conv64_wgrad_ws_floats(l) - 2 * AMAX_N);
        // the bias gradient (column sums of lo or of hi) rides in the kernel: both tensors pass through its registers anyway
        static const bool bias_apart = diag_env("ARVAE_CONV64_BIAS_APART") != nullptr;     // A/B: the separate channel-sum launches
        const bool ride = dbias != nullptr && (bias_side == 1 || bias_side == 2) && bias_done != nullptr && !bias_apart &&
                          (bias_side == 1 ? l->clo : l->chi) <= 64;
        if (ride) {
            g.bias_ws = reinterpret_cast<float *>(am) - (int64_t)slices * 64;
            g.bias_side = bias_side;
            *bias_done = true;
        }
        if (amax_lo == nullptr || !pl) {
            if (int rc = conv64_operand_amax(g.lo, (int64_t)l->n * l->lh * l->lw * l->clo, am, s)) return rc;
            amax_lo = am;
        }
        if (amax_hi == nullptr || !ph) {
            if (int rc = conv64_operand_amax(g.hi, (int64_t)l->n * l->hh * l->hw * l->chi, am + AMAX_N, s)) return rc;
            amax_hi = am + AMAX_N;
        }
        if (pl && ph) ARVAE_LAUNCH((conv_wgrad_pairs_h2_kernel<true, true>), grid, dim3(256), lds, s, g, ipw, amax_lo, amax_hi);
        else if (pl) ARVAE_LAUNCH((conv_wgrad_pairs_h2_kernel<true, false>), grid, dim3(256), lds, s, g, ipw, amax_lo, amax_hi);
        else if (ph) ARVAE_LAUNCH((conv_wgrad_pairs_h2_kernel<false, true>), grid, dim3(256), lds, s, g, ipw, amax_lo, amax_hi);
        else ARVAE_LAUNCH((conv_wgrad_pairs_h2_kernel<false, false>), grid, dim3(256), lds, s, g, ipw, amax_lo, amax_hi);
        const int count = taps * l->clo * l->chi, nbias = ride ? (bias_side == 1 ? l->clo : l->chi) : 0;
        ARVAE_LAUNCH(conv64_wgrad_reduce_kernel, dim3(((count + nbias) * 4 + 255) / 256), dim3(256), 0, s, ws, slices, taps, l->clo, l->chi, dwt,
                     g.bias_ws, nbias, dbias);
        return check_launch((l->chi >= 64 && l->clo >= 64) ? "conv64_wgrad(pairs, wide)" : "conv64_wgrad(pairs, narrow)");
    }
    if (conv64_wgrad_rows_fits(l)) {
        const int ipw = wr_img_per_wg(l), slices = (l->n + ipw - 1) / ipw;
        const dim3 grid((l->chi + 31) / 32, slices);
        const size_t lds = (2 * 3 * WR_APLANE + WR_RING * 3 * WR_HPLANE) * sizeof(unsigned short);
        static std::once_flag attr;
        std::call_once(attr, [&] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wgrad_rows_x3_kernel<true, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wgrad_rows_x3_kernel<true, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wgrad_rows_x3_kernel<false, true, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(conv_wgrad_rows_x3_kernel<false, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        });
        if (pl && ph) ARVAE_LAUNCH((conv_wgrad_rows_x3_kernel<true, true, 4>), grid, dim3(256), lds, s, g, ipw);
        else if (pl) ARVAE_LAUNCH((conv_wgrad_rows_x3_kernel<true, false, 4>), grid, dim3(256), lds, s, g, ipw);
        else if (ph) ARVAE_LAUNCH((conv_wgrad_rows_x3_kernel<false, true, 4>), grid, dim3(256), lds, s, g, ipw);
        else ARVAE_LAUNCH((conv_wgrad_rows_x3_kernel<false, false, 4>), grid, dim3(256), lds, s, g, ipw);
        const int count = taps * l->clo * l->chi;
        ARVAE_LAUNCH(conv64_wgrad_reduce_kernel, dim3((count * 4 + 255) / 256), dim3(256), 0, s, ws, slices, taps, l->clo, l->chi, dwt);
        return check_launch("conv64_wgrad(rows)");
    }
    const int slices = (int)(((int64_t)l->n * l->lh * l->lw + C64_WG_SLICE - 1) / C64_WG_SLICE);
    const dim3 grid(taps, slices);
    if (pl && ph) ARVAE_LAUNCH((conv_wgrad_x3_kernel<true, true>), grid, dim3(256), 0, s, g);
    else if (pl) ARVAE_LAUNCH((conv_wgrad_x3_kernel<true, false>), grid, dim3(256), 0, s, g);
    else if (ph) ARVAE_LAUNCH((conv_wgrad_x3_kernel<false, true>), grid, dim3(256), 0, s, g);
    else ARVAE_LAUNCH((conv_wgrad_x3_kernel<false, false>), grid, dim3(256), 0, s, g);
    const int count = taps * l->clo * l->chi;
    ARVAE_LAUNCH(conv64_wgrad_reduce_kernel, dim3((count * 4 + 255) / 256), dim3(256), 0, s, ws, slices, taps, l->clo, l->chi, dwt);
    return check_launch("conv64_wgrad");
}

}  // namespace arvae

#ifdef S8_STAMPS
extern "C" int arvae_debug_s8_stamps(unsigned long long *out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(arvae::g_s8_stamps), sizeof(unsigned long long) * 8) == hipSuccess ? 0 : -1;
}
#endif
