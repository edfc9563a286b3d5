// Shared helpers for the gfx950 kernels of libarvae_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/arvae_hip.h"

namespace arvae {

// ---- error reporting -------------------------------------------------------------------------
extern thread_local char g_last_error[512];

int fail(int code, const char *fmt, ...);
int check_launch(const char *what);
void prof_gap();           // closes the interval before a launch, so that a kernel's profile time excludes the launch gap
bool profiling_active();   // arvae_profile_begin() is recording: keep every kernel on the caller's stream

// every kernel launch of the library: when the opt-in timeline is recording, an event right before the launch separates
// the kernel's own time from whatever the stream was doing (or not doing) before it
#define ARVAE_LAUNCH(...)                  \
    do {                                   \
        ::arvae::prof_gap();               \
        hipLaunchKernelGGL(__VA_ARGS__);   \
    } while (0)

#define ARVAE_REQUIRE(cond, ...)                                   \
    do {                                                           \
        if (!(cond)) return ::arvae::fail(ARVAE_E_INVALID, __VA_ARGS__); \
    } while (0)

// stream of the ABI call in progress on this thread (lets check_launch() timestamp kernels when profiling)
extern thread_local hipStream_t g_cur_stream;
static inline hipStream_t as_stream(arvae_stream_t s) {
    g_cur_stream = reinterpret_cast<hipStream_t>(s);
    return g_cur_stream;
}

// ---- division by a runtime constant (n < 2^31) -----------------------------------------------
struct FastDiv {
    uint32_t d, mul, sh;
    FastDiv() : d(1), mul(0), sh(0) {}
    explicit FastDiv(uint32_t div) : d(div), mul(0), sh(0) {
        if (div > 1) {
            uint32_t shift = 0;
            while ((1ull << shift) < div) ++shift;
            mul = (uint32_t)(((1ull << (31 + shift)) + div - 1) / div);
            sh = shift - 1;
        }
    }
    __device__ __forceinline__ uint32_t div(uint32_t n) const { return d == 1 ? n : (__umulhi(n, mul) >> sh); }
    __device__ __forceinline__ void divmod(uint32_t n, uint32_t &q, uint32_t &r) const {
        q = div(n);
        r = n - q * d;
    }
};

// ---- activations ------------------------------------------------------------------------------
constexpr float kSeluAlpha = 1.6732632423543772f;
constexpr float kSeluScale = 1.0507009873554805f;

__device__ __forceinline__ float act_fwd(float x, int act) {
    if (act == ARVAE_ACT_RELU) return fmaxf(x, 0.f);
    // exp on v_exp_f32 (2^x, 1 ulp; the argument is <= 0 here): expf() is ~30 instructions, and the 41 M SELU outputs of the first
    // Morpho-MNIST layer alone spent ~35 us per step in it
    if (act == ARVAE_ACT_SELU)
        return x > 0.f ? kSeluScale * x : (kSeluScale * kSeluAlpha) * (__builtin_amdgcn_exp2f(x * 1.4426950408889634f) - 1.f);
    return x;
}
// derivative evaluated from the activation OUTPUT y
__device__ __forceinline__ float act_bwd_from_out(float y, int act) {
    if (act == ARVAE_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == ARVAE_ACT_SELU) return y > 0.f ? kSeluScale : y + kSeluScale * kSeluAlpha;
    return 1.f;
}

// the same without control flow (selects on the wave-uniform `act`): for code that must keep many loads in flight
__device__ __forceinline__ float act_fwd_sel(float x, int act) {
    const float relu = fmaxf(x, 0.f);
    const float selu = x > 0.f ? kSeluScale * x : (kSeluScale * kSeluAlpha) * (__builtin_amdgcn_exp2f(fminf(x, 0.f) * 1.4426950408889634f) - 1.f);
    return act == ARVAE_ACT_RELU ? relu : (act == ARVAE_ACT_SELU ? selu : x);
}
__device__ __forceinline__ float act_bwd_from_out_sel(float y, int act) {
    const float relu = y > 0.f ? 1.f : 0.f, selu = y > 0.f ? kSeluScale : y + kSeluScale * kSeluAlpha;
    return act == ARVAE_ACT_RELU ? relu : (act == ARVAE_ACT_SELU ? selu : 1.f);
}

// Epilogue arithmetic in coefficient form, for kernels whose epilogue is bound by its vector instruction count (conv64s.hip,
// conv64.hip's 8-channel kernel, link_gemm.hip's single-channel kernel): no selects on the (launch-uniform) activation kind and
// no branch on the sign (the compiler turns `x > 0 ? a : b(exp)` into an exec-mask branch per value):
//   act(x)  = pos * max(x, 0) + neg * (exp2(min(x, 0) * log2 e) - 1)            [none: x itself, one select]
//   act'(y) = y > 0 ? dpos : y * dslope + dneg          (from the saved output; the keep-mask's factor 2 folded in)
// with the same roundings as common.h's act_fwd / act_bwd_from_out_sel (SELU exponential on v_exp_f32, 1 ulp).
struct ActCoef {
    float pos, neg;
    bool none;
};
__device__ __forceinline__ ActCoef act_coef(int act) {
    return ActCoef{act == ARVAE_ACT_SELU ? kSeluScale : 1.f, act == ARVAE_ACT_SELU ? kSeluScale * kSeluAlpha : 0.f,
                   act != ARVAE_ACT_SELU && act != ARVAE_ACT_RELU};
}
__device__ __forceinline__ float act_fwd_coef(float x, const ActCoef &a) {
    const float e = __builtin_amdgcn_exp2f(fminf(x, 0.f) * 1.4426950408889634f) - 1.f;
    const float r = fmaf(a.neg, e, a.pos * fmaxf(x, 0.f));
    return a.none ? x : r;
}
struct GateCoef { float pos, slope, neg; };
// k2 = 2 with a keep-mask (the saved output is then the kept activation times two: Operand::apply), else 1
__device__ __forceinline__ GateCoef gate_coef(int act, bool masked) {
    const float k2 = masked ? 2.f : 1.f;
    if (act == ARVAE_ACT_SELU) return GateCoef{k2 * kSeluScale, 1.f, k2 * (kSeluScale * kSeluAlpha)};   // (y / k2 + sa) * k2
    if (act == ARVAE_ACT_RELU) return GateCoef{k2, 0.f, 0.f};
    return GateCoef{k2, 0.f, k2};
}
__device__ __forceinline__ float gate_deriv(float y, const GateCoef &c) { return y > 0.f ? c.pos : fmaf(y, c.slope, c.neg); }

// one form for the three epilogues, result *= gate_deriv(y, c) * keep byte: a data gradient's gate (gate_coef), a forward keep-mask
// (d = 2; y reads as zero), neither (d = 1, keep bytes of ones)
__device__ __forceinline__ GateCoef epilogue_coef(const float *gate_y, int gate_act, const uint8_t *gate_mask, const uint8_t *fwd_mask) {
    if (gate_y != nullptr) return gate_coef(gate_act, gate_mask != nullptr);
    return fwd_mask != nullptr ? GateCoef{2.f, 0.f, 2.f} : GateCoef{1.f, 0.f, 1.f};
}

// device view of arvae_operand_t
struct Operand {
    const float *v;
    const float *y;
    const uint8_t *mask;
    int act;
    const float *scale = nullptr;   // optional device scalar multiplying the operand; honoured by the conv_c1 kernels only
    __device__ __forceinline__ float at(int64_t i) const {
        float r = v[i];
        if (y != nullptr) {
            float yy = y[i];
            if (mask != nullptr) {
                r *= 2.f * (float)mask[i];
                yy *= 0.5f;
            }
            r *= act_bwd_from_out(yy, act);
        }
        return r;
    }
    // elements i and i + 1 (i even, the arrays 8-byte aligned) with one load per array
    __device__ __forceinline__ float2 at2(int64_t i) const {
        float2 r = *reinterpret_cast<const float2 *>(v + i);
        if (y != nullptr) {
            float2 yy = *reinterpret_cast<const float2 *>(y + i);
            if (mask != nullptr) {
                const uchar2 m = *reinterpret_cast<const uchar2 *>(mask + i);
                r.x *= 2.f * (float)m.x; r.y *= 2.f * (float)m.y;
                yy.x *= 0.5f; yy.y *= 0.5f;
            }
            r.x *= act_bwd_from_out(yy.x, act); r.y *= act_bwd_from_out(yy.y, act);
        }
        return r;
    }
    // elements i .. i + 3 (i a multiple of 4, the arrays 16-byte aligned)
    __device__ __forceinline__ float4 at4(int64_t i) const {
        float4 r = *reinterpret_cast<const float4 *>(v + i);
        if (y != nullptr) {
            float4 yy = *reinterpret_cast<const float4 *>(y + i);
            if (mask != nullptr) {
                const uchar4 m = *reinterpret_cast<const uchar4 *>(mask + i);
                r.x *= 2.f * (float)m.x; r.y *= 2.f * (float)m.y; r.z *= 2.f * (float)m.z; r.w *= 2.f * (float)m.w;
                yy.x *= 0.5f; yy.y *= 0.5f; yy.z *= 0.5f; yy.w *= 0.5f;
            }
            r.x *= act_bwd_from_out(yy.x, act); r.y *= act_bwd_from_out(yy.y, act);
            r.z *= act_bwd_from_out(yy.z, act); r.w *= act_bwd_from_out(yy.w, act);
        }
        return r;
    }
    // Two-phase access for latency-bound loops: fetch<MODE>() only loads (MODE 0: v; 1: v and y; 2: v, y and the keep-mask),
    // apply<MODE>() only computes -- Operand::at() branches on y / mask / act between its loads, so a loop over at() is one
    // memory round trip per ELEMENT (measured: 8 us per 16 elements in the Linear weight gradient).
    template <int MODE> __device__ __forceinline__ void fetch(int64_t i, float &rv, float &ry, float &rm) const {
        rv = v[i];
        ry = MODE >= 1 ? y[i] : 0.f;
        rm = MODE == 2 ? (float)mask[i] : 1.f;
    }
    template <int MODE> __device__ __forceinline__ float apply(float rv, float ry, float rm) const {
        if (MODE == 2) { rv *= 2.f * rm; ry *= 0.5f; }
        if (MODE >= 1) rv *= act_bwd_from_out_sel(ry, act);
        return rv;
    }
    __host__ __device__ __forceinline__ int mode() const { return y == nullptr ? 0 : (mask == nullptr ? 1 : 2); }
};
// "gate" of a data-gradient epilogue in the general form: the saved OUTPUT y of the layer that produced the tensor the gradient
// belongs to, that layer's activation and its dropout keep-mask.  result *= act'(y) * (mask ? 2 mask : 1): the gradient leaves
// the kernel already w.r.t. the producer's pre-activation.
struct GateOp {
    const float *y = nullptr;
    const uint8_t *mask = nullptr;
    int act = 0;
};
static inline Operand make_operand(const arvae_operand_t *o) { return Operand{o->v, o->y, o->mask, o->act}; }

// relu_bits16: sign bits of a 32-channel ReLU output, the compact form of a "gate" for the data-gradient kernels.
// One uint16 per (pixel, half), index pixel * 2 + half; bit 4g + j <-> channel 8g + 4*half + j, i.e. exactly the 16
// channels a lane (pixel, half) of the 32x32x2 MFMA holds when the weight is the A operand (conv32.hip, conv_c1.hip).

// one pixel of the image reconstruction term (image_vae_trainer.py:623-655): loss and correct-count accumulate,
// dl = d loss / d logit (already divided by the batch size)
// Hardware exp2/log2/rcp based (about 1e-7 relative, ~25 instructions instead of ~120 with the libm versions, which
// made the reconstruction term ALU-bound): log1p(e) = log(u) - ((u - 1) - e) / u with u = 1 + e keeps the bits
// that 1 + e rounds away.
template <int DIST>
__device__ __forceinline__ void recon_elem(float l, float x, float inv_b, float &loss, float &corr, float &dl) {
    const float e = __expf(-fabsf(l));
    const float u = 1.f + e;
    const float r = __frcp_rn(u);
    const float sig = l >= 0.f ? r : e * r;
    if (DIST == ARVAE_RECON_BERNOULLI) {
        loss += fmaxf(l, 0.f) - l * x + (__logf(u) - ((u - 1.f) - e) * r);
        dl = (sig - x) * inv_b;
    } else {
        const float df = sig - x;
        loss += df * df;
        dl = 2.f * df * sig * (1.f - sig) * inv_b;
    }
    corr += ((l >= 0.f) == (x >= 0.5f)) ? 1.f : 0.f;
}

// ---- a barrier for LDS traffic only ------------------------------------------------------------
// __syncthreads() is a workgroup-scope release / acquire of ALL memory: the compiler puts s_waitcnt vmcnt(0) in front of the
// s_barrier, i.e. every wave first waits for its outstanding GLOBAL loads and stores -- including the operands it has just
// prefetched for the next step of a recurrence (stamped in gru_seq_bwd_x3_kernel: 3000 of a step's 8500 cycles sat there).  Where
// the waves of a workgroup exchange data through LDS only, this waits for the LDS queue alone.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// ---- reductions -------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// sum over a 256-thread block; result valid in every thread. `red` = 4 floats of LDS.
__device__ __forceinline__ float block_sum_256(float v, float *red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

}  // namespace arvae
