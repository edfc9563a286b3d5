// Whole-sequence GRU kernels of the MeasureVAE path: all T time steps of one GRU layer (any number of independent
// directions / parameter sets) in ONE launch, forward and backward-through-time.
//
// Reference: nn.GRU inside measurevae/encoder.py:27-34,113-118 (2-layer bidirectional, 24 ticks) and
// measurevae/decoder.py:338-368,436-525 (beat RNN 4 steps, tick RNN 6 steps per beat), gate order r | z | n:
//     r = sigmoid(gi_r + gh_r);  z = sigmoid(gi_z + gh_z);  n = tanh(gi_n + r * gh_n);  h' = (1-z)*n + z*h
// with gi = W_ih x + b_ih computed for all time steps by one dense launch beforehand and gh = W_hh h + b_hh here.
//
// Every batch row is an independent recurrence, so a workgroup owns 16 rows for the whole sequence and never talks
// to another workgroup: H/16 waves, wave w owns hidden units [16w, 16w+16) of all three gates and keeps its
// 3 x 16 x H slice of W_hh in registers (96 VGPRs at H = 128) for all T steps.  Per step: h (16 x H, LDS, double
// buffered) times the slice on v_mfma_f32_16x16x4_f32 (the 16x16 result tile = 4 rows x 1 unit per lane for each
// gate, so the gate math is lane-local), the new h goes back to LDS, one barrier.  gi of the next step is prefetched
// while the MFMAs run.  The backward kernel keeps W_hh^T the same way and carries dL/dh in registers.
//
// The weight gradients (dW_hh = dgh^T h_prev, dW_ih = dgi^T x) and the input gradients are ordinary dense launches
// over all T*R rows afterwards (ops.py).
#include "common.h"

namespace arvae {

constexpr int GRU_SEQ_MAX = 4;

struct GruSeq {
    // forward
    const float *gi;        // [T][R][3H]  (gi_tstride floats between steps; 0 = the same block every step)
    int64_t gi_tstride;
    const float *w_hh;      // [3H][H]
    const float *b_hh;      // [3H]
    const float *h0;        // [R][H] or null (zeros)
    float *h_all;           // h of step t, row r, unit j at h_all[(t*R + r) * h_stride + j]
    int64_t h_stride;
    float *saved;           // [T][R][4][H] : r, z, n, gh_n
    int reverse;            // process t = T-1 .. 0
    // backward
    const float *dh_all;    // gradient w.r.t. h_all, same addressing with dh_stride; may be null
    int64_t dh_stride;
    float *dgi;             // [T][R][3H]
    float *dgh;             // [T][R][3H]
    float *dh0;             // [R][H] or null
};
struct GruSeqBatch {
    GruSeq seq[GRU_SEQ_MAX];
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float fast_sigmoid(float x) { return __frcp_rn(1.f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x) {
    // 1 - 2 / (1 + e^{2x}); saturates correctly at both ends (e -> inf gives 1, e -> 0 gives -1)
    return 1.f - 2.f * __frcp_rn(1.f + __expf(2.f * x));
}

template <int H>
__global__ __launch_bounds__(H * 4) void gru_seq_fwd_kernel(GruSeqBatch batch, int T, int R) {
    constexpr int KQ = H / 16;             // groups of 16 k values (4 MFMAs each)
    constexpr int HS = H + 4;              // LDS row stride (floats)
    __shared__ float hbuf[2][16][HS];
    const GruSeq &s = batch.seq[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, quad = lane >> 4;
    const int unit = 16 * w + col;
    const int row0 = blockIdx.x * 16;

    f32x4 wreg[3][KQ];
#pragma unroll
    for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq)
            wreg[g][kq] = *reinterpret_cast<const f32x4 *>(s.w_hh + (int64_t)(g * H + unit) * H + 16 * kq + 4 * quad);
    const float bh_r = s.b_hh[unit], bh_z = s.b_hh[H + unit], bh_n = s.b_hh[2 * H + unit];

    int rows[4];
    bool live[4];
    float h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + 4 * quad + i;
        live[i] = r < R;
        rows[i] = live[i] ? r : R - 1;
        h[i] = (s.h0 != nullptr && live[i]) ? s.h0[(int64_t)rows[i] * H + unit] : 0.f;
        hbuf[0][4 * quad + i][unit] = h[i];
    }
    float gi_next[4][3];
    {
        const int t0 = s.reverse ? T - 1 : 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float *p = s.gi + t0 * s.gi_tstride + (int64_t)rows[i] * 3 * H + unit;
            gi_next[i][0] = p[0]; gi_next[i][1] = p[H]; gi_next[i][2] = p[2 * H];
        }
    }
    __syncthreads();

    for (int step = 0; step < T; ++step) {
        const int t = s.reverse ? T - 1 - step : step;
        const int cur = step & 1;
        float gi[4][3];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 3; ++g) gi[i][g] = gi_next[i][g];
        if (step + 1 < T) {
            const int tn = s.reverse ? t - 1 : t + 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float *p = s.gi + tn * s.gi_tstride + (int64_t)rows[i] * 3 * H + unit;
                gi_next[i][0] = p[0]; gi_next[i][1] = p[H]; gi_next[i][2] = p[2 * H];
            }
        }
        f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(&hbuf[cur][col][16 * kq + 4 * quad]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int g = 0; g < 3; ++g)
                    acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], wreg[g][kq][j], acc[g], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float r = fast_sigmoid(gi[i][0] + acc[0][i] + bh_r);
            const float z = fast_sigmoid(gi[i][1] + acc[1][i] + bh_z);
            const float ghn = acc[2][i] + bh_n;
            const float n = fast_tanh(gi[i][2] + r * ghn);
            const float hn = (1.f - z) * n + z * h[i];
            h[i] = hn;
            hbuf[cur ^ 1][4 * quad + i][unit] = hn;
            if (live[i]) {
                const int64_t tr = (int64_t)t * R + rows[i];
                s.h_all[tr * s.h_stride + unit] = hn;
                float *sv = s.saved + tr * 4 * H + unit;
                sv[0] = r; sv[H] = z; sv[2 * H] = n; sv[3 * H] = ghn;
            }
        }
        __syncthreads();
    }
}

// Backward through time.  Per step (in the reverse of the forward's processing order):
//   g = dh_all[t] + carry;  dpn = g (1-z)(1-n^2);  dpz = g (h_prev - n) z (1-z);  dpr = dpn gh_n r (1-r)
//   dgi = [dpr, dpz, dpn];  dgh = [dpr, dpz, dpn r];  carry = g z + dgh . W_hh
template <int H>
__global__ __launch_bounds__(H * 4) void gru_seq_bwd_kernel(GruSeqBatch batch, int T, int R) {
    constexpr int KQ = 3 * H / 16;
    constexpr int DS = 3 * H + 4;
    __shared__ float dbuf[2][16][DS];
    const GruSeq &s = batch.seq[blockIdx.y];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int col = lane & 15, quad = lane >> 4;
    const int unit = 16 * w + col;
    const int row0 = blockIdx.x * 16;

    // B[k = c][n = unit] = W_hh[c][unit], c = 16 kq + 4 quad + j
    f32x4 wreg[KQ];
#pragma unroll
    for (int kq = 0; kq < KQ; ++kq)
#pragma unroll
        for (int j = 0; j < 4; ++j) wreg[kq][j] = s.w_hh[(int64_t)(16 * kq + 4 * quad + j) * H + unit];

    int rows[4];
    bool live[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + 4 * quad + i;
        live[i] = r < R;
        rows[i] = live[i] ? r : R - 1;
    }
    float carry[4] = {0.f, 0.f, 0.f, 0.f};

    // operands of one step: dh, r, z, n, gh_n, h_prev
    float nx[4][6];
    auto fetch = [&](int step) {
        const int t = s.reverse ? step : T - 1 - step;
        const bool has_prev = step + 1 < T;
        const int tp = s.reverse ? t + 1 : t - 1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t tr = (int64_t)t * R + rows[i];
            nx[i][0] = s.dh_all != nullptr ? s.dh_all[tr * s.dh_stride + unit] : 0.f;
            const float *sv = s.saved + tr * 4 * H + unit;
            nx[i][1] = sv[0]; nx[i][2] = sv[H]; nx[i][3] = sv[2 * H]; nx[i][4] = sv[3 * H];
            if (has_prev) nx[i][5] = s.h_all[((int64_t)tp * R + rows[i]) * s.h_stride + unit];
            else nx[i][5] = s.h0 != nullptr ? s.h0[(int64_t)rows[i] * H + unit] : 0.f;
        }
    };
    fetch(0);

    for (int step = 0; step < T; ++step) {
        const int t = s.reverse ? step : T - 1 - step;
        const int cur = step & 1;
        float gz[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float g = live[i] ? nx[i][0] + carry[i] : 0.f;
            const float r = nx[i][1], z = nx[i][2], n = nx[i][3], ghn = nx[i][4], hp = nx[i][5];
            const float dpn = g * (1.f - z) * (1.f - n * n);
            const float dpz = g * (hp - n) * z * (1.f - z);
            const float dpr = dpn * ghn * r * (1.f - r);
            const float dhn = dpn * r;
            gz[i] = g * z;
            float *d = &dbuf[cur][4 * quad + i][unit];
            d[0] = dpr; d[H] = dpz; d[2 * H] = dhn;
            if (live[i]) {
                const int64_t o = ((int64_t)t * R + rows[i]) * 3 * H + unit;
                s.dgi[o] = dpr; s.dgi[o + H] = dpz; s.dgi[o + 2 * H] = dpn;
                s.dgh[o] = dpr; s.dgh[o + H] = dpz; s.dgh[o + 2 * H] = dhn;
            }
        }
        if (step + 1 < T) fetch(step + 1);
        __syncthreads();
        f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kq = 0; kq < KQ; ++kq) {
            const f32x4 a = *reinterpret_cast<const f32x4 *>(&dbuf[cur][col][16 * kq + 4 * quad]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[kq % 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], wreg[kq][j], acc[kq % 3], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) carry[i] = gz[i] + (acc[0][i] + acc[1][i] + acc[2][i]);
    }
    if (s.dh0 != nullptr)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (live[i]) s.dh0[(int64_t)rows[i] * H + unit] = carry[i];
}

static int fill_batch(GruSeqBatch *b, const arvae_gru_seq_t *seqs, int nseq) {
    for (int i = 0; i < nseq; ++i) {
        const arvae_gru_seq_t &q = seqs[i];
        GruSeq &s = b->seq[i];
        s.gi = q.gi; s.gi_tstride = q.gi_tstride; s.w_hh = q.w_hh; s.b_hh = q.b_hh; s.h0 = q.h0;
        s.h_all = q.h_all; s.h_stride = q.h_stride; s.saved = q.saved; s.reverse = q.reverse;
        s.dh_all = q.dh_all; s.dh_stride = q.dh_stride; s.dgi = q.dgi; s.dgh = q.dgh; s.dh0 = q.dh0;
    }
    return 0;
}

}  // namespace arvae

using namespace arvae;

extern "C" int arvae_gru_seq_supported(int32_t hidden) { return hidden == 32 || hidden == 64 || hidden == 128; }

extern "C" int arvae_gru_seq_fwd(const arvae_gru_seq_t *seqs, int32_t nseq, int32_t steps, int32_t rows, int32_t hidden,
                                 arvae_stream_t stream) {
    ARVAE_REQUIRE(seqs != nullptr && nseq >= 1 && nseq <= GRU_SEQ_MAX, "gru_seq_fwd: 1..%d sequences per launch", GRU_SEQ_MAX);
    ARVAE_REQUIRE(steps >= 1 && rows >= 1, "gru_seq_fwd: empty sequence");
    ARVAE_REQUIRE(arvae_gru_seq_supported(hidden), "gru_seq_fwd: hidden size %d is not built (32, 64, 128)", hidden);
    for (int i = 0; i < nseq; ++i)
        ARVAE_REQUIRE(seqs[i].gi && seqs[i].w_hh && seqs[i].b_hh && seqs[i].h_all && seqs[i].saved, "gru_seq_fwd: null pointer");
    GruSeqBatch b{};
    fill_batch(&b, seqs, nseq);
    hipStream_t st = as_stream(stream);
    const dim3 grid((rows + 15) / 16, nseq);
    prof_gap();
    if (hidden == 128) hipLaunchKernelGGL(gru_seq_fwd_kernel<128>, grid, dim3(512), 0, st, b, steps, rows);
    else if (hidden == 64) hipLaunchKernelGGL(gru_seq_fwd_kernel<64>, grid, dim3(256), 0, st, b, steps, rows);
    else hipLaunchKernelGGL(gru_seq_fwd_kernel<32>, grid, dim3(128), 0, st, b, steps, rows);
    return check_launch("gru_seq_fwd_kernel");
}

extern "C" int arvae_gru_seq_bwd(const arvae_gru_seq_t *seqs, int32_t nseq, int32_t steps, int32_t rows, int32_t hidden,
                                 arvae_stream_t stream) {
    ARVAE_REQUIRE(seqs != nullptr && nseq >= 1 && nseq <= GRU_SEQ_MAX, "gru_seq_bwd: 1..%d sequences per launch", GRU_SEQ_MAX);
    ARVAE_REQUIRE(steps >= 1 && rows >= 1, "gru_seq_bwd: empty sequence");
    ARVAE_REQUIRE(arvae_gru_seq_supported(hidden), "gru_seq_bwd: hidden size %d is not built (32, 64, 128)", hidden);
    for (int i = 0; i < nseq; ++i)
        ARVAE_REQUIRE(seqs[i].w_hh && seqs[i].h_all && seqs[i].saved && seqs[i].dgi && seqs[i].dgh, "gru_seq_bwd: null pointer");
    GruSeqBatch b{};
    fill_batch(&b, seqs, nseq);
    hipStream_t st = as_stream(stream);
    const dim3 grid((rows + 15) / 16, nseq);
    prof_gap();
    if (hidden == 128) hipLaunchKernelGGL(gru_seq_bwd_kernel<128>, grid, dim3(512), 0, st, b, steps, rows);
    else if (hidden == 64) hipLaunchKernelGGL(gru_seq_bwd_kernel<64>, grid, dim3(256), 0, st, b, steps, rows);
    else hipLaunchKernelGGL(gru_seq_bwd_kernel<32>, grid, dim3(128), 0, st, b, steps, rows);
    return check_launch("gru_seq_bwd_kernel");
}
