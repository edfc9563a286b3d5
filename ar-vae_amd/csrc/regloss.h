// The all-pairs attribute regularisation (reference utils/trainer.py:369-403) as a device function, so that its workgroups
// can ride in another kernel's grid (conv32.hip pairs them with the first decoder convolution of the fused forward pass).
// grid = (row blocks, R).  The column vectors z_cols[:,d] / lab_cols[:,d] are staged in LDS in chunks (zero N x N traffic to
// HBM); each wavefront owns REG_ROWS_PER_WAVE rows, its 64 lanes stride over the staged columns and the two sums
// (|t-s| and (1-t^2) sgn(t-s)) are reduced with wave shuffles.
#pragma once
#include "common.h"

namespace arvae {

constexpr int REG_ROWS_PER_WAVE = 2;
constexpr int REG_ROWS_PER_BLOCK = 4 * REG_ROWS_PER_WAVE;
constexpr int REG_CHUNK = 2048;   // columns staged per pass: 2 * 8 KB of LDS

struct RegDims { int d[16]; };

struct RegArgs {
    const float *zr, *lr;        // this rank's rows: z [n_rows][ldz], labels [n_rows][ldl]
    int64_t n_rows;
    const float *zc, *lc;        // the columns they are compared with (the same arrays unless data parallel)
    int64_t n_cols, ldz, ldl;
    RegDims dims;
    float delta;
    float *row_loss, *row_grad;  // [R][n_rows] each
};

// t = tanh(x) and 1 - t^2 from ONE hardware exponential and ONE reciprocal (v_exp_f32, v_rcp_f32: 1 ulp each): with
// e = exp(-2 |x|), t = sign(x) (1 - e) / (1 + e) and 1 - t^2 = 4 e / (1 + e)^2 -- nine vector instructions where ocml's tanhf is ~30
// with branches (the regulariser's workgroups ride in a convolution's grid and compete with its waves for the vector ALU), and
// no cancellation in 1 - t^2 where |x| is large.  |t - tanh(x)| <= ~2e-7.
__device__ __forceinline__ void tanh_sech2(float x, float &t, float &s2) {
    const float e = __builtin_amdgcn_exp2f(-2.885390081777927f * fabsf(x));      // exp(-2 |x|) = 2^(-2 log2(e) |x|)
    const float r = __builtin_amdgcn_rcpf(1.f + e);
    const float m = (1.f - e) * r;
    t = x < 0.f ? -m : m;
    s2 = 4.f * e * r * r;
}

// workgroup (bx, by) of the grid; xs / as: REG_CHUNK floats of LDS each; 256 threads
__device__ __forceinline__ void reg_loss_block(const RegArgs &p, const int bx, const int by, float *xs, float *as) {
    const float *__restrict__ zr = p.zr, *__restrict__ lr = p.lr, *__restrict__ zc = p.zc, *__restrict__ lc = p.lc;
    const int64_t n_rows = p.n_rows, n_cols = p.n_cols, ldz = p.ldz, ldl = p.ldl;
    const float delta = p.delta;
    int d = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q)                                 // constant indices into the by-value argument block
        if (q == by) d = p.dims.d[q];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = (int64_t)bx * REG_ROWS_PER_BLOCK + wave * REG_ROWS_PER_WAVE;
    float xi[REG_ROWS_PER_WAVE], ai[REG_ROWS_PER_WAVE], sl[REG_ROWS_PER_WAVE], sg[REG_ROWS_PER_WAVE];
#pragma unroll
    for (int r = 0; r < REG_ROWS_PER_WAVE; ++r) {
        const int64_t row = row0 + r;
        xi[r] = row < n_rows ? zr[row * ldz + d] : 0.f;
        ai[r] = row < n_rows ? lr[row * ldl + d] : 0.f;
        sl[r] = sg[r] = 0.f;
    }
    for (int64_t c0 = 0; c0 < n_cols; c0 += REG_CHUNK) {
        const int cn = (int)min((int64_t)REG_CHUNK, n_cols - c0);
        __syncthreads();
        for (int j = threadIdx.x; j < cn; j += 256) {
            xs[j] = zc[(c0 + j) * ldz + d];
            as[j] = lc[(c0 + j) * ldl + d];
        }
        __syncthreads();
        for (int j = lane; j < cn; j += 64) {
            const float xj = xs[j], aj = as[j];
#pragma unroll
            for (int r = 0; r < REG_ROWS_PER_WAVE; ++r) {
                float t, s2;
                tanh_sech2(delta * (xi[r] - xj), t, s2);
                const float da = ai[r] - aj;
                const float s = da > 0.f ? 1.f : (da < 0.f ? -1.f : 0.f);
                const float e = t - s;
                sl[r] += fabsf(e);
                sg[r] += s2 * (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f));
            }
        }
    }
#pragma unroll
    for (int r = 0; r < REG_ROWS_PER_WAVE; ++r) {
        const float l = wave_sum(sl[r]), g = wave_sum(sg[r]);
        const int64_t row = row0 + r;
        if (lane == 0 && row < n_rows) {
            p.row_loss[(int64_t)by * n_rows + row] = l;
            p.row_grad[(int64_t)by * n_rows + row] = g;
        }
    }
}

}  // namespace arvae
