// Kernels of the MeasureVAE path that are not GEMMs: GRU gate math (forward / backward), embedding gather
// and its gradient, top-1 feedback, column concat / split, dropout masks, and the measure attribute labels.
// The GEMMs of the GRUs (W_ih x, W_hh h) run on the dense MFMA kernels (dense.hip).
//
// GRU cell (PyTorch gate order r, z, n stacked in rows; reference measurevae/encoder.py:27-34,
// measurevae/decoder.py:338-368 via nn.GRU):
//     r = sigmoid(gi_r + gh_r);  z = sigmoid(gi_z + gh_z);  n = tanh(gi_n + r * gh_n);  h' = (1-z)*n + z*h
#include "common.h"
#include "attributes.h"

namespace arvae {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// gi, gh: [B, 3H] (bias already added); h_prev: [B, H] or null (zeros).  saved: [4][B][H] = r, z, n, gh_n
__global__ __launch_bounds__(256) void gru_gates_fwd_kernel(const float *__restrict__ gi, const float *__restrict__ gh,
                                                             const float *__restrict__ h_prev, int batch, int hid,
                                                             float *__restrict__ h_new, float *__restrict__ saved) {
    const int64_t total = (int64_t)batch * hid;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / hid), j = (int)(i - (int64_t)b * hid);
        const float *gib = gi + (int64_t)b * 3 * hid, *ghb = gh + (int64_t)b * 3 * hid;
        const float r = sigmoidf_(gib[j] + ghb[j]);
        const float z = sigmoidf_(gib[hid + j] + ghb[hid + j]);
        const float ghn = ghb[2 * hid + j];
        const float n = tanhf(gib[2 * hid + j] + r * ghn);
        const float hp = h_prev != nullptr ? h_prev[i] : 0.f;
        h_new[i] = (1.f - z) * n + z * hp;
        if (saved != nullptr) {
            saved[i] = r;
            saved[total + i] = z;
            saved[2 * total + i] = n;
            saved[3 * total + i] = ghn;
        }
    }
}

// dh: gradient w.r.t. h'.  Outputs dgi, dgh [B,3H] and dh_prev [B,H] (= dh * z, the direct path only).
__global__ __launch_bounds__(256) void gru_gates_bwd_kernel(const float *__restrict__ dh, const float *__restrict__ saved,
                                                             const float *__restrict__ h_prev, int batch, int hid,
                                                             float *__restrict__ dgi, float *__restrict__ dgh,
                                                             float *__restrict__ dh_prev) {
    const int64_t total = (int64_t)batch * hid;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / hid), j = (int)(i - (int64_t)b * hid);
        const float r = saved[i], z = saved[total + i], n = saved[2 * total + i], ghn = saved[3 * total + i];
        const float hp = h_prev != nullptr ? h_prev[i] : 0.f;
        const float g = dh[i];
        const float dpn = g * (1.f - z) * (1.f - n * n);
        const float dpz = g * (hp - n) * z * (1.f - z);
        const float dpr = dpn * ghn * r * (1.f - r);
        float *dgib = dgi + (int64_t)b * 3 * hid, *dghb = dgh + (int64_t)b * 3 * hid;
        dgib[j] = dpr; dghb[j] = dpr;
        dgib[hid + j] = dpz; dghb[hid + j] = dpz;
        dgib[2 * hid + j] = dpn; dghb[2 * hid + j] = dpn * r;
        dh_prev[i] = g * z;
    }
}

// out[(t*B + b) or (b*T + t)][:] = table[idx[b*T + t]][:]
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t *__restrict__ idx, const float *__restrict__ table,
                                                         int batch, int steps, int dim, int vocab, int time_major,
                                                         float *__restrict__ out) {
    const int64_t total = (int64_t)batch * steps * dim;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int d = (int)(i % dim);
        const int64_t row = i / dim;                         // output row
        const int b = time_major ? (int)(row % batch) : (int)(row / steps);
        const int t = time_major ? (int)(row / batch) : (int)(row % steps);
        int64_t v = idx[(int64_t)b * steps + t];
        v = v < 0 ? 0 : (v >= vocab ? vocab - 1 : v);
        out[i] = table[v * dim + d];
    }
}
// the same for rows that are multiples of four floats (wide per-vocabulary projection tables): a wave per output row, 16 bytes
// per lane and step, the row's index read once
__global__ __launch_bounds__(256) void embed_fwd4_kernel(const int64_t *__restrict__ idx, const float4 *__restrict__ table,
                                                          int batch, int steps, int dim4, int vocab, int time_major,
                                                          float4 *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t rows = (int64_t)batch * steps, nw = (int64_t)gridDim.x * 4;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += nw) {
        const int b = time_major ? (int)(row % batch) : (int)(row / steps);
        const int t = time_major ? (int)(row / batch) : (int)(row % steps);
        int64_t v = idx[(int64_t)b * steps + t];
        v = v < 0 ? 0 : (v >= vocab ? vocab - 1 : v);
        const float4 *src = table + v * dim4;
        float4 *dst = out + row * dim4;
        for (int d = lane; d < dim4; d += 64) dst[d] = src[d];
    }
}

// ... with the beat RNN's constant input riding in the grid's last workgroups (attributes.h)
__global__ __launch_bounds__(256) void embed_fwd4_beat_kernel(const int64_t *__restrict__ idx, const float4 *__restrict__ table,
                                                               int batch, int steps, int dim4, int vocab, int time_major,
                                                               float4 *__restrict__ out, BeatInput beat, int nb) {
    if ((int)blockIdx.x >= nb) {
        beat_input_items(beat, (int64_t)((int)blockIdx.x - nb) * 256 + threadIdx.x, (int64_t)((int)gridDim.x - nb) * 256);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int64_t rows = (int64_t)batch * steps, nw = (int64_t)nb * 4;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += nw) {
        const int b = time_major ? (int)(row % batch) : (int)(row / steps);
        const int t = time_major ? (int)(row / batch) : (int)(row % steps);
        int64_t v = idx[(int64_t)b * steps + t];
        v = v < 0 ? 0 : (v >= vocab ? vocab - 1 : v);
        const float4 *src = table + v * dim4;
        float4 *dst = out + row * dim4;
        for (int d = lane; d < dim4; d += 64) dst[d] = src[d];
    }
}

// dtable[v][d] += sum over positions with idx == v of g[row][d], in two launches with a fixed summation order:
// (1) a workgroup stages EMBED_POS positions (indices + gradient rows) in LDS and one thread per (v, d) pair sums the rows
// whose index is v; (2) the per-workgroup partials are added in workgroup order.
constexpr int EMBED_POS = 64;
__global__ __launch_bounds__(256) void embed_bwd_partial_kernel(const int64_t *__restrict__ idx, const float *__restrict__ g,
                                                                 int batch, int steps, int dim, int vocab, int time_major,
                                                                 float *__restrict__ partial) {
    extern __shared__ float sg[];                            // [EMBED_POS][dim + 1] then int sidx[EMBED_POS]
    const int pitch = dim + 1;
    int *sidx = reinterpret_cast<int *>(sg + EMBED_POS * pitch);
    const int64_t n = (int64_t)batch * steps;
    // element e = (position slot, d): consecutive threads read consecutive floats of a gradient row
    for (int e = threadIdx.x; e < EMBED_POS * dim; e += 256) {
        const int slot = e / dim, d = e - slot * dim;
        const int64_t pos = (int64_t)blockIdx.x * EMBED_POS + slot;            // pos = b*steps + t
        float v = 0.f;
        if (pos < n) {
            const int b = (int)(pos / steps), t = (int)(pos % steps);
            const int64_t row = time_major ? (int64_t)t * batch + b : pos;
            v = g[row * dim + d];
            if (d == 0) sidx[slot] = (int)idx[pos];
        } else if (d == 0) {
            sidx[slot] = -1;
        }
        sg[slot * pitch + d] = v;
    }
    __syncthreads();
    for (int pair = threadIdx.x; pair < vocab * dim; pair += 256) {
        const int v = pair / dim, d = pair - v * dim;
        float s = 0.f;
#pragma unroll 8
        for (int p = 0; p < EMBED_POS; ++p) s += sidx[p] == v ? sg[p * pitch + d] : 0.f;
        partial[(int64_t)blockIdx.x * vocab * dim + pair] = s;
    }
}
__global__ __launch_bounds__(256) void embed_bwd_reduce_kernel(const float *__restrict__ partial, int blocks, int count,
                                                                float *__restrict__ dtable, int accumulate) {
    // 4 lanes per table element, lane q sums blocks q, q+4, ... with 8 loads in flight; fixed-order lane sum
    const int i = (blockIdx.x * 256 + threadIdx.x) >> 2, q = threadIdx.x & 3;
    const int ic = i < count ? i : 0;
    float s = 0.f;
    for (int z0 = q; z0 < blocks; z0 += 32) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int z = z0 + 4 * u;
            const float v = partial[(int64_t)(z < blocks ? z : 0) * count + ic];
            t[u] = z < blocks ? v : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s += t[u];
    }
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (q == 0 && i < count) dtable[i] = accumulate ? dtable[i] + s : s;
}

// The same for WIDE rows (dim > 128: a gradient w.r.t. rows that were looked up from a per-vocabulary projection table, e.g. the
// encoder's layer-0 input projection P = table W_ih^T + b of both directions, 768 columns).  Workgroup (column slice of 32,
// position group of EMBED_WSPLIT): thread = (column, one of 8 position lanes) walks its positions -- 128-byte segments per
// lane group, EMBED_WFLY rows in flight --, adding into its own LDS cell acc[v][thread] (no conflicts, position order); the
// eight lanes of a column are then summed in lane order: partial [group][v][dim], EMBED_WSPLIT of them for
// embed_bwd_reduce_kernel (1.7 MB at 35 x 768, where one partial per 32 positions was 20 MB).
constexpr int EMBED_WSPLIT = 16, EMBED_WCOLS = 32, EMBED_WLANES = 8, EMBED_WFLY = 16;
// (time_major 2 = TICK ORDER, arvae_tick_gi_bwd: position i IS gradient row i = (j, beat, b) of [tpb][beats][batch], its token the
// PREVIOUS tick's note idx[b][tpb*beat + j - 1], or the extra entry `vocab - 1` -- the learned start vector -- at tick 0)
__global__ __launch_bounds__(256) void embed_bwd_wide_kernel(const int64_t *__restrict__ idx, const float *__restrict__ g,
                                                              int batch, int steps, int dim, int vocab, int time_major,
                                                              float *__restrict__ partial, int beats, int tpb) {
    extern __shared__ float acc[];                           // [vocab][256] | int srow[per] | int sidx[per]
    const int n = batch * steps;
    const int col = threadIdx.x & (EMBED_WCOLS - 1), pl = threadIdx.x / EMBED_WCOLS;
    const int d = blockIdx.x * EMBED_WCOLS + col;
    const bool dok = d < dim;
    // this group's positions: [p_lo, p_hi), lane pl takes p_lo + pl, + 8, ...; their gradient rows and (clamped) tokens are worked
    // out once per workgroup (the divisions), not once per thread and position
    const int per = (n + EMBED_WSPLIT - 1) / EMBED_WSPLIT;
    const int p_lo = (int)blockIdx.y * per, p_hi = min(p_lo + per, n), cnt = max(p_hi - p_lo, 0);
    int *srow = reinterpret_cast<int *>(acc + vocab * 256), *sidx = srow + per;
    for (int i = threadIdx.x; i < cnt; i += 256) {
        const int pos = p_lo + i;
        int v;
        if (time_major == 2) {
            const int j = pos / (beats * batch), rem = pos - j * beats * batch, beat = rem / batch, b = rem - beat * batch;
            const int t = tpb * beat + j;
            srow[i] = pos;
            v = t == 0 ? vocab - 1 : (int)idx[b * steps + t - 1];
            v = v < 0 ? 0 : (v >= vocab - 1 && t != 0 ? vocab - 2 : v);
        } else {
            const int b = pos / steps, t = pos - b * steps;
            srow[i] = time_major ? t * batch + b : pos;
            v = (int)idx[pos];
            v = v < 0 ? 0 : (v >= vocab ? vocab - 1 : v);
        }
        sidx[i] = v;
    }
    for (int v = 0; v < vocab; ++v) acc[v * 256 + threadIdx.x] = 0.f;
    __syncthreads();
    for (int i0 = pl; i0 < cnt; i0 += EMBED_WLANES * EMBED_WFLY) {
        float gv[EMBED_WFLY];
#pragma unroll
        for (int u = 0; u < EMBED_WFLY; ++u) {
            const int i = i0 + u * EMBED_WLANES;
            gv[u] = (dok && i < cnt) ? g[(int64_t)srow[i < cnt ? i : 0] * dim + d] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < EMBED_WFLY; ++u) {
            const int i = i0 + u * EMBED_WLANES;
            if (i < cnt) acc[sidx[i] * 256 + threadIdx.x] += gv[u];
        }
    }
    __syncthreads();
    for (int e = threadIdx.x; e < vocab * EMBED_WCOLS; e += 256) {
        const int v = e / EMBED_WCOLS, c = e - v * EMBED_WCOLS;
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < EMBED_WLANES; ++q) s += acc[v * 256 + q * EMBED_WCOLS + c];
        const int dd = blockIdx.x * EMBED_WCOLS + c;
        if (dd < dim) partial[((int64_t)blockIdx.y * vocab + v) * dim + dd] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// The tick RNN's layer-0 input projection, re-associated (measurevae/decoder.py:459-505: per tick W_ih0 [embedding of the
// previous note | beat embedding] + b_ih0, 24 ticks x batch rows of 138 inputs).  The note half takes only vocab + 1 distinct
// values (the vocabulary's embeddings and the learned start vector x_0), the beat half one per (beat, measure): ONE small product
//     G = X W_ih0^T,   X = [ table | 0 ]  (vocab rows)
//                          [  x_0  | 0 ]  (1 row)
//                          [   0   | beat_emb ]  (beats*batch rows)
// and per tick gi = G[previous note] + G[vocab + 1 + beat*batch + b] + b_ih0 (tick_gi_fwd).  Backward: the per-tick gradients are
// summed over the ticks of a beat (beat rows) and per previous note (the wide segment sum above, tick order) into dG; dX = dG W_ih0
// splits into the embedding table's, x_0's and the beat embedding's gradients (tick_rows_bwd).
__global__ __launch_bounds__(256) void tick_rows_fwd_kernel(const float *__restrict__ table, const float *__restrict__ x0,
                                                             const float *__restrict__ beat_emb, int64_t beat_stride, int vocab, int emb,
                                                             int hidden, int rows, float *__restrict__ x) {
    const int cols = emb + hidden;
    const int64_t total = (int64_t)(vocab + 1 + rows) * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
        float v = 0.f;
        if (r < vocab) v = c < emb ? table[r * emb + c] : 0.f;
        else if (r == vocab) v = c < emb ? x0[c] : 0.f;
        else v = c >= emb ? beat_emb[(int64_t)(r - vocab - 1) * beat_stride + c - emb] : 0.f;
        x[i] = v;
    }
}
__global__ __launch_bounds__(256) void tick_rows_bwd_kernel(const float *__restrict__ dx, int vocab, int emb, int hidden, int rows,
                                                             float *__restrict__ dtable, float *__restrict__ dx0,
                                                             float *__restrict__ dbeat, int64_t dbeat_stride) {
    const int cols = emb + hidden;
    const int64_t n_tab = (int64_t)(vocab + 1) * emb, total = n_tab + (int64_t)rows * hidden;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        if (i < n_tab) {
            const int r = (int)(i / emb), c = (int)(i - (int64_t)r * emb);
            const float v = dx[(int64_t)r * cols + c];
            if (r < vocab) { if (dtable != nullptr) dtable[r * emb + c] += v; }
            else if (dx0 != nullptr) dx0[c] += v;
        } else {
            const int64_t k = i - n_tab;
            const int r = (int)(k / hidden), c = (int)(k - (int64_t)r * hidden);
            dbeat[(int64_t)r * dbeat_stride + c] = dx[(int64_t)(vocab + 1 + r) * cols + emb + c];
        }
    }
}
// gi[(j*beats + beat)*batch + b] = G[prev(b, tpb*beat + j)] + G[vocab + 1 + beat*batch + b] + bias: a wave per output row
__global__ __launch_bounds__(256) void tick_gi_fwd_kernel(const float4 *__restrict__ gs, const int64_t *__restrict__ tokens,
                                                           const float4 *__restrict__ bias, int batch, int beats, int tpb, int vocab,
                                                           int cols4, float4 *__restrict__ gi, int64_t *__restrict__ copy_to) {
    const int lane = threadIdx.x & 63, steps = beats * tpb, rows_b = beats * batch;
    const int64_t rows = (int64_t)tpb * rows_b, nw = (int64_t)gridDim.x * 4;
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += nw) {
        const int j = (int)(row / rows_b), rem = (int)(row - (int64_t)j * rows_b), beat = rem / batch, b = rem - beat * batch;
        const int t = tpb * beat + j;
        int64_t v = t == 0 ? vocab : tokens[(int64_t)b * steps + t - 1];
        // (teacher forcing in the whole-model executor: the notes fed back ARE the score; its copy rides here, one element per row)
        if (copy_to != nullptr && lane == 0) copy_to[(int64_t)b * steps + t] = tokens[(int64_t)b * steps + t];
        v = v < 0 ? 0 : (v > vocab ? vocab : v);
        if (t != 0 && v == vocab) v = vocab - 1;
        const float4 *pn = gs + v * cols4, *pb = gs + (int64_t)(vocab + 1 + rem) * cols4;
        float4 *dst = gi + row * cols4;
        for (int d = lane; d < cols4; d += 64) {
            const float4 a = pn[d], c = pb[d], e = bias != nullptr ? bias[d] : make_float4(0.f, 0.f, 0.f, 0.f);
            dst[d] = make_float4(a.x + c.x + e.x, a.y + c.y + e.y, a.z + c.z + e.z, a.w + c.w + e.w);
        }
    }
}
// beat rows of dG: the sum over a beat's ticks, dG[vocab + 1 + r] = sum_j dgi[j*rows_b + r] (tick order of the sum: fixed)
__global__ __launch_bounds__(256) void tick_gi_bwd_beat_kernel(const float4 *__restrict__ dgi, int rows_b, int tpb, int cols4,
                                                                float4 *__restrict__ dg_beat) {
    const int64_t total = (int64_t)rows_b * cols4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        float4 s = dgi[i];
        for (int j = 1; j < tpb; ++j) {
            const float4 v = dgi[(int64_t)j * total + i];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        dg_beat[i] = s;
    }
}


// idx[b] = argmax_j w[b][j], lowest index on ties (reference decoder.py:506-507 topk(1))
__global__ __launch_bounds__(256) void row_argmax_kernel(const float *__restrict__ w, int rows, int cols,
                                                          int64_t *__restrict__ idx) {
    for (int r = blockIdx.x * 256 + threadIdx.x; r < rows; r += gridDim.x * 256) {
        const float *row = w + (int64_t)r * cols;
        float mx = row[0];
        int arg = 0;
        for (int j = 1; j < cols; ++j)
            if (row[j] > mx) { mx = row[j]; arg = j; }
        idx[r] = arg;
    }
}

// out[r] = [a[r] | b[r]]  and the adjoint split
__global__ __launch_bounds__(256) void concat_cols_kernel(const float *__restrict__ a, const float *__restrict__ b, int64_t rows,
                                                           int ca, int cb, float *__restrict__ out) {
    const int c = ca + cb;
    const int64_t total = rows * c;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / c;
        const int j = (int)(i - r * c);
        out[i] = j < ca ? a[r * ca + j] : b[r * cb + (j - ca)];
    }
}
__global__ __launch_bounds__(256) void split_cols_kernel(const float *__restrict__ g, int64_t rows, int ca, int cb,
                                                          float *__restrict__ da, float *__restrict__ db, int accumulate_b) {
    const int c = ca + cb;
    const int64_t total = rows * c;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / c;
        const int j = (int)(i - r * c);
        if (j < ca) {
            if (da != nullptr) da[r * ca + j] = g[i];
        } else if (db != nullptr) {
            if (accumulate_b) db[r * cb + (j - ca)] += g[i];
            else db[r * cb + (j - ca)] = g[i];
        }
    }
}

// y = alpha * x * (mask ? mask : 1) + (accumulate ? y : 0)
__global__ __launch_bounds__(256) void scale_mask_kernel(const float *__restrict__ x, const uint8_t *__restrict__ mask,
                                                          float alpha, int64_t count, int accumulate, float *__restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) {
        const float v = alpha * x[i] * (mask != nullptr ? (float)mask[i] : 1.f);
        y[i] = accumulate ? y[i] + v : v;
    }
}

// broadcast rows: y[r][:] = v[:]  (learned start vectors x_0 / b_0) and its adjoint column sum
__global__ __launch_bounds__(256) void broadcast_rows_kernel(const float *__restrict__ v, int64_t rows, int cols, float *__restrict__ y) {
    const int64_t total = rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) y[i] = v[i % cols];
}

// measure attributes (reference bar_dataset.py:338-500 via measure_vae_trainer.py:167-186): one lane per measure
//   out[b] = [rhythmic complexity, pitch range / 26, note density, contour / 26]
__global__ __launch_bounds__(64) void measure_attributes_kernel(AttrArgs a) {
    measure_attributes_rows(a, blockIdx.x * 64 + threadIdx.x, gridDim.x * 64);
}

static inline int blocks_for(int64_t n, int cap = 2048) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace arvae

using namespace arvae;

extern "C" int arvae_gru_gates_fwd(const float *gi, const float *gh, const float *h_prev, int32_t batch, int32_t hidden,
                                   float *h_new, float *saved, arvae_stream_t stream) {
    ARVAE_REQUIRE(gi && gh && h_new && batch > 0 && hidden > 0, "gru_gates_fwd: bad argument");
    ARVAE_LAUNCH(gru_gates_fwd_kernel, dim3(blocks_for((int64_t)batch * hidden)), dim3(256), 0, as_stream(stream), gi,
                       gh, h_prev, batch, hidden, h_new, saved);
    return check_launch("gru_gates_fwd_kernel");
}

extern "C" int arvae_gru_gates_bwd(const float *dh, const float *saved, const float *h_prev, int32_t batch, int32_t hidden,
                                   float *dgi, float *dgh, float *dh_prev, arvae_stream_t stream) {
    ARVAE_REQUIRE(dh && saved && dgi && dgh && dh_prev && batch > 0 && hidden > 0, "gru_gates_bwd: bad argument");
    ARVAE_LAUNCH(gru_gates_bwd_kernel, dim3(blocks_for((int64_t)batch * hidden)), dim3(256), 0, as_stream(stream), dh,
                       saved, h_prev, batch, hidden, dgi, dgh, dh_prev);
    return check_launch("gru_gates_bwd_kernel");
}

extern "C" int arvae_embed_fwd(const int64_t *idx, const float *table, int32_t batch, int32_t steps, int32_t dim,
                               int32_t vocab, int32_t time_major, float *out, arvae_stream_t stream) {
    ARVAE_REQUIRE(idx && table && out && batch > 0 && steps > 0 && dim > 0 && vocab > 0, "embed_fwd: bad argument");
    if (dim % 4 == 0 && dim >= 64 && (reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(out)) % 16 == 0) {
        ARVAE_LAUNCH(embed_fwd4_kernel, dim3((unsigned)(((int64_t)batch * steps + 3) / 4)), dim3(256), 0, as_stream(stream), idx,
                     reinterpret_cast<const float4 *>(table), batch, steps, dim / 4, vocab, time_major, reinterpret_cast<float4 *>(out));
        return check_launch("embed_fwd4_kernel");
    }
    ARVAE_LAUNCH(embed_fwd_kernel, dim3(blocks_for((int64_t)batch * steps * dim)), dim3(256), 0, as_stream(stream), idx,
                       table, batch, steps, dim, vocab, time_major, out);
    return check_launch("embed_fwd_kernel");
}

// arvae_embed_fwd (wide rows) with the beat RNN's constant input in the same launch (plan_measure.hip); false: not that case
namespace arvae {
bool embed_fwd_with_beat(const int64_t *idx, const float *table, int32_t batch, int32_t steps, int32_t dim, int32_t vocab, int32_t time_major,
                         float *out, const BeatInput &beat, hipStream_t s, int *rc) {
    if (!(dim % 4 == 0 && dim >= 64 && (reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(out)) % 16 == 0)) return false;
    const int nb = (int)(((int64_t)batch * steps + 3) / 4);
    const int64_t items = (int64_t)beat.batch * beat.cols + beat.rows;
    const int nbeat = (int)((items + 255) / 256 > 512 ? 512 : (items + 255) / 256);
    ARVAE_LAUNCH(embed_fwd4_beat_kernel, dim3((unsigned)(nb + nbeat)), dim3(256), 0, s, idx, reinterpret_cast<const float4 *>(table), batch, steps,
                 dim / 4, vocab, time_major, reinterpret_cast<float4 *>(out), beat, nb);
    *rc = check_launch("embed_fwd4_kernel(+ beat input)");
    return true;
}
}  // namespace arvae

extern "C" int64_t arvae_embed_bwd_ws_floats(int32_t batch, int32_t steps, int32_t dim, int32_t vocab) {
    if (dim > 128) return (int64_t)EMBED_WSPLIT * vocab * dim;
    const int64_t blocks = ((int64_t)batch * steps + EMBED_POS - 1) / EMBED_POS;
    return blocks * vocab * dim;
}

extern "C" int arvae_embed_bwd(const int64_t *idx, const float *g, int32_t batch, int32_t steps, int32_t dim,
                               int32_t vocab, int32_t time_major, float *dtable, int32_t accumulate, float *ws, arvae_stream_t stream) {
    ARVAE_REQUIRE(idx && g && dtable && ws && batch > 0 && steps > 0 && dim > 0 && vocab > 0,
                  "embed_bwd: bad argument (ws = arvae_embed_bwd_ws_floats() floats)");
    hipStream_t st = as_stream(stream);
    if (dim <= 128) {
        const int blocks = (int)(((int64_t)batch * steps + EMBED_POS - 1) / EMBED_POS);
        const size_t lds = (size_t)EMBED_POS * (dim + 1) * sizeof(float) + EMBED_POS * sizeof(int);
        ARVAE_LAUNCH(embed_bwd_partial_kernel, dim3(blocks), dim3(256), lds, st, idx, g, batch, steps, dim, vocab, time_major, ws);
        ARVAE_LAUNCH(embed_bwd_reduce_kernel, dim3((vocab * dim * 4 + 255) / 256), dim3(256), 0, st, ws, blocks, vocab * dim, dtable, accumulate);
        return check_launch("embed_bwd_kernel");
    }
    // wide rows (lookups of a per-vocabulary projection table): column-parallel segment sums
    const int per = (int)(((int64_t)batch * steps + EMBED_WSPLIT - 1) / EMBED_WSPLIT);
    const size_t lds = (size_t)vocab * 256 * sizeof(float) + (size_t)per * 2 * sizeof(int);
    ARVAE_REQUIRE(lds <= 64 * 1024 && (int64_t)batch * steps < (1 << 30),
                  "embed_bwd: %d vocabulary entries x %d positions do not fit the wide-row kernel", vocab, batch * steps);
    ARVAE_LAUNCH(embed_bwd_wide_kernel, dim3((dim + EMBED_WCOLS - 1) / EMBED_WCOLS, EMBED_WSPLIT), dim3(256), lds, st, idx, g, batch, steps,
                 dim, vocab, time_major, ws, 0, 0);
    ARVAE_LAUNCH(embed_bwd_reduce_kernel, dim3((vocab * dim * 4 + 255) / 256), dim3(256), 0, st, ws, EMBED_WSPLIT, vocab * dim, dtable,
                 accumulate);
    return check_launch("embed_bwd_wide_kernel");
}

extern "C" int arvae_tick_rows_fwd(const float *table, const float *x0, const float *beat_emb, int64_t beat_emb_stride, int32_t vocab, int32_t emb,
                                   int32_t hidden, int32_t rows, float *x_small, arvae_stream_t stream) {
    ARVAE_REQUIRE(table && x0 && beat_emb && x_small && vocab > 0 && emb > 0 && hidden > 0 && rows > 0, "tick_rows_fwd: bad argument");
    ARVAE_LAUNCH(tick_rows_fwd_kernel, dim3(blocks_for((int64_t)(vocab + 1 + rows) * (emb + hidden))), dim3(256), 0, as_stream(stream), table,
                 x0, beat_emb, beat_emb_stride != 0 ? beat_emb_stride : (int64_t)hidden, vocab, emb, hidden, rows, x_small);
    return check_launch("tick_rows_fwd_kernel");
}

extern "C" int arvae_tick_rows_bwd(const float *dx_small, int32_t vocab, int32_t emb, int32_t hidden, int32_t rows, float *dtable,
                                   float *dx0, float *dbeat_emb, int64_t dbeat_emb_stride, arvae_stream_t stream) {
    ARVAE_REQUIRE(dx_small && dbeat_emb && vocab > 0 && emb > 0 && hidden > 0 && rows > 0, "tick_rows_bwd: bad argument");
    ARVAE_LAUNCH(tick_rows_bwd_kernel, dim3(blocks_for((int64_t)(vocab + 1) * emb + (int64_t)rows * hidden)), dim3(256), 0,
                 as_stream(stream), dx_small, vocab, emb, hidden, rows, dtable, dx0, dbeat_emb, dbeat_emb_stride != 0 ? dbeat_emb_stride : (int64_t)hidden);
    return check_launch("tick_rows_bwd_kernel");
}

// arvae_tick_gi_fwd with the tokens also copied to `copy_to` (may be null; must not alias `tokens`)
namespace arvae {
int tick_gi_fwd_copy(const float *g_small, const int64_t *tokens, const float *bias, int32_t batch, int32_t beats, int32_t ticks_per_beat,
                     int32_t vocab, int32_t cols, float *gi, int64_t *copy_to, hipStream_t s) {
    ARVAE_REQUIRE(g_small && tokens && gi && batch > 0 && beats > 0 && ticks_per_beat > 0 && vocab > 0 && cols > 0 && cols % 4 == 0,
                  "tick_gi_fwd: bad argument (cols must be a multiple of 4)");
    ARVAE_REQUIRE(copy_to != tokens, "tick_gi_fwd: the token copy must not alias its source");
    const int64_t rows = (int64_t)ticks_per_beat * beats * batch;
    ARVAE_LAUNCH(tick_gi_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s,
                 reinterpret_cast<const float4 *>(g_small), tokens, reinterpret_cast<const float4 *>(bias), batch, beats, ticks_per_beat,
                 vocab, cols / 4, reinterpret_cast<float4 *>(gi), copy_to);
    return check_launch("tick_gi_fwd_kernel");
}
}  // namespace arvae

extern "C" int arvae_tick_gi_fwd(const float *g_small, const int64_t *tokens, const float *bias, int32_t batch, int32_t beats,
                                 int32_t ticks_per_beat, int32_t vocab, int32_t cols, float *gi, arvae_stream_t stream) {
    return arvae::tick_gi_fwd_copy(g_small, tokens, bias, batch, beats, ticks_per_beat, vocab, cols, gi, nullptr, as_stream(stream));
}

extern "C" int64_t arvae_tick_gi_bwd_ws_floats(int32_t vocab, int32_t cols) { return (int64_t)EMBED_WSPLIT * (vocab + 1) * cols; }

extern "C" int arvae_tick_gi_bwd(const float *dgi, const int64_t *tokens, int32_t batch, int32_t beats, int32_t ticks_per_beat,
                                 int32_t vocab, int32_t cols, float *dg_small, float *ws, arvae_stream_t stream) {
    ARVAE_REQUIRE(dgi && tokens && dg_small && ws && batch > 0 && beats > 0 && ticks_per_beat > 0 && vocab > 0 && cols > 0 && cols % 4 == 0,
                  "tick_gi_bwd: bad argument (cols must be a multiple of 4)");
    hipStream_t st = as_stream(stream);
    const int rows_b = beats * batch, nv = vocab + 1;
    const int64_t n = (int64_t)ticks_per_beat * rows_b;
    const int per = (int)((n + EMBED_WSPLIT - 1) / EMBED_WSPLIT);
    const size_t lds = (size_t)nv * 256 * sizeof(float) + (size_t)per * 2 * sizeof(int);
    ARVAE_REQUIRE(lds <= 64 * 1024 && n < (1 << 30), "tick_gi_bwd: %d vocabulary entries x %lld rows do not fit the segment-sum kernel", nv,
                  (long long)n);
    ARVAE_LAUNCH(tick_gi_bwd_beat_kernel, dim3(blocks_for((int64_t)rows_b * (cols / 4))), dim3(256), 0, st,
                 reinterpret_cast<const float4 *>(dgi), rows_b, ticks_per_beat, cols / 4,
                 reinterpret_cast<float4 *>(dg_small + (int64_t)nv * cols));
    // (positions = gradient rows in tick order; `batch`/`steps` as the token array has them)
    ARVAE_LAUNCH(embed_bwd_wide_kernel, dim3((cols + EMBED_WCOLS - 1) / EMBED_WCOLS, EMBED_WSPLIT), dim3(256), lds, st, tokens, dgi,
                 batch, beats * ticks_per_beat, cols, nv, 2, ws, beats, ticks_per_beat);
    ARVAE_LAUNCH(embed_bwd_reduce_kernel, dim3((nv * cols * 4 + 255) / 256), dim3(256), 0, st, ws, EMBED_WSPLIT, nv * cols, dg_small, 0);
    return check_launch("tick_gi_bwd_kernel");
}

extern "C" int arvae_row_argmax(const float *w, int32_t rows, int32_t cols, int64_t *idx, arvae_stream_t stream) {
    ARVAE_REQUIRE(w && idx && rows > 0 && cols > 0, "row_argmax: bad argument");
    ARVAE_LAUNCH(row_argmax_kernel, dim3(blocks_for(rows)), dim3(256), 0, as_stream(stream), w, rows, cols, idx);
    return check_launch("row_argmax_kernel");
}

extern "C" int arvae_concat_cols(const float *a, const float *b, int64_t rows, int32_t ca, int32_t cb, float *out,
                                 arvae_stream_t stream) {
    ARVAE_REQUIRE(a && b && out && rows > 0 && ca > 0 && cb > 0, "concat_cols: bad argument");
    ARVAE_LAUNCH(concat_cols_kernel, dim3(blocks_for(rows * (ca + cb))), dim3(256), 0, as_stream(stream), a, b, rows, ca,
                       cb, out);
    return check_launch("concat_cols_kernel");
}

extern "C" int arvae_split_cols(const float *g, int64_t rows, int32_t ca, int32_t cb, float *da, float *db,
                                int32_t accumulate_b, arvae_stream_t stream) {
    ARVAE_REQUIRE(g && rows > 0 && ca > 0 && cb > 0, "split_cols: bad argument");
    ARVAE_LAUNCH(split_cols_kernel, dim3(blocks_for(rows * (ca + cb))), dim3(256), 0, as_stream(stream), g, rows, ca, cb,
                       da, db, accumulate_b);
    return check_launch("split_cols_kernel");
}

extern "C" int arvae_scale_mask(const float *x, const uint8_t *mask, float alpha, int64_t count, int32_t accumulate,
                                float *y, arvae_stream_t stream) {
    ARVAE_REQUIRE(x && y && count > 0, "scale_mask: bad argument");
    ARVAE_LAUNCH(scale_mask_kernel, dim3(blocks_for(count)), dim3(256), 0, as_stream(stream), x, mask, alpha, count,
                       accumulate, y);
    return check_launch("scale_mask_kernel");
}

extern "C" int arvae_broadcast_rows(const float *v, int64_t rows, int32_t cols, float *y, arvae_stream_t stream) {
    ARVAE_REQUIRE(v && y && rows > 0 && cols > 0, "broadcast_rows: bad argument");
    ARVAE_LAUNCH(broadcast_rows_kernel, dim3(blocks_for(rows * cols)), dim3(256), 0, as_stream(stream), v, rows, cols, y);
    return check_launch("broadcast_rows_kernel");
}

extern "C" int arvae_measure_attributes(const int64_t *score, int32_t batch, int32_t steps, const int32_t *midi_lut,
                                             const uint8_t *is_note, const uint8_t *is_density_note, int32_t vocab,
                                             const float *rhythm_weights, float rhythm_norm, float *out,
                                             arvae_stream_t stream) {
    ARVAE_REQUIRE(score && midi_lut && is_note && is_density_note && rhythm_weights && out && batch > 0 && steps > 0 &&
                      vocab > 0 && rhythm_norm > 0.f, "measure_attributes: bad argument");
    ARVAE_LAUNCH(measure_attributes_kernel, dim3((batch + 63) / 64), dim3(64), 0, as_stream(stream),
                 AttrArgs{score, batch, steps, midi_lut, is_note, is_density_note, vocab, rhythm_weights, rhythm_norm, out});
    return check_launch("measure_attributes_kernel");
}
