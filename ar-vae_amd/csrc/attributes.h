// Small independent jobs of the MeasureVAE step as device functions, so that they can ride in another launch's grid.
// measure attributes (reference bar_dataset.py:338-500 via measure_vae_trainer.py:167-186): one lane per measure,
//   out[b] = [rhythmic complexity, pitch range / 26, note density, contour / 26]
// so that the labels can ride in another launch's grid (losses.hip: the cross-entropy launch of the MeasureVAE executor carries them;
// they depend on the score alone).  sequence.hip's measure_attributes_kernel is the launch of its own.
#pragma once
#include <cstdint>
#include "common.h"

namespace arvae {

struct AttrArgs {
    const int64_t *score;        // [batch][steps]
    int batch, steps;
    const int32_t *midi;
    const uint8_t *is_note, *is_dens;
    int vocab;
    const float *rhy_w;
    float rhy_norm;
    float *out;                  // [batch][4]; null: nothing to do
};

// measures first, first + stride, ...
__device__ __forceinline__ void measure_attributes_rows(const AttrArgs &a, int first, int stride) {
    // (the walk along a measure is a dependent chain of table look-ups: a measure's notes are requested first, then every
    // table entry they select -- two memory round trips per chunk of 8 ticks instead of two per tick: 10.5 -> ~3 us at B = 256)
    constexpr int CH = 8;
    const int steps = a.steps, vocab = a.vocab;
    for (int b = first; b < a.batch; b += stride) {
        float rhy = 0.f;
        int dens = 0, count = 0, first_m = 0, last = 0, lo = 0, hi = 0;
        for (int t0 = 0; t0 < steps; t0 += CH) {
            int v[CH], dn[CH], nt[CH], md[CH];
            float rw[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                const int t = t0 + u < steps ? t0 + u : steps - 1;
                const int64_t x = a.score[(int64_t)b * steps + t];
                v[u] = (int)(x < 0 ? 0 : (x >= vocab ? vocab - 1 : x));
                rw[u] = a.rhy_w[t];
            }
#pragma unroll
            for (int u = 0; u < CH; ++u) { dn[u] = a.is_dens[v[u]]; nt[u] = a.is_note[v[u]]; md[u] = a.midi[v[u]]; }
#pragma unroll
            for (int u = 0; u < CH; ++u) {
                if (t0 + u >= steps) continue;
                dens += dn[u];
                if (nt[u]) {
                    rhy += rw[u];
                    const int m = md[u];
                    if (count == 0) { first_m = lo = hi = m; }
                    last = m;
                    lo = m < lo ? m : lo;
                    hi = m > hi ? m : hi;
                    ++count;
                }
            }
        }
        float *o = a.out + (int64_t)b * 4;
        o[0] = rhy / a.rhy_norm;
        o[1] = count >= 2 ? (float)(hi - lo) / 26.f : 0.f;
        o[2] = (float)dens / (float)steps;
        o[3] = count >= 2 ? (float)(last - first_m) / 26.f : 0.f;
    }
}

// the beat RNN's constant input b_0 (decoder.py:436-440): its copies x0b[rows] (what the weight gradient reads) and its projection
// gi[b][c] = b_0 * w[c] + bias[c], the same row for every measure -- a function of the parameters alone, so it rides in the first
// lookup launch of the forward pass (sequence.hip embed_fwd4_beat_kernel) instead of being a launch of its own
struct BeatInput {
    const float *b0, *w, *bias;
    int batch, cols, rows;
    float *x0b, *gi;             // gi null: nothing to do
};
__device__ __forceinline__ void beat_input_items(const BeatInput &p, int64_t first, int64_t stride) {
    const float v = p.b0[0];
    const int64_t n_gi = (int64_t)p.batch * p.cols, total = n_gi + p.rows;
    for (int64_t i = first; i < total; i += stride) {
        if (i < n_gi) {
            const int c = (int)(i % p.cols);
            p.gi[i] = fmaf(v, p.w[c], p.bias[c]);
        } else {
            p.x0b[i - n_gi] = v;
        }
    }
}

}  // namespace arvae
