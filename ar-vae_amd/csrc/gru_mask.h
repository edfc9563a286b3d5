// Dropout between two stacked GRU layers inside the lower layer's sequence launches (round 5; gru_seq.hip, used by plan_measure.hip).
// nn.GRU(dropout = p) feeds layer l + 1 with keep * mask * h of layer l (measurevae/encoder.py:27-34, decoder.py:338-368).  Forward: the
// recurrence that produces h also stores the masked copy (one keep byte loaded and one more store per element and step -- a lane has
// ONE element per step since the kernels own four rows per workgroup, gru_seq.hip gru_seq_fwd_h2_kernel).  Backward: the gradient
// that arrives is the one w.r.t. the masked copy; the recurrence multiplies it by keep * mask as it loads it.  Six ~5 us launches of
// a MeasureVAE step are gone (scale_mask_kernel x 4, scale_mask_tick_kernel x 2).
#pragma once
#include <cstdint>
#include "../../include/arvae_hip.h"

namespace arvae {

struct GruSeqMask {
    const uint8_t *mask;         // keep bytes (0 / 1); null: this sequence has no dropout on its output
    float keep;                  // 1 / (1 - p)
    float *h_masked;             // forward: keep * mask * h, addressed like h_all with hm_stride floats per row
    int64_t hm_stride;
    // byte offset of the keep byte of (step t, row r, unit j): t * tstride + (r / group) * gstride + (r % group) * rstride + j
    // (the tick RNN's rows are (beat, measure) pairs over ticks-in-beat steps, its masks are ordered (tick, measure))
    int64_t tstride, rstride, gstride;
    int32_t group;               // 0: no groups (r / group = 0, r % group = r)
};

// true when the default (fp16 two-term) sequence kernels run: the diagnostic build's fp32 / bf16 alternatives take no masks
bool gru_seq_masks_supported();
// arvae_gru_seq_fwd / _bwd with masks[i] for seqs[i] (masks may be null: the plain calls)
int gru_seq_fwd_masked(const arvae_gru_seq_t *seqs, const GruSeqMask *masks, int32_t nseq, int32_t steps, int32_t rows, int32_t hidden,
                       arvae_stream_t stream);
int gru_seq_bwd_masked(const arvae_gru_seq_t *seqs, const GruSeqMask *masks, int32_t nseq, int32_t steps, int32_t rows, int32_t hidden,
                       arvae_stream_t stream);

}  // namespace arvae
