"""Measure attribute labels restated in closed form (test infrastructure; see oracle/__init__.py).

Follows reference measure_vae_trainer.py:167-186 -> bar_dataset.py:
  get_rhy_complexity           :442-468  (weights bar_dataset_helpers.py:21-30)
  get_pitch_range_in_measure   :360-390
  get_note_density_in_measure  :338-358  (does NOT exclude the `None` symbol)
  get_contour                  :470-500
The music21 pitch lookup is replaced by an index -> MIDI table.
"""
import numpy as np

RHY_COEFFS = np.array([0.20, 1, 2, 0.5, 2, 1, 0.67, 1, 2, 0.5, 2, 1,
                       0.25, 1, 2, 0.5, 2, 1, 0.67, 1, 2, 0.5, 2, 1], np.float64)


def attribute_labels(score, midi_lut, is_note, is_density_note):
    """score (B,24) int -> (B,4) float32 [rhy_complexity, pitch_range, note_density, contour]."""
    score = np.asarray(score)
    b, t = score.shape
    onset = is_note[score].astype(bool)
    w = RHY_COEFFS.astype(np.float32)
    out = np.zeros((b, 4), np.float32)
    out[:, 0] = (w[None, :] * onset.astype(np.float32)).sum(1, dtype=np.float32) / w.sum(dtype=np.float32)
    out[:, 2] = is_density_note[score].sum(1).astype(np.float32) / np.float32(t)
    midi = midi_lut[score]
    for i in range(b):
        notes = midi[i][onset[i]]
        if notes.size >= 2:
            out[i, 1] = np.float32(notes.max() - notes.min()) / np.float32(26)
            out[i, 3] = np.float32(notes[-1] - notes[0]) / np.float32(26)
    return out
