"""Deterministic synthetic inputs and weights for the AR-VAE training path.

Everything here is plain numpy driven by ``numpy.random.RandomState`` so that
the golden-vector generator (tests/golden/make_goldens.py, which imports the
reference), the CPU oracle, the parity tests and bench.py all see bit-identical
tensors without storing them.  No torch RNG is involved.

Shapes / distributions follow SURVEY.md section 8(d):
  * dSprites-shaped batch   (reference loader: data/dataloaders/dsprites_dataset.py:38-53)
  * Morpho-MNIST-shaped     (reference loader: data/dataloaders/mnist_dataset.py:60-82)
  * 24-tick measure batch   (reference loader: data/dataloaders/bar_dataset.py:179-222)
"""
import math
import zlib

import numpy as np

# ----------------------------------------------------------------------------
# measure vocabulary used by every synthetic MeasureVAE config (V = 35)
# ----------------------------------------------------------------------------
SLUR, START, END, REST, NONE = '__', 'START', 'END', 'rest', None
_PITCH_CLASSES = ['C', 'C#', 'D', 'D#', 'E', 'F', 'F#', 'G', 'G#', 'A', 'A#', 'B']
MIDI_LO, MIDI_HI = 55, 84          # reference pitch range: bar_dataset.py:22


def midi_to_name(midi):
    return f'{_PITCH_CLASSES[midi % 12]}{midi // 12 - 1}'


def name_to_midi(name):
    """Inverse of midi_to_name (what music21.pitch.Pitch(name).midi returns)."""
    octave = int(name[-1])
    return 12 * (octave + 1) + _PITCH_CLASSES.index(name[:-1])


def measure_vocabulary():
    """index2note / note2index dicts in the reference's format (V = 35)."""
    symbols = [SLUR, START, END, REST, NONE] + [midi_to_name(m) for m in range(MIDI_LO, MIDI_HI + 1)]
    index2note = {i: s for i, s in enumerate(symbols)}
    note2index = {s: i for i, s in enumerate(symbols)}
    return index2note, note2index


def measure_tables():
    """(midi_lut int32[V], is_note uint8[V], is_density_note uint8[V]).

    is_note excludes slur/START/END/rest/None (bar_dataset.py:452-461);
    is_density_note does NOT exclude None (bar_dataset.py:348-356)."""
    index2note, _ = measure_vocabulary()
    v = len(index2note)
    midi = np.zeros(v, np.int32)
    is_note = np.zeros(v, np.uint8)
    is_dens = np.zeros(v, np.uint8)
    for i, s in index2note.items():
        if s in (SLUR, START, END, REST, NONE):
            is_dens[i] = 1 if s is NONE else 0
            continue
        midi[i] = name_to_midi(s)
        is_note[i] = 1
        is_dens[i] = 1
    return midi, is_note, is_dens


# ----------------------------------------------------------------------------
# inputs
# ----------------------------------------------------------------------------
def dsprites_batch(batch, seed=1234):
    """x (B,1,64,64) float32 in {0,1} with ~10 % foreground, labels (B,6) on the
    real factor grid co1 sh3 sc6 or40 x32 y32 (so ties occur as in real data)."""
    rs = np.random.RandomState(seed)
    x = (rs.random_sample((batch, 1, 64, 64)) < 0.10).astype(np.float32)
    lab = np.empty((batch, 6), np.float32)
    lab[:, 0] = 1.0
    lab[:, 1] = rs.randint(1, 4, batch)
    lab[:, 2] = 0.5 + 0.1 * rs.randint(0, 6, batch)
    lab[:, 3] = 2.0 * math.pi * rs.randint(0, 40, batch) / 39.0
    lab[:, 4] = rs.randint(0, 32, batch) / 31.0
    lab[:, 5] = rs.randint(0, 32, batch) / 31.0
    return x, lab


_MNIST_RANGES = [(0, 9), (0, 350), (0, 100), (0, 15), (-1.2, 1.2), (0, 30), (0, 30)]


def mnist_batch(batch, seed=4321):
    """x (B,1,28,28) float32 grey ink on ~19 % of pixels, labels (B,7)
    (ranges: reference image_vae_trainer.py:30-38)."""
    rs = np.random.RandomState(seed)
    ink = rs.random_sample((batch, 1, 28, 28)) < 0.19
    x = (rs.random_sample((batch, 1, 28, 28)) * ink).astype(np.float32)
    lab = np.empty((batch, 7), np.float32)
    lab[:, 0] = rs.randint(0, 10, batch)
    for c in range(1, 7):
        lo, hi = _MNIST_RANGES[c]
        lab[:, c] = lo + (hi - lo) * rs.random_sample(batch)
    return x, lab


def measure_batch(batch, seed=5, p_slur=0.6):
    """score (B,24) int64; P(slur)=0.6, the other V-1 symbols uniform."""
    rs = np.random.RandomState(seed)
    v = len(measure_vocabulary()[0])
    other = rs.randint(1, v, (batch, 24))
    slur = rs.random_sample((batch, 24)) < p_slur
    return np.where(slur, 0, other).astype(np.int64)


def normal_noise(shape, seed):
    """Explicit epsilon for the reparameterisation (parity runs pass it in)."""
    return np.random.RandomState(seed).standard_normal(shape).astype(np.float32)


def dropout_masks(shapes, seed, p=0.5):
    """Explicit Bernoulli(1-p) keep masks, uint8 in {0,1}."""
    rs = np.random.RandomState(seed)
    return [(rs.random_sample(s) >= p).astype(np.uint8) for s in shapes]


# ----------------------------------------------------------------------------
# weights
# ----------------------------------------------------------------------------
def synth_tensor(name, shape, seed=0, gain=1.6):
    """Deterministic parameter tensor keyed by (name, seed): order independent.

    Weights ~ N(0, gain*sqrt(2/(fan_in+fan_out))) (Xavier-normal shaped, the
    reference's init utils/model.py:90-97, with a gain that keeps ReLU stacks
    alive); biases and the learned start vectors ~ U(-0.1, 0.1)."""
    rs = np.random.RandomState((zlib.crc32(name.encode()) ^ (seed * 2654435761)) & 0x7fffffff)
    shape = tuple(shape)
    leaf = name.split('.')[-1]
    if 'weight' not in leaf:
        return rs.uniform(-0.1, 0.1, shape).astype(np.float32)
    if len(shape) == 1:
        fan_in = fan_out = shape[0]
    else:
        rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
        fan_in, fan_out = shape[1] * rf, shape[0] * rf
    std = gain * math.sqrt(2.0 / (fan_in + fan_out))
    return (rs.standard_normal(shape) * std).astype(np.float32)


def synth_state(shapes, seed=0, gain=1.6):
    """{name: ndarray} for an ordered {name: shape} mapping."""
    return {k: synth_tensor(k, s, seed, gain) for k, s in shapes.items()}


def sample_indices(name, numel, k=16):
    """k deterministic flat indices into a tensor called `name` (golden spot checks)."""
    rs = np.random.RandomState(zlib.crc32(('idx:' + name).encode()) & 0x7fffffff)
    return rs.randint(0, numel, k)
