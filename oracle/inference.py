"""CPU restatement of the evaluation-only entry points (test infrastructure; see oracle/__init__.py).

SURVEY.md section 8(f) row N4: encoder-only / decoder-only passes that feed the reference's
evaluation code.  Pinned by tests/golden/inference_{dsprites,mnist,measure}.npz, which
tests/golden/make_goldens.py produced by running the reference's own methods:
  image   : imagevae/image_vae_trainer.py:274-287 (compute_representations), :381-403
            (compute_latent_interpolations{,2d}), :595-621 (loss_and_acc_test)
  measure : measurevae/measure_vae_trainer.py:188-206 (compute_representations), :281-288
            (decode_latent_codes), :290-308 (compute_latent_interpolations), :367-397 (loss_and_acc_test)
All models are in eval mode (no dropout); eps is the explicit noise of each batch's rsample().
"""
import numpy as np
import torch

from . import attributes, image_vae, losses, measure_vae

IMAGE_ATTRS = {'dsprites': ['shape', 'scale', 'orientation', 'posx', 'posy'],          # column = index + 1
               'mnist': ['area', 'length', 'thickness', 'slant', 'width', 'height']}
MEASURE_ATTRS = ['rhy_complexity', 'pitch_range', 'note_density', 'contour']


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _params(state):
    return {k: _t(v) for k, v in state.items()}


# ---------------------------------------------------------------- image models
def image_representations(kind, state, batches, eps):
    """batches: [(x, labels)], eps: one (B, Z) array per batch -> (codes, attribute columns, names).
    The 'color' / 'digit_identity' column 0 is dropped (image_vae_trainer.py:264-272)."""
    p = _params(state)
    codes, attrs = [], []
    with torch.no_grad():
        for (x, lab), e in zip(batches, eps):
            codes.append(image_vae.forward(kind, p, _t(x), _t(e))[3].numpy())
            attrs.append(np.asarray(lab))
    names = IMAGE_ATTRS[kind]
    return np.concatenate(codes), np.concatenate(attrs)[:, 1:1 + len(names)], names


def image_test_loss(kind, state, batches, eps, dec_dist='bernoulli'):
    """mean over batches of the reconstruction term alone and of the pixel accuracy."""
    p = _params(state)
    loss = acc = 0.0
    with torch.no_grad():
        for (x, _), e in zip(batches, eps):
            logits = image_vae.forward(kind, p, _t(x), _t(e))[0]
            rec = losses.bce_with_logits_per_batch if dec_dist == 'bernoulli' else losses.gaussian_recon_per_batch
            loss += float(rec(logits, _t(x)))
            acc += float(losses.pixel_accuracy(logits, _t(x)))
    return loss / len(batches), acc / len(batches)


def image_interpolations(kind, state, code, dim1, num_points):
    """sigmoid(decode(z)) for z = code with dimension dim1 swept over linspace(-4, 4, num_points)."""
    z = _t(np.asarray(code, np.float32)).reshape(1, -1).repeat(num_points, 1)
    z[:, dim1] = torch.linspace(-4.0, 4.0, num_points)
    with torch.no_grad():
        return torch.sigmoid(image_vae.decode(kind, _params(state), z)).numpy()


def image_interpolations2d(kind, state, code, dim1, dim2, num_points):
    """the same over the meshgrid of two dimensions, dim1 varying slowest ('ij' indexing, as torch.meshgrid defaults)."""
    x = torch.linspace(-4.0, 4.0, num_points)
    z = _t(np.asarray(code, np.float32)).reshape(1, -1).repeat(num_points * num_points, 1)
    z[:, dim1] = x.repeat_interleave(num_points)
    z[:, dim2] = x.repeat(num_points)
    with torch.no_grad():
        return torch.sigmoid(image_vae.decode(kind, _params(state), z)).numpy()


# ---------------------------------------------------------------- MeasureVAE
def measure_representations(state, scores, eps, tables):
    """scores: [(B, 24) int64], eps per batch, tables = synthetic.measure_tables() -> (codes, attributes (n, 4), names)."""
    p = _params(state)
    codes, attrs = [], []
    with torch.no_grad():
        for s, e in zip(scores, eps):
            codes.append(measure_vae.forward(p, _t(s), _t(e), False)[4].numpy())
            attrs.append(attributes.attribute_labels(np.asarray(s), *tables))
    return np.concatenate(codes), np.concatenate(attrs), MEASURE_ATTRS


def measure_test_loss(state, scores, eps):
    """mean over batches of the cross-entropy term alone and of the top-1 accuracy (free-running decoder: train=False)."""
    p = _params(state)
    loss = acc = 0.0
    with torch.no_grad():
        for s, e in zip(scores, eps):
            w = measure_vae.forward(p, _t(s), _t(e), False)[0]
            loss += float(losses.cross_entropy_mean(w, _t(s)))
            acc += float(losses.top1_accuracy(w, _t(s)))
    return loss / len(scores), acc / len(scores)


def measure_decode(state, codes):
    """(n, Z) latent codes -> note indices (n, 1, 24): the decoder alone, argmax feedback, dummy all-zero score."""
    z = _t(np.asarray(codes, np.float32))
    with torch.no_grad():
        return measure_vae.decode(_params(state), z, torch.zeros(z.shape[0], 24, dtype=torch.int64), False)[1].numpy()


def measure_interpolations(state, code, dim1, num_points):
    """note indices (num_points, 24) of the sweep of dimension dim1 over linspace(-4, 4, num_points), one decode per point."""
    z = _t(np.asarray(code, np.float32)).reshape(1, -1).repeat(num_points, 1)
    z[:, dim1] = torch.linspace(-4.0, 4.0, num_points)
    return np.concatenate([measure_decode(state, z[n:n + 1].numpy()) for n in range(num_points)])[:, 0, :]
