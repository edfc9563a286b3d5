"""Pins the CPU oracle against the golden vectors produced by the reference
(tests/golden/make_goldens.py).  Runs anywhere (no GPU, no /root/reference)."""
import os

import numpy as np
import pytest
import torch

from arvae_amd import synthetic as syn
from oracle import attributes, image_vae, losses, measure_vae, step

torch.set_num_threads(1)


def G(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def close(a, b, rtol=1e-4, atol=0.0):
    np.testing.assert_allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=atol)


# ---------------------------------------------------------------- G1
@pytest.mark.parametrize('n', [7, 64, 512])
def test_reg_loss_closed_form(golden_dir, n):
    g = G(golden_dir, 'reg_loss.npz')
    x = g[f'n{n}/x']
    for tag in ('cont', 'ties'):
        a = g[f'n{n}/a_{tag}']
        for delta in (1.0, 10.0):
            for gamma in (1.0, 10.0):
                key = f'n{n}_{tag}_d{delta:g}_g{gamma:g}'
                loss, grad = losses.reg_loss_closed_form(x, a, gamma, delta)
                close(loss, g[f'{key}/loss'], rtol=2e-6)
                close(grad, g[f'{key}/grad'], rtol=1e-4, atol=2e-7 * gamma * delta)
                # differentiable torch form agrees too
                z = torch.from_numpy(np.stack([x * 0, x], 1)).requires_grad_(True)
                lab = torch.from_numpy(np.stack([a * 0, a], 1))
                lt = losses.reg_loss(z, lab, (1,), gamma, delta)
                lt.backward()
                close(lt.item(), g[f'{key}/loss'], rtol=2e-6)
                close(z.grad[:, 1].numpy(), g[f'{key}/grad'], rtol=1e-4, atol=2e-7 * gamma * delta)


def test_reg_loss_row_block_equals_global(golden_dir):
    """DP identity: shard-row losses sum to the global loss and the
    row-block gradient equals the global gradient rows."""
    g = G(golden_dir, 'reg_loss.npz')
    x, a = g['n64/x'], g['n64/a_ties']
    loss, grad = losses.reg_loss_closed_form(x, a, 10.0, 1.0)
    parts = [losses.reg_loss_closed_form(x[s], a[s], 10.0, 1.0, cols_x=x, cols_a=a)
             for s in (slice(0, 32), slice(32, 64))]
    close(sum(p[0] for p in parts), loss, rtol=1e-12)
    close(np.concatenate([p[1] for p in parts]), grad, rtol=1e-12, atol=1e-15)


# ---------------------------------------------------------------- G2
@pytest.mark.parametrize('b,z', [(8, 10), (64, 10), (32, 32)])
def test_latent_head(golden_dir, b, z):
    g = G(golden_dir, 'latent_head.npz')
    mu, ls, eps = (g[f'b{b}_z{z}/{k}'] for k in ('mu', 'log_std', 'eps'))
    w = syn.normal_noise((b, z), seed=99)
    close(mu + eps * np.exp(ls), g[f'b{b}_z{z}/z'], rtol=1e-6, atol=1e-6)
    for c in (0.0, 25.0):
        for beta in (4.0, 0.001):
            key = f'b{b}_z{z}_c{c:g}_beta{beta:g}'
            val, dmu, dls = losses.kld_closed_form(mu, ls, beta, c)
            close(val, g[f'{key}/kld'], rtol=1e-5)
            # golden grads include d(sum(z*w)): dz/dmu = 1, dz/dls = eps*sigma
            close(dmu + w, g[f'{key}/dmu'], rtol=1e-4, atol=1e-6)
            close(dls + w * eps * np.exp(ls), g[f'{key}/dls'], rtol=1e-4, atol=1e-6)
            mt, st = torch.from_numpy(mu), torch.from_numpy(np.exp(ls))
            close(losses.kld_loss(mt, st, beta, c).item(), g[f'{key}/kld'], rtol=1e-5)


# ---------------------------------------------------------------- G3
@pytest.mark.parametrize('b,hw', [(4, 64), (16, 28)])
def test_recon_terms(golden_dir, b, hw):
    g = G(golden_dir, 'recon.npz')
    rs = np.random.RandomState(b * hw)
    logits = (3.0 * rs.standard_normal((b, 1, hw, hw))).astype(np.float32)
    logits.ravel()[::97] = 0.0
    x = (rs.random_sample((b, 1, hw, hw)) < 0.2).astype(np.float32)
    if hw == 28:
        x = (x * rs.random_sample(x.shape)).astype(np.float32)
    for dist, fn in (('bernoulli', losses.bce_with_logits_per_batch), ('gaussian', losses.gaussian_recon_per_batch)):
        lt = torch.from_numpy(logits).requires_grad_(True)
        loss = fn(lt, torch.from_numpy(x))
        loss.backward()
        close(loss.item(), g[f'b{b}_{hw}_{dist}/loss'], rtol=1e-5)
        close(lt.grad.numpy().ravel()[::131], g[f'b{b}_{hw}_{dist}/dlogits_samp'], rtol=1e-4, atol=1e-7)
        close(lt.grad.double().abs().sum().item(), g[f'b{b}_{hw}_{dist}/dlogits_abs_sum'], rtol=1e-5)
    close(losses.pixel_accuracy(torch.from_numpy(logits), torch.from_numpy(x)).item(), g[f'b{b}_{hw}/acc'], rtol=1e-6)


@pytest.mark.parametrize('b', [5, 32])
def test_cross_entropy(golden_dir, b):
    g = G(golden_dir, 'recon.npz')
    w = torch.from_numpy(g[f'ce_b{b}/w']).requires_grad_(True)
    tgt = torch.from_numpy(g[f'ce_b{b}/tgt'])
    ce = losses.cross_entropy_mean(w, tgt)
    ce.backward()
    close(ce.item(), g[f'ce_b{b}/loss'], rtol=1e-5)
    close(losses.top1_accuracy(w.detach(), tgt).item(), g[f'ce_b{b}/acc'], rtol=1e-6)
    close(w.grad.numpy().ravel()[::37], g[f'ce_b{b}/dw_samp'], rtol=1e-4, atol=1e-8)


# ---------------------------------------------------------------- G4 / G5
def check_step(res, g, state_before, names, grad_rtol=1e-3, samp_atol=1e-4):
    t = res['terms']
    for k in ('recons', 'dist', 'reg', 'loss', 'acc'):
        close(t[k], g[k], rtol=1e-4)
    rows = g['z'].shape[0]                                      # headline-size fixtures keep the first rows + sums
    close(t['z'][:rows], g['z'], rtol=0, atol=1e-4)
    close(t['mu'][:rows], g['mu'], rtol=0, atol=1e-4)
    close(t['sigma'][:rows], g['sigma'], rtol=1e-4, atol=1e-5)
    if 'z_sum' in g:
        for k in ('z', 'mu', 'sigma'):
            full = np.asarray(t[k], np.float64)
            close(full.sum(), g[f'{k}_sum'], rtol=1e-4, atol=1e-4 * full.shape[0])
            close(np.abs(full).sum(), g[f'{k}_abs_sum'], rtol=1e-4)
    for name in names:
        gr = res['grads'][name].astype(np.float64).ravel()
        gn = float(g[f'gnorm/{name}'])
        close(np.sqrt((gr * gr).sum()), gn, rtol=grad_rtol)
        idx = syn.sample_indices(name, gr.size)
        close(gr[idx], g[f'gsamp/{name}'], rtol=grad_rtol, atol=samp_atol * gn / np.sqrt(gr.size) + 1e-7)
        d = (res['params'][name].astype(np.float64) - state_before[name].astype(np.float64)).ravel()
        close(np.sqrt((d * d).sum()), g[f'dnorm/{name}'], rtol=2e-3)


IMAGE_CASES = [
    ('dsprites_step_b8.npz', 'dsprites', 8, 1, 1234, 11, 4.0, 0.0, 'bernoulli', None, 1.6),
    ('dsprites_step_b64.npz', 'dsprites', 64, 1, 1234, 12, 4.0, 0.0, 'bernoulli', None, 1.6),
    ('dsprites_step_b8_cap_gauss.npz', 'dsprites', 8, 2, 77, 13, 1.0, 25.0, 'gaussian', None, 1.6),
    ('mnist_step_eval.npz', 'mnist', 8, 3, 4321, 14, 1.0, 0.0, 'bernoulli', None, 0.7),
    ('mnist_step_train.npz', 'mnist', 8, 3, 4321, 15, 1.0, 0.0, 'bernoulli', 21, 0.7),
    # BASELINE.json configs[1] and [2] at their own batch sizes (tests/golden/make_goldens.py gen_headline_steps)
    ('dsprites_step_b512.npz', 'dsprites', 512, 1, 1234, 1, 4.0, 0.0, 'bernoulli', None, 1.6),
    ('mnist_step_train_b1024.npz', 'mnist', 1024, 3, 4321, 15, 1.0, 0.0, 'bernoulli', 21, 0.7),
]


@pytest.mark.parametrize('case', IMAGE_CASES, ids=[c[0][:-4] for c in IMAGE_CASES])
def test_image_step(golden_dir, case):
    fname, kind, b, wseed, xseed, eseed, beta, cap, dist, mseed, gain = case
    g = G(golden_dir, fname)
    state = syn.synth_state(image_vae.SHAPES[kind], wseed, gain)
    x, lab = (syn.dsprites_batch if kind == 'dsprites' else syn.mnist_batch)(b, seed=xseed)
    eps = syn.normal_noise((b, image_vae.Z_DIM[kind]), seed=eseed)
    dims = (1, 2, 3, 4, 5) if kind == 'dsprites' else (1, 2, 3, 4, 5, 6)
    masks = None if mseed is None else syn.dropout_masks([(b,) + s for s in image_vae.MNIST_MASK_SHAPES], mseed)
    res = step.image_step(kind, state, x, lab, eps, dims, beta, 10.0, 1.0, capacity=cap, dec_dist=dist, masks=masks)
    # sampled gradient entries: two fp32 CPU runs (the reference with its thread count, the oracle with one thread) sum a
    # 1024-image batch in different orders; entries near zero move by ~3e-4 of the tensor's RMS
    check_step(res, g, state, list(state), samp_atol=1e-4 if b <= 64 else 4e-4)
    lg = res['terms']['logits'].ravel()
    close(lg.astype(np.float64).sum(), g['logits_sum'], rtol=1e-4, atol=1e-2)
    close(lg[syn.sample_indices('logits', lg.size, 64)], g['logits_samp'], rtol=1e-4, atol=1e-4)


# ---------------------------------------------------------------- G6 / G7
def measure_state(wseed):
    state = syn.synth_state(measure_vae.shapes(), wseed)
    state['decoder.tick_emb_to_note_emb.0.bias'] = state['decoder.tick_emb_to_note_emb.0.bias'] + np.float32(0.5)
    state['decoder.tick_emb_to_note_emb.0.weight'] = state['decoder.tick_emb_to_note_emb.0.weight'] * np.float32(3.0)
    return state


MEASURE_CASES = [('measure_step_tf.npz', 5, 31, True, 16), ('measure_step_free.npz', 5, 32, False, 16),
                 ('measure_step_eval.npz', 6, 33, False, 16),
                 # BASELINE.json configs[4] at its own batch size
                 ('measure_step_tf_b256.npz', 5, 1, True, 256), ('measure_step_free_b256.npz', 5, 2, False, 256)]


@pytest.mark.parametrize('case', MEASURE_CASES, ids=[c[0][:-4] for c in MEASURE_CASES])
def test_measure_step(golden_dir, case):
    fname, sseed, eseed, teacher, b = case
    g = G(golden_dir, fname)
    state = measure_state(4)
    score = syn.measure_batch(b, seed=sseed)
    eps = syn.normal_noise((b, 32), seed=eseed)
    attr = attributes.attribute_labels(score, *syn.measure_tables())
    close(attr, g['attr'], rtol=1e-6, atol=1e-7)
    res = step.measure_step(state, score, eps, attr, (0, 1, 2, 3), 0.001, 1.0, 10.0, teacher)
    np.testing.assert_array_equal(res['terms']['samples'], g['samples'])
    check_step(res, g, state, list(state), grad_rtol=2e-3, samp_atol=1e-4 if b <= 64 else 4e-4)
    w = res['terms']['weights']
    close(w[0], g['weights_row0'], rtol=1e-4, atol=1e-5)
    close(w.ravel()[syn.sample_indices('weights', w.size, 128)], g['weights_samp'], rtol=1e-4, atol=1e-5)


def test_attribute_labels(golden_dir):
    g = G(golden_dir, 'attributes.npz')
    attr = attributes.attribute_labels(g['score'], *syn.measure_tables())
    close(attr, g['attr'], rtol=1e-6, atol=1e-7)
    assert attr[0].tolist() == [0, 0, 0, 0]            # all slur
    assert attr[2, 2] == 1.0 and attr[2, 0] == 0.0     # all `None`: density counts them, rhythm does not


# ---------------------------------------------------------------- G9: evaluation-only inference (N4)
INFER = {'dsprites': dict(wseed=1, gain=1.6, xseed=600, eseed=40, mk=syn.dsprites_batch),
         'mnist': dict(wseed=3, gain=0.7, xseed=700, eseed=50, mk=syn.mnist_batch)}


def image_inference_inputs(kind):
    """the synthetic loader make_goldens.py::image_inference ran the reference on (3 batches of 16)."""
    c = INFER[kind]
    state = syn.synth_state(image_vae.SHAPES[kind], c['wseed'], c['gain'])
    batches = [c['mk'](16, seed=c['xseed'] + i) for i in range(3)]
    eps = [syn.normal_noise((16, image_vae.Z_DIM[kind]), seed=c['eseed'] + i) for i in range(3)]
    return state, batches, eps


def measure_inference_inputs():
    state = measure_state(4)
    scores = [syn.measure_batch(16, seed=800 + i) for i in range(3)]
    eps = [syn.normal_noise((16, 32), seed=60 + i) for i in range(3)]
    return state, scores, eps


@pytest.mark.parametrize('kind', ['dsprites', 'mnist'])
def test_image_inference_entry_points(golden_dir, kind):
    from oracle import inference
    g = G(golden_dir, f'inference_{kind}.npz')
    state, batches, eps = image_inference_inputs(kind)
    codes, attrs, names = inference.image_representations(kind, state, batches, eps)
    close(codes, g['codes'], rtol=1e-5, atol=1e-5)
    np.testing.assert_array_equal(attrs, g['attrs'])
    assert list(names) == [str(n) for n in g['names']]
    loss, acc = inference.image_test_loss(kind, state, batches, eps)
    close(loss, g['test_loss'], rtol=1e-5)
    close(acc, g['test_acc'], rtol=1e-6)
    row = inference.image_interpolations(kind, state, g['codes'][3], 2, 5)
    close(row, g['row'], rtol=1e-4, atol=1e-6)
    grid = inference.image_interpolations2d(kind, state, g['codes'][5], 1, 4, 3).reshape(9, -1)
    close(grid[:, ::4], g['grid_samp'], rtol=1e-4, atol=1e-6)
    close(grid.astype(np.float64).sum(1), g['grid_sum'], rtol=1e-5)


def test_measure_inference_entry_points(golden_dir):
    from oracle import inference
    g = G(golden_dir, 'inference_measure.npz')
    state, scores, eps = measure_inference_inputs()
    codes, attrs, names = inference.measure_representations(state, scores, eps, syn.measure_tables())
    close(codes, g['codes'], rtol=1e-5, atol=1e-5)
    close(attrs, g['attrs'], rtol=1e-6, atol=1e-7)
    assert list(names) == [str(n) for n in g['names']]
    loss, acc = inference.measure_test_loss(state, scores, eps)
    close(loss, g['test_loss'], rtol=1e-5)
    close(acc, g['test_acc'], rtol=1e-6)
    np.testing.assert_array_equal(inference.measure_decode(state, g['codes'][:8]), g['notes'])
    np.testing.assert_array_equal(inference.measure_interpolations(state, g['codes'][3], 3, 5), g['sweep'])
    assert len(np.unique(g['sweep'])) > 2 and len(np.unique(g['notes'])) > 3      # non-trivial decodes


# ---------------------------------------------------------------- Philox4x32-10 (device RNG of eps / dropout masks)
def test_philox_known_answers():
    """oracle/philox.py against the known-answer vectors of the Random123 distribution (kat_vectors: philox4x32, 10 rounds)"""
    from oracle import philox
    kat = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0], [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, want in kat:
        assert [int(v) for v in philox.philox4x32_10(ctr, key)] == want
    n = philox.normal(1 << 16, seed=1234, offset=3)
    assert abs(float(n.mean())) < 0.02 and abs(float(n.std()) - 1.0) < 0.02 and np.isfinite(n).all()
    m = philox.keep_mask(1 << 16, 0.5, seed=9)
    assert abs(float(m.mean()) - 0.5) < 0.01 and set(np.unique(m)) == {0, 1}
