"""SURVEY.md section 8(f) row N4 on the GPU: the evaluation-only entry points of the two trainers (encoder-only /
decoder-only passes through the C-ABI) against the goldens tests/golden/make_goldens.py produced by running the REFERENCE's
own methods (image_vae_trainer.py:274-287,381-403,595-621; measure_vae_trainer.py:188-206,281-308,367-397) and against the
oracle's restatement of them (oracle/inference.py) on the same synthetic loaders."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import arvae_amd  # noqa: E402,F401
from arvae_amd import synthetic as syn  # noqa: E402
from test_oracle_golden import G, close, image_inference_inputs, measure_inference_inputs  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def nchw(t):
    """the HIP models hand out images as (B, 1, H, W) tensors; host copy for numpy"""
    return t.detach().cpu().reshape(t.shape[0], 1, t.shape[-2], t.shape[-1]) if t.dim() == 4 else t.detach().cpu()


class DspritesDataset:
    pass


class MorphoMnistDataset:
    pass


@pytest.mark.parametrize('kind', ['dsprites', 'mnist'])
def test_image_inference_vs_reference_golden_and_oracle(dev, golden_dir, kind):
    from arvae_amd.image_vae import DspritesVAE, MnistVAE
    from arvae_amd.image_vae_trainer import ImageVAETrainer
    from oracle import inference
    g = G(golden_dir, f'inference_{kind}.npz')
    state, batches, eps = image_inference_inputs(kind)
    model = DspritesVAE() if kind == 'dsprites' else MnistVAE()
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    dims = (1, 2, 3, 4, 5) if kind == 'dsprites' else (1, 2, 3, 4, 5, 6)
    trainer = ImageVAETrainer(DspritesDataset() if kind == 'dsprites' else MorphoMnistDataset(), model, reg_type=('all',),
                              reg_dim=dims, beta=1.0)
    trainer.cuda()
    model.eval()
    if kind == 'mnist':       # (inputs, digit labels, morpho labels): image_vae_trainer.py:126-130
        loader = [(torch.from_numpy(x), torch.zeros(len(x), dtype=torch.int64), torch.from_numpy(lab)) for x, lab in batches]
    else:
        loader = [(torch.from_numpy(x), torch.from_numpy(lab)) for x, lab in batches]

    for e in eps:
        model.push_noise(torch.from_numpy(e))
    codes, attrs, names = trainer.compute_representations(loader)
    o_codes, o_attrs, o_names = inference.image_representations(kind, state, batches, eps)
    close(codes, g['codes'], rtol=1e-4, atol=1e-4)
    close(codes, o_codes, rtol=1e-4, atol=1e-4)
    np.testing.assert_array_equal(attrs, g['attrs'])
    assert list(names) == [str(n) for n in g['names']] == list(o_names)

    for e in eps:
        model.push_noise(torch.from_numpy(e))
    loss, acc = trainer.loss_and_acc_test(loader)
    close(loss, g['test_loss'], rtol=1e-4)
    close(acc, g['test_acc'], rtol=1e-5)
    o_loss, o_acc = inference.image_test_loss(kind, state, batches, eps)
    close(loss, o_loss, rtol=1e-4)
    close(acc, o_acc, rtol=1e-5)

    row = trainer.compute_latent_interpolations(g['codes'][3], dim1=2, num_points=5).cpu().numpy()
    assert row.shape == g['row'].shape
    close(row, g['row'], rtol=1e-4, atol=1e-5)
    close(row, inference.image_interpolations(kind, state, g['codes'][3], 2, 5), rtol=1e-4, atol=1e-5)
    grid = trainer.compute_latent_interpolations2d(g['codes'][5], dim1=1, dim2=4, num_points=3).cpu().numpy().reshape(9, -1)
    close(grid[:, ::4], g['grid_samp'], rtol=1e-4, atol=1e-5)
    close(grid.astype(np.float64).sum(1), g['grid_sum'], rtol=1e-4)
    # decode(z) / encode(x) themselves, full tensors against the oracle
    from oracle import image_vae as o_vae
    p = {k: torch.from_numpy(v) for k, v in state.items()}
    z = torch.from_numpy(g['codes'][:7].copy())
    with torch.no_grad():
        close(nchw(model.decode(z.to(dev))), o_vae.decode(kind, p, z).numpy(), rtol=1e-4, atol=1e-4)
        x = torch.from_numpy(batches[1][0])
        dist = model.encode(x.to(dev))
        mu, log_std = o_vae.encode(kind, p, x)
        close(dist.loc.cpu(), mu.numpy(), rtol=1e-4, atol=1e-4)
        close(dist.scale.cpu(), torch.exp(log_std).numpy(), rtol=1e-4, atol=1e-6)


class _FolkDataset:
    """stand-in for data.dataloaders.bar_dataset.FolkNBarDataset: the trainer reads class_name, the vocabulary and n_bars"""
    class_name = '4by4_FolkNBarDataset_1_'
    n_bars = 1

    def __init__(self):
        self.index2note_dicts, self.note2index_dicts = syn.measure_vocabulary()


def test_measure_inference_vs_reference_golden_and_oracle(dev, golden_dir):
    from arvae_amd.measure_vae import MeasureVAE
    from arvae_amd.measure_vae_trainer import MeasureVAETrainer
    from oracle import inference
    g = G(golden_dir, 'inference_measure.npz')
    state, scores, eps = measure_inference_inputs()
    ds = _FolkDataset()
    model = MeasureVAE(ds, 10, 2, 2, 128, 0.5, 32, 2, 128, 0.5, False, 'folk')
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    trainer = MeasureVAETrainer(ds, model, lr=1e-4, reg_type=('all',), reg_dim=(0, 1, 2, 3), beta=0.001, gamma=1.0,
                                capacity=0.0, rand=0, delta=10.0)
    trainer.cuda()
    model.eval()
    loader = [(torch.from_numpy(s), torch.from_numpy(s)) for s in scores]

    for e in eps:
        model.push_noise(torch.from_numpy(e))
    codes, attrs, names = trainer.compute_representations(loader)
    close(codes, g['codes'], rtol=1e-4, atol=1e-4)
    close(attrs, g['attrs'], rtol=1e-6, atol=1e-7)
    assert list(names) == [str(n) for n in g['names']]
    o_codes, o_attrs, _ = inference.measure_representations(state, scores, eps, syn.measure_tables())
    close(codes, o_codes, rtol=1e-4, atol=1e-4)
    close(attrs, o_attrs, rtol=1e-6, atol=1e-7)

    for e in eps:
        model.push_noise(torch.from_numpy(e))
    loss, acc = trainer.loss_and_acc_test(loader)
    close(loss, g['test_loss'], rtol=1e-4)
    close(acc, g['test_acc'], rtol=1e-6)

    _, notes = trainer.decode_latent_codes(torch.from_numpy(g['codes'][:8].copy()))
    np.testing.assert_array_equal(notes.cpu().numpy(), g['notes'])
    np.testing.assert_array_equal(notes.cpu().numpy(), inference.measure_decode(state, g['codes'][:8]))
    _, sweep = trainer.compute_latent_interpolations(g['codes'][3], None, dim1=3, num_points=5)
    np.testing.assert_array_equal(sweep.cpu().numpy(), g['sweep'])
