#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference; it never travels to the
GPU box):

    python tests/golden/make_goldens.py

The reference modules are imported unmodified from /root/reference on
PyTorch-CPU.  Third-party packages that are absent here and only serve
logging / plotting / dataset parsing (tensorboardX, torchvision, seaborn,
music21, ...) are replaced by ``MagicMock`` entries in ``sys.modules``; every
arithmetic op on the training path runs in real PyTorch.  The only behavioural
stubs are:
  * ``music21.pitch.Pitch(name).midi``  -> arvae_amd.synthetic.name_to_midi
  * ``torch.distributions.normal._standard_normal`` -> explicit epsilon queue
    (Normal.rsample() draws its noise there: reference mnist_vae.py:79)
  * ``nn.Dropout`` modules of MnistVAE -> explicit keep-masks (train-mode case)
Inputs and weights come from arvae_amd.synthetic (numpy RandomState), so the
fixtures hold only OUTPUTS (a few KB each); tests regenerate the inputs.

What is written:
  G1 reg_loss.npz      Trainer.reg_loss_sign / compute_reg_loss  (utils/trainer.py:369-403)
  G2 latent_head.npz   rsample + compute_kld_loss               (mnist_vae.py:74-87, utils/trainer.py:354-367)
  G3 recon.npz         reconstruction_loss / mean_accuracy       (image_vae_trainer.py:623-655)
                       mean_crossentropy_loss / mean_accuracy    (utils/trainer.py:247-282)
  G4 dsprites_step_b{8,64}.npz   full ImageVAETrainer step + Adam
  G5 mnist_step_{eval,train}.npz full step, dropout off / explicit masks
  G6 measure_step_{tf,free,eval}.npz  MeasureVAETrainer step (V=35)
  G8 dsprites_step_b512.npz, mnist_step_train_b1024.npz, measure_step_{tf,free}_b256.npz
                       the same steps at BASELINE.json's batch sizes (configs[1], [2], [4]); z / mu / sigma keep their first
                       HEAD_ROWS rows (+ whole-tensor sums), everything else as in G4-G6
  G7 attributes.npz    compute_attribute_labels                   (measure_vae_trainer.py:167-186)
  G9 inference_{dsprites,mnist,measure}.npz   the evaluation-only entry points:
                       compute_representations, loss_and_acc_test, compute_latent_interpolations{,2d}
                       (image_vae_trainer.py:274-287,381-403,595-621), decode_latent_codes,
                       compute_latent_interpolations (measure_vae_trainer.py:188-206,281-308,367-397)
                       on synthetic loaders; make_grid / the music21 converters are replaced by pass-throughs
"""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = '/root/reference'
sys.path.insert(0, REPO)

import torch  # noqa: E402
from torch import nn  # noqa: E402

from arvae_amd import synthetic as syn  # noqa: E402


# ----------------------------------------------------------------------------
# import the reference with stubs for absent third-party packages
# ----------------------------------------------------------------------------
def install_stubs():
    for name in ['tensorboardX', 'torchvision', 'torchvision.utils', 'torchvision.models',
                 'torchvision.datasets', 'torchvision.transforms', 'seaborn', 'pypianoroll',
                 'pretty_midi', 'skimage', 'skimage.morphology', 'skimage.transform',
                 'skimage.filters', 'skimage.measure', 'skimage.draw',
                 'music21.abcFormat', 'music21.meter', 'music21.note']:
        sys.modules.setdefault(name, MagicMock())
    resnet = types.ModuleType('torchvision.models.resnet')

    class ResNet(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    class BasicBlock(nn.Module):
        pass
    resnet.ResNet, resnet.BasicBlock = ResNet, BasicBlock
    sys.modules['torchvision.models.resnet'] = resnet

    music21 = types.ModuleType('music21')
    pitch = types.ModuleType('music21.pitch')

    class Pitch:
        def __init__(self, name):
            self.midi = syn.name_to_midi(name)
    pitch.Pitch = Pitch
    music21.pitch = pitch
    for sub in ('abcFormat', 'meter', 'note'):
        setattr(music21, sub, sys.modules['music21.' + sub])
    music21.__getattr__ = lambda name: MagicMock()
    sys.modules['music21'] = music21
    sys.modules['music21.pitch'] = pitch


install_stubs()
sys.path.insert(0, REFERENCE)
torch.set_num_threads(1)                     # fixed CPU summation order
os.chdir('/tmp')                             # the reference creates dirs relative to cwd

from utils.trainer import Trainer  # noqa: E402
from imagevae.dsprites_vae import DspritesVAE  # noqa: E402
from imagevae.mnist_vae import MnistVAE  # noqa: E402
from imagevae.image_vae_trainer import ImageVAETrainer  # noqa: E402
from measurevae.measure_vae import MeasureVAE  # noqa: E402
from measurevae.measure_vae_trainer import MeasureVAETrainer  # noqa: E402
from data.dataloaders.bar_dataset import FolkNBarDataset  # noqa: E402
import torch.distributions.normal as _tdn  # noqa: E402


class EpsQueue:
    """Feeds explicit noise to Normal.rsample()."""

    def __init__(self):
        self.queue = []
        self._orig = _tdn._standard_normal

    def push(self, eps):
        self.queue.append(torch.from_numpy(np.asarray(eps)))

    def __call__(self, shape, dtype, device):
        if self.queue:
            e = self.queue.pop(0)
            assert tuple(e.shape) == tuple(shape), (e.shape, shape)
            return e.to(dtype)
        return self._orig(shape, dtype=dtype, device=device)


EPS = EpsQueue()
_tdn._standard_normal = EPS


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print(f'wrote {name}: {os.path.getsize(path) / 1024:.1f} KB, {len(arrays)} arrays')


def load_synth_weights(model, seed, gain=1.6):
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    state = {k: t(v) for k, v in syn.synth_state(shapes, seed, gain).items()}
    model.load_state_dict(state)
    return shapes


HEAD_ROWS = 64          # rows of z / mu / sigma the headline-size fixtures keep


def head_rows(out, keys=('z', 'mu', 'sigma')):
    """a headline-size fixture keeps the first HEAD_ROWS rows of the per-row outputs and their float64 sums"""
    for k in keys:
        full = np.asarray(out[k])
        out[f'{k}_sum'] = full.astype(np.float64).sum()
        out[f'{k}_abs_sum'] = np.abs(full.astype(np.float64)).sum()
        out[k] = full[:HEAD_ROWS]
    return out


def grad_and_update_summaries(model, before):
    """per-parameter grad L2 norm, 16 sampled grad entries, post-Adam deltas."""
    out = {}
    for name, p in model.named_parameters():
        g = p.grad.detach().double().numpy().ravel()
        idx = syn.sample_indices(name, g.size)
        out[f'gnorm/{name}'] = np.sqrt((g * g).sum())
        out[f'gsamp/{name}'] = p.grad.detach().numpy().ravel()[idx]
        d = (p.detach().double().numpy() - before[name].astype(np.float64)).ravel()
        out[f'dnorm/{name}'] = np.sqrt((d * d).sum())
        out[f'dsamp/{name}'] = (p.detach().numpy().ravel() - before[name].ravel())[idx]
    return out


# ----------------------------------------------------------------------------
# G1 - G3: op level
# ----------------------------------------------------------------------------
def gen_reg_loss():
    out = {}
    for n in (7, 64, 512):
        rs = np.random.RandomState(100 + n)
        x = rs.standard_normal(n).astype(np.float32)
        a_cont = rs.standard_normal(n).astype(np.float32)
        a_ties = rs.randint(1, 4, n).astype(np.float32)           # many ties (dSprites 'shape')
        for tag, a in (('cont', a_cont), ('ties', a_ties)):
            for delta in (1.0, 10.0):
                for gamma in (1.0, 10.0):
                    z = t(np.stack([x * 0, x], 1)).requires_grad_(True)   # column 1 is regularised
                    loss = Trainer.compute_reg_loss(z, t(a), 1, gamma=gamma, factor=delta)
                    loss.backward()
                    key = f'n{n}_{tag}_d{delta:g}_g{gamma:g}'
                    out[f'{key}/loss'] = loss.item()
                    out[f'{key}/grad'] = z.grad[:, 1].numpy().copy()
        out[f'n{n}/x'] = x
        out[f'n{n}/a_cont'] = a_cont
        out[f'n{n}/a_ties'] = a_ties
    save('reg_loss.npz', **out)


def gen_latent_head():
    out = {}
    for b, z in ((8, 10), (64, 10), (32, 32)):
        rs = np.random.RandomState(7 * b + z)
        mu = rs.standard_normal((b, z)).astype(np.float32)
        ls = (0.5 * rs.standard_normal((b, z)) - 0.5).astype(np.float32)
        eps = syn.normal_noise((b, z), seed=b + z)
        for c in (0.0, 25.0):
            for beta in (4.0, 0.001):
                mu_t = t(mu).requires_grad_(True)
                ls_t = t(ls).requires_grad_(True)
                dist = torch.distributions.Normal(loc=mu_t, scale=torch.exp(ls_t))
                EPS.push(eps)
                zt = dist.rsample()
                prior = torch.distributions.Normal(torch.zeros_like(mu_t), torch.ones_like(mu_t))
                kld = Trainer.compute_kld_loss(dist, prior, beta=beta, c=torch.FloatTensor([c]))
                # a z-dependent scalar so d(z)/d(mu, log_std) is exercised too
                w = t(syn.normal_noise((b, z), seed=99))
                (kld.sum() + (zt * w).sum()).backward()
                key = f'b{b}_z{z}_c{c:g}_beta{beta:g}'
                out[f'{key}/kld'] = kld.detach().numpy()
                out[f'{key}/dmu'] = mu_t.grad.numpy().copy()
                out[f'{key}/dls'] = ls_t.grad.numpy().copy()
                out[f'b{b}_z{z}/z'] = zt.detach().numpy()
                out[f'b{b}_z{z}/sigma'] = dist.scale.detach().numpy()
        out[f'b{b}_z{z}/mu'] = mu
        out[f'b{b}_z{z}/log_std'] = ls
        out[f'b{b}_z{z}/eps'] = eps
    save('latent_head.npz', **out)


def gen_recon():
    out = {}
    for b, hw in ((4, 64), (16, 28)):
        rs = np.random.RandomState(b * hw)
        logits = (3.0 * rs.standard_normal((b, 1, hw, hw))).astype(np.float32)
        logits.ravel()[::97] = 0.0                                  # exact zeros: accuracy edge l >= 0
        x = (rs.random_sample((b, 1, hw, hw)) < 0.2).astype(np.float32)
        if hw == 28:
            x = (x * rs.random_sample(x.shape)).astype(np.float32)  # grey levels
        for dist in ('bernoulli', 'gaussian'):
            lt = t(logits).requires_grad_(True)
            loss = ImageVAETrainer.reconstruction_loss(t(x), lt, dist)
            loss.backward()
            out[f'b{b}_{hw}_{dist}/loss'] = loss.item()
            out[f'b{b}_{hw}_{dist}/dlogits_samp'] = lt.grad.numpy().ravel()[::131].copy()
            out[f'b{b}_{hw}_{dist}/dlogits_abs_sum'] = lt.grad.double().abs().sum().item()
        acc = ImageVAETrainer.mean_accuracy(torch.sigmoid(t(logits)), t(x))
        out[f'b{b}_{hw}/acc'] = acc.item()
    # cross entropy on ReLU-ed logits (utils/trainer.py:247-282)
    for b, v in ((5, 35), (32, 35)):
        rs = np.random.RandomState(b + v)
        w = np.maximum(2.0 * rs.standard_normal((b, 24, v)), 0).astype(np.float32)
        tgt = rs.randint(0, v, (b, 24)).astype(np.int64)
        wt = t(w).requires_grad_(True)
        ce = Trainer.mean_crossentropy_loss(wt, t(tgt))
        ce.backward()
        out[f'ce_b{b}/loss'] = ce.item()
        out[f'ce_b{b}/acc'] = Trainer.mean_accuracy(t(w), t(tgt)).item()
        out[f'ce_b{b}/dw_samp'] = wt.grad.numpy().ravel()[::37].copy()
        out[f'ce_b{b}/w'] = w
        out[f'ce_b{b}/tgt'] = tgt
    save('recon.npz', **out)


# ----------------------------------------------------------------------------
# G4 / G5: image VAE full steps
# ----------------------------------------------------------------------------
GAIN = {'dsprites': 1.6, 'mnist': 0.7}      # keeps sigma = exp(log_std) O(1) in both stacks


class DspritesDataset:          # ImageVAETrainer sniffs dataset.__class__.__name__
    pass


class MorphoMnistDataset:
    pass


class MaskedDropout(nn.Module):
    """Dropout with an explicit keep-mask: y = x * mask / (1 - p), p = 0.5."""

    def __init__(self, mask):
        super().__init__()
        self.mask = mask

    def forward(self, x):
        return x * self.mask.to(x.dtype).view_as(x) * 2.0


def image_step(kind, batch, mode, wseed, xseed, eseed, beta, gamma, delta, capacity=0.0,
               dec_dist='bernoulli', mask_seed=None):
    if kind == 'dsprites':
        model, dataset = DspritesVAE(), DspritesDataset()
        x, lab = syn.dsprites_batch(batch, seed=xseed)
        reg_dim, reg_type = (1, 2, 3, 4, 5), ('all',)
    else:
        model, dataset = MnistVAE(), MorphoMnistDataset()
        x, lab = syn.mnist_batch(batch, seed=xseed)
        reg_dim, reg_type = (1, 2, 3, 4, 5, 6), ('all',)
    load_synth_weights(model, wseed, gain=GAIN[kind])
    trainer = ImageVAETrainer(dataset, model, lr=1e-4, reg_type=reg_type, reg_dim=reg_dim,
                              dec_dist=dec_dist, beta=beta, gamma=gamma, capacity=capacity,
                              rand=0, delta=delta)
    if mode == 'train':
        model.train()
    else:
        model.eval()
    if mask_seed is not None:                # MNIST train mode with explicit masks
        shapes = [(batch, 64, 25, 25), (batch, 64, 22, 22), (batch, 8, 19, 19),
                  (batch, 64, 22, 22), (batch, 64, 25, 25)]
        masks = [t(m) for m in syn.dropout_masks(shapes, mask_seed)]
        model.enc_conv[2], model.enc_conv[5], model.enc_conv[8] = (MaskedDropout(m) for m in masks[:3])
        model.dec_conv[2], model.dec_conv[5] = (MaskedDropout(m) for m in masks[3:])
    eps = syn.normal_noise((batch, model.z_dim), seed=eseed)
    before = {k: v.detach().numpy().copy() for k, v in model.named_parameters()}
    inputs, labels = t(x), t(lab)

    # pass 1: split terms (mirrors image_vae_trainer.py:157-180)
    EPS.push(eps)
    outputs, z_dist, prior_dist, z_tilde, _ = model(inputs)
    recons = trainer.reconstruction_loss(inputs, outputs, trainer.dec_dist)
    dist_loss = trainer.compute_kld_loss(z_dist, prior_dist, beta=trainer.beta, c=trainer.capacity)
    reg = sum(trainer.compute_reg_loss(z_tilde, labels[:, d], d, gamma=trainer.gamma,
                                       factor=trainer.delta) for d in reg_dim)
    # pass 2: the trainer's own step (utils/trainer.py:126-147)
    EPS.push(eps)
    trainer.zero_grad()
    loss, acc = trainer.loss_and_acc_for_batch((inputs, labels), epoch_num=0, batch_num=0,
                                               train=(mode == 'train'))
    loss.backward()
    trainer.step()
    assert abs(loss.item() - (recons + dist_loss + reg).item()) <= 1e-5 * abs(loss.item())

    lg = outputs.detach().numpy().ravel()
    out = dict(recons=recons.item(), dist=dist_loss.item(), reg=reg.item(), loss=loss.item(),
               acc=acc.item(), z=z_tilde.detach().numpy(), mu=z_dist.loc.detach().numpy(),
               sigma=z_dist.scale.detach().numpy(), logits_sum=lg.astype(np.float64).sum(),
               logits_abs_sum=np.abs(lg.astype(np.float64)).sum(),
               logits_samp=lg[syn.sample_indices('logits', lg.size, 64)])
    out.update(grad_and_update_summaries(model, before))
    return out


def gen_image_steps():
    save('dsprites_step_b8.npz', **image_step('dsprites', 8, 'train', wseed=1, xseed=1234, eseed=11,
                                              beta=4.0, gamma=10.0, delta=1.0))
    save('dsprites_step_b64.npz', **image_step('dsprites', 64, 'train', wseed=1, xseed=1234, eseed=12,
                                               beta=4.0, gamma=10.0, delta=1.0))
    save('dsprites_step_b8_cap_gauss.npz', **image_step('dsprites', 8, 'train', wseed=2, xseed=77, eseed=13,
                                                        beta=1.0, gamma=10.0, delta=1.0, capacity=25.0,
                                                        dec_dist='gaussian'))
    save('mnist_step_eval.npz', **image_step('mnist', 8, 'eval', wseed=3, xseed=4321, eseed=14,
                                             beta=1.0, gamma=10.0, delta=1.0))
    save('mnist_step_train.npz', **image_step('mnist', 8, 'train', wseed=3, xseed=4321, eseed=15,
                                              beta=1.0, gamma=10.0, delta=1.0, mask_seed=21))


def gen_headline_steps():
    """BASELINE.json configs[1] / [2] / [4] at their own batch sizes; seeds = the `..._at_baseline_batch_*` GPU tests"""
    save('dsprites_step_b512.npz', **head_rows(image_step('dsprites', 512, 'train', wseed=1, xseed=1234, eseed=1,
                                                          beta=4.0, gamma=10.0, delta=1.0)))
    save('mnist_step_train_b1024.npz', **head_rows(image_step('mnist', 1024, 'train', wseed=3, xseed=4321, eseed=15,
                                                              beta=1.0, gamma=10.0, delta=1.0, mask_seed=21)))
    save('measure_step_tf_b256.npz', **head_rows(measure_step(256, 'train', wseed=4, sseed=5, eseed=1, teacher=True)))
    save('measure_step_free_b256.npz', **head_rows(measure_step(256, 'train', wseed=4, sseed=5, eseed=2, teacher=False)))   # eseed 1: a 3e-5 top-1 margin


# ----------------------------------------------------------------------------
# G6 / G7: MeasureVAE
# ----------------------------------------------------------------------------
def folk_dataset():
    ds = object.__new__(FolkNBarDataset)
    ds.index2note_dicts, ds.note2index_dicts = syn.measure_vocabulary()
    ds.n_bars = 1
    ds.class_name = '4by4_FolkNBarDataset_1_'
    return ds


def measure_step(batch, mode, wseed, sseed, eseed, teacher):
    ds = folk_dataset()
    model = MeasureVAE(dataset=ds, note_embedding_dim=10, metadata_embedding_dim=2,
                       num_encoder_layers=2, encoder_hidden_size=128, encoder_dropout_prob=0.0,
                       latent_space_dim=32, num_decoder_layers=2, decoder_hidden_size=128,
                       decoder_dropout_prob=0.0, has_metadata=False, dataset_type='folk')
    shapes = load_synth_weights(model, wseed)
    # keep a positive top-1 margin on the ReLU-ed logits (tie-breaking)
    with torch.no_grad():
        model.decoder.tick_emb_to_note_emb[0].bias.add_(0.5)
        model.decoder.tick_emb_to_note_emb[0].weight.mul_(3.0)
    trainer = MeasureVAETrainer(ds, model, lr=1e-4, reg_type=('all',), reg_dim=(0, 1, 2, 3),
                                beta=0.001, gamma=1.0, capacity=0.0, rand=0, delta=10.0)
    model.train() if mode == 'train' else model.eval()
    model.decoder.teacher_forcing_prob = 1.0 if teacher else 0.0   # forces the coin (decoder.py:427-428)
    score = t(syn.measure_batch(batch, seed=sseed))
    eps = syn.normal_noise((batch, 32), seed=eseed)
    before = {k: v.detach().numpy().copy() for k, v in model.named_parameters()}

    EPS.push(eps)
    weights, samples, z_dist, prior_dist, z_tilde, _ = model(score, score, train=(mode == 'train'))
    recons = trainer.reconstruction_loss(x=score, x_recons=weights)
    dist_loss = trainer.compute_kld_loss(z_dist, prior_dist, trainer.beta)
    attr = trainer.compute_attribute_labels(score)
    reg = sum(trainer.compute_reg_loss(z_tilde, attr[:, d], d, gamma=trainer.gamma, factor=trainer.delta)
              for d in (0, 1, 2, 3))
    top2 = weights.detach().topk(2, dim=2)[0]
    margin = (top2[..., 0] - top2[..., 1]).min().item()
    assert margin > 1e-4, f'top-1 margin too small for a stable golden: {margin}'

    EPS.push(eps)
    trainer.zero_grad()
    loss, acc = trainer.loss_and_acc_for_batch((score, score), epoch_num=0, batch_num=0,
                                               train=(mode == 'train'))
    loss.backward()
    trainer.step()
    assert abs(loss.item() - (recons + dist_loss + reg).item()) <= 1e-5 * abs(loss.item())

    wn = weights.detach().numpy()
    out = dict(recons=recons.item(), dist=dist_loss.item(), reg=reg.item(), loss=loss.item(),
               acc=acc.item(), z=z_tilde.detach().numpy(), mu=z_dist.loc.detach().numpy(),
               sigma=z_dist.scale.detach().numpy(), samples=samples.numpy(),
               attr=attr.numpy(), margin=margin,
               weights_sum=wn.astype(np.float64).sum(),
               weights_samp=wn.ravel()[syn.sample_indices('weights', wn.size, 128)],
               weights_row0=wn[0])
    out.update(grad_and_update_summaries(model, before))
    return out


def gen_measure_steps():
    save('measure_step_tf.npz', **measure_step(16, 'train', wseed=4, sseed=5, eseed=31, teacher=True))
    save('measure_step_free.npz', **measure_step(16, 'train', wseed=4, sseed=5, eseed=32, teacher=False))
    save('measure_step_eval.npz', **measure_step(16, 'eval', wseed=4, sseed=6, eseed=33, teacher=False))


def gen_attributes():
    ds = folk_dataset()
    trainer = types.SimpleNamespace(dataset=ds, attr_dict={'rhy_complexity': 0, 'pitch_range': 1,
                                                           'note_density': 2, 'contour': 3})
    score = syn.measure_batch(64, seed=9)
    score[0, :] = 0                                   # all slur -> zero notes
    score[1, :] = 0
    score[1, 5] = 7                                   # a single note -> range/contour 0
    score[2, :] = 4                                   # all None: counts for density only
    score[3, :] = np.arange(5, 29)                    # 24 rising notes
    attr = MeasureVAETrainer.compute_attribute_labels(trainer, t(score))
    save('attributes.npz', score=score, attr=attr.numpy())


# ----------------------------------------------------------------------------
# G9: evaluation-only inference (N4)
# ----------------------------------------------------------------------------
INFER_BATCHES, INFER_BATCH = 3, 16


def image_inference(kind, wseed, xseed, eseed):
    import imagevae.image_vae_trainer as ivt
    if kind == 'dsprites':
        model, dataset, mk = DspritesVAE(), DspritesDataset(), syn.dsprites_batch
        reg_dim = (1, 2, 3, 4, 5)
    else:
        model, dataset, mk = MnistVAE(), MorphoMnistDataset(), syn.mnist_batch
        reg_dim = (1, 2, 3, 4, 5, 6)
    load_synth_weights(model, wseed, gain=GAIN[kind])
    trainer = ImageVAETrainer(dataset, model, lr=1e-4, reg_type=('all',), reg_dim=reg_dim, beta=1.0,
                              gamma=10.0, capacity=0.0, rand=0, delta=1.0)
    model.eval()
    loader = []
    for i in range(INFER_BATCHES):
        x, lab = mk(INFER_BATCH, seed=xseed + i)
        if kind == 'mnist':                          # (inputs, digit labels, morpho labels): image_vae_trainer.py:126-130
            loader.append((t(x), torch.zeros(INFER_BATCH, dtype=torch.int64), t(lab)))
        else:
            loader.append((t(x), t(lab)))
    eps = [syn.normal_noise((INFER_BATCH, model.z_dim), seed=eseed + i) for i in range(INFER_BATCHES)]
    out = {}
    with torch.no_grad():
        for e in eps:
            EPS.push(e)
        codes, attrs, names = trainer.compute_representations(loader)
        for e in eps:
            EPS.push(e)
        loss, acc = trainer.loss_and_acc_test(loader)
        # the sweeps return make_grid(...) of the decoded probabilities: the (absent) torchvision helper only tiles
        # images for plotting, so it is replaced by a pass-through and the fixture holds the decoded batch itself
        keep = ivt.make_grid
        ivt.make_grid = lambda tensor, **kw: tensor
        try:
            row = trainer.compute_latent_interpolations(codes[3], dim1=2, num_points=5)
            grid = trainer.compute_latent_interpolations2d(codes[5], dim1=1, dim2=4, num_points=3)
        finally:
            ivt.make_grid = keep
    assert not EPS.queue
    g2 = grid.numpy().reshape(grid.shape[0], -1)          # 9 images: every 4th pixel + per-image sums keep the file small
    out.update(codes=codes, attrs=attrs, names=np.array(names), test_loss=float(loss), test_acc=float(acc),
               row=row.numpy(), grid_samp=g2[:, ::4].copy(), grid_sum=g2.astype(np.float64).sum(1))
    return out


def measure_inference(wseed, sseed, eseed):
    ds = folk_dataset()
    ds.beat_subdivisions, ds.seq_size_in_beats = 6, 4
    ds.tensor_to_m21score = lambda tensor_score: None          # music21 rendering: not on the path
    ds.concatenate_scores = lambda scores: None
    model = MeasureVAE(dataset=ds, note_embedding_dim=10, metadata_embedding_dim=2,
                       num_encoder_layers=2, encoder_hidden_size=128, encoder_dropout_prob=0.5,
                       latent_space_dim=32, num_decoder_layers=2, decoder_hidden_size=128,
                       decoder_dropout_prob=0.5, has_metadata=False, dataset_type='folk')
    load_synth_weights(model, wseed)
    with torch.no_grad():
        model.decoder.tick_emb_to_note_emb[0].bias.add_(0.5)
        model.decoder.tick_emb_to_note_emb[0].weight.mul_(3.0)
    trainer = MeasureVAETrainer(ds, model, lr=1e-4, reg_type=('all',), reg_dim=(0, 1, 2, 3),
                                beta=0.001, gamma=1.0, capacity=0.0, rand=0, delta=10.0)
    model.eval()
    loader = []
    for i in range(INFER_BATCHES):
        sc = t(syn.measure_batch(INFER_BATCH, seed=sseed + i))
        loader.append((sc, sc))
    eps = [syn.normal_noise((INFER_BATCH, 32), seed=eseed + i) for i in range(INFER_BATCHES)]
    with torch.no_grad():
        for e in eps:
            EPS.push(e)
        codes, attrs, names = trainer.compute_representations(loader)
        for e in eps:
            EPS.push(e)
        loss, acc = trainer.loss_and_acc_test(loader)
        _, notes = trainer.decode_latent_codes(t(codes[:8]))
        _, sweep = trainer.compute_latent_interpolations(codes[3], None, dim1=3, num_points=5)
    assert not EPS.queue
    return dict(codes=codes, attrs=attrs, names=np.array(names), test_loss=float(loss), test_acc=float(acc),
                notes=notes.numpy(), sweep=sweep.numpy())


def gen_inference():
    save('inference_dsprites.npz', **image_inference('dsprites', wseed=1, xseed=600, eseed=40))
    save('inference_mnist.npz', **image_inference('mnist', wseed=3, xseed=700, eseed=50))
    save('inference_measure.npz', **measure_inference(wseed=4, sseed=800, eseed=60))


if __name__ == '__main__':
    gen_reg_loss()
    gen_latent_head()
    gen_recon()
    gen_image_steps()
    gen_measure_steps()
    gen_headline_steps()
    gen_attributes()
    gen_inference()
