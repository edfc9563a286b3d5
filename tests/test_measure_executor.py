"""The whole-model MeasureVAE executor (arvae_measure_vae_forward / _backward, csrc/plan_measure.hip) against the per-layer
autograd path that issues the same launches one by one (ar-vae_amd/measure_vae.py): loss terms, notes fed back and every
parameter gradient, teacher-forced and free-running, with and without dropout, explicit and library-drawn noise.
The per-layer path itself is held against the oracle and the reference's goldens in tests/test_hip_parity.py -- and so is the
executor, which those tests reach through MeasureVAETrainer.loss_and_acc_for_batch.  Needs a real MI355X."""
import numpy as np
import pytest
import torch

from arvae_amd import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.fail('gpu-marked test needs a GPU (the HIP path has no CPU fallback)')
    return torch.device('cuda:0')


class _FolkDataset:
    class_name = '4by4_FolkNBarDataset_1_'
    n_bars = 1

    def __init__(self):
        self.index2note_dicts, self.note2index_dicts = syn.measure_vocabulary()

    def __repr__(self):
        return self.class_name


def _trainer(hid, zdim, dropout, reg=True, seed=11):
    from arvae_amd.measure_vae import MeasureVAE
    from arvae_amd.measure_vae_trainer import MeasureVAETrainer
    torch.manual_seed(seed)
    ds = _FolkDataset()
    model = MeasureVAE(ds, 10, 2, 2, hid, dropout, zdim, 2, hid, dropout, False, 'folk')
    with torch.no_grad():                                     # spread the logits so that the argmax varies
        model.decoder.tick_emb_to_note_emb[0].weight.mul_(3.0)
        model.decoder.tick_emb_to_note_emb[0].bias.add_(0.4)
    kw = dict(reg_type=('all',), reg_dim=(0, 1, 2, 3)) if reg else {}
    trainer = MeasureVAETrainer(ds, model, lr=1e-4, beta=0.001, gamma=1.0, capacity=0.0, rand=0, delta=10.0, **kw)
    trainer.cuda()
    return trainer, model


def _masks(b, hid, seed):
    gen = torch.Generator().manual_seed(seed)
    return [(torch.rand(24, b, 2 * hid, generator=gen) >= 0.5).to(torch.uint8),
            (torch.rand(4, b, hid, generator=gen) >= 0.5).to(torch.uint8),
            (torch.rand(24, b, hid, generator=gen) >= 0.5).to(torch.uint8)]


def _one_step(trainer, model, score, fused, teacher, eps=None, masks=None, train=True):
    trainer.use_fused_step = fused
    model.train(train)
    model.decoder.teacher_forcing_prob = 2.0 if teacher else -1.0
    if eps is not None:
        model.push_noise(eps)
    if masks is not None and train:
        model.encoder.push_dropout_mask(masks[0].to(score.device))
        model.decoder.push_dropout_masks(masks[1].to(score.device), masks[2].to(score.device))
    trainer.zero_grad()
    assert (trainer.fused_executor(score) is not None) == fused
    if train:
        loss, acc = trainer.loss_and_acc_for_batch((score, score), 0, 0, True)
        trainer.backward(loss)
    else:
        with torch.no_grad():
            loss, acc = trainer.loss_and_acc_for_batch((score, score), 0, 0, False)
    torch.cuda.synchronize()
    terms = {k: (None if v is None else float(v)) for k, v in trainer.last_terms.items()}
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()} if train else {}
    return float(loss.detach()), float(acc), terms, grads


def _same(a, b, what, rel=2e-5):
    la, aa, ta, ga = a
    lb, ab, tb, gb = b
    assert abs(la - lb) <= 1e-5 * abs(lb) + 1e-7, (what, la, lb)
    assert abs(aa - ab) <= 1e-6, (what, aa, ab)
    for k in tb:
        if tb[k] is None:
            assert ta[k] is None
        else:
            assert abs(ta[k] - tb[k]) <= 1e-5 * abs(tb[k]) + 1e-8, (what, k, ta[k], tb[k])
    assert ga.keys() == gb.keys()
    for k, want in gb.items():
        got = ga[k]
        assert torch.isfinite(got).all(), (what, k)
        err, ref = float((got - want).norm()), float(want.norm())
        assert err <= rel * ref + 1e-9, (what, k, err, ref)


@pytest.mark.parametrize('b,hid,zdim,dropout', [(256, 128, 32, 0.5), (21, 128, 32, 0.5), (5, 64, 16, 0.0), (37, 64, 32, 0.5),
                                                (96, 128, 32, 0.0), (100, 64, 16, 0.5),
                                                # (the recurrences at 8 and 16 rows per workgroup, with dropout between their layers)
                                                (700, 64, 32, 0.5), (1400, 64, 16, 0.5)])
@pytest.mark.parametrize('teacher', [True, False], ids=['teacher_forced', 'free_running'])
def test_executor_step_matches_the_per_layer_path(dev, b, hid, zdim, dropout, teacher):
    """same explicit noise and keep-masks: the two paths' loss terms, accuracy and all gradients agree (the launches are the
    same; only a few summation orders differ: the constant beat input's gradients, the cross entropy's row order)"""
    trainer, model = _trainer(hid, zdim, dropout)
    score = torch.from_numpy(syn.measure_batch(b, seed=18 + b)).to(dev)
    eps = torch.from_numpy(syn.normal_noise((b, zdim), seed=19))
    masks = _masks(b, hid, 3) if dropout > 0 else None
    ref = _one_step(trainer, model, score, False, teacher, eps, masks)
    got = _one_step(trainer, model, score, True, teacher, eps, masks)
    _same(got, ref, (b, hid, teacher))
    assert ref[2]['reg'] is not None and ref[2]['reg'] > 0


def test_executor_without_regulariser_and_in_eval_mode(dev):
    trainer, model = _trainer(128, 32, 0.5, reg=False)
    b = 64
    score = torch.from_numpy(syn.measure_batch(b, seed=5)).to(dev)
    eps = torch.from_numpy(syn.normal_noise((b, 32), seed=6))
    masks = _masks(b, 128, 7)
    _same(_one_step(trainer, model, score, True, True, eps, masks), _one_step(trainer, model, score, False, True, eps, masks), 'no reg')
    # evaluation: no dropout, free-running (measure_vae_trainer.py:367-397 calls the model with train=False)
    _same(_one_step(trainer, model, score, True, False, eps, None, train=False),
          _one_step(trainer, model, score, False, False, eps, None, train=False), 'eval')


def test_executor_draws_the_per_layer_paths_noise(dev):
    """no explicit noise: the executor draws the encoder keep-mask, eps and the decoder keep-masks from the same Philox streams,
    in the same order, as the per-layer path's ops.keep_mask / ops.normal_noise launches -- the steps agree draw for draw"""
    from arvae_amd import ops
    b = 48
    score = torch.from_numpy(syn.measure_batch(b, seed=9)).to(dev)
    res = {}
    for fused in (False, True):
        trainer, model = _trainer(128, 32, 0.5)
        ops.rng_reseed(123)
        steps = []
        for teacher in (True, False, True):
            steps.append(_one_step(trainer, model, score, fused, teacher))
        res[fused] = steps
    for i, (got, ref) in enumerate(zip(res[True], res[False])):
        _same(got, ref, ('rng', i))
    assert res[True][0][0] != res[True][2][0]                 # (a fresh draw per step)


def test_executor_pass_in_flight_keeps_its_workspace(dev):
    """two forwards before the first backward (two losses summed): each pass reads the activations its own forward wrote"""
    trainer, model = _trainer(128, 32, 0.0)
    b = 32
    s1 = torch.from_numpy(syn.measure_batch(b, seed=1)).to(dev)
    s2 = torch.from_numpy(syn.measure_batch(b, seed=2)).to(dev)
    e1 = torch.from_numpy(syn.normal_noise((b, 32), seed=3))
    e2 = torch.from_numpy(syn.normal_noise((b, 32), seed=4))
    model.train()
    model.decoder.teacher_forcing_prob = 2.0
    out = {}
    for fused in (False, True):
        trainer.use_fused_step = fused
        trainer.zero_grad()
        model.push_noise(e1)
        l1, _ = trainer.loss_and_acc_for_batch((s1, s1), 0, 0, True)
        model.push_noise(e2)
        l2, _ = trainer.loss_and_acc_for_batch((s2, s2), 0, 1, True)
        trainer.backward(l1 + l2)
        torch.cuda.synchronize()
        out[fused] = (float(l1.detach()), float(l2.detach()), {k: p.grad.detach().clone() for k, p in model.named_parameters()})
    assert abs(out[True][0] - out[False][0]) <= 1e-5 * abs(out[False][0]) and abs(out[True][1] - out[False][1]) <= 1e-5 * abs(out[False][1])
    for k, want in out[False][2].items():
        assert float((out[True][2][k] - want).norm()) <= 2e-5 * float(want.norm()) + 1e-9, k


def test_executor_rejects_bad_arguments(dev):
    import ctypes
    from arvae_amd import _lib
    from arvae_amd.fused_measure import FusedMeasureVAE
    trainer, model = _trainer(128, 32, 0.5)
    lib = _lib.load()
    assert FusedMeasureVAE.supports(model, trainer.optimizer, (0, 1, 2, 3)) is None
    assert FusedMeasureVAE.supports(model, trainer.optimizer, (0, 7)) is not None
    fused = FusedMeasureVAE(model, trainer.optimizer, (0, 1, 2, 3), 0.001, 1.0, 10.0)
    d = fused.descriptor()
    assert lib.arvae_measure_vae_ws_floats(ctypes.byref(d), 256) > 0
    assert lib.arvae_measure_vae_ws_floats(ctypes.byref(d), 0) == -1
    assert lib.arvae_measure_vae_forward(ctypes.byref(d), 8, None, None, None, None, None, 1, None, None, None, None, None, None, None,
                                         None, None, 0, None) == -1
    assert b'null pointer' in lib.arvae_last_error_string()
    assert not fused.fits(4096) and fused.fits(1024)
    # a model whose paired layers are NOT back to back in the arena (torch's parameters() order): per-layer path
    from arvae_amd.optim import FlatAdam
    other = FlatAdam(model.parameters(), lr=1e-4)
    assert 'adjacent' in FusedMeasureVAE.supports(model, other, ())


def test_trainer_falls_back_to_the_per_layer_path(dev):
    """shapes the executor is not built for (a 35-note vocabulary at decoder hidden size 32: no one-launch free-running decoder) and
    debug checks keep the per-layer path; a single measure (B = 1) and the largest batch whose positions fit the segment sums run
    through the executor"""
    from arvae_amd.measure_vae import MeasureVAE
    from arvae_amd.measure_vae_trainer import MeasureVAETrainer
    torch.manual_seed(2)
    ds = _FolkDataset()
    model = MeasureVAE(ds, 10, 2, 2, 64, 0.0, 16, 2, 32, 0.0, False, 'folk')
    trainer = MeasureVAETrainer(ds, model, lr=1e-4, reg_type=('all',), reg_dim=(0, 1), beta=0.001, gamma=1.0, capacity=0.0, rand=0, delta=10.0)
    trainer.cuda()
    model.train()
    score = torch.from_numpy(syn.measure_batch(8, seed=3)).to(dev)
    assert trainer.fused_executor(score) is None
    trainer.zero_grad()
    loss, _ = trainer.loss_and_acc_for_batch((score, score), 0, 0, True)
    trainer.backward(loss)
    assert torch.isfinite(loss) and trainer.optimizer.grad_arena.abs().sum() > 0
    # B = 1 and B = 1024 through the executor, against the per-layer path
    trainer2, model2 = _trainer(128, 32, 0.0)
    for b in (1, 1024):
        sc = torch.from_numpy(syn.measure_batch(b, seed=40 + b)).to(dev)
        eps = torch.from_numpy(syn.normal_noise((b, 32), seed=41))
        _same(_one_step(trainer2, model2, sc, True, True, eps), _one_step(trainer2, model2, sc, False, True, eps), ('edge', b))
    # 24 x 4096 positions do not fit the segment sums' LDS (V = 35: fits() turns false above B = 2389): the executor declines and
    # the per-layer path must take the whole step -- both lookups fall back to embedding + GEMM (the encoder's did not)
    big = torch.from_numpy(syn.measure_batch(4096, seed=1)).to(dev)
    trainer2.use_fused_step = True
    assert trainer2._fused_binding() is not None and not trainer2._fused_binding().fits(4096)
    assert trainer2.fused_executor(big) is None
    model2.train()
    model2.decoder.teacher_forcing_prob = 2.0
    trainer2.zero_grad()
    loss, _ = trainer2.loss_and_acc_for_batch((big, big), 0, 0, True)
    trainer2.backward(loss)
    torch.cuda.synchronize()
    assert torch.isfinite(loss) and torch.isfinite(trainer2.optimizer.grad_arena).all() and trainer2.optimizer.grad_arena.abs().sum() > 0


def test_round5_recurrences_match_the_round4_kernels(dev):
    """The executor's step under the default recurrences -- 4 batch rows per workgroup, the backward pass on per-row-scaled two-term
    fp16, the dropout between stacked layers inside the launches -- against the diagnostic library's round-4 forms of all three
    (ARVAE_GRU_WIDE: 16 rows; ARVAE_GRU_BF16_BWD: three-term bf16; ARVAE_GRU_MASK_APART: mask launches): the same loss terms, and
    gradients that agree to the arithmetic's 1e-5."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, json; sys.path.insert(0, %r)\n"
        "import torch\n"
        "from arvae_amd import synthetic as syn\n"
        "from tests.test_measure_executor import _trainer, _masks, _one_step\n"
        "dev = torch.device('cuda:0')\n"
        "out = {}\n"
        "for teacher in (True, False):\n"
        "    trainer, model = _trainer(128, 32, 0.5)\n"
        "    score = torch.from_numpy(syn.measure_batch(256, seed=77)).to(dev)\n"
        "    eps = torch.from_numpy(syn.normal_noise((256, 32), seed=78))\n"
        "    loss, acc, terms, grads = _one_step(trainer, model, score, True, teacher, eps, _masks(256, 128, 5))\n"
        "    out[str(teacher)] = {'loss': loss, 'acc': acc, 'gn': {k: float(v.norm()) for k, v in grads.items()},\n"
        "                         'g0': {k: float(v.flatten()[0]) for k, v in grads.items()}}\n"
        "print(json.dumps(out))\n" % root)
    diag = os.path.join(root, 'ar-vae_amd', 'libarvae_hip_diag.so')
    old = {'ARVAE_LIB': diag, 'ARVAE_GRU_WIDE': '1', 'ARVAE_GRU_BF16_BWD': '1', 'ARVAE_GRU_MASK_APART': '1'}
    res = {}
    for name, env in (('round5', {}), ('round4', old)):
        r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr[-2000:]
        res[name] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    for teacher in ('True', 'False'):
        a, b = res['round5'][teacher], res['round4'][teacher]
        assert abs(a['loss'] - b['loss']) <= 1e-5 * abs(b['loss']), (teacher, a['loss'], b['loss'])
        assert abs(a['acc'] - b['acc']) <= 1e-6
        for k, v in b['gn'].items():
            assert abs(a['gn'][k] - v) <= 2e-5 * v + 1e-10, (teacher, k, a['gn'][k], v)
            assert abs(a['g0'][k] - b['g0'][k]) <= 1e-4 * abs(b['g0'][k]) + 2e-5 * v / max(1.0, v) + 1e-9, (teacher, k, a['g0'][k], b['g0'][k])
