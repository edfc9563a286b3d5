"""Device-resident datasets, the uint8 gather kernel, evaluation inference and the training CLIs on tiny synthetic files
written in the reference's on-disk formats."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import arvae_amd  # noqa: E402,F401
from arvae_amd import synthetic as syn  # noqa: E402
from arvae_amd.data import formats  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dev():
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    return torch.device('cuda:0')


def write_dsprites(folder, n, seed=0):
    x, lab = syn.dsprites_batch(n, seed=seed)
    imgs = (x[:, 0] > 0.5).astype(np.uint8)
    np.savez(os.path.join(folder, 'dsprites_ndarray_co1sh3sc6or40x32y32_64x64.npz'), imgs=imgs,
             latents_values=lab.astype(np.float64))
    return imgs, lab.astype(np.float32)


def write_mnist(folder, n_train, n_test, seed=0):
    plain = os.path.join(folder, 'mnist_data', 'plain')
    os.makedirs(plain)
    out = {}
    for split, n in (('train', n_train), ('t10k', n_test)):
        x, lab = syn.mnist_batch(n, seed=seed + len(split))
        imgs = np.clip(np.rint(x[:, 0] * 255.0), 0, 255).astype(np.uint8)
        digits = (np.arange(n) % 10).astype(np.uint8)
        formats.save_idx(imgs, os.path.join(plain, split + '-images-idx3-ubyte.gz'))
        formats.save_idx(digits, os.path.join(plain, split + '-labels-idx1-ubyte.gz'))
        np.savetxt(os.path.join(plain, split + '-morpho.csv'), lab, delimiter=',', header='index,area,length,thickness,slant,width,height', comments='')
        out[split] = (imgs, digits, lab.astype(np.float32))
    return out


def write_folk(folder, n, seed=0):
    raw = os.path.join(folder, 'folk_raw_data')
    os.makedirs(raw)
    score = torch.from_numpy(syn.measure_batch(n, seed=seed)).int()
    torch.save(torch.utils.data.TensorDataset(score, score), os.path.join(raw, '4by4_FolkNBarDataset_1_train'))
    i2n, n2i = syn.measure_vocabulary()
    with open(os.path.join(raw, 'index_dicts.txt'), 'w') as f:
        f.write(repr(i2n) + '\n' + repr(n2i) + '\n')
    return score.numpy()


def test_gather_rows_u8_is_exact(dev):
    from arvae_amd import ops
    rs = np.random.RandomState(3)
    src = rs.randint(0, 256, (97, 1, 28, 28)).astype(np.uint8)
    idx = rs.randint(0, 97, 301)
    got = ops.gather_rows_u8(torch.from_numpy(src).to(dev), torch.from_numpy(idx).to(dev), 1.0 / 255.0).cpu().numpy()
    want = src[idx].astype(np.float32) * np.float32(1.0 / 255.0)
    assert got.shape == (301, 1, 28, 28) and np.array_equal(got, want)
    with pytest.raises(TypeError):
        ops.gather_rows_u8(torch.zeros(4, 8, device=dev), torch.zeros(2, dtype=torch.int64, device=dev))


def test_dsprites_loaders_split_and_content(dev, tmp_path):
    from arvae_amd.data import DspritesDataset
    imgs, lab = write_dsprites(str(tmp_path), 200)
    ds = DspritesDataset(path=str(tmp_path / 'dsprites_ndarray_co1sh3sc6or40x32y32_64x64.npz'), device=dev)
    tr, va, ev = ds.data_loaders(batch_size=32, split=(0.70, 0.20))
    cut2 = int((0.70 + 0.20) * 200)                                  # 179, not 180: the reference's float arithmetic
    assert (len(tr), len(va), len(ev)) == (5, 2, 1)                  # 140 / 39 / 21 rows
    xs, ls = zip(*[(x.cpu().numpy(), l.cpu().numpy()) for x, l in ev])
    assert np.array_equal(np.concatenate(xs)[:, 0], imgs[cut2:].astype(np.float32)) and np.array_equal(np.concatenate(ls), lab[cut2:])
    seen = np.concatenate([l.cpu().numpy() for _, l in tr])
    assert seen.shape == (140, 6)                                    # a permutation of the first 140 rows
    assert np.array_equal(np.sort(seen.sum(1)), np.sort(lab[:140].sum(1)))
    x0, l0 = next(iter(tr))
    assert x0.dtype == torch.float32 and x0.shape == (32, 1, 64, 64) and set(np.unique(x0.cpu().numpy())) <= {0.0, 1.0}


def test_mnist_and_folk_loaders(dev, tmp_path):
    from arvae_amd.data import FolkNBarDataset, MorphoMnistDataset
    ref = write_mnist(str(tmp_path), 50, 30)
    ds = MorphoMnistDataset(root_dir=str(tmp_path / 'mnist_data' / 'plain'), device=dev)
    tr, va, ev = ds.data_loaders(batch_size=16)
    assert (len(tr), len(va), len(ev)) == (4, 2, 2)
    x, d, m = next(iter(va))
    imgs, digits, lab = ref['t10k']
    assert np.array_equal(x.cpu().numpy()[:, 0], imgs[:16].astype(np.float32) * np.float32(1 / 255.0))
    assert np.array_equal(d.cpu().numpy(), digits[:16].astype(np.int64)) and np.allclose(m.cpu().numpy(), lab[:16])
    score = write_folk(str(tmp_path), 100)
    fd = FolkNBarDataset(dataset_dir=str(tmp_path / 'folk_raw_data'), device=dev)
    assert repr(fd) == '4by4_FolkNBarDataset_1_' and fd.index2note_dicts[0] == '__'
    tr, va, ev = fd.data_loaders(batch_size=8, split=(0.70, 0.20))
    assert (len(tr), len(va), len(ev)) == (8, 2, 1)                  # drop_last: 70 -> 8, 20 -> 2, 10 -> 1
    s, meta = next(iter(va))
    assert np.array_equal(s.cpu().numpy(), score[70:78]) and torch.equal(s, meta)


def _run_cli(script, args, env_dir, **extra_env):
    env = dict(os.environ, ARVAE_DATA_DIR=str(env_dir), ARVAE_MODEL_DIR=str(env_dir / 'models'), **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, script)] + args, capture_output=True, text=True, timeout=600, env=env,
                       cwd=str(env_dir))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    start = r.stdout.index('{\n')
    return json.JSONDecoder().raw_decode(r.stdout[start:])[0], r.stdout      # (RCCL may print a banner after the summary)


def test_train_image_cli_dsprites_end_to_end(dev, tmp_path):
    write_dsprites(str(tmp_path), 400)
    summary, out = _run_cli('train_image_vae.py', ['-d', 'dsprites', '--num_epochs', '2', '--batch_size', '64', '--rand', '3',
                                                   '-r', 'all'], tmp_path)
    assert 'Num Train Batches:  5' in out and out.count('saved') >= 2
    assert summary['attributes'] == ['shape', 'scale', 'orientation', 'posx', 'posy'] and summary['num_codes'] == 400 - int((0.80 + 0.15) * 400)
    assert np.isfinite(summary['test_loss']) and 0.0 <= summary['test_acc'] <= 1.0
    name = summary['model']
    assert os.path.exists(tmp_path / 'models' / name / (name + '.pt'))
    # --test reloads the checkpoint without training
    again, _ = _run_cli('train_image_vae.py', ['-d', 'dsprites', '--test', '--rand', '3', '-r', 'all'], tmp_path)
    assert again['model'] == name and again['test_loss'] == pytest.approx(summary['test_loss'], rel=5e-2)    # eps is redrawn


def test_train_image_cli_mnist_single_attribute(dev, tmp_path):
    write_mnist(str(tmp_path), 96, 64)
    summary, _ = _run_cli('train_image_vae.py', ['--num_epochs', '1', '--batch_size', '32', '--rand', '0', '-r', 'slant'], tmp_path)
    assert summary['num_codes'] == 64 and len(summary['latent_mean_abs']) == 16 and np.isfinite(summary['test_loss'])


def test_train_measure_cli_end_to_end(dev, tmp_path):
    write_folk(str(tmp_path), 400)                   # evaluation split: 20 measures -> one batch of 16 (drop_last)
    summary, _ = _run_cli('train_measure_vae.py', ['--num_epochs', '1', '--batch_size', '16', '--rand', '1', '-r', 'all'], tmp_path)
    assert summary['attributes'] == ['rhy_complexity', 'pitch_range', 'note_density', 'contour']
    assert summary['num_codes'] == 16 and np.isfinite(summary['test_loss'])


def test_decoder_sweeps(dev):
    from arvae_amd.image_vae import DspritesVAE
    from arvae_amd.image_vae_trainer import ImageVAETrainer

    class DspritesDataset:
        pass
    model = DspritesVAE()
    trainer = ImageVAETrainer(DspritesDataset(), model, reg_type=('all',), reg_dim=(1, 2, 3, 4, 5))
    trainer.cuda()
    z = np.zeros(10, np.float32)
    row = trainer.compute_latent_interpolations(z, dim1=2, num_points=7)
    grid = trainer.compute_latent_interpolations2d(z, dim1=1, dim2=4, num_points=5)
    assert row.shape == (7, 1, 64, 64) and grid.shape == (25, 1, 64, 64)
    assert float(row.min()) >= 0.0 and float(row.max()) <= 1.0
    # the middle of an odd sweep is the unperturbed code
    mid = torch.sigmoid(model.decode(torch.zeros(1, 10, device=dev)))
    assert torch.allclose(row[3], mid[0], atol=1e-6)


def test_train_cli_data_parallel_code_path(dev, tmp_path):
    """ARVAE_FORCE_DP=1: the CLI joins an RCCL group of one rank, shards its loaders (rank 0 of 1) and all-reduces the
    gradient arena: the run must train, save and evaluate exactly as the single-process one does."""
    write_dsprites(str(tmp_path), 400)
    args = ['-d', 'dsprites', '--num_epochs', '1', '--batch_size', '64', '--rand', '3', '-r', 'all']
    plain, _ = _run_cli('train_image_vae.py', args, tmp_path)
    name = plain['model']
    ckpt = tmp_path / 'models' / name / (name + '.pt')
    w_plain = {k: v.clone() for k, v in torch.load(ckpt, map_location='cpu').items()}
    forced, out = _run_cli('train_image_vae.py', args, tmp_path, ARVAE_FORCE_DP='1')
    assert 'Num Train Batches:  5' in out and forced['num_codes'] == plain['num_codes'] and forced['model'] == name
    # same seed, rank 0 of 1: the device's Philox streams (eps of every training / evaluation step), the shuffling order and
    # the batches are the same in both runs, so the five Adam steps and the evaluation must agree to rounding -- the
    # data-parallel run takes the row-block regulariser, the all-gathers and the all-reduce (identities on one rank)
    w_forced = torch.load(ckpt, map_location='cpu')
    for k, v in w_plain.items():
        d = (w_forced[k].double() - v.double()).norm()
        assert d <= 1e-5 * v.double().norm() + 1e-7, k
    assert forced['test_loss'] == pytest.approx(plain['test_loss'], rel=1e-4)
    assert forced['test_acc'] == pytest.approx(plain['test_acc'], abs=1e-4)


def test_train_cli_two_ranks_on_this_box(dev, tmp_path):
    """train_image_vae.py as TWO ranks (RANK / WORLD_SIZE / LOCAL_RANK as torch.distributed.run sets them; both ranks on cuda:0
    when the box has one GPU, collectives staged through the host: ARVAE_DP_TRANSPORT=staged): every rank walks its half of
    rank 0's shuffled global batches, the epoch statistics are the ranks' means, rank 0 saves and evaluates alone."""
    import socket
    write_dsprites(str(tmp_path), 400)
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    args = ['-d', 'dsprites', '--num_epochs', '1', '--batch_size', '32', '--rand', '3', '-r', 'all']
    n_gpu = torch.cuda.device_count()
    procs = []
    for r in range(2):
        env = dict(os.environ, ARVAE_DATA_DIR=str(tmp_path), ARVAE_MODEL_DIR=str(tmp_path / 'models'), RANK=str(r), WORLD_SIZE='2',
                   LOCAL_RANK=str(r % n_gpu), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   ARVAE_DP_TRANSPORT='staged' if n_gpu < 2 else 'library')
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'train_image_vae.py')] + args, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True, env=env, cwd=str(tmp_path)))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    # 280 training rows, global batch 2 x 32: four full global batches + a tail of 24 -> 12 rows per rank
    assert 'Num Train Batches:  5' in outs[0] and 'Num Train Batches' not in outs[1]
    summary = json.JSONDecoder().raw_decode(outs[0][outs[0].index('{\n'):])[0]
    assert summary['num_codes'] == 400 - int((0.80 + 0.15) * 400) and np.isfinite(summary['test_loss'])
    assert '{\n' not in outs[1]                                      # the other rank neither evaluates nor prints a summary


def test_measure_inference_for_evaluation(dev, tmp_path):
    """N4, measure side (measure_vae_trainer.py:281-308,367-397): decoder-only passes, the reconstruction-only test loss and
    the representation record the host-side metric suite reads."""
    from arvae_amd.data import FolkNBarDataset
    from arvae_amd.measure_vae import MeasureVAE
    from arvae_amd.measure_vae_trainer import MeasureVAETrainer
    write_folk(str(tmp_path), 300)
    ds = FolkNBarDataset(dataset_dir=str(tmp_path / 'folk_raw_data'), device=dev)
    torch.manual_seed(0)
    model = MeasureVAE(ds, 10, 2, 2, 64, 0.5, 16, 2, 64, 0.5, False, 'folk')
    trainer = MeasureVAETrainer(ds, model, reg_type=('all',), reg_dim=(0, 1, 2, 3))
    trainer.cuda()
    model.eval()
    z = torch.randn(5, 16)
    score, notes = trainer.decode_latent_codes(z)
    assert score is None and notes.shape == (5, 1, 24) and notes.dtype == torch.int64
    assert int(notes.min()) >= 0 and int(notes.max()) < len(ds.note2index_dicts)
    _, again = trainer.decode_latent_codes(z)
    assert torch.equal(notes, again)                                 # eval mode, argmax decoding: deterministic
    _, sweep = trainer.compute_latent_interpolations(z[2].numpy(), dim1=3, num_points=5)
    assert sweep.shape == (5, 24)
    zz = z[2:3].repeat(5, 1)
    zz[:, 3] = torch.linspace(-4.0, 4.0, 5)
    assert torch.equal(sweep, trainer.decode_latent_codes(zz)[1].squeeze(1))
    with pytest.raises(AssertionError):
        trainer.compute_latent_interpolations(z[0].numpy(), num_points=4)
    # reconstruction-only test loss = mean over batches of the CE the training step reports as its first term
    _, _, ev = ds.data_loaders(batch_size=8, split=(0.70, 0.20))
    try:
        model.encoder.static_eps = torch.zeros(8, 16, device=dev)    # z = mu in both passes: the two numbers are comparable
        loss, acc = trainer.loss_and_acc_test(ev)
        want = []
        with torch.no_grad():
            for batch in ev:
                s, m = trainer.process_batch_data(batch)
                trainer.loss_and_acc_for_batch((s, m), 0, 0, train=False)
                want.append(float(trainer.last_terms['recons']))
    finally:
        type(model.encoder).static_eps = None
        model.encoder.static_eps = None
    assert loss == pytest.approx(np.mean(want), rel=1e-5) and 0.0 <= acc <= 1.0
    codes, attrs, names = trainer.save_representations(str(tmp_path / 'rep.json'), data_loader=ev)
    rec = json.load(open(tmp_path / 'rep.json'))
    assert rec['attr_list'] == ['rhy_complexity', 'pitch_range', 'note_density', 'contour']
    assert np.asarray(rec['latent_codes']).shape == codes.shape == (len(want) * 8, 16) and np.asarray(rec['attributes']).shape == (len(want) * 8, 4)
    model.filepath = str(tmp_path / 'models' / 'm' / 'm.pt')
    metrics = trainer.compute_eval_metrics(batch_size=8)
    assert os.path.exists(tmp_path / 'models' / 'm' / 'results_dict.json') and 'test_loss' in metrics
    assert trainer.compute_eval_metrics(batch_size=8) == json.load(open(tmp_path / 'models' / 'm' / 'results_dict.json'))
