"""Dataset file formats and the CLI flag surface (N2): host-side logic, runs anywhere."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import arvae_amd  # noqa: E402,F401
from arvae_amd.data import formats  # noqa: E402


def test_idx_round_trip_and_bad_magic(tmp_path):
    rs = np.random.RandomState(0)
    for shape, name in (((7, 28, 28), 'a-images-idx3-ubyte.gz'), ((7,), 'a-labels-idx1-ubyte.gz'), ((3, 5), 'plain-idx')):
        arr = rs.randint(0, 256, shape).astype(np.uint8)
        formats.save_idx(arr, tmp_path / name)
        back = formats.load_idx(tmp_path / name)
        assert back.dtype == np.uint8 and back.shape == shape and (back == arr).all()
    (tmp_path / 'bad').write_bytes(b'\x01\x02\x08\x01' + b'\x00' * 8)
    with pytest.raises(ValueError):
        formats.load_idx(tmp_path / 'bad')
    formats.save_idx(np.zeros((4, 4), np.uint8), tmp_path / 'short')
    data = (tmp_path / 'short').read_bytes()
    (tmp_path / 'short').write_bytes(data[:-3])
    with pytest.raises(ValueError):
        formats.load_idx(tmp_path / 'short')


def test_idx_big_endian_int32(tmp_path):
    import struct
    vals = np.array([[1, -2, 300000], [4, 5, 6]], dtype=np.int32)
    with open(tmp_path / 'i32', 'wb') as f:
        f.write(bytes([0, 0, 0x0C, 2]) + struct.pack('>II', 2, 3) + vals.astype('>i4').tobytes())
    assert (formats.load_idx(tmp_path / 'i32') == vals).all()


def test_dsprites_npz(tmp_path):
    rs = np.random.RandomState(1)
    imgs = (rs.rand(10, 64, 64) > 0.5).astype(np.uint8)
    lat = rs.rand(10, 6)
    np.savez(tmp_path / 'd.npz', imgs=imgs, latents_values=lat, latents_classes=np.zeros((10, 6), np.int64))
    a, b = formats.load_dsprites_npz(tmp_path / 'd.npz')
    assert a.dtype == np.uint8 and (a == imgs).all() and b.dtype == np.float32 and np.allclose(b, lat.astype(np.float32))
    np.savez(tmp_path / 'e.npz', other=imgs)
    with pytest.raises(ValueError):
        formats.load_dsprites_npz(tmp_path / 'e.npz')


def test_morpho_csv_and_dicts_and_measures(tmp_path):
    (tmp_path / 'm.csv').write_text('index,area,length,thickness,slant,width,height\n0,1.5,2,3,4,5,6\n1,7,8,9,10,11,12.25\n')
    t = formats.load_morpho_csv(tmp_path / 'm.csv')
    assert t.shape == (2, 7) and t.dtype == np.float32 and t[1, 6] == 12.25
    (tmp_path / 'index_dicts.txt').write_text("{0: '__', 1: 'C4'}\n{'__': 0, 'C4': 1}\n")
    i2n, n2i = formats.load_index_dicts(tmp_path / 'index_dicts.txt')
    assert i2n[1] == 'C4' and n2i['__'] == 0
    (tmp_path / 'evil.txt').write_text("__import__('os').system('true')\n{}\n")
    with pytest.raises(ValueError):
        formats.load_index_dicts(tmp_path / 'evil.txt')          # literals only: nothing is evaluated
    score = torch.arange(48, dtype=torch.int32).reshape(2, 24)
    torch.save(torch.utils.data.TensorDataset(score, score), tmp_path / 'folk')
    back = formats.load_measure_tensor(tmp_path / 'folk')
    assert back.dtype == np.int64 and (back == score.numpy()).all()


def _load_script(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, name + '.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_image_cli_flag_surface_and_reg_dims():
    """Names and defaults of the reference's click options (train_image_vae.py:12-46)."""
    mod = _load_script('train_image_vae')
    got = {p.name: p.default for p in mod.main.params}
    assert got == {'dataset_type': 'mnist', 'batch_size': 128, 'num_epochs': 100, 'lr': 1e-4, 'beta': 4.0, 'capacity': 0.0,
                   'gamma': 10.0, 'delta': 1.0, 'dec_dist': 'bernoulli', 'train': True, 'log': False, 'rand': None,
                   'reg_type': None}
    from arvae_amd.image_vae_trainer import DSPRITES_REG_TYPE, MNIST_REG_TYPES
    assert mod.reg_dims_for((), MNIST_REG_TYPES) == (0,)
    assert mod.reg_dims_for(('all',), MNIST_REG_TYPES) == (1, 2, 3, 4, 5, 6)
    assert mod.reg_dims_for(('all',), DSPRITES_REG_TYPE) == (1, 2, 3, 4, 5)
    assert mod.reg_dims_for(('slant',), MNIST_REG_TYPES) == (4,)
    assert mod.reg_dims_for(('posx', 'scale'), DSPRITES_REG_TYPE) == (4, 2)


def test_measure_cli_flag_surface():
    mod = _load_script('train_measure_vae')
    got = {p.name: p.default for p in mod.main.params}
    assert got == {'dataset_type': 'folk', 'note_embedding_dim': 10, 'metadata_embedding_dim': 2, 'num_encoder_layers': 2,
                   'encoder_hidden_size': 128, 'encoder_dropout_prob': 0.5, 'has_metadata': False, 'latent_space_dim': 32,
                   'num_decoder_layers': 2, 'decoder_hidden_size': 128, 'decoder_dropout_prob': 0.5, 'batch_size': 256,
                   'num_epochs': 30, 'lr': 1e-4, 'beta': 0.001, 'capacity': 0.0, 'gamma': 1.0, 'delta': 10.0, 'train': True,
                   'log': False, 'rand': None, 'reg_type': None}
    from arvae_amd.measure_vae_trainer import MUSIC_REG_TYPE
    assert mod.reg_dims_for(('all',), MUSIC_REG_TYPE) == (0, 1, 2, 3)


def test_device_loader_refuses_cpu_tensors():
    from arvae_amd.data import DeviceLoader
    with pytest.raises(RuntimeError):
        DeviceLoader((torch.zeros(4, 2),), 0, 4, 2, shuffle=False)


def test_sharded_loader_plan_partitions_the_rows():
    """data-parallel loaders (data/loaders.py): the ranks' row ranges of every global batch are disjoint, equal in size and
    cover the split except for a tail smaller than the world size; world = 1 is the plain ceil(n / batch) walk."""
    from arvae_amd.data.loaders import DeviceLoader
    for n, bs, world, drop in [(1000, 64, 1, False), (1000, 64, 1, True), (1003, 32, 4, False), (515, 64, 8, False), (515, 64, 8, True),
                               (7, 8, 2, False)]:
        plan = DeviceLoader.plan(n, bs, world, drop)
        seen = []
        for offset, rows in plan:
            assert 0 < rows <= bs
            for r in range(world):
                seen.extend(range(offset + r * rows, offset + (r + 1) * rows))
        assert len(seen) == len(set(seen)) and (not seen or max(seen) < n)
        if world == 1:
            assert len(plan) == (n // bs if drop else -(-n // bs)) and (drop or len(seen) == n)
        elif not drop:
            assert n - len(seen) < world
