"""Morpho-MNIST (reference data/dataloaders/mnist_dataset.py)."""
import os

import torch

from . import formats
from .loaders import DeviceLoader


class MorphoMnistDataset:
    """`train` and `t10k` splits of MNIST with the Morpho-MNIST measurement table.

    Batches are (image fp32 (B,1,28,28) in [0,1], digit label uint8->int64 (B,), morpho fp32 (B,7)); validation and
    evaluation loaders both walk the t10k split in order (mnist_dataset.py:24-41).
    """

    def __init__(self, root_dir=None, device=None):
        root = os.environ.get('ARVAE_DATA_DIR', os.path.join(os.getcwd(), 'data'))
        self.root_dir = root_dir or os.path.join(root, 'mnist_data', 'plain')
        self.device = device
        self._splits = {}

    def _split(self, name):
        if name not in self._splits:
            def part(suffix):
                p = os.path.join(self.root_dir, name + suffix)
                if not os.path.exists(p):
                    raise FileNotFoundError(f'{p}: Morpho-MNIST file not found (set ARVAE_DATA_DIR or pass root_dir=)')
                return p
            images = formats.load_idx(part('-images-idx3-ubyte.gz'))
            labels = formats.load_idx(part('-labels-idx1-ubyte.gz'))
            morpho = formats.load_morpho_csv(part('-morpho.csv'))
            if not (len(images) == len(labels) == len(morpho)):
                raise ValueError(f'{self.root_dir}/{name}: images, labels and morpho table differ in length')
            dev = torch.device(self.device if self.device is not None else 'cuda')
            self._splits[name] = (torch.from_numpy(images).unsqueeze(1).to(dev),
                                  torch.from_numpy(labels.astype('int64')).to(dev),
                                  torch.from_numpy(morpho).to(dev))
        return self._splits[name]

    def data_loaders(self, batch_size, split=(0.85, 0.10), shard=None):
        train, test = self._split('train'), self._split('t10k')
        scale = 1.0 / 255.0
        return (DeviceLoader(train, 0, train[0].shape[0], batch_size, shuffle=True, u8_scale=scale, shard=shard),
                DeviceLoader(test, 0, test[0].shape[0], batch_size, shuffle=False, u8_scale=scale, shard=shard),
                DeviceLoader(test, 0, test[0].shape[0], batch_size, shuffle=False, u8_scale=scale, shard=shard))
