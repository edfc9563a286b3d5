"""dSprites (reference data/dataloaders/dsprites_dataset.py)."""
import os

import torch

from . import formats
from .loaders import DeviceLoader

DSPRITES_FILE = 'dsprites_ndarray_co1sh3sc6or40x32y32_64x64.npz'


class DspritesDataset:
    """737 280 binary 64x64 sprites + 6 generative factors.

    The set stays uint8 in HBM (3.0 GB instead of the reference's 12 GB fp32 host copy).  As in the reference the
    train / validation / evaluation split is by position in the file: its `np.random.shuffle` acts on a scratch array
    after the image and label arrays were already copied out of it (dsprites_dataset.py:44-53), so nothing is shuffled
    before the split; only the training loader shuffles, per epoch.
    """

    def __init__(self, path=None, device=None):
        root = os.environ.get('ARVAE_DATA_DIR', os.path.join(os.getcwd(), 'data'))
        self.data_path = path or os.path.join(root, DSPRITES_FILE)
        self.device = device
        self.images = self.latents = None

    def load_dataset(self):
        if not os.path.exists(self.data_path):
            raise FileNotFoundError(f'{self.data_path}: dSprites archive not found (set ARVAE_DATA_DIR or pass path=)')
        imgs, latents = formats.load_dsprites_npz(self.data_path)
        dev = torch.device(self.device if self.device is not None else 'cuda')
        self.images = torch.from_numpy(imgs).unsqueeze(1).to(dev)          # (N,1,64,64) uint8
        self.latents = torch.from_numpy(latents).to(dev)                   # (N,6) fp32

    def __len__(self):
        if self.images is None:
            self.load_dataset()
        return self.images.shape[0]

    def data_loaders(self, batch_size, split=(0.80, 0.15), shard=None):
        assert sum(split) < 1
        n = len(self)
        a, b = split
        cut1, cut2 = int(a * n), int((a + b) * n)
        cols = (self.images, self.latents)
        return (DeviceLoader(cols, 0, cut1, batch_size, shuffle=True, shard=shard),
                DeviceLoader(cols, cut1, cut2, batch_size, shuffle=True, shard=shard),
                DeviceLoader(cols, cut2, n, batch_size, shuffle=False, shard=shard))
