"""One-bar folk measures (reference data/dataloaders/bar_dataset.py, FolkNBarDataset): the already-built tensor file
and its vocabulary.  Parsing ABC tunes with music21 (the offline step that writes these files) is out of scope."""
import os

import torch

from . import formats
from .loaders import DeviceLoader


class FolkNBarDataset:
    class_name = '4by4_FolkNBarDataset_1_'

    def __init__(self, dataset_type='train', is_short=False, num_bars=1, dataset_dir=None, device=None):
        if num_bars != 1:
            raise ValueError('only the one-bar dataset the AR-VAE experiments use is supported')
        root = os.environ.get('ARVAE_DATA_DIR', os.path.join(os.getcwd(), 'data'))
        self.dataset_dir_path = dataset_dir or os.path.join(root, 'folk_raw_data')
        self.n_bars, self.num_bars = num_bars, num_bars
        self.dataset_type, self.device = dataset_type, device
        self.dict_path = os.path.join(self.dataset_dir_path, 'index_dicts.txt')
        self.dataset_path = os.path.join(self.dataset_dir_path, self.class_name + dataset_type + ('_short' if is_short else ''))
        self.index2note_dicts = self.note2index_dicts = None
        self.score = None
        if os.path.exists(self.dict_path):
            self.compute_dicts()

    def __repr__(self):
        return self.class_name

    def compute_dicts(self):
        if not os.path.exists(self.dict_path):
            raise FileNotFoundError(f'{self.dict_path}: vocabulary file not found')
        self.index2note_dicts, self.note2index_dicts = formats.load_index_dicts(self.dict_path)

    def get_dataset(self):
        if self.score is None:
            if not os.path.exists(self.dataset_path):
                raise FileNotFoundError(f'{self.dataset_path}: measure tensor not found (set ARVAE_DATA_DIR)')
            if self.index2note_dicts is None:
                self.compute_dicts()
            dev = torch.device(self.device if self.device is not None else 'cuda')
            self.score = torch.from_numpy(formats.load_measure_tensor(self.dataset_path)).to(dev)
        return self.score

    def data_loaders(self, batch_size, split=(0.85, 0.10), shard=None):
        assert sum(split) < 1
        score = self.get_dataset()
        n = score.shape[0]
        a, b = split
        cut1, cut2 = int(a * n), int((a + b) * n)
        cols = (score, score)                        # (score, metadata placeholder), as the reference stores it
        return (DeviceLoader(cols, 0, cut1, batch_size, shuffle=True, drop_last=True, shard=shard),
                DeviceLoader(cols, cut1, cut2, batch_size, shuffle=False, drop_last=True, shard=shard),
                DeviceLoader(cols, cut2, n, batch_size, shuffle=False, drop_last=True, shard=shard))
