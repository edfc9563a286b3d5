"""Model base class: checkpoint naming / save / load / Xavier init.

API surface of the reference's utils/model.py:6-97 (filepath, trainer_config,
update_filepath, update_trainer_config, save, save_checkpoint, load,
xavier_initialization).  state_dict keys and tensor layouts are the
reference's, so its checkpoints load unchanged.
"""
import os

import torch
from torch import nn


def models_root():
    """<repo>/models unless ARVAE_MODEL_DIR is set (reference: <repo>/models, utils/model.py:26-33)."""
    env = os.environ.get('ARVAE_MODEL_DIR')
    if env:
        return env
    return os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'models')


class Model(nn.Module):
    def __init__(self, filepath=None):
        super().__init__()
        self.filepath = filepath
        self.trainer_config = ''

    def forward(self, *args, **kwargs):
        raise NotImplementedError

    def __repr__(self):
        raise NotImplementedError

    # -- naming ---------------------------------------------------------------------------------
    def update_filepath(self):
        name = repr(self)
        self.filepath = os.path.join(models_root(), name, name + '.pt')

    def update_trainer_config(self, config):
        self.trainer_config = config
        self.update_filepath()

    # -- persistence ----------------------------------------------------------------------------
    def _ensure_dir(self):
        folder = os.path.dirname(self.filepath)
        os.makedirs(folder, exist_ok=True)
        return folder

    def save(self):
        self._ensure_dir()
        torch.save(self.state_dict(), self.filepath)
        print(f'Model {repr(self)} saved')

    def save_checkpoint(self, epoch_num):
        folder = self._ensure_dir()
        torch.save(self.state_dict(), os.path.join(folder, f'{repr(self)}_{epoch_num}.pt'))

    def load(self, cpu=False):
        state = torch.load(self.filepath, map_location='cpu' if cpu else None)
        self.load_state_dict(state)

    # -- init -----------------------------------------------------------------------------------
    def xavier_initialization(self):
        """xavier_normal_ on every parameter whose name contains 'weight' (utils/model.py:90-97)."""
        for name, param in self.named_parameters():
            if 'weight' in name:
                nn.init.xavier_normal_(param)


class ParamLayer(nn.Module):
    """Holds `weight` (+ `bias`) under the reference's state_dict names; the math runs in HIP kernels."""

    def __init__(self, weight_shape, bias_len, fan_in):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(*weight_shape))
        self.bias = nn.Parameter(torch.empty(bias_len))
        bound = 1.0 / (fan_in ** 0.5)                       # PyTorch's default bias init
        nn.init.uniform_(self.bias, -bound, bound)
        nn.init.kaiming_uniform_(self.weight, a=5 ** 0.5)


class LayerStack(nn.Module):
    """Children registered under explicit indices so keys read `enc_conv.0.weight`, `enc_conv.3.bias`, ..."""

    def __init__(self, layers):
        super().__init__()
        for idx, layer in layers:
            self.add_module(str(idx), layer)

    def __getitem__(self, idx):
        return self._modules[str(idx)]
