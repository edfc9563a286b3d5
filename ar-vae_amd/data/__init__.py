"""Dataset file formats and device-resident minibatch loaders.

The reference keeps each dataset as fp32 in host RAM behind a torch DataLoader.  Here a dataset is parsed once
(`formats.py`), moved to HBM in its on-disk precision (uint8 images, fp32 label columns, int32 measures) and cut
into minibatches on the device (`loaders.DeviceLoader`): no host work and no PCIe traffic per step.
"""
from .dsprites_dataset import DspritesDataset
from .folk_dataset import FolkNBarDataset
from .loaders import DeviceLoader
from .mnist_dataset import MorphoMnistDataset

__all__ = ['DspritesDataset', 'MorphoMnistDataset', 'FolkNBarDataset', 'DeviceLoader']
