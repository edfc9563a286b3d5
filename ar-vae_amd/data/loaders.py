"""Device-resident minibatch iteration."""
import torch

from .. import ops


class DeviceLoader:
    """Stands in for torch's DataLoader over tensors that already live in HBM.

    `columns` is a tuple of device tensors sharing their first dimension; a uint8 column is expanded to fp32 by the
    library's gather kernel (arvae_gather_rows_u8) times `u8_scale`, every other column is row-gathered as it is.
    Rows [lo, hi) of the columns belong to this loader.  Yields tuples in column order, like a TensorDataset loader.
    """

    def __init__(self, columns, lo, hi, batch_size, shuffle, drop_last=False, u8_scale=1.0, generator=None):
        if not columns or any(c.shape[0] != columns[0].shape[0] for c in columns):
            raise ValueError('columns must be non-empty and share their first dimension')
        if any(not c.is_cuda for c in columns):
            raise RuntimeError('DeviceLoader iterates device-resident tensors: move the dataset to the GPU first')
        if not (0 <= lo <= hi <= columns[0].shape[0]) or batch_size <= 0:
            raise ValueError('bad row range or batch size')
        self.columns, self.lo, self.hi = tuple(columns), int(lo), int(hi)
        self.batch_size, self.shuffle, self.drop_last = int(batch_size), bool(shuffle), bool(drop_last)
        self.u8_scale, self.generator = float(u8_scale), generator

    def __len__(self):
        n = self.hi - self.lo
        return n // self.batch_size if self.drop_last else -(-n // self.batch_size)

    def __iter__(self):
        dev = self.columns[0].device
        n = self.hi - self.lo
        if self.shuffle:
            order = torch.randperm(n, device=dev, generator=self.generator) + self.lo
        else:
            order = torch.arange(self.lo, self.hi, device=dev)
        for b in range(len(self)):
            idx = order[b * self.batch_size:(b + 1) * self.batch_size]
            yield tuple(ops.gather_rows_u8(c, idx, self.u8_scale) if c.dtype == torch.uint8 else c.index_select(0, idx)
                        for c in self.columns)
