"""Device-resident minibatch iteration."""
import torch

from .. import ops


class DeviceLoader:
    """Stands in for torch's DataLoader over tensors that already live in HBM.

    `columns` is a tuple of device tensors sharing their first dimension; a uint8 column is expanded to fp32 by the
    library's gather kernel (arvae_gather_rows_u8) times `u8_scale`, every other column is row-gathered as it is.
    Rows [lo, hi) of the columns belong to this loader.  Yields tuples in column order, like a TensorDataset loader.
    """

    def __init__(self, columns, lo, hi, batch_size, shuffle, drop_last=False, u8_scale=1.0, generator=None, shard=None):
        if not columns or any(c.shape[0] != columns[0].shape[0] for c in columns):
            raise ValueError('columns must be non-empty and share their first dimension')
        if any(not c.is_cuda for c in columns):
            raise RuntimeError('DeviceLoader iterates device-resident tensors: move the dataset to the GPU first')
        if not (0 <= lo <= hi <= columns[0].shape[0]) or batch_size <= 0:
            raise ValueError('bad row range or batch size')
        self.columns, self.lo, self.hi = tuple(columns), int(lo), int(hi)
        self.batch_size, self.shuffle, self.drop_last = int(batch_size), bool(shuffle), bool(drop_last)
        self.u8_scale, self.generator = float(u8_scale), generator
        # data parallel: shard = (rank, world).  `batch_size` is the PER-RANK batch; a global batch is `world` consecutive
        # slices of the (rank-0) shuffled order, of which this loader yields its own; a trailing global batch that does
        # not divide evenly is cut to the largest size every rank can take (equal shards keep mean-of-ranks == global mean)
        self.rank, self.world = (int(shard[0]), int(shard[1])) if shard is not None else (0, 1)
        self.comm = shard[2] if shard is not None and len(shard) > 2 else None       # shard = (rank, world[, communicator])
        if not (0 <= self.rank < self.world):
            raise ValueError('shard = (rank, world) with 0 <= rank < world')

    @staticmethod
    def plan(n, batch_size, world=1, drop_last=False):
        """[(offset of the global batch in the order, rows per rank)] for n rows: what every rank of a sharded loader walks;
        rank r takes order[offset + r * rows : offset + (r + 1) * rows]."""
        g = batch_size * world
        out = [(b * g, batch_size) for b in range(n // g)]
        tail = (n - (n // g) * g) // world
        if tail > 0 and not drop_last:
            out.append(((n // g) * g, tail))
        return out

    def __len__(self):
        return len(self.plan(self.hi - self.lo, self.batch_size, self.world, self.drop_last))

    def __iter__(self):
        dev = self.columns[0].device
        n = self.hi - self.lo
        if self.shuffle:
            order = torch.randperm(n, device=dev, generator=self.generator) + self.lo
            if self.world > 1:                           # every rank walks rank 0's permutation
                if self.comm is None:
                    raise RuntimeError('a sharded, shuffling loader needs the communicator: shard = (rank, world, comm)')
                self.comm.broadcast(order, src=0)        # parallel.LibraryComm / TorchComm
        else:
            order = torch.arange(self.lo, self.hi, device=dev)
        for offset, rows in self.plan(n, self.batch_size, self.world, self.drop_last):
            idx = order[offset + self.rank * rows:offset + (self.rank + 1) * rows]
            yield tuple(ops.gather_rows_u8(c, idx, self.u8_scale) if c.dtype == torch.uint8 else c.index_select(0, idx)
                        for c in self.columns)
