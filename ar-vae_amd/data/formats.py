"""Parsers for the three on-disk formats the reference trains from.  numpy / stdlib only: no GPU needed.

* dSprites      `dsprites_ndarray_co1sh3sc6or40x32y32_64x64.npz`: `imgs` uint8 (N,64,64) in {0,1},
                `latents_values` float64 (N,6) = color, shape, scale, orientation, posX, posY
                (data/dataloaders/dsprites_dataset.py:38-53)
* Morpho-MNIST  `<split>-images-idx3-ubyte.gz`, `<split>-labels-idx1-ubyte.gz` (IDX, big-endian header) and
                `<split>-morpho.csv` with a header line and 7 numeric columns per digit
                (data/dataloaders/mnist_dataset.py:60-82; column meaning: image_vae_trainer.py:20-28)
* Folk measures a `torch.save`d TensorDataset holding one int tensor (N, 24) twice, plus `index_dicts.txt`: two lines,
                the Python literals of index2note_dicts and note2index_dicts (data/dataloaders/bar_dataset.py:804-841)
"""
import ast
import gzip
import struct

import numpy as np

_IDX_DTYPES = {0x08: np.uint8, 0x09: np.int8, 0x0B: '>i2', 0x0C: '>i4', 0x0D: '>f4', 0x0E: '>f8'}


def load_dsprites_npz(path):
    """-> (imgs uint8 (N,64,64), latents float32 (N,6))."""
    with np.load(path, encoding='bytes', allow_pickle=True) as data:
        if 'imgs' not in data or 'latents_values' not in data:
            raise ValueError(f'{path}: not a dSprites archive (needs `imgs` and `latents_values`)')
        imgs = np.ascontiguousarray(data['imgs'], dtype=np.uint8)
        latents = np.ascontiguousarray(data['latents_values'], dtype=np.float32)
    if imgs.ndim != 3 or latents.ndim != 2 or len(imgs) != len(latents):
        raise ValueError(f'{path}: imgs {imgs.shape} / latents_values {latents.shape} do not match')
    return imgs, latents


def load_idx(path):
    """IDX file (optionally gzip-compressed) -> numpy array of its stored dtype and shape."""
    opener = gzip.open if str(path).endswith('.gz') else open
    with opener(path, 'rb') as f:
        head = f.read(4)
        if len(head) != 4 or head[0] != 0 or head[1] != 0 or head[2] not in _IDX_DTYPES:
            raise ValueError(f'{path}: bad IDX magic {head!r}')
        ndim = head[3]
        dims = struct.unpack('>' + 'I' * ndim, f.read(4 * ndim)) if ndim else ()
        dtype = np.dtype(_IDX_DTYPES[head[2]])
        count = int(np.prod(dims, dtype=np.int64)) if ndim else 1
        raw = f.read(count * dtype.itemsize)
    if len(raw) != count * dtype.itemsize:
        raise ValueError(f'{path}: truncated IDX payload ({len(raw)} of {count * dtype.itemsize} bytes)')
    return np.frombuffer(raw, dtype=dtype).reshape(dims).astype(dtype.newbyteorder('='), copy=True)


def save_idx(array, path):
    """Inverse of load_idx for uint8 arrays (used by the tests to fabricate MNIST-shaped files)."""
    array = np.ascontiguousarray(array, dtype=np.uint8)
    opener = gzip.open if str(path).endswith('.gz') else open
    with opener(path, 'wb') as f:
        f.write(bytes([0, 0, 0x08, array.ndim]))
        f.write(struct.pack('>' + 'I' * array.ndim, *array.shape))
        f.write(array.tobytes())


def load_morpho_csv(path):
    """Header line + numeric rows -> float32 (N, n_columns)."""
    table = np.loadtxt(path, delimiter=',', skiprows=1, dtype=np.float64, ndmin=2)
    return table.astype(np.float32)


def load_index_dicts(path):
    """-> (index2note, note2index): two single-line Python dict literals."""
    with open(path, 'r') as f:
        lines = [ln.rstrip('\n') for ln in f if ln.strip()]
    if len(lines) != 2:
        raise ValueError(f'{path}: expected two dictionary lines, found {len(lines)}')
    return ast.literal_eval(lines[0]), ast.literal_eval(lines[1])


def load_measure_tensor(path):
    """`torch.save`d TensorDataset (or tensor / tuple of tensors) -> int64 numpy (N, ticks)."""
    import torch
    obj = torch.load(path, map_location='cpu', weights_only=False)
    tensors = getattr(obj, 'tensors', obj)
    score = tensors[0] if isinstance(tensors, (tuple, list)) else tensors
    score = torch.as_tensor(score)
    if score.dim() != 2:
        raise ValueError(f'{path}: expected an (N, ticks) measure tensor, found {tuple(score.shape)}')
    return score.to(torch.int64).numpy()
