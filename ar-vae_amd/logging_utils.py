"""Scalar/image logging used by the trainers when log=True.

The reference logs through tensorboardX.SummaryWriter (utils/trainer.py:48-56).
tensorboardX / tensorboard may be absent; then a JSON-lines writer with the
same add_scalar / add_image calls is used so training never depends on them.
"""
import json
import os

import torch


class JsonlWriter:
    def __init__(self, logdir):
        os.makedirs(logdir, exist_ok=True)
        self.path = os.path.join(logdir, 'scalars.jsonl')
        self.logdir = logdir

    def add_scalar(self, tag, value, step):
        with open(self.path, 'a') as f:
            f.write(json.dumps({'tag': tag, 'value': float(value), 'step': int(step)}) + '\n')

    def add_image(self, tag, image, step):
        torch.save(image, os.path.join(self.logdir, f"{tag.replace('/', '_')}_{int(step)}.pt"))

    def close(self):
        pass


def make_writer(logdir):
    try:
        from tensorboardX import SummaryWriter
        return SummaryWriter(logdir=logdir)
    except ImportError:
        try:
            from torch.utils.tensorboard import SummaryWriter
            return SummaryWriter(log_dir=logdir)
        except ImportError:
            return JsonlWriter(logdir)


def image_grid(images, nrow, pad_value=1.0, padding=2):
    """(N,C,H,W) -> one (C, rows*(H+pad)+pad, nrow*(W+pad)+pad) grid (what torchvision.utils.make_grid builds)."""
    n, c, h, w = images.shape
    rows = (n + nrow - 1) // nrow
    grid = torch.full((c, rows * (h + padding) + padding, nrow * (w + padding) + padding), pad_value)
    for i in range(n):
        r, col = divmod(i, nrow)
        y, x = padding + r * (h + padding), padding + col * (w + padding)
        grid[:, y:y + h, x:x + w] = images[i]
    return grid
