"""Functional CPU restatement of the conv AR-VAEs (test infrastructure; see oracle/__init__.py).

Parameters are a flat {state_dict key: tensor} mapping with the reference's key
names and layouts (conv OIHW, conv-transpose IOHW, linear [out,in]).
  dSprites stack : reference imagevae/dsprites_vae.py:12-46
  MNIST stack    : reference imagevae/mnist_vae.py:16-47
  forward        : reference imagevae/mnist_vae.py:59-105
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

SELU_ALPHA = 1.6732632423543772
SELU_SCALE = 1.0507009873554805

DSPRITES_SHAPES = OrderedDict([
    ('enc_conv.0.weight', (32, 1, 4, 4)), ('enc_conv.0.bias', (32,)),
    ('enc_conv.2.weight', (32, 32, 4, 4)), ('enc_conv.2.bias', (32,)),
    ('enc_conv.4.weight', (32, 32, 4, 4)), ('enc_conv.4.bias', (32,)),
    ('enc_conv.6.weight', (32, 32, 4, 4)), ('enc_conv.6.bias', (32,)),
    ('enc_lin.0.weight', (256, 512)), ('enc_lin.0.bias', (256,)),
    ('enc_lin.2.weight', (256, 256)), ('enc_lin.2.bias', (256,)),
    ('enc_mean.weight', (10, 256)), ('enc_mean.bias', (10,)),
    ('enc_log_std.weight', (10, 256)), ('enc_log_std.bias', (10,)),
    ('dec_lin.0.weight', (256, 10)), ('dec_lin.0.bias', (256,)),
    ('dec_lin.2.weight', (256, 256)), ('dec_lin.2.bias', (256,)),
    ('dec_lin.4.weight', (512, 256)), ('dec_lin.4.bias', (512,)),
    ('dec_conv.0.weight', (32, 32, 4, 4)), ('dec_conv.0.bias', (32,)),
    ('dec_conv.2.weight', (32, 32, 4, 4)), ('dec_conv.2.bias', (32,)),
    ('dec_conv.4.weight', (32, 32, 4, 4)), ('dec_conv.4.bias', (32,)),
    ('dec_conv.6.weight', (32, 1, 4, 4)), ('dec_conv.6.bias', (1,)),
])

MNIST_SHAPES = OrderedDict([
    ('enc_conv.0.weight', (64, 1, 4, 4)), ('enc_conv.0.bias', (64,)),
    ('enc_conv.3.weight', (64, 64, 4, 4)), ('enc_conv.3.bias', (64,)),
    ('enc_conv.6.weight', (8, 64, 4, 4)), ('enc_conv.6.bias', (8,)),
    ('enc_lin.0.weight', (256, 2888)), ('enc_lin.0.bias', (256,)),
    ('enc_mean.weight', (16, 256)), ('enc_mean.bias', (16,)),
    ('enc_log_std.weight', (16, 256)), ('enc_log_std.bias', (16,)),
    ('dec_lin.0.weight', (256, 16)), ('dec_lin.0.bias', (256,)),
    ('dec_lin.2.weight', (2888, 256)), ('dec_lin.2.bias', (2888,)),
    ('dec_conv.0.weight', (8, 64, 4, 4)), ('dec_conv.0.bias', (64,)),
    ('dec_conv.3.weight', (64, 64, 4, 4)), ('dec_conv.3.bias', (64,)),
    ('dec_conv.6.weight', (64, 1, 4, 4)), ('dec_conv.6.bias', (1,)),
])

SHAPES = {'dsprites': DSPRITES_SHAPES, 'mnist': MNIST_SHAPES}
Z_DIM = {'dsprites': 10, 'mnist': 16}
MNIST_MASK_SHAPES = [(64, 25, 25), (64, 22, 22), (8, 19, 19), (64, 22, 22), (64, 25, 25)]


def selu(x):
    return SELU_SCALE * torch.where(x > 0, x, SELU_ALPHA * (torch.exp(x) - 1.0))


def _drop(h, mask):
    """explicit keep-mask dropout, p = 0.5: y = h * mask * 2; mask None = off."""
    return h if mask is None else h * mask.to(h.dtype).reshape(h.shape) * 2.0


def encode(kind, p, x, masks=None):
    """-> (mu, log_std).  reference mnist_vae.py:59-66."""
    b = x.shape[0]
    if kind == 'dsprites':
        h = x
        for i in (0, 2, 4, 6):
            h = F.relu(F.conv2d(h, p[f'enc_conv.{i}.weight'], p[f'enc_conv.{i}.bias'], stride=2, padding=1))
        h = h.reshape(b, -1)
        h = F.relu(F.linear(h, p['enc_lin.0.weight'], p['enc_lin.0.bias']))
        h = F.relu(F.linear(h, p['enc_lin.2.weight'], p['enc_lin.2.bias']))
    else:
        h = x
        for k, i in enumerate((0, 3, 6)):
            h = selu(F.conv2d(h, p[f'enc_conv.{i}.weight'], p[f'enc_conv.{i}.bias']))
            h = _drop(h, None if masks is None else masks[k])
        h = h.reshape(b, -1)
        h = selu(F.linear(h, p['enc_lin.0.weight'], p['enc_lin.0.bias']))
    mu = F.linear(h, p['enc_mean.weight'], p['enc_mean.bias'])
    log_std = F.linear(h, p['enc_log_std.weight'], p['enc_log_std.bias'])
    return mu, log_std


def decode(kind, p, z, masks=None):
    """-> logits (B,1,H,W).  reference mnist_vae.py:68-72."""
    b = z.shape[0]
    if kind == 'dsprites':
        h = z
        for i in (0, 2, 4):
            h = F.relu(F.linear(h, p[f'dec_lin.{i}.weight'], p[f'dec_lin.{i}.bias']))
        h = h.reshape(b, 32, 4, 4)
        for i in (0, 2, 4):
            h = F.relu(F.conv_transpose2d(h, p[f'dec_conv.{i}.weight'], p[f'dec_conv.{i}.bias'],
                                          stride=2, padding=1))
        return F.conv_transpose2d(h, p['dec_conv.6.weight'], p['dec_conv.6.bias'], stride=2, padding=1)
    h = selu(F.linear(z, p['dec_lin.0.weight'], p['dec_lin.0.bias']))
    h = selu(F.linear(h, p['dec_lin.2.weight'], p['dec_lin.2.bias']))
    h = h.reshape(b, 8, 19, 19)
    for k, i in enumerate((0, 3)):
        h = selu(F.conv_transpose2d(h, p[f'dec_conv.{i}.weight'], p[f'dec_conv.{i}.bias']))
        h = _drop(h, None if masks is None else masks[3 + k])
    return F.conv_transpose2d(h, p['dec_conv.6.weight'], p['dec_conv.6.bias'])


def forward(kind, p, x, eps, masks=None):
    """-> (logits, mu, sigma, z) with z = mu + eps * exp(log_std)
    (rsample: reference mnist_vae.py:74-87; the second, unused z_prior draw is
    not modelled -- it only advances the reference's RNG)."""
    mu, log_std = encode(kind, p, x, masks)
    sigma = torch.exp(log_std)
    z = mu + eps * sigma
    return decode(kind, p, z, masks), mu, sigma, z
