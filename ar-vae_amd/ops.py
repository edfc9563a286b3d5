"""torch.autograd wrappers over the libarvae_hip.so C-ABI (PyTorch = bookkeeping only).

Every op takes fp32 contiguous HIP tensors and launches hand-written gfx950
kernels on the current stream.  There is no CPU path: CPU tensors raise.

Activation tensors are channels-last ([N, H, W, C]; [B, F] for dense layers).
"""
import ctypes
import os
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib
from ._lib import LinkDesc, OperandDesc

ACT_NONE, ACT_RELU, ACT_SELU = 0, 1, 2
RECON_DIST = {'bernoulli': 0, 'gaussian': 1}


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError('arvae_amd ops run on the GPU only (HIP kernels, no CPU fallback); '
                               'got a CPU tensor -- call .cuda() on the model/inputs')
        if t.dtype not in (torch.float32, torch.uint8, torch.int64):
            raise TypeError(f'unsupported dtype {t.dtype}')
        if not t.is_contiguous():
            raise ValueError('arvae_amd ops need contiguous tensors')


def _operand(v, y=None, mask=None, act=ACT_NONE):
    return OperandDesc(_ptr(v), _ptr(y), _ptr(mask), act)


@dataclass(frozen=True)
class Link:
    """Static geometry of a strided link (see include/arvae_hip.h): hi [.,hh,hw,chi] <-> lo [.,lh,lw,clo]."""
    hh: int
    hw: int
    chi: int
    lh: int
    lw: int
    clo: int
    kh: int = 1
    kw: int = 1
    stride: int = 1
    pad: int = 0
    hi_perm: Tuple[int, int] = (0, 0)
    lo_perm: Tuple[int, int] = (0, 0)

    def desc(self, n):
        return LinkDesc(n, self.hh, self.hw, self.chi, self.lh, self.lw, self.clo, self.kh, self.kw, self.stride,
                        self.pad, self.hi_perm[0], self.hi_perm[1], self.lo_perm[0], self.lo_perm[1])

    @staticmethod
    def dense(in_features, out_features, in_perm=(0, 0), out_perm=(0, 0)):
        return Link(1, 1, in_features, 1, 1, out_features, hi_perm=in_perm, lo_perm=out_perm)

    def hi_shape(self, n):
        return (n, self.chi) if self.hh == self.hw == self.lh == self.lw == 1 else (n, self.hh, self.hw, self.chi)

    def lo_shape(self, n):
        return (n, self.clo) if self.hh == self.hw == self.lh == self.lw == 1 else (n, self.lh, self.lw, self.clo)


# ------------------------------------------------------------------------------------------------
# per-kernel timing hook (bench.py): HIP events on the launch stream around every library call
# ------------------------------------------------------------------------------------------------
_PROFILE = None


def profile_begin():
    global _PROFILE
    _PROFILE = []


def profile_end():
    """-> {kernel family: dict(calls, ms, flop, bytes)} aggregated over the recorded launches."""
    global _PROFILE
    rec, _PROFILE = _PROFILE, None
    torch.cuda.synchronize()
    out = {}
    for name, flop, nbytes, e0, e1 in rec or []:
        d = out.setdefault(name, dict(calls=0, ms=0.0, flop=0.0, bytes=0.0))
        d['calls'] += 1
        d['ms'] += e0.elapsed_time(e1)
        d['flop'] += flop
        d['bytes'] += nbytes
    return out


class _timed:
    def __init__(self, name, flop=0.0, nbytes=0.0):
        self.args = (name, flop, nbytes)

    def __enter__(self):
        if _PROFILE is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if _PROFILE is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _PROFILE.append(self.args + (self.e0, e1))


def _link_cost(link, n, op):
    """(algorithmic FLOP, layer-boundary bytes) of one link launch."""
    macs = n * link.lh * link.lw * link.clo * link.chi * link.kh * link.kw
    hi_b, lo_b = 4 * n * link.hh * link.hw * link.chi, 4 * n * link.lh * link.lw * link.clo
    wt_b = 4 * link.clo * link.chi * link.kh * link.kw
    return 2.0 * macs, float(hi_b + lo_b + wt_b)


# ------------------------------------------------------------------------------------------------
# raw launches (no autograd)
# ------------------------------------------------------------------------------------------------
def _link_ws(lib, desc, device):
    """the caller-owned scratch arvae_link_down / _up ask for (most links: none)"""
    n = lib.arvae_link_ws_floats(ctypes.byref(desc))
    return torch.empty(n, device=device, dtype=torch.float32) if n else None


def link_down(link: Link, n, hi_op, wt, bias, act, out_mask, out=None):
    lib = _lib.load()
    lo = out if out is not None else torch.empty(link.lo_shape(n), device=wt.device, dtype=torch.float32)
    d = link.desc(n)
    ws = _link_ws(lib, d, wt.device)
    with _timed('link_gemm<down>', *_link_cost(link, n, hi_op)):
        _lib.check(lib.arvae_link_down(ctypes.byref(d), ctypes.byref(hi_op), _ptr(wt), _ptr(bias), act,
                                       _ptr(out_mask), _ptr(lo), _ptr(ws), _stream()), 'link_down')
    return lo


def link_up(link: Link, n, lo_op, wt, bias, act, out_mask, out=None):
    lib = _lib.load()
    hi = out if out is not None else torch.empty(link.hi_shape(n), device=wt.device, dtype=torch.float32)
    d = link.desc(n)
    name = 'up_single_channel' if (link.chi == 1 and not lo_op.y) else 'link_gemm<up>'
    ws = _link_ws(lib, d, wt.device)
    with _timed(name, *_link_cost(link, n, lo_op)):
        _lib.check(lib.arvae_link_up(ctypes.byref(d), ctypes.byref(lo_op), _ptr(wt), _ptr(bias), act,
                                     _ptr(out_mask), _ptr(hi), _ptr(ws), _stream()), 'link_up')
    return hi


def link_wgrad(link: Link, n, lo_op, hi_op, dwt, dbias=None, bias_side=0):
    """dwt += weight gradient; dbias += bias gradient (bias_side 1: sums of lo, 2: sums of hi). Accumulates."""
    lib = _lib.load()
    d = link.desc(n)
    nws = lib.arvae_link_wgrad_ws_floats(ctypes.byref(d))
    ws = torch.empty(nws, device=dwt.device, dtype=torch.float32) if nws else None
    with _timed('link_wgrad', *_link_cost(link, n, lo_op)):
        _lib.check(lib.arvae_link_wgrad(ctypes.byref(d), ctypes.byref(lo_op), ctypes.byref(hi_op), _ptr(dwt),
                                        _ptr(dbias), bias_side if dbias is not None else 0, _ptr(ws), _stream()),
                   'link_wgrad')
    return dwt


def channel_sum(op, rows, channels, perm, out):
    """out += per-channel sums (accumulates)."""
    lib = _lib.load()
    ws = torch.empty(lib.arvae_channel_sum_ws_floats(rows, channels), device=out.device, dtype=torch.float32)
    with _timed('channel_sum', float(rows * channels), 4.0 * rows * channels * (2 if op.y else 1)):
        _lib.check(lib.arvae_channel_sum(ctypes.byref(op), rows, channels, perm[0], perm[1], _ptr(out), _ptr(ws),
                                         _stream()), 'channel_sum')
    return out


LONG_BATCH_ROWS = 2048      # csrc/dense.h DENSE_SPLIT_MIN_ROWS: Linear layers over this many rows run on the rows-GEMM kernels


def _plain_gradient(g, y, mask, act, link, n):
    """operand of a Linear layer's output gradient; for long batches with an activation / mask to fold in, the folded
    gradient is written out once so that both the data- and the weight-gradient run on the plain-operand kernels."""
    dense_long = n >= LONG_BATCH_ROWS and link.hh == link.hw == link.lh == link.lw == 1
    # the wide stride-1 convolution kernels (csrc/conv64.hip) gather their operands once per tap
    wide_conv = link.stride == 1 and link.kh * link.kw > 1 and (link.chi % 32 == 0 or link.clo % 32 == 0)
    if (act != ACT_NONE or mask is not None) and (dense_long or wide_conv):
        lib = _lib.load()
        out = torch.empty_like(g)
        op = _operand(g, y, mask, act)
        _lib.check(lib.arvae_operand_apply(ctypes.byref(op), g.numel(), _ptr(out), _stream()), 'operand_apply')
        return _operand(out), out
    if act == ACT_NONE and mask is None:
        return _operand(g), g
    return _operand(g, y, mask, act), g


# ------------------------------------------------------------------------------------------------
# batch-sized Linear weight gradients of one backward pass, launched together when the pass ends
# ------------------------------------------------------------------------------------------------
_WGRAD_QUEUE = []
DEFER_DENSE_WGRADS = True


def _flush_dense_wgrads():
    """one launch for every queued Linear weight gradient (csrc/dense.hip dense_wgrad_batch_kernel)."""
    global _WGRAD_QUEUE
    if not _WGRAD_QUEUE:
        return
    queue, _WGRAD_QUEUE = _WGRAD_QUEUE, []
    lib = _lib.load()
    jobs = (_lib.DenseWgradJob * len(queue))()
    for j, (gop, x, dw, db, rows, n_in, n_out, _keep) in zip(jobs, queue):
        j.g, j.x, j.dw, j.dbias = gop, _ptr(x), _ptr(dw), _ptr(db)
        j.rows, j.n_in, j.n_out = rows, n_in, n_out
    with _timed('dense_wgrad_batch', 0.0, 0.0):
        _lib.check(lib.arvae_dense_wgrad_batch(jobs, len(queue), _stream()), 'dense_wgrad_batch')


def _defer_dense_wgrad(link, n, gop, keep, x, dw, db):
    """queue dw += g^T x (db += colsum g) until the running backward pass ends; False if it cannot be queued."""
    if not DEFER_DENSE_WGRADS or n >= LONG_BATCH_ROWS or link.hi_perm[0] or link.lo_perm[0]:
        return False
    if not (link.hh == link.hw == link.lh == link.lw == link.kh == link.kw == 1):
        return False
    if any(q[2].data_ptr() == dw.data_ptr() for q in _WGRAD_QUEUE):
        return False          # two jobs of one launch must not add into the same tensor (a weight used at several steps)
    if not _WGRAD_QUEUE:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_flush_dense_wgrads)
        except RuntimeError:                                  # not inside a backward pass
            return False
    _WGRAD_QUEUE.append((gop, x, dw, db, n, link.chi, link.clo, keep))
    return True


# bumped by every gradient the kernels add straight into a parameter's `.grad` buffer (autograd gets None back, so the
# optimizer's post-accumulate hooks do not fire): FlatAdam compares it with the value it saw at its last step()
GRAD_WRITE_EPOCH = [0]


def _grad_target(param):
    """Where a parameter gradient is accumulated.  When the parameter already owns a `.grad` buffer (the
    trainer's flat gradient arena after zero_grad()), the kernels add straight into it and autograd gets
    None back -- no zero-fill and no AccumulateGrad add per tensor.  Otherwise a fresh zeroed tensor is
    returned through autograd as usual."""
    g = param.grad
    if g is not None and g.is_contiguous() and g.dtype == torch.float32 and g.device == param.device:
        GRAD_WRITE_EPOCH[0] += 1          # no AccumulateGrad (hence no hook) will run for this write: optim.FlatAdam looks here
        return g, True
    return torch.zeros_like(param), False


# ------------------------------------------------------------------------------------------------
# layers
# ------------------------------------------------------------------------------------------------
class _LinkDownFn(Function):
    """nn.Conv2d / nn.Linear forward (+bias, activation, dropout keep-mask) and its backward."""

    @staticmethod
    def forward(ctx, hi, wt, bias, link, act, mask):
        _dev(hi, wt, bias, mask)
        n = hi.shape[0]
        lo = link_down(link, n, _operand(hi), wt, bias, act, mask)
        ctx.link, ctx.act, ctx.n = link, act, n
        ctx.save_for_backward(hi, wt, lo, mask)
        ctx.wt_ref, ctx.bias_ref = wt, bias
        return lo

    @staticmethod
    @once_differentiable
    def backward(ctx, g_lo):
        hi, wt, lo, mask = ctx.saved_tensors
        link, n = ctx.link, ctx.n
        g_lo = g_lo.contiguous()
        gop, _keep = _plain_gradient(g_lo, lo, mask, ctx.act, link, n)
        d_hi = d_wt = d_bias = None
        if ctx.needs_input_grad[0]:
            d_hi = link_up(link, n, gop, wt, None, ACT_NONE, None)
        want_bias = ctx.bias_ref is not None and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            buf, direct = _grad_target(ctx.wt_ref)
            bbuf, bdirect = _grad_target(ctx.bias_ref) if want_bias else (None, True)
            # queued launches add into the parameters' own .grad buffers; a gradient handed back through autograd
            # must be complete when backward() returns it, so those go out immediately
            if not (direct and bdirect and _defer_dense_wgrad(link, n, gop, (g_lo, _keep, lo, mask, hi), hi, buf, bbuf)):
                link_wgrad(link, n, gop, _operand(hi), buf, bbuf, 1)
            d_wt = None if direct else buf
            d_bias = None if bdirect else bbuf
        elif want_bias:
            bbuf, bdirect = _grad_target(ctx.bias_ref)
            channel_sum(gop, n * link.lh * link.lw, link.clo, link.lo_perm, bbuf)
            d_bias = None if bdirect else bbuf
        return d_hi, d_wt, d_bias, None, None, None


class _LinkUpFn(Function):
    """nn.ConvTranspose2d forward (+bias, activation, dropout keep-mask) and its backward."""

    @staticmethod
    def forward(ctx, lo, wt, bias, link, act, mask):
        _dev(lo, wt, bias, mask)
        n = lo.shape[0]
        hi = link_up(link, n, _operand(lo), wt, bias, act, mask)
        ctx.link, ctx.act, ctx.n = link, act, n
        ctx.save_for_backward(lo, wt, hi, mask)
        ctx.wt_ref, ctx.bias_ref = wt, bias
        return hi

    @staticmethod
    @once_differentiable
    def backward(ctx, g_hi):
        lo, wt, hi, mask = ctx.saved_tensors
        link, n = ctx.link, ctx.n
        g_hi = g_hi.contiguous()
        gop, _keep = _plain_gradient(g_hi, hi, mask, ctx.act, link, n)
        d_lo = d_wt = d_bias = None
        if ctx.needs_input_grad[0]:
            d_lo = link_down(link, n, gop, wt, None, ACT_NONE, None)
        want_bias = ctx.bias_ref is not None and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            buf, direct = _grad_target(ctx.wt_ref)
            bbuf, bdirect = _grad_target(ctx.bias_ref) if want_bias else (None, True)
            link_wgrad(link, n, _operand(lo), gop, buf, bbuf, 2)
            d_wt = None if direct else buf
            d_bias = None if bdirect else bbuf
        elif want_bias:
            bbuf, bdirect = _grad_target(ctx.bias_ref)
            channel_sum(gop, n * link.hh * link.hw, link.chi, link.hi_perm, bbuf)
            d_bias = None if bdirect else bbuf
        return d_lo, d_wt, d_bias, None, None, None


def conv_down(hi, wt, bias, link, act=ACT_NONE, mask=None):
    return _LinkDownFn.apply(hi, wt, bias, link, act, mask)


def conv_up(lo, wt, bias, link, act=ACT_NONE, mask=None):
    return _LinkUpFn.apply(lo, wt, bias, link, act, mask)


def dense(x, wt, bias, link, act=ACT_NONE):
    return _LinkDownFn.apply(x, wt, bias, link, act, None)


# ------------------------------------------------------------------------------------------------
# latent head / KL
# ------------------------------------------------------------------------------------------------
class _LatentFn(Function):
    @staticmethod
    def forward(ctx, mu, log_std, eps):
        _dev(mu, log_std, eps)
        lib = _lib.load()
        sigma, z = torch.empty_like(mu), torch.empty_like(mu)
        _lib.check(lib.arvae_latent_fwd(_ptr(mu), _ptr(log_std), _ptr(eps), mu.numel(), _ptr(sigma), _ptr(z),
                                        _stream()), 'latent_fwd')
        ctx.save_for_backward(eps, sigma)
        return sigma, z

    @staticmethod
    @once_differentiable
    def backward(ctx, g_sigma, g_z):
        eps, sigma = ctx.saved_tensors
        lib = _lib.load()
        g_sigma = None if g_sigma is None else g_sigma.contiguous()
        g_z = None if g_z is None else g_z.contiguous()
        d_mu, d_ls = torch.empty_like(sigma), torch.empty_like(sigma)
        _lib.check(lib.arvae_latent_bwd(_ptr(g_z), _ptr(g_sigma), _ptr(eps), _ptr(sigma), sigma.numel(), _ptr(d_mu),
                                        _ptr(d_ls), _stream()), 'latent_bwd')
        return d_mu, d_ls, None


def latent_head(mu, log_std, eps):
    """-> (sigma, z) with sigma = exp(log_std), z = mu + eps * sigma."""
    return _LatentFn.apply(mu.contiguous(), log_std.contiguous(), eps.contiguous())


class _KLDFn(Function):
    @staticmethod
    def forward(ctx, mu, sigma, prior_mu, prior_sigma, beta, capacity):
        _dev(mu, sigma, prior_mu, prior_sigma, capacity)
        lib = _lib.load()
        b, zd = mu.shape
        out = torch.empty(2, device=mu.device, dtype=torch.float32)
        _lib.check(lib.arvae_kld_fwd(_ptr(mu), _ptr(sigma), _ptr(prior_mu), _ptr(prior_sigma), b, zd, beta,
                                     _ptr(capacity), _ptr(out), _stream()), 'kld_fwd')
        ctx.beta = beta
        ctx.save_for_backward(mu, sigma, prior_mu, prior_sigma, out, capacity)
        return out[0:1] if capacity is not None else out[0]

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        mu, sigma, prior_mu, prior_sigma, out, capacity = ctx.saved_tensors
        lib = _lib.load()
        b, zd = mu.shape
        g = g.reshape(1).contiguous()
        d_mu, d_sigma = torch.empty_like(mu), torch.empty_like(mu)
        _lib.check(lib.arvae_kld_bwd(_ptr(g), _ptr(mu), _ptr(sigma), _ptr(prior_mu), _ptr(prior_sigma), b, zd,
                                     ctx.beta, _ptr(out), _ptr(capacity), _ptr(d_mu), _ptr(d_sigma), _stream()),
                   'kld_bwd')
        return d_mu, d_sigma, None, None, None, None


def kld_loss(mu, sigma, beta, capacity=None, prior_mu=None, prior_sigma=None):
    """beta * |mean_b sum_z KL(N(mu,sigma) || prior) - c|; prior None = N(0,1).
    capacity: 1-element tensor (result has shape (1,), like the reference) or None."""
    if capacity is not None:
        capacity = capacity.reshape(1).to(torch.float32).contiguous()
    return _KLDFn.apply(mu.contiguous(), sigma.contiguous(), prior_mu, prior_sigma, float(beta), capacity)


# ------------------------------------------------------------------------------------------------
# attribute regularisation
# ------------------------------------------------------------------------------------------------
class _RegLossFn(Function):
    @staticmethod
    def forward(ctx, z, labels, dims, gamma, delta, z_cols, lab_cols):
        _dev(z, labels, z_cols, lab_cols)
        lib = _lib.load()
        n_rows, ldz = z.shape
        ldl = labels.shape[1]
        zc = z if z_cols is None else z_cols
        lc = labels if lab_cols is None else lab_cols
        if zc.shape[1] != ldz or lc.shape[1] != ldl or zc.shape[0] != lc.shape[0]:
            raise ValueError('column tensors must have the row tensors\' widths')
        r = len(dims)
        ws = torch.empty(lib.arvae_reg_loss_ws_floats(n_rows, r), device=z.device, dtype=torch.float32)
        loss = torch.empty((), device=z.device, dtype=torch.float32)
        dz = torch.empty_like(z)
        cdims = (ctypes.c_int32 * r)(*dims)
        with _timed('reg_loss', 0.0, 8.0 * (n_rows + zc.shape[0]) * r):
            _lib.check(lib.arvae_reg_loss(_ptr(z), _ptr(labels), n_rows, _ptr(zc), _ptr(lc), zc.shape[0], ldz, ldl,
                                          cdims, r, gamma, delta, _ptr(ws), _ptr(loss), _ptr(dz), _stream()),
                       'reg_loss')
        ctx.save_for_backward(dz)
        return loss

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (dz,) = ctx.saved_tensors
        return scale_by_scalar(g, dz), None, None, None, None, None, None


def reg_loss(z, labels, dims, gamma, delta, z_cols=None, lab_cols=None):
    """sum over dims of gamma * mean_ij |tanh(delta (z_i - z_j)) - sign(a_i - a_j)|, all dims in one launch.
    z[:, d] pairs with labels[:, d].  z_cols/lab_cols: the all-gathered global batch under data parallelism
    (rows = this rank's samples); the mean is then over len(cols)^2 pairs."""
    dims = tuple(int(d) for d in dims)
    if len(dims) == 0:
        raise ValueError('reg_loss needs at least one dimension')
    return _RegLossFn.apply(z.contiguous(), labels.contiguous().to(torch.float32), dims, float(gamma), float(delta),
                            None if z_cols is None else z_cols.contiguous(),
                            None if lab_cols is None else lab_cols.contiguous().to(torch.float32))


def scale_by_scalar(g, x):
    lib = _lib.load()
    g = g.reshape(1).contiguous()
    y = torch.empty_like(x)
    _lib.check(lib.arvae_scale_by_scalar(_ptr(g), _ptr(x), x.numel(), _ptr(y), _stream()), 'scale_by_scalar')
    return y


# ------------------------------------------------------------------------------------------------
# reconstruction terms
# ------------------------------------------------------------------------------------------------
class _ImageReconFn(Function):
    @staticmethod
    def forward(ctx, logits, x, dist):
        _dev(logits, x)
        lib = _lib.load()
        count, batch = logits.numel(), logits.shape[0]
        ws = torch.empty(lib.arvae_recon_ws_floats(count), device=x.device, dtype=torch.float32)
        out = torch.empty(2, device=x.device, dtype=torch.float32)
        need = ctx.needs_input_grad[0]
        dl = torch.empty_like(logits) if need else None
        with _timed('image_recon', 0.0, 4.0 * count * (3 if need else 2)):
            _lib.check(lib.arvae_image_recon(_ptr(logits), _ptr(x), count, batch, dist, _ptr(ws), _ptr(out),
                                             _ptr(dl), _stream()), 'image_recon')
        if need:
            ctx.save_for_backward(dl)
        loss, acc = out[0], out[1]
        ctx.mark_non_differentiable(acc)
        return loss, acc

    @staticmethod
    @once_differentiable
    def backward(ctx, g, _g_acc):
        (dl,) = ctx.saved_tensors
        return scale_by_scalar(g, dl), None, None


def image_recon(logits, x, dist='bernoulli'):
    """-> (sum_all(term)/batch, pixel accuracy)."""
    if dist not in RECON_DIST:
        raise AttributeError('invalid dist')
    return _ImageReconFn.apply(logits.contiguous(), x.contiguous(), RECON_DIST[dist])


class _TokenReconFn(Function):
    @staticmethod
    def forward(ctx, weights, targets):
        _dev(weights, targets)
        lib = _lib.load()
        vocab = weights.shape[-1]
        rows = weights.numel() // vocab
        ws = torch.empty(lib.arvae_recon_ws_floats(rows), device=weights.device, dtype=torch.float32)
        out = torch.empty(2, device=weights.device, dtype=torch.float32)
        need = ctx.needs_input_grad[0]
        dw = torch.empty_like(weights) if need else None
        _lib.check(lib.arvae_token_recon(_ptr(weights), _ptr(targets), rows, vocab, _ptr(ws), _ptr(out), _ptr(dw),
                                         _stream()), 'token_recon')
        if need:
            ctx.save_for_backward(dw)
        loss, acc = out[0], out[1]
        ctx.mark_non_differentiable(acc)
        return loss, acc

    @staticmethod
    @once_differentiable
    def backward(ctx, g, _g_acc):
        (dw,) = ctx.saved_tensors
        return scale_by_scalar(g, dw), None


def token_recon(weights, targets):
    """-> (mean cross entropy over rows, top-1 accuracy) for weights [..., V], int64 targets [...]."""
    if targets.dtype != torch.int64:
        raise TypeError('targets must be int64')
    return _TokenReconFn.apply(weights.contiguous(), targets.contiguous())


# ------------------------------------------------------------------------------------------------
# Adam
# ------------------------------------------------------------------------------------------------
def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0, zero_grad=False, status=None):
    """one Adam update of the flat arenas; zero_grad: g is cleared by the same kernel once it has been consumed;
    status (int32 device tensor of >= 5 words, optional): while status[0] != 0 the launch leaves p, m, v alone and counts
    the skipped update in status[4] (include/arvae_hip.h, arvae_adam_step)"""
    _dev(p, g, m, v)
    lib = _lib.load()
    with _timed('adam', 0.0, 28.0 * p.numel()):
        _lib.check(lib.arvae_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), int(step), lr, beta1, beta2,
                                       eps, grad_scale, int(bool(zero_grad)), _ptr(status), _stream()), 'adam_step')


# ------------------------------------------------------------------------------------------------
# MeasureVAE building blocks (GRU gate math, embedding, concat/split, masks, argmax, attribute labels)
# ------------------------------------------------------------------------------------------------
class _GruGatesFn(Function):
    """h' = GRU cell gates given gi = W_ih x + b_ih and gh = W_hh h + b_hh ([B, 3H], gate order r|z|n)."""

    @staticmethod
    def forward(ctx, gi, gh, h_prev):
        _dev(gi, gh, h_prev)
        lib = _lib.load()
        b, h3 = gi.shape
        hid = h3 // 3
        h_new = torch.empty(b, hid, device=gi.device, dtype=torch.float32)
        saved = torch.empty(4, b, hid, device=gi.device, dtype=torch.float32)
        _lib.check(lib.arvae_gru_gates_fwd(_ptr(gi), _ptr(gh), _ptr(h_prev), b, hid, _ptr(h_new), _ptr(saved),
                                           _stream()), 'gru_gates_fwd')
        ctx.save_for_backward(saved, h_prev)
        ctx.dims = (b, hid)
        return h_new

    @staticmethod
    @once_differentiable
    def backward(ctx, dh):
        saved, h_prev = ctx.saved_tensors
        lib = _lib.load()
        b, hid = ctx.dims
        dh = dh.contiguous()
        dgi = torch.empty(b, 3 * hid, device=dh.device, dtype=torch.float32)
        dgh = torch.empty_like(dgi)
        dhp = torch.empty(b, hid, device=dh.device, dtype=torch.float32)
        _lib.check(lib.arvae_gru_gates_bwd(_ptr(dh), _ptr(saved), _ptr(h_prev), b, hid, _ptr(dgi), _ptr(dgh), _ptr(dhp),
                                           _stream()), 'gru_gates_bwd')
        return dgi, dgh, (dhp if h_prev is not None else None)


def gru_gates(gi, gh, h_prev=None):
    return _GruGatesFn.apply(gi.contiguous(), gh.contiguous(), None if h_prev is None else h_prev.contiguous())


def gru_sequence_supported(hidden):
    return bool(_lib.load().arvae_gru_seq_supported(int(hidden)))


def _gru_seq_descs(n):
    return (_lib.GruSeqDesc * n)()


class _GruSeqFn(Function):
    """All time steps of one GRU layer (1..4 directions / parameter sets) in one launch; see csrc/gru_seq.hip.

    per direction d the inputs are gi_d (T, R, 3H) -- or (R, 3H) when the same projection feeds every step --,
    w_hh_d, b_hh_d, h0_d (R, H) or None.  With gi_merged (T, R, ndir*3H) the directions read their input projection from
    columns [3H d, 3H (d+1)) of that ONE tensor (both directions of a bidirectional layer projected by one GEMM,
    dense_pair) and the per-direction gi_d are None; its gradient comes back as one tensor as well.
    Outputs: (T, R, ndir*H) with direction d in columns [dH, (d+1)H), and the final states (R, ndir*H) = nn.GRU's h_n (each
    direction's last processed step), written by the sequence launch itself."""

    @staticmethod
    def forward(ctx, steps, reverse, want_finals, gi_merged, *tensors):
        ndir = len(reverse)
        gis, whs, bhs, h0s = tensors[0::4], tensors[1::4], tensors[2::4], tensors[3::4]
        _dev(*[t for t in tensors if t is not None])
        lib = _lib.load()
        hid = whs[0].shape[1]
        rows = gi_merged.shape[1] if gi_merged is not None else gis[0].shape[-2]
        dev = whs[0].device
        out = torch.empty(steps, rows, ndir * hid, device=dev, dtype=torch.float32)
        saved = torch.empty(ndir, steps, rows, 4 * hid, device=dev, dtype=torch.float32)
        finals = torch.empty(rows, ndir * hid, device=dev, dtype=torch.float32) if want_finals else None
        descs = _gru_seq_descs(ndir)
        for d in range(ndir):
            q = descs[d]
            if gi_merged is not None:
                q.gi = gi_merged.data_ptr() + 4 * d * 3 * hid
                q.gi_tstride, q.gi_rstride = rows * ndir * 3 * hid, ndir * 3 * hid
            else:
                q.gi, q.gi_tstride = _ptr(gis[d]), (rows * 3 * hid if gis[d].dim() == 3 else 0)
            q.w_hh, q.b_hh, q.h0 = _ptr(whs[d]), _ptr(bhs[d]), _ptr(h0s[d])
            q.h_all, q.h_stride = out.data_ptr() + 4 * d * hid, ndir * hid
            q.saved, q.reverse = saved[d].data_ptr(), int(reverse[d])
            if finals is not None:
                q.h_fin, q.h_fin_stride = finals.data_ptr() + 4 * d * hid, ndir * hid
        with _timed('gru_seq_fwd', 2.0 * ndir * steps * rows * 3 * hid * hid, 4.0 * ndir * steps * rows * 8 * hid):
            _lib.check(lib.arvae_gru_seq_fwd(descs, ndir, steps, rows, hid, _stream()), 'gru_seq_fwd')
        ctx.save_for_backward(out, saved, *whs, *[h for h in h0s if h is not None])
        ctx.h0_present = [h is not None for h in h0s]
        ctx.gi_const = [g is not None and g.dim() == 2 for g in gis]
        ctx.merged = gi_merged is not None
        ctx.refs = (whs, bhs)
        ctx.geom = (steps, rows, hid, tuple(reverse))
        ctx.set_materialize_grads(False)
        if not want_finals:
            none = out.new_empty(0)                              # (an empty tensor: no launch)
            ctx.mark_non_differentiable(none)
            return out, none
        return out, finals

    @staticmethod
    @once_differentiable
    def backward(ctx, d_out, d_fin):
        steps, rows, hid, reverse = ctx.geom
        ndir = len(reverse)
        out, saved = ctx.saved_tensors[:2]
        whs = ctx.saved_tensors[2:2 + ndir]
        h0_it = iter(ctx.saved_tensors[2 + ndir:])
        h0s = [next(h0_it) if p else None for p in ctx.h0_present]
        lib = _lib.load()
        dev = out.device
        d_out = None if d_out is None else d_out.contiguous()
        d_fin = None if (d_fin is None or d_fin.numel() == 0) else d_fin.contiguous()
        if ctx.merged:                                           # one (T, R, ndir * 3H) gradient for the one projection
            dgi_all = torch.empty(steps, rows, ndir * 3 * hid, device=dev, dtype=torch.float32)
            dgi = None
        else:
            dgi_all = None
            dgi = torch.empty(ndir, steps, rows, 3 * hid, device=dev, dtype=torch.float32)
        dgh = torch.empty(ndir, steps, rows, 3 * hid, device=dev, dtype=torch.float32)
        h_prev = torch.empty(ndir, steps, rows, hid, device=dev, dtype=torch.float32)
        base = 4                                                 # inputs: steps, reverse, want_finals, gi_merged, then 4 per direction
        dh0 = [torch.empty(rows, hid, device=dev, dtype=torch.float32)
               if (h0s[d] is not None and ctx.needs_input_grad[base + 4 * d + 3]) else None for d in range(ndir)]
        descs = _gru_seq_descs(ndir)
        for d in range(ndir):
            q = descs[d]
            q.w_hh, q.h0 = _ptr(whs[d]), _ptr(h0s[d])
            q.h_all, q.h_stride = out.data_ptr() + 4 * d * hid, ndir * hid
            q.saved, q.reverse = saved[d].data_ptr(), int(reverse[d])
            if d_out is not None:
                q.dh_all, q.dh_stride = d_out.data_ptr() + 4 * d * hid, ndir * hid
            if d_fin is not None:
                q.dh_last, q.dh_last_stride = d_fin.data_ptr() + 4 * d * hid, ndir * hid
            if ctx.merged:
                q.dgi, q.dgi_rstride = dgi_all.data_ptr() + 4 * d * 3 * hid, ndir * 3 * hid
            else:
                q.dgi = dgi[d].data_ptr()
            q.dgh, q.dh0 = dgh[d].data_ptr(), _ptr(dh0[d])
            q.h_prev_out = h_prev[d].data_ptr()
        with _timed('gru_seq_bwd', 2.0 * ndir * steps * rows * 3 * hid * hid, 4.0 * ndir * steps * rows * 12 * hid):
            _lib.check(lib.arvae_gru_seq_bwd(descs, ndir, steps, rows, hid, _stream()), 'gru_seq_bwd')
        grads = [None, None, None, dgi_all if (ctx.merged and ctx.needs_input_grad[3]) else None]
        link = Link.dense(hid, 3 * hid)
        w_refs, b_refs = ctx.refs
        for d in range(ndir):
            d_w = d_b = None
            if ctx.needs_input_grad[base + 4 * d + 1]:
                buf, direct = _grad_target(w_refs[d])
                bbuf, bdirect = _grad_target(b_refs[d])
                link_wgrad(link, steps * rows, _operand(dgh[d].view(steps * rows, 3 * hid)),
                           _operand(h_prev[d].view(steps * rows, hid)), buf, bbuf, 1)
                d_w, d_b = (None if direct else buf), (None if bdirect else bbuf)
            g_gi = None
            if not ctx.merged and ctx.needs_input_grad[base + 4 * d]:
                g_gi = dgi[d].sum(0) if ctx.gi_const[d] else dgi[d]
            grads += [g_gi, d_w, d_b, dh0[d]]
        return tuple(grads)


def gru_sequence(steps, directions, finals=True, merged_gi=None):
    """directions: list of (gi, w_hh, b_hh, h0, reverse) -> (outputs (T, R, ndir*H), final states (R, ndir*H) or, with
    finals=False, None).  merged_gi (T, R, ndir*3H): the directions' input projections as one tensor (their gi entries None)."""
    flat = []
    for gi, w_hh, b_hh, h0, _ in directions:
        flat += [None if gi is None else gi.contiguous(), w_hh, b_hh, None if h0 is None else h0.contiguous()]
    out, fin = _GruSeqFn.apply(int(steps), tuple(bool(d[4]) for d in directions), bool(finals),
                               None if merged_gi is None else merged_gi.contiguous(), *flat)
    return out, (fin if finals else None)


def _adjacent(a, b):
    """b starts where a ends, in the same allocation (two parameters of the trainer's flat arena laid out back to back)"""
    return (a is not None and b is not None and a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype and a.device == b.device
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
            and a.data_ptr() + a.numel() * a.element_size() == b.data_ptr())


class _DensePairFn(Function):
    """two Linear layers (+ one activation) on the same input whose weights (and biases) are adjacent in memory, as one
    [out_a + out_b, in] layer"""

    @staticmethod
    def forward(ctx, x, w_a, w_b, b_a, b_b, act):
        _dev(x, w_a, w_b, b_a, b_b)
        n, n_in, n_out = x.shape[0], w_a.shape[1], w_a.shape[0] + w_b.shape[0]
        link = Link.dense(n_in, n_out)
        w_cat = w_a.detach().as_strided((n_out, n_in), (n_in, 1))
        b_cat = b_a.detach().as_strided((n_out,), (1,))
        out = link_down(link, n, _operand(x), w_cat, b_cat, act, None)
        ctx.link, ctx.n, ctx.act = link, n, act
        ctx.save_for_backward(x, w_cat, out)
        ctx.refs = (w_a, w_b, b_a, b_b)
        ctx.x_ref = x if (x.is_leaf and x.requires_grad) else None
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        x, w_cat, out = ctx.saved_tensors
        link, n = ctx.link, ctx.n
        g = g.contiguous()
        gop, keep = _plain_gradient(g, out, None, ctx.act, link, n)
        d_x = link_up(link, n, gop, w_cat, None, ACT_NONE, None) if ctx.needs_input_grad[0] else None
        x_ref = ctx.x_ref
        if d_x is not None and x_ref is not None and x_ref.grad is not None and x_ref.grad.shape == d_x.shape:
            # the input is itself a parameter with a gradient buffer (an embedding table projected as a whole): added here, on
            # this stream, instead of by an AccumulateGrad node (whose stream is not the capture's: graphed.py)
            _grad_target(x_ref)
            x_ref.grad.add_(d_x)
            d_x = None
        w_a, w_b, b_a, b_b = ctx.refs
        if ctx.needs_input_grad[1]:
            _grad_target(w_a), _grad_target(b_a)                 # (the arena is written without autograd's accumulation)
            n_out, n_in = w_cat.shape
            gw = w_a.grad.as_strided((n_out, n_in), (n_in, 1))
            gb = b_a.grad.as_strided((n_out,), (1,))
            if not _defer_dense_wgrad(link, n, gop, (g, keep, out, x), x, gw, gb):
                link_wgrad(link, n, gop, _operand(x), gw, gb, 1)
        return d_x, None, None, None, None, None


def dense_pair(x, w_a, b_a, w_b, b_b, act=ACT_NONE):
    """act(x (n, in) times [w_a; w_b]^T + [b_a; b_b]) as ONE product -> (n, out_a + out_b) when the two layers' weights, biases
    and (when gradients are on) gradient buffers sit back to back in memory -- the trainer's flat arena in
    Model.arena_parameters() order -- ; None when they do not (the caller runs the layers one by one)."""
    if not (_adjacent(w_a, w_b) and _adjacent(b_a, b_b) and w_a.shape[1] == w_b.shape[1]):
        return None
    if torch.is_grad_enabled() and (w_a.requires_grad or w_b.requires_grad):
        if not (w_a.requires_grad and w_b.requires_grad and _adjacent(w_a.grad, w_b.grad) and _adjacent(b_a.grad, b_b.grad)):
            return None
    return _DensePairFn.apply(x, w_a, w_b, b_a, b_b, int(act))


def tick_free_run_supported(hidden, vocab):
    """True when the one-launch free-running tick decoder is built for (hidden, vocab); else go tick by tick."""
    return bool(_lib.load().arvae_tick_free_run_supported(int(hidden), int(vocab)))


def tick_free_run(weights, h0_l0, h0_l1, gib, ptab, mask, keep_scale, batch, beats, ticks_per_beat):
    """tokens (B, beats*ticks_per_beat) int64 of the free-running tick decoder; no autograd (csrc/gru_seq.hip).
    weights = (w_hh0, b_hh0, w_ih1, b_ih1, w_hh1, b_hh1, w_out, b_out)."""
    _dev(*weights, h0_l0, h0_l1, gib, ptab, mask)
    lib = _lib.load()
    hid, vocab = weights[0].shape[1], weights[6].shape[0]
    tw = _lib.TickWeights(*[_ptr(t) for t in weights])
    tokens = torch.empty(batch, beats * ticks_per_beat, device=gib.device, dtype=torch.int64)
    ws = torch.empty(lib.arvae_tick_free_run_ws_floats(hid), device=gib.device, dtype=torch.float32)
    with _timed('tick_free_run', 2.0 * batch * beats * ticks_per_beat * (9 * hid * hid + vocab * hid), 0.0):
        _lib.check(lib.arvae_tick_free_run(ctypes.byref(tw), _ptr(h0_l0), _ptr(h0_l1), 0, _ptr(gib), _ptr(ptab), _ptr(mask),
                                           float(keep_scale), batch, beats, ticks_per_beat, hid, vocab, _ptr(tokens),
                                           _ptr(ws), _stream()), 'tick_free_run')
    return tokens


class _EmbedFn(Function):
    @staticmethod
    def forward(ctx, idx, table, time_major):
        _dev(idx, table)
        lib = _lib.load()
        b, steps = idx.shape
        v, dim = table.shape
        rows = (steps, b) if time_major else (b, steps)
        out = torch.empty(rows + (dim,), device=table.device, dtype=torch.float32)
        _lib.check(lib.arvae_embed_fwd(_ptr(idx), _ptr(table), b, steps, dim, v, int(time_major), _ptr(out), _stream()),
                   'embed_fwd')
        ctx.save_for_backward(idx)
        ctx.table_ref, ctx.time_major = table, time_major
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        lib = _lib.load()
        table = ctx.table_ref
        b, steps = idx.shape
        v, dim = table.shape
        if table.is_leaf and table.grad is not None:
            buf, direct = _grad_target(table)                    # the parameter's own gradient buffer: add into it
        else:                                                    # (a projection table computed upstream: a fresh, overwritten tensor)
            buf, direct = torch.empty_like(table), False
        ws = torch.empty(lib.arvae_embed_bwd_ws_floats(b, steps, dim, v), device=buf.device, dtype=torch.float32)
        _lib.check(lib.arvae_embed_bwd(_ptr(idx), _ptr(g.contiguous()), b, steps, dim, v, int(ctx.time_major), _ptr(buf),
                                       int(direct), _ptr(ws), _stream()), 'embed_bwd')
        return None, (None if direct else buf), None


def embed(idx, table, time_major=False):
    """nn.Embedding lookup of int64 idx [B, T] -> [B, T, D] (or [T, B, D] when time_major)."""
    if idx.dtype != torch.int64:
        raise TypeError('embedding indices must be int64')
    return _EmbedFn.apply(idx.contiguous(), table, bool(time_major))


class _TickInputFn(Function):
    """gi0 (ticks_per_beat, beats*batch, 3H) of the tick RNN's first layer from (embedding table, x_0, beat embeddings, W_ih0,
    b_ih0, fed-back tokens): one small product over vocab + 1 + beats*batch rows and a gather, instead of a whole-sequence GEMM
    over 24*batch rows of concatenated inputs (include/arvae_hip.h, arvae_tick_*; measurevae/decoder.py:459-505)."""

    @staticmethod
    def forward(ctx, table, x0, beat_emb, w_ih, b_ih, tokens, beats, tpb):
        _dev(table, x0, beat_emb, w_ih, b_ih, tokens)
        lib = _lib.load()
        vocab, emb = table.shape
        rows, hid = beat_emb.shape
        batch = rows // beats
        cols = w_ih.shape[0]
        n = vocab + 1 + rows
        dev = table.device
        x_small = torch.empty(n, emb + hid, device=dev, dtype=torch.float32)
        _lib.check(lib.arvae_tick_rows_fwd(_ptr(table), _ptr(x0), _ptr(beat_emb), 0, vocab, emb, hid, rows, _ptr(x_small), _stream()),
                   'tick_rows_fwd')
        link = Link.dense(emb + hid, cols)
        g_small = link_down(link, n, _operand(x_small), w_ih, None, ACT_NONE, None)
        gi = torch.empty(tpb, rows, cols, device=dev, dtype=torch.float32)
        with _timed('tick_gi_fwd'):
            _lib.check(lib.arvae_tick_gi_fwd(_ptr(g_small), _ptr(tokens), _ptr(b_ih), batch, beats, tpb, vocab, cols, _ptr(gi), _stream()),
                       'tick_gi_fwd')
        ctx.save_for_backward(x_small, tokens, w_ih)
        ctx.refs = (table, x0, w_ih, b_ih)
        ctx.geom = (vocab, emb, hid, rows, batch, beats, tpb, cols, link, n)
        return gi

    @staticmethod
    @once_differentiable
    def backward(ctx, d_gi):
        x_small, tokens, w_ih = ctx.saved_tensors
        table, x0, w_ref, b_ref = ctx.refs
        vocab, emb, hid, rows, batch, beats, tpb, cols, link, n = ctx.geom
        lib = _lib.load()
        dev = x_small.device
        d_gi = d_gi.contiguous()
        dg = torch.empty(n, cols, device=dev, dtype=torch.float32)
        ws = torch.empty(lib.arvae_tick_gi_bwd_ws_floats(vocab, cols), device=dev, dtype=torch.float32)
        with _timed('tick_gi_bwd'):
            _lib.check(lib.arvae_tick_gi_bwd(_ptr(d_gi), _ptr(tokens), batch, beats, tpb, vocab, cols, _ptr(dg), _ptr(ws), _stream()),
                       'tick_gi_bwd')
        gop = _operand(dg)
        # every tick row carries the bias once and exactly one note entry: db = the column sums of the vocab + 1 note rows of dG
        d_b = None
        if ctx.needs_input_grad[4]:
            bbuf, bdirect = _grad_target(b_ref)
            channel_sum(_operand(dg[:vocab + 1]), vocab + 1, cols, (0, 0), bbuf)
            d_b = None if bdirect else bbuf
        d_w = None
        if ctx.needs_input_grad[3]:
            buf, direct = _grad_target(w_ref)
            if not (direct and _defer_dense_wgrad(link, n, gop, (dg, x_small), x_small, buf, None)):
                link_wgrad(link, n, gop, _operand(x_small), buf, None, 0)
            d_w = None if direct else buf
        dx = link_up(link, n, gop, w_ih, None, ACT_NONE, None)
        tbuf, tdirect = _grad_target(table) if ctx.needs_input_grad[0] else (None, True)
        xbuf, xdirect = _grad_target(x0) if ctx.needs_input_grad[1] else (None, True)
        d_beat = torch.empty(rows, hid, device=dev, dtype=torch.float32)
        _lib.check(lib.arvae_tick_rows_bwd(_ptr(dx), vocab, emb, hid, rows, _ptr(tbuf), _ptr(xbuf), _ptr(d_beat), 0, _stream()),
                   'tick_rows_bwd')
        return (None if tdirect else tbuf), (None if xdirect else xbuf), d_beat, d_w, d_b, None, None, None


def tick_input_projection(table, x0, beat_emb, w_ih, b_ih, tokens, beats, ticks_per_beat):
    """-> gi0 (ticks_per_beat, beats*batch, 3H); tokens (batch, beats*ticks_per_beat) int64, beat_emb (beats*batch, H)"""
    return _TickInputFn.apply(table, x0, beat_emb.contiguous(), w_ih, b_ih, tokens.contiguous(), int(beats), int(ticks_per_beat))


def row_argmax(w):
    """top-1 index per row of a [rows, cols] tensor, lowest index on ties; int64 [rows]."""
    _dev(w)
    lib = _lib.load()
    w = w.contiguous()
    idx = torch.empty(w.shape[0], device=w.device, dtype=torch.int64)
    _lib.check(lib.arvae_row_argmax(_ptr(w), w.shape[0], w.shape[1], _ptr(idx), _stream()), 'row_argmax')
    return idx


class _ConcatFn(Function):
    @staticmethod
    def forward(ctx, a, b):
        _dev(a, b)
        lib = _lib.load()
        rows, ca, cb = a.shape[0], a.shape[1], b.shape[1]
        out = torch.empty(rows, ca + cb, device=a.device, dtype=torch.float32)
        _lib.check(lib.arvae_concat_cols(_ptr(a), _ptr(b), rows, ca, cb, _ptr(out), _stream()), 'concat_cols')
        ctx.dims = (rows, ca, cb)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        lib = _lib.load()
        rows, ca, cb = ctx.dims
        g = g.contiguous()
        da = torch.empty(rows, ca, device=g.device, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        db = torch.empty(rows, cb, device=g.device, dtype=torch.float32) if ctx.needs_input_grad[1] else None
        _lib.check(lib.arvae_split_cols(_ptr(g), rows, ca, cb, _ptr(da), _ptr(db), 0, _stream()), 'split_cols')
        return da, db


def concat_cols(a, b):
    """torch.cat((a, b), dim=1) for 2-D tensors."""
    return _ConcatFn.apply(a.contiguous(), b.contiguous())


class _SplitFn(Function):
    @staticmethod
    def forward(ctx, x, ca):
        _dev(x)
        lib = _lib.load()
        rows, c = x.shape
        cb = c - ca
        a = torch.empty(rows, ca, device=x.device, dtype=torch.float32)
        b = torch.empty(rows, cb, device=x.device, dtype=torch.float32)
        _lib.check(lib.arvae_split_cols(_ptr(x), rows, ca, cb, _ptr(a), _ptr(b), 0, _stream()), 'split_cols')
        ctx.dims = (rows, ca, cb)
        return a, b

    @staticmethod
    @once_differentiable
    def backward(ctx, ga, gb):
        lib = _lib.load()
        rows, ca, cb = ctx.dims
        dev = (ga if ga is not None else gb).device
        ga = torch.zeros(rows, ca, device=dev) if ga is None else ga.contiguous()
        gb = torch.zeros(rows, cb, device=dev) if gb is None else gb.contiguous()
        out = torch.empty(rows, ca + cb, device=dev, dtype=torch.float32)
        _lib.check(lib.arvae_concat_cols(_ptr(ga), _ptr(gb), rows, ca, cb, _ptr(out), _stream()), 'concat_cols')
        return out, None


def split_cols(x, ca):
    """(x[:, :ca], x[:, ca:]) as contiguous tensors."""
    return _SplitFn.apply(x.contiguous(), int(ca))


class _ScaleMaskFn(Function):
    @staticmethod
    def forward(ctx, x, mask, alpha):
        _dev(x, mask)
        lib = _lib.load()
        y = torch.empty_like(x)
        _lib.check(lib.arvae_scale_mask(_ptr(x), _ptr(mask), alpha, x.numel(), 0, _ptr(y), _stream()), 'scale_mask')
        ctx.save_for_backward(mask)
        ctx.alpha = alpha
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (mask,) = ctx.saved_tensors
        lib = _lib.load()
        g = g.contiguous()
        dx = torch.empty_like(g)
        _lib.check(lib.arvae_scale_mask(_ptr(g), _ptr(mask), ctx.alpha, g.numel(), 0, _ptr(dx), _stream()), 'scale_mask')
        return dx, None, None


# ------------------------------------------------------------------------------------------------
# debug-mode failure checks (ARVAE_CHECK=1): the reference's NaN-weight scan and note-index range check
# ------------------------------------------------------------------------------------------------
def checks_enabled():
    return os.environ.get('ARVAE_CHECK', '0') == '1'


def check_finite(params, what):
    """raise ValueError when any tensor of `params` holds a NaN / infinity (measurevae/encoder.py:101-106, decoder.py:420-425):
    one counting launch per tensor into one device word, ONE host sync (the reference syncs once per parameter)."""
    params = [p for p in params if p is not None and p.numel() > 0]
    if not params or torch.cuda.is_current_stream_capturing():
        return
    lib = _lib.load()
    flag = torch.zeros(1, dtype=torch.int32, device=params[0].device)
    for p in params:
        t = p.detach().contiguous()
        _dev(t)
        _lib.check(lib.arvae_count_nonfinite(_ptr(t), t.numel(), _ptr(flag), _stream()), 'count_nonfinite')
    bad = int(flag)
    if bad:
        print(f'{what} has become nan')
        raise ValueError(f'{what}: {bad} non-finite weight value(s)')


def check_index(indices, num_notes):
    """raise ValueError unless every index is in [0, num_notes) (Decoder.check_index, measurevae/decoder.py:30-41)"""
    if torch.cuda.is_current_stream_capturing():
        return
    idx = indices.contiguous()
    _dev(idx)
    flag = torch.zeros(1, dtype=torch.int32, device=idx.device)
    _lib.check(_lib.load().arvae_count_out_of_range(_ptr(idx), idx.numel(), 0, int(num_notes), _ptr(flag), _stream()),
               'count_out_of_range')
    bad = int(flag)
    if bad:
        print('Invalid Values of Indices: ', int(idx.min()), int(idx.max()))
        raise ValueError(f'{bad} note index(es) outside [0, {num_notes})')


# ------------------------------------------------------------------------------------------------
# random draws: the library's counter-based Philox4x32-10 stream (csrc/rng.h), no torch RNG kernels
# ------------------------------------------------------------------------------------------------
class RngState:
    """Every draw of the path (eps, dropout keep-masks) is element i of the stream keyed by
    (seed, offset, step + *dev_step): seed = torch.initial_seed() salted with the data-parallel rank, so the trainers'
    torch.manual_seed(rand) (image_vae_trainer.py:103) selects the stream as it does in the reference and every rank of a
    data-parallel run draws DIFFERENT noise and keep-masks for its rows ("RNG under DP": rank 0's
    stream is the single-process one); `offset` counts the draws since the stream was last selected (`rng_reseed`, called
    where the trainers call torch.manual_seed, restarts it -- the reference's generator reset).
    Under HIP-graph replay the host numbers are frozen into the captured launches, so graphed.GraphedStep advances the device
    word `dev_step` inside the graph and every replay sees fresh values."""
    offset = 0
    rank = 0                           # data-parallel rank (parallel.DataParallel sets it)
    dev_step = None                    # int32 tensor (1,) on the device, or None


RANK_SALT = 0x9E3779B97F4A7C15         # odd 64-bit constant: rank r's key = seed + r * RANK_SALT (mod 2^64)


def rng_seed():
    return (torch.initial_seed() + RngState.rank * RANK_SALT) & 0xFFFFFFFFFFFFFFFF


def rng_set_rank(rank):
    """select this process's sub-stream (0 = the single-process stream)"""
    RngState.rank = int(rank)


def rng_reseed(seed):
    """torch.manual_seed(seed) + restart of the library's stream position: what the trainers' constructors call, so that
    training seed k in a loop over seeds draws what a fresh process with --rand k draws"""
    torch.manual_seed(seed)
    RngState.offset = 0
    if RngState.dev_step is not None:
        RngState.dev_step.zero_()


def rng_next_offset():
    off = RngState.offset
    RngState.offset = (off + 1) & 0xFFFFFFFF
    return off


def rng_device_step(device, create=False):
    """the device word a captured graph advances (None until a graph asked for one)"""
    if create and (RngState.dev_step is None or RngState.dev_step.device != torch.device(device)):
        RngState.dev_step = torch.zeros(1, dtype=torch.int32, device=device)
    d = RngState.dev_step
    return d if d is not None and d.device == torch.device(device) else None


def rng_advance_device_step():
    """one more step of the device-side stream position (recorded into a graph being captured)"""
    if RngState.dev_step is not None:
        RngState.dev_step.add_(1)


def normal_noise(shape, device):
    """eps ~ N(0, 1) for z_dist.rsample() (mnist_vae.py:79, measure_vae.py:116): one library launch"""
    out = torch.empty(shape, dtype=torch.float32, device=device)
    _dev(out)
    _lib.check(_lib.load().arvae_philox_normal(_ptr(out), out.numel(), rng_seed(), rng_next_offset(), 0,
                                                _ptr(rng_device_step(device)), _stream()), 'philox_normal')
    return out


def keep_mask(shape, p, device):
    """uint8 keep-mask (1 with probability 1 - p) of nn.Dropout(p) / nn.GRU(dropout=p): one library launch"""
    out = torch.empty(shape, dtype=torch.uint8, device=device)
    _dev(out)
    _lib.check(_lib.load().arvae_philox_keep_mask(_ptr(out), out.numel(), 1.0 - float(p), rng_seed(), rng_next_offset(), 0,
                                                   _ptr(rng_device_step(device)), _stream()), 'philox_keep_mask')
    return out


def keep_masks(shapes, p, device):
    """the keep-masks of several Dropout layers of one step as ONE launch (arvae_philox_keep_masks); mask j is the step's next
    draw, byte for byte what keep_mask(shapes[j], ...) called in the same order would return"""
    if len(shapes) > 8:
        return [keep_mask(s, p, device) for s in shapes]
    outs = [torch.empty(s, dtype=torch.uint8, device=device) for s in shapes]
    _dev(*outs)
    n = len(outs)
    ptrs = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
    counts = (ctypes.c_int64 * n)(*[o.numel() for o in outs])
    offsets = (ctypes.c_uint32 * n)(*[rng_next_offset() for _ in outs])
    _lib.check(_lib.load().arvae_philox_keep_masks(n, ptrs, counts, 1.0 - float(p), rng_seed(), offsets, 0, _ptr(rng_device_step(device)),
                                                   _stream()), 'philox_keep_masks')
    return outs


def dropout_mask(x, mask, p=0.5):
    """y = x * mask / (1 - p) for an explicit uint8 keep-mask (None: identity)."""
    if mask is None:
        return x
    return _ScaleMaskFn.apply(x.contiguous(), mask.contiguous(), 1.0 / (1.0 - p))


class _BroadcastFn(Function):
    @staticmethod
    def forward(ctx, v, rows):
        _dev(v)
        lib = _lib.load()
        cols = v.numel()
        y = torch.empty(rows, cols, device=v.device, dtype=torch.float32)
        _lib.check(lib.arvae_broadcast_rows(_ptr(v), rows, cols, _ptr(y), _stream()), 'broadcast_rows')
        ctx.v_ref, ctx.dims = v, (rows, cols)
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        rows, cols = ctx.dims
        buf, direct = _grad_target(ctx.v_ref)
        channel_sum(_operand(g.contiguous()), rows, cols, (0, 0), buf.view(-1))
        return (None if direct else buf), None


def broadcast_rows(v, rows):
    """v[None, :].expand(rows, -1) materialised (the learned start vectors)."""
    return _BroadcastFn.apply(v, int(rows))


def gather_rows_u8(src, idx, scale=1.0):
    """src (N, ...) uint8 on the device, idx (B,) int64 -> (B, ...) fp32 = scale * src[idx]."""
    _dev(src, idx)
    if src.dtype != torch.uint8 or idx.dtype != torch.int64:
        raise TypeError('gather_rows_u8 needs a uint8 source and int64 indices')
    lib = _lib.load()
    src, idx = src.contiguous(), idx.contiguous()
    row = src[0].numel()
    out = torch.empty((idx.numel(),) + tuple(src.shape[1:]), device=src.device, dtype=torch.float32)
    _lib.check(lib.arvae_gather_rows_u8(_ptr(src), src.shape[0], row, _ptr(idx), idx.numel(), float(scale), _ptr(out),
                                        _stream()), 'gather_rows_u8')
    return out


def measure_attributes(score, tables, rhythm_weights, rhythm_norm):
    """(B, 24) int64 measures -> (B, 4) [rhythmic complexity, pitch range, note density, contour]."""
    _dev(score)
    lib = _lib.load()
    midi, is_note, is_dens = tables
    b, steps = score.shape
    out = torch.empty(b, 4, device=score.device, dtype=torch.float32)
    _lib.check(lib.arvae_measure_attributes(_ptr(score.contiguous()), b, steps, _ptr(midi), _ptr(is_note), _ptr(is_dens),
                                            midi.numel(), _ptr(rhythm_weights), float(rhythm_norm), _ptr(out), _stream()),
               'measure_attributes')
    return out
