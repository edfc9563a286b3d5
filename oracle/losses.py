"""Loss terms of the AR-VAE step, restated on CPU (test infrastructure; see oracle/__init__.py).

numpy float64 closed forms (independent of autograd) + differentiable torch
versions used by oracle.step.
"""
import numpy as np
import torch


# ----------------------------------------------------------------------------
# attribute-regularisation loss   (reference utils/trainer.py:369-403)
# ----------------------------------------------------------------------------
def reg_loss_closed_form(x, a, gamma=1.0, delta=1.0, cols_x=None, cols_a=None):
    """float64 loss and analytic gradient of
        L = gamma * mean_{i,j} | tanh(delta (x_i - x_j)) - sign(a_i - a_j) |
    (diagonal included, N^2 pairs: utils/trainer.py:390-401).

    Row-block form for data-parallel runs: `x`, `a` are the LOCAL rows and
    `cols_x`, `cols_a` the gathered global columns (defaults: same as rows);
    the mean is over len(cols)^2 pairs and the returned gradient is the full
    d(global L)/d x_i for the local rows (pair term is symmetric under i<->j, so
    row + column contributions = 2 x row sum).
    """
    x = np.asarray(x, np.float64)
    a = np.asarray(a, np.float64)
    cx = x if cols_x is None else np.asarray(cols_x, np.float64)
    ca = a if cols_a is None else np.asarray(cols_a, np.float64)
    n = cx.shape[0]
    t = np.tanh(delta * (x[:, None] - cx[None, :]))
    s = np.sign(a[:, None] - ca[None, :])
    loss = gamma * np.abs(t - s).sum() / (n * n)
    grad = (2.0 * gamma * delta / (n * n)) * ((1.0 - t * t) * np.sign(t - s)).sum(1)
    return loss, grad


def reg_loss(z, labels, dims, gamma, delta):
    """Differentiable torch form, summed over regularised dims:
    sum_d gamma * L1(tanh(delta * dz_d), sign(da_d)) with z[:, d] paired with
    labels[:, d] (image_vae_trainer.py:171-180, measure_vae_trainer.py:135-139)."""
    total = z.new_zeros(())
    for d in dims:
        a = labels[:, d]
        x = z[:, d]
        dx = x[:, None] - x[None, :]
        da = a[:, None] - a[None, :]
        total = total + gamma * (torch.tanh(delta * dx) - torch.sign(da)).abs().mean()
    return total


# ----------------------------------------------------------------------------
# KL term   (reference utils/trainer.py:354-367; torch _kl_normal_normal vs N(0,1))
# ----------------------------------------------------------------------------
def kld_rows(mu, sigma):
    """per-row sum of 0.5 (sigma^2 + mu^2 - 1 - log sigma^2)."""
    var = sigma * sigma
    return (0.5 * (var + mu * mu - 1.0 - torch.log(var))).sum(1)


def kld_loss(mu, sigma, beta, c=0.0):
    """beta * | mean_rows(KL) - c |."""
    return beta * (kld_rows(mu, sigma).mean() - c).abs()


def kld_closed_form(mu, log_std, beta, c=0.0):
    """float64 value and gradients wrt (mu, log_std) with sigma = exp(log_std)."""
    mu = np.asarray(mu, np.float64)
    ls = np.asarray(log_std, np.float64)
    var = np.exp(2.0 * ls)
    b = mu.shape[0]
    kl = (0.5 * (var + mu * mu - 1.0 - 2.0 * ls)).sum(1).mean()
    sgn = np.sign(kl - c)
    return beta * abs(kl - c), beta * sgn * mu / b, beta * sgn * (var - 1.0) / b


# ----------------------------------------------------------------------------
# reconstruction terms
# ----------------------------------------------------------------------------
class _BCEWithLogitsSum(torch.autograd.Function):
    """sum of max(l,0) - l x + log1p(exp(-|l|)); closed-form gradient sigmoid(l) - x
    (autograd through clamp/abs would give the wrong sub-gradient at l == 0)."""

    @staticmethod
    def forward(ctx, logits, x):
        ctx.save_for_backward(logits, x)
        return (torch.clamp(logits, min=0) - logits * x + torch.log1p(torch.exp(-logits.abs()))).sum()

    @staticmethod
    def backward(ctx, g):
        logits, x = ctx.saved_tensors
        return g * (torch.sigmoid(logits) - x), None


def bce_with_logits_per_batch(logits, x):
    """sum over pixels AND batch of [max(l,0) - l x + log1p(exp(-|l|))], / B
    (reference image_vae_trainer.py:626-629)."""
    return _BCEWithLogitsSum.apply(logits, x) / x.shape[0]


def gaussian_recon_per_batch(logits, x):
    """sum (sigmoid(l) - x)^2 / B   (image_vae_trainer.py:630-634)."""
    return ((torch.sigmoid(logits) - x) ** 2).sum() / x.shape[0]


def pixel_accuracy(logits, x):
    """mean [(sigmoid(l) >= .5) == (x >= .5)] == mean [(l >= 0) == (x >= .5)]
    (image_vae_trainer.py:639-655)."""
    return ((logits >= 0) == (x >= 0.5)).float().mean()


def cross_entropy_mean(weights, targets):
    """mean over B*T rows of -log_softmax(w)[target]   (utils/trainer.py:247-264)."""
    v = weights.shape[-1]
    w = weights.reshape(-1, v)
    tgt = targets.reshape(-1)
    lse = torch.logsumexp(w, dim=1)
    return (lse - w.gather(1, tgt[:, None])[:, 0]).mean()


def top1_accuracy(weights, targets):
    """mean(argmax == target), first index on ties   (utils/trainer.py:266-282)."""
    v = weights.shape[-1]
    return (weights.reshape(-1, v).argmax(1) == targets.reshape(-1)).float().mean()


# ----------------------------------------------------------------------------
# Adam   (reference utils/trainer.py:31-34 -> torch.optim.Adam defaults)
# ----------------------------------------------------------------------------
def adam_step(p, g, m, v, step, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8):
    """One Adam update on numpy arrays (float32 in, float32 out; math in the
    same order as torch: eps is added AFTER the bias-corrected sqrt).
    `step` is the 1-based step count.  Returns (p, m, v)."""
    g = g.astype(np.float32)
    m = (beta1 * m + (1.0 - beta1) * g).astype(np.float32)
    v = (beta2 * v + (1.0 - beta2) * g * g).astype(np.float32)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    denom = np.sqrt(v) / np.float32(np.sqrt(bc2)) + np.float32(eps)
    p = (p - np.float32(lr / bc1) * (m / denom)).astype(np.float32)
    return p, m, v


def reg_loss_row_block(z_rows, lab_rows, dims, gamma, delta, z_cols=None, lab_cols=None):
    """Differentiable row-block form used under data parallelism: rows = this rank's
    samples, cols = the gathered global batch (constants).  VALUE = this rank's share of the global loss
    (mean over len(cols)^2 pairs); GRADIENT w.r.t. z_rows = the FULL d(global loss)/d z_i, i.e. twice the
    row-role gradient (the pair term is symmetric under i <-> j)."""
    zc = z_rows.detach() if z_cols is None else z_cols.detach()
    lc = lab_rows if lab_cols is None else lab_cols
    n = zc.shape[0]
    total = z_rows.new_zeros(())
    for d in dims:
        dx = z_rows[:, d][:, None] - zc[:, d][None, :]
        da = lab_rows[:, d][:, None] - lc[:, d][None, :]
        total = total + gamma * (torch.tanh(delta * dx) - torch.sign(da)).abs().sum() / (n * n)
    return 2.0 * total - total.detach()
