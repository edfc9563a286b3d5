"""One full training step on CPU: loss terms, gradients, Adam update
(test infrastructure; see oracle/__init__.py).

Mirrors the reference's step driver utils/trainer.py:126-147 around
  image  : imagevae/image_vae_trainer.py:137-217
  measure: measurevae/measure_vae_trainer.py:95-165
"""
import numpy as np
import torch

from . import image_vae, losses, measure_vae


def _to_params(state):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).clone().requires_grad_(True) for k, v in state.items()}


def _finish(params, terms, loss, lr, adam_state, step_no):
    loss.backward()
    grads = {k: v.grad.detach().numpy().copy() for k, v in params.items()}
    new_params, new_state = {}, {}
    for k, v in params.items():
        m, vv = adam_state.get(k, (np.zeros_like(grads[k]), np.zeros_like(grads[k]))) if adam_state else \
            (np.zeros_like(grads[k]), np.zeros_like(grads[k]))
        p, m, vv = losses.adam_step(v.detach().numpy(), grads[k], m, vv, step_no, lr=lr)
        new_params[k] = p
        new_state[k] = (m, vv)
    terms['loss'] = float(loss.detach())
    return dict(terms=terms, grads=grads, params=new_params, adam=new_state)


def image_step(kind, state, x, labels, eps, reg_dims, beta, gamma, delta, capacity=0.0,
               dec_dist='bernoulli', masks=None, lr=1e-4, adam_state=None, step_no=1):
    """state: {key: ndarray}.  Returns dict(terms, grads, params, adam) plus
    'z', 'mu', 'sigma', 'logits' as ndarrays under terms['...']."""
    p = _to_params(state)
    xt, lt, et = (torch.from_numpy(np.ascontiguousarray(a)) for a in (x, labels, eps))
    mt = None if masks is None else [torch.from_numpy(m) for m in masks]
    logits, mu, sigma, z = image_vae.forward(kind, p, xt, et, mt)
    if dec_dist == 'bernoulli':
        recons = losses.bce_with_logits_per_batch(logits, xt)
    elif dec_dist == 'gaussian':
        recons = losses.gaussian_recon_per_batch(logits, xt)
    else:
        raise AttributeError('invalid dist')
    dist = losses.kld_loss(mu, sigma, beta, capacity)
    reg = losses.reg_loss(z, lt, reg_dims, gamma, delta) if len(reg_dims) else z.new_zeros(())
    loss = recons + dist + reg
    terms = dict(recons=float(recons.detach()), dist=float(dist.detach()), reg=float(reg.detach()),
                 acc=float(losses.pixel_accuracy(logits.detach(), xt)),
                 z=z.detach().numpy().copy(), mu=mu.detach().numpy().copy(),
                 sigma=sigma.detach().numpy().copy(), logits=logits.detach().numpy().copy())
    return _finish(p, terms, loss, lr, adam_state, step_no)


def measure_step(state, score, eps, attr, reg_dims, beta, gamma, delta, teacher_forced,
                 masks=None, lr=1e-4, adam_state=None, step_no=1):
    p = _to_params(state)
    st = torch.from_numpy(np.ascontiguousarray(score))
    et = torch.from_numpy(np.ascontiguousarray(eps))
    at = torch.from_numpy(np.ascontiguousarray(attr))
    weights, samples, mu, sigma, z = measure_vae.forward(p, st, et, teacher_forced, masks)
    recons = losses.cross_entropy_mean(weights, st)
    dist = losses.kld_loss(mu, sigma, beta, 0.0)
    reg = losses.reg_loss(z, at, reg_dims, gamma, delta) if len(reg_dims) else z.new_zeros(())
    loss = recons + dist + reg
    terms = dict(recons=float(recons.detach()), dist=float(dist.detach()), reg=float(reg.detach()),
                 acc=float(losses.top1_accuracy(weights.detach(), st)),
                 z=z.detach().numpy().copy(), mu=mu.detach().numpy().copy(),
                 sigma=sigma.detach().numpy().copy(), weights=weights.detach().numpy().copy(),
                 samples=samples.numpy().copy())
    return _finish(p, terms, loss, lr, adam_state, step_no)
