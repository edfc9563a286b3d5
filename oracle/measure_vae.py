"""Functional CPU restatement of MeasureVAE (test infrastructure; see oracle/__init__.py).

GRU cells are written out by hand from the state_dict tensors (PyTorch gate
order r, z, n stacked in rows):
    r = sigmoid(W_ir x + b_ir + W_hr h + b_hr)
    z = sigmoid(W_iz x + b_iz + W_hz h + b_hz)
    n = tanh   (W_in x + b_in + r * (W_hn h + b_hn))
    h' = (1 - z) * n + z * h
  encoder : reference measurevae/encoder.py:94-124
  decoder : reference measurevae/decoder.py:388-525 (HierarchicalDecoder)
  wiring  : reference measurevae/measure_vae.py:97-131
Inter-layer GRU dropout is modelled with optional explicit keep-masks (parity
cases run p = 0 or eval mode).
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

from .image_vae import selu

TICKS, BEATS, TICKS_PER_BEAT = 24, 4, 6


def shapes(v=35, emb=10, hid=128, zdim=32):
    """state_dict key -> shape, in the reference's registration order."""
    s = OrderedDict()
    g = 3 * hid
    for layer, inp in ((0, emb), (1, 2 * hid)):
        for suf in ('', '_reverse'):
            s[f'encoder.lstm.weight_ih_l{layer}{suf}'] = (g, inp)
            s[f'encoder.lstm.weight_hh_l{layer}{suf}'] = (g, hid)
            s[f'encoder.lstm.bias_ih_l{layer}{suf}'] = (g,)
            s[f'encoder.lstm.bias_hh_l{layer}{suf}'] = (g,)
    s['encoder.note_embedding_layer.weight'] = (v, emb)
    for head in ('linear_mean', 'linear_log_std'):
        s[f'encoder.{head}.0.weight'] = (2 * hid, 4 * hid)
        s[f'encoder.{head}.0.bias'] = (2 * hid,)
        s[f'encoder.{head}.2.weight'] = (zdim, 2 * hid)
        s[f'encoder.{head}.2.bias'] = (zdim,)
    s['decoder.b_0'] = (1,)
    s['decoder.x_0'] = (emb,)
    s['decoder.note_embedding_layer.weight'] = (v, emb)
    s['decoder.z_to_beat_rnn_input.0.weight'] = (2 * hid, zdim)
    s['decoder.z_to_beat_rnn_input.0.bias'] = (2 * hid,)
    for rnn, inp in (('rnn_beat', 1),):
        for layer, i in ((0, inp), (1, hid)):
            s[f'decoder.{rnn}.weight_ih_l{layer}'] = (g, i)
            s[f'decoder.{rnn}.weight_hh_l{layer}'] = (g, hid)
            s[f'decoder.{rnn}.bias_ih_l{layer}'] = (g,)
            s[f'decoder.{rnn}.bias_hh_l{layer}'] = (g,)
    s['decoder.beat_emb_to_tick_rnn_hidden.0.weight'] = (2 * hid, hid)
    s['decoder.beat_emb_to_tick_rnn_hidden.0.bias'] = (2 * hid,)
    s['decoder.beat_emb_to_tick_rnn_input.0.weight'] = (hid, hid)
    s['decoder.beat_emb_to_tick_rnn_input.0.bias'] = (hid,)
    for layer, i in ((0, emb + hid), (1, hid)):
        s[f'decoder.rnn_tick.weight_ih_l{layer}'] = (g, i)
        s[f'decoder.rnn_tick.weight_hh_l{layer}'] = (g, hid)
        s[f'decoder.rnn_tick.bias_ih_l{layer}'] = (g,)
        s[f'decoder.rnn_tick.bias_hh_l{layer}'] = (g,)
    s['decoder.tick_emb_to_note_emb.0.weight'] = (v, hid)
    s['decoder.tick_emb_to_note_emb.0.bias'] = (v,)
    return s


def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    hid = h.shape[1]
    gi = F.linear(x, w_ih, b_ih)
    gh = F.linear(h, w_hh, b_hh)
    r = torch.sigmoid(gi[:, :hid] + gh[:, :hid])
    z = torch.sigmoid(gi[:, hid:2 * hid] + gh[:, hid:2 * hid])
    n = torch.tanh(gi[:, 2 * hid:] + r * gh[:, 2 * hid:])
    return (1.0 - z) * n + z * h


def _cell(p, prefix, x, h):
    return gru_cell(x, h, p[prefix.format('weight_ih')], p[prefix.format('weight_hh')],
                    p[prefix.format('bias_ih')], p[prefix.format('bias_hh')])


def _keep(h, mask):
    return h if mask is None else h * mask.to(h.dtype) * 2.0


def encode(p, score, enc_mask=None):
    """score (B,24) int64 -> (mu, log_std).  enc_mask: optional (B,24,256) keep
    mask on the layer-0 outputs (nn.GRU dropout acts between layers only)."""
    b = score.shape[0]
    hid = p['encoder.lstm.weight_hh_l0'].shape[1]
    x = p['encoder.note_embedding_layer.weight'][score]                 # (B,24,emb)
    finals = []
    seq = x
    for layer in (0, 1):
        outs = []
        for suf in ('', '_reverse'):
            h = x.new_zeros(b, hid)
            steps = range(TICKS) if suf == '' else range(TICKS - 1, -1, -1)
            hs = [None] * TICKS
            for t in steps:
                h = _cell(p, f'encoder.lstm.{{}}_l{layer}{suf}', seq[:, t], h)
                hs[t] = h
            finals.append(h)
            outs.append(torch.stack(hs, 1))
        seq = torch.cat(outs, 2)                                        # (B,24,2H)
        if layer == 0:
            seq = _keep(seq, enc_mask)
    hcat = torch.cat(finals, 1)          # [l0 fwd, l0 rev, l1 fwd, l1 rev] (encoder.py:116-117)

    def head(name):
        t = selu(F.linear(hcat, p[f'encoder.{name}.0.weight'], p[f'encoder.{name}.0.bias']))
        return F.linear(t, p[f'encoder.{name}.2.weight'], p[f'encoder.{name}.2.bias'])
    return head('linear_mean'), head('linear_log_std')


def _split_hidden(flat, hid):
    """(B, 2H) -> [layer0 (B,H), layer1 (B,H)]: view(B,2,H).transpose(0,1)  (decoder.py:402-404)."""
    return [flat[:, :hid], flat[:, hid:2 * hid]]


def decode(p, z, score, teacher_forced, beat_mask=None, tick_masks=None):
    """-> (weights (B,24,V) >= 0, samples (B,1,24) int64).

    teacher_forced: feed score[:, t] back (decoder.py:492-495); else argmax of
    the ReLU-ed logits (decoder.py:506-507; lowest index on ties here)."""
    b = z.shape[0]
    hid = p['decoder.rnn_beat.weight_hh_l0'].shape[1]
    h = _split_hidden(selu(F.linear(z, p['decoder.z_to_beat_rnn_input.0.weight'],
                                    p['decoder.z_to_beat_rnn_input.0.bias'])), hid)
    b0 = p['decoder.b_0'].reshape(1, 1).expand(b, 1)
    beat_out = []
    for i in range(BEATS):
        h[0] = _cell(p, 'decoder.rnn_beat.{}_l0', b0, h[0])
        mid = _keep(h[0], None if beat_mask is None else beat_mask[:, i])
        h[1] = _cell(p, 'decoder.rnn_beat.{}_l1', mid, h[1])
        beat_out.append(h[1])
    prev = p['decoder.x_0'].reshape(1, -1).expand(b, -1)
    weights, samples = [], []
    for i in range(BEATS):
        bo = beat_out[i]
        th = _split_hidden(selu(F.linear(bo, p['decoder.beat_emb_to_tick_rnn_hidden.0.weight'],
                                         p['decoder.beat_emb_to_tick_rnn_hidden.0.bias'])), hid)
        bemb = selu(F.linear(bo, p['decoder.beat_emb_to_tick_rnn_input.0.weight'],
                             p['decoder.beat_emb_to_tick_rnn_input.0.bias']))
        for j in range(TICKS_PER_BEAT):
            t = i * TICKS_PER_BEAT + j
            inp = torch.cat((prev, bemb), 1)
            th[0] = _cell(p, 'decoder.rnn_tick.{}_l0', inp, th[0])
            mid = _keep(th[0], None if tick_masks is None else tick_masks[:, t])
            th[1] = _cell(p, 'decoder.rnn_tick.{}_l1', mid, th[1])
            probs = F.relu(F.linear(th[1], p['decoder.tick_emb_to_note_emb.0.weight'],
                                    p['decoder.tick_emb_to_note_emb.0.bias']))
            idx = score[:, t] if teacher_forced else probs.detach().argmax(1)
            prev = p['decoder.note_embedding_layer.weight'][idx]
            weights.append(probs)
            samples.append(idx)
    return torch.stack(weights, 1), torch.stack(samples, 1)[:, None, :]


def forward(p, score, eps, teacher_forced, masks=None):
    """-> (weights, samples, mu, sigma, z)   (measure_vae.py:97-131)."""
    masks = masks or {}
    mu, log_std = encode(p, score, masks.get('enc'))
    sigma = torch.exp(log_std)
    z = mu + eps * sigma
    weights, samples = decode(p, z, score, teacher_forced, masks.get('beat'), masks.get('tick'))
    return weights, samples, mu, sigma, z
