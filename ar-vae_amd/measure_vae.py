"""MeasureVAE (GRU encoder + hierarchical GRU decoder over 24-tick measures) on the HIP kernels.

Same class names, constructor arguments, forward contract and state_dict keys as the reference
(measurevae/encoder.py:8-124, measurevae/decoder.py:7-51,309-525, measurevae/measure_vae.py:11-131):
    MeasureVAE.forward(score, metadata, train) -> (weights (B,24,V), samples (B,1,24), z_dist, prior_dist, z_tilde, z_prior)
Every GRU layer runs as whole-sequence launches (csrc/gru_seq.hip: all time steps of a layer, both directions, in one
kernel forward and one backward; input projections and weight gradients as whole-sequence GEMMs); the free-running
decoder gets its tokens from one more launch.  Hidden sizes other than 32 / 64 / 128 (or ARVAE_GRU_STEPWISE=1) fall back
to one launch per GRU cell and time step (csrc/sequence.hip + the dense kernels).  autograd only chains the launches.
"""
import os
from collections import deque

import torch
from torch import distributions, nn

from . import ops
from .model import LayerStack, Model, ParamLayer
from .ops import ACT_NONE, ACT_RELU, ACT_SELU, Link


class GRUParams(nn.Module):
    """Parameters of an nn.GRU under torch's names (weight_ih_l0, weight_hh_l0_reverse, ...)."""

    def __init__(self, input_size, hidden_size, num_layers, bidirectional):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        self.num_directions = 2 if bidirectional else 1
        bound = 1.0 / (hidden_size ** 0.5)
        for layer in range(num_layers):
            in_l = input_size if layer == 0 else hidden_size * self.num_directions
            for suf in ([''] + (['_reverse'] if bidirectional else [])):
                for name, shape in ((f'weight_ih_l{layer}{suf}', (3 * hidden_size, in_l)),
                                    (f'weight_hh_l{layer}{suf}', (3 * hidden_size, hidden_size)),
                                    (f'bias_ih_l{layer}{suf}', (3 * hidden_size,)),
                                    (f'bias_hh_l{layer}{suf}', (3 * hidden_size,))):
                    p = nn.Parameter(torch.empty(*shape))
                    nn.init.uniform_(p, -bound, bound)
                    self.register_parameter(name, p)

    def cell(self, layer, suffix=''):
        g = lambda n: getattr(self, f'{n}_l{layer}{suffix}')
        return g('weight_ih'), g('weight_hh'), g('bias_ih'), g('bias_hh')


class Embedding(nn.Module):
    def __init__(self, num, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(num, dim))


def _dense_layer(fin, fout):
    return LayerStack([(0, ParamLayer((fout, fin), fout, fin))])


def _lin(x, layer, act):
    w = layer.weight
    return ops.dense(x, w, layer.bias, Link.dense(w.shape[1], w.shape[0]), act)


def _gru_step(x_proj, h, w_hh, b_hh):
    """one GRU time step given the input projection gi = W_ih x + b_ih."""
    gh = ops.dense(h, w_hh, b_hh, Link.dense(w_hh.shape[1], w_hh.shape[0]), ACT_NONE)
    return ops.gru_gates(x_proj, gh, h)


def _use_sequence_kernels(hidden):
    """whole-sequence GRU launches (csrc/gru_seq.hip) when the hidden size is built; ARVAE_GRU_STEPWISE=1 keeps the
    one-launch-per-time-step path (A/B measurements, and the only path for other hidden sizes)."""
    return os.environ.get('ARVAE_GRU_STEPWISE', '0') != '1' and ops.gru_sequence_supported(hidden)


class Encoder(Model):
    def __init__(self, note_embedding_dim, rnn_hidden_size, num_layers, num_notes, dropout, bidirectional, z_dim,
                 rnn_class=nn.GRU):
        super().__init__()
        if not bidirectional or num_layers != 2:
            raise NotImplementedError('the HIP encoder implements the reference configuration: 2-layer bidirectional GRU')
        self.bidirectional, self.num_directions = bidirectional, 2
        self.note_embedding_dim, self.num_layers = note_embedding_dim, num_layers
        self.rnn_hidden_size, self.z_dim, self.dropout, self.rnn_class = rnn_hidden_size, z_dim, dropout, rnn_class
        self.num_notes = num_notes
        self.lstm = GRUParams(note_embedding_dim, rnn_hidden_size, num_layers, bidirectional)
        self.note_embedding_layer = Embedding(num_notes, note_embedding_dim)
        feat = rnn_hidden_size * 2 * num_layers
        self.linear_mean = LayerStack([(0, ParamLayer((rnn_hidden_size * 2, feat), rnn_hidden_size * 2, feat)),
                                       (2, ParamLayer((z_dim, rnn_hidden_size * 2), z_dim, rnn_hidden_size * 2))])
        self.linear_log_std = LayerStack([(0, ParamLayer((rnn_hidden_size * 2, feat), rnn_hidden_size * 2, feat)),
                                          (2, ParamLayer((z_dim, rnn_hidden_size * 2), z_dim, rnn_hidden_size * 2))])
        self.xavier_initialization()
        self._mask_queue = deque()
        self._eps_queue = deque()

    def __repr__(self):
        return (f'Encoder({self.note_embedding_dim},{self.rnn_class},{self.num_layers},{self.rnn_hidden_size},'
                f'{self.dropout},{self.bidirectional},{self.z_dim},)')

    def push_dropout_mask(self, mask):
        """explicit keep-mask (24, B, 2H) uint8 for the layer-0 outputs (parity runs)."""
        self._mask_queue.append(mask)

    def embed_forward(self, score_tensor):
        return ops.embed(score_tensor, self.note_embedding_layer.weight)

    def _layer(self, seq, layer):
        """seq: list of T tensors (B, in) -> (list of T (B, 2H), [final fwd, final rev])."""
        steps, b = len(seq), seq[0].shape[0]
        x_all = torch.cat(seq, 0)                                    # (T*B, in), time-major rows
        outs, finals = [], []
        for suffix in ('', '_reverse'):
            w_ih, w_hh, b_ih, b_hh = self.lstm.cell(layer, suffix)
            gi_all = ops.dense(x_all, w_ih, b_ih, Link.dense(w_ih.shape[1], w_ih.shape[0]), ACT_NONE)
            gi = torch.unbind(gi_all.view(steps, b, -1), 0)
            h = torch.zeros(b, self.rnn_hidden_size, device=x_all.device)
            hs = [None] * steps
            for t in (range(steps) if suffix == '' else range(steps - 1, -1, -1)):
                h = _gru_step(gi[t], h, w_hh, b_hh)
                hs[t] = h
            outs.append(hs)
            finals.append(h)
        return [ops.concat_cols(f, r) for f, r in zip(*outs)], finals

    def push_noise(self, eps):
        """use `eps` (B, z_dim) for the next reparameterised sample instead of drawing it."""
        self._eps_queue.append(eps)

    static_eps = None                                        # tests: fixed noise buffer for eager-vs-graph comparisons

    def forward(self, score_tensor):
        """score (B, 24) int64 -> Normal(mu, exp(log_std)); the reparameterised sample computed by the same
        fused kernel travels with the distribution object (z_dist._arvae_sample)."""
        if ops.checks_enabled():                             # the reference's NaN scan of the weights (encoder.py:101-106)
            ops.check_finite([p for n, p in self.named_parameters() if 'weight' in n], 'Encoder')
        mu, log_std = self.encode_params(score_tensor)
        if self._eps_queue:
            eps = self._eps_queue.popleft().to(mu.device, torch.float32).contiguous()
        elif self.static_eps is not None:                    # a device buffer read at run time (visible to a captured graph)
            eps = self.static_eps
        else:
            eps = ops.normal_noise(mu.shape, mu.device)
        sigma, z = ops.latent_head(mu, log_std, eps)
        z_dist = distributions.Normal(loc=mu, scale=sigma, validate_args=False)
        z_dist._arvae_sample = z
        return z_dist

    def _layer_sequence(self, x_all, steps, b, layer):
        """x_all (T*B, in), time-major rows -> (T, B, 2H): both directions of one layer in one sequence launch."""
        wf, whf, bf, bhf = self.lstm.cell(layer, '')
        wr, whr, br, bhr = self.lstm.cell(layer, '_reverse')
        gi_all = ops.dense_pair(x_all, wf, bf, wr, br)           # (T*B, 2 * 3H) when the two projections are adjacent in memory
        if gi_all is not None:
            return ops.gru_sequence(steps, [(None, whf, bhf, None, False), (None, whr, bhr, None, True)],
                                    merged_gi=gi_all.view(steps, b, -1))
        dirs = []
        for suffix in ('', '_reverse'):
            w_ih, w_hh, b_ih, b_hh = self.lstm.cell(layer, suffix)
            gi = ops.dense(x_all, w_ih, b_ih, Link.dense(w_ih.shape[1], w_ih.shape[0]), ACT_NONE)
            dirs.append((gi.view(steps, b, -1), w_hh, b_hh, None, suffix != ''))
        return ops.gru_sequence(steps, dirs)

    def _first_layer_by_lookup(self, score_tensor, steps, b):
        """Layer 0 sees embedded tokens, so its input projection takes only `num_notes` distinct values: P = table W_ih^T + b_ih
        (both directions side by side, num_notes x 6H: one small product) and gi[t, b] = P[score[b, t]] (a lookup, no
        whole-sequence GEMM); backward the per-position gradients are summed per token into dP (a segment sum) and the
        weight / bias / embedding gradients follow from dP on num_notes rows instead of T*B.  The same algebra as
        encoder.py:111-114 (embedding, then nn.GRU's W_ih x + b_ih), re-associated.  -> (out, finals) or (None, None) when the
        two directions' projections are not adjacent in memory (ops.dense_pair)."""
        # the lookup's backward is the wide-row segment sum (csrc/sequence.hip, embed_bwd_wide_kernel): its per-token accumulators
        # and this batch's positions must fit 64 KB of LDS -- the decoder's lookup and FusedMeasureVAE.fits() test the same bound
        if self.num_notes * 1024 + (steps * b + 15) // 16 * 8 > 65536:
            return None, None
        wf, whf, bf, bhf = self.lstm.cell(0, '')
        wr, whr, br, bhr = self.lstm.cell(0, '_reverse')
        ptab = ops.dense_pair(self.note_embedding_layer.weight, wf, bf, wr, br)            # (num_notes, 2 * 3H)
        if ptab is None:
            return None, None
        gi_all = ops.embed(score_tensor, ptab, time_major=True)                            # (T, B, 2 * 3H)
        return ops.gru_sequence(steps, [(None, whf, bhf, None, False), (None, whr, bhr, None, True)], merged_gi=gi_all)

    def encode_params(self, score_tensor):
        """-> (mu, log_std)"""
        b, steps = score_tensor.shape
        hid = self.rnn_hidden_size
        # (T, B, E); with the lookup path of layer 0 nothing reads it and autograd drops the launch's backward
        emb = None if _use_sequence_kernels(hid) else ops.embed(score_tensor, self.note_embedding_layer.weight, time_major=True)
        dropping = self.training and self.dropout > 0
        mask = None
        if dropping:
            dev = score_tensor.device
            mask = self._mask_queue.popleft().to(dev) if self._mask_queue else ops.keep_mask((steps, b, 2 * hid), self.dropout, dev)
        if _use_sequence_kernels(hid):
            out0, fin0 = self._first_layer_by_lookup(score_tensor, steps, b)
            if out0 is None:
                emb = ops.embed(score_tensor, self.note_embedding_layer.weight, time_major=True)
                out0, fin0 = self._layer_sequence(emb.view(steps * b, -1), steps, b, 0)
            mid = out0.view(steps * b, 2 * hid)
            if dropping:
                mid = ops.dropout_mask(mid, mask.contiguous().view(steps * b, 2 * hid), self.dropout)
            _, fin1 = self._layer_sequence(mid, steps, b, 1)
            # h_n of nn.GRU: (layer 0 fwd, layer 0 rev, layer 1 fwd, layer 1 rev), each direction's LAST processed step
            hidden = ops.concat_cols(fin0, fin1)
        else:
            seq = list(torch.unbind(emb, 0))
            seq, finals0 = self._layer(seq, 0)
            if dropping:
                seq = [ops.dropout_mask(s, m, self.dropout) for s, m in zip(seq, torch.unbind(mask.contiguous(), 0))]
            _, finals1 = self._layer(seq, 1)
            hidden = ops.concat_cols(ops.concat_cols(finals0[0], finals0[1]), ops.concat_cols(finals1[0], finals1[1]))
        la, lb = self.linear_mean[0], self.linear_log_std[0]
        both = ops.dense_pair(hidden, la.weight, la.bias, lb.weight, lb.bias, ACT_SELU)      # (B, 2 * 2H) or None
        if both is not None:
            h_mu, h_ls = ops.split_cols(both, la.weight.shape[0])
        else:
            h_mu, h_ls = _lin(hidden, la, ACT_SELU), _lin(hidden, lb, ACT_SELU)
        mu = _lin(h_mu, self.linear_mean[2], ACT_NONE)
        log_std = _lin(h_ls, self.linear_log_std[2], ACT_NONE)
        return mu, log_std


class Decoder(nn.Module):
    def __init__(self, note_embedding_dim, num_notes, z_dim):
        super().__init__()
        self.name = 'DecoderABC'
        self.num_notes, self.note_embedding_dim, self.z_dim = num_notes, note_embedding_dim, z_dim

    def xavier_initialization(self):
        for name, param in self.named_parameters():
            if 'weight' in name:
                nn.init.xavier_normal_(param)


class HierarchicalDecoder(Decoder):
    def __init__(self, note_embedding_dim, num_notes, z_dim, num_layers, rnn_hidden_size, dropout, rnn_class=nn.GRU):
        super().__init__(note_embedding_dim, num_notes, z_dim)
        if num_layers != 2:
            raise NotImplementedError('the HIP decoder implements the reference configuration: 2-layer GRUs')
        self.name = 'HierarchicalDecoder'
        self.rnn_class, self.num_layers, self.rnn_hidden_size, self.dropout = rnn_class, num_layers, rnn_hidden_size, dropout
        h = rnn_hidden_size
        self.b_0 = nn.Parameter(torch.zeros(1))
        self.x_0 = nn.Parameter(torch.zeros(note_embedding_dim))
        self.note_embedding_layer = Embedding(num_notes, note_embedding_dim)
        self.z_to_beat_rnn_input = _dense_layer(z_dim, h * num_layers)
        self.beat_rnn_input_dim = 1
        self.rnn_beat = GRUParams(1, h, num_layers, False)
        self.beat_emb_to_tick_rnn_hidden = _dense_layer(h, h * num_layers)
        self.beat_emb_to_tick_rnn_input = _dense_layer(h, h)
        self.rnn_tick = GRUParams(note_embedding_dim + h, h, num_layers, False)
        self.tick_emb_to_note_emb = _dense_layer(h, num_notes)
        self.use_teacher_forcing = True
        self.teacher_forcing_prob = 0.5
        self.sampling = 'argmax'
        self.xavier_initialization()
        self._mask_queue = deque()

    def __repr__(self):
        return f'{self.name}{self.note_embedding_dim},{self.rnn_class},{self.num_layers},{self.rnn_hidden_size},{self.dropout},)'

    def push_dropout_masks(self, beat_mask, tick_mask):
        """explicit keep-masks (4, B, H) and (24, B, H) uint8 for the layer-0 hidden states."""
        self._mask_queue.append((beat_mask, tick_mask))

    def hidden_init(self, inp, rnn_type):
        """(B, feats) -> [layer-0 hidden, layer-1 hidden]  (view(B, 2, H).transpose(0, 1), decoder.py:388-406)."""
        if rnn_type == 'beat':
            flat = _lin(inp, self.z_to_beat_rnn_input[0], ACT_SELU)
        elif rnn_type == 'tick':
            flat = _lin(inp, self.beat_emb_to_tick_rnn_hidden[0], ACT_SELU)
        else:
            raise ValueError
        return list(ops.split_cols(flat, self.rnn_hidden_size))

    def _two_layer_step(self, rnn, gi0, h, mask):
        w_hh0, b_hh0 = rnn.cell(0)[1], rnn.cell(0)[3]
        h0 = _gru_step(gi0, h[0], w_hh0, b_hh0)
        mid = ops.dropout_mask(h0, mask, self.dropout) if mask is not None else h0
        w_ih1, w_hh1, b_ih1, b_hh1 = rnn.cell(1)
        gi1 = ops.dense(mid, w_ih1, b_ih1, Link.dense(w_ih1.shape[1], w_ih1.shape[0]), ACT_NONE)
        h1 = _gru_step(gi1, h[1], w_hh1, b_hh1)
        return [h0, h1]

    def forward(self, z, score_tensor, train):
        if z.size(1) != self.z_dim or z.size(0) != score_tensor.size(0):
            raise AssertionError('latent / score shape mismatch')
        if ops.checks_enabled():                             # the reference's NaN scan of the weights (decoder.py:420-425)
            ops.check_finite([p for n, p in self.named_parameters() if 'weight' in n], 'Decoder')
        weights, samples = self._forward(z, score_tensor, train)
        if ops.checks_enabled():                             # Decoder.check_index on every fed-back note (decoder.py:30-41)
            ops.check_index(samples, self.num_notes)
        return weights, samples

    def _forward(self, z, score_tensor, train):
        if self.use_teacher_forcing and train:
            teacher_forced = torch.rand(1).item() < self.teacher_forcing_prob       # host coin (decoder.py:427-428)
        else:
            teacher_forced = False
        if train and self.sampling != 'argmax':
            raise NotImplementedError('only argmax sampling is implemented')
        b = z.size(0)
        masks = (None, None)
        if self.training and self.dropout > 0:
            if self._mask_queue:
                masks = tuple(m.to(z.device) for m in self._mask_queue.popleft())
            else:
                h = self.rnn_hidden_size                    # one draw for the beat (4 steps) and the tick (24 steps) masks
                both = ops.keep_mask((4 + 24, b, h), self.dropout, z.device)
                masks = (both[:4], both[4:])
        if _use_sequence_kernels(self.rnn_hidden_size):
            beat_out = self.beat_rnn_sequence(z, 4, masks[0])
            return self.tick_rnn_sequence(score_tensor, beat_out, 6, teacher_forced, masks[1])
        beat_out = self.forward_beat_rnn(z, 4, masks[0])
        return self.forward_tick_rnn(score_tensor, beat_out, 6, teacher_forced, 'argmax', masks[1])

    # ---- whole-sequence path ------------------------------------------------------------------------------------
    def _two_layer_sequence(self, rnn, steps, gi0, h0, mask):
        """2-layer unidirectional GRU over `steps`: gi0 (T, R, 3H) or (R, 3H); h0 = [layer-0, layer-1] initial states;
        mask (T*R, H) keep-mask on the layer-0 outputs (nn.GRU's inter-layer dropout).  -> layer-1 outputs (T, R, H)"""
        hid = self.rnn_hidden_size
        out0, _ = ops.gru_sequence(steps, [(gi0, rnn.cell(0)[1], rnn.cell(0)[3], h0[0], False)], finals=False)
        rows = out0.shape[1]
        mid = out0.view(steps * rows, hid)
        if mask is not None:
            mid = ops.dropout_mask(mid, mask, self.dropout)
        w_ih1, w_hh1, b_ih1, b_hh1 = rnn.cell(1)
        gi1 = ops.dense(mid, w_ih1, b_ih1, Link.dense(w_ih1.shape[1], w_ih1.shape[0]), ACT_NONE).view(steps, rows, -1)
        return ops.gru_sequence(steps, [(gi1, w_hh1, b_hh1, h0[1], False)], finals=False)[0]

    def beat_rnn_sequence(self, z, seq_len, mask=None):
        """-> (4, B, H) beat embeddings (decoder.py:436-457)"""
        b = z.size(0)
        h = self.hidden_init(z, 'beat')
        w_ih0, _, b_ih0, _ = self.rnn_beat.cell(0)
        x0 = ops.broadcast_rows(self.b_0, b)
        gi0 = ops.dense(x0, w_ih0, b_ih0, Link.dense(1, w_ih0.shape[0]), ACT_NONE)           # the same input every beat
        m = None if mask is None else mask.contiguous().view(seq_len * b, -1)
        return self._two_layer_sequence(self.rnn_beat, seq_len, gi0, h, m)

    def tick_rnn_sequence(self, score_tensor, beat_out, tick_seq_len, teacher_forced, mask=None):
        """The tick RNN restarts from a beat-dependent hidden state at every beat (decoder.py:459-525), so given the
        fed-back tokens the four beats are independent 6-step sequences: they run as ONE sequence launch per layer over
        4*B rows (row = beat*B + b).  With teacher forcing the fed-back tokens are the score; otherwise they come from
        the free-running pass `_free_running_tokens` (argmax is not differentiated, decoder.py:506-516), and the same
        graph is then evaluated on those tokens."""
        nb, b = beat_out.shape[0], beat_out.shape[1]
        hid, steps = self.rnn_hidden_size, tick_seq_len
        ticks = nb * steps
        bo = beat_out.view(nb * b, hid)
        la, lb = self.beat_emb_to_tick_rnn_hidden[0], self.beat_emb_to_tick_rnn_input[0]
        both = ops.dense_pair(bo, la.weight, la.bias, lb.weight, lb.bias, ACT_SELU)           # (4B, 2H + H) or None
        if both is not None:
            flat, beat_emb = ops.split_cols(both, la.weight.shape[0])
            h0 = list(ops.split_cols(flat, hid))
        else:
            h0 = self.hidden_init(bo, 'tick')
            beat_emb = _lin(bo, lb, ACT_SELU)                                                  # (4B, H)
        m = None
        if mask is not None:                                                                   # (24, B, H) -> rows (j, beat, b)
            m = mask.view(nb, steps, b, hid).transpose(0, 1).contiguous().view(steps * nb * b, hid)
        if self.use_teacher_forcing and teacher_forced:
            tokens = score_tensor
        else:
            with torch.no_grad():
                tokens = self._free_running_tokens(beat_out.detach(), h0, beat_emb, mask)
        w_ih0, _, b_ih0, _ = self.rnn_tick.cell(0)
        if (3 * hid) % 4 == 0 and (self.num_notes + 1) * 1024 + (steps * nb * b + 15) // 16 * 8 <= 65536:    # (the segment sum's LDS)
            # layer-0 input projection by lookup: W_ih0 applied once to the vocabulary's embeddings, x_0 and the beat embeddings
            # (num_notes + 1 + 4B rows), each tick's row gathered and added (ops.tick_input_projection)
            gi0 = ops.tick_input_projection(self.note_embedding_layer.weight, self.x_0, beat_emb, w_ih0, b_ih0, tokens, nb, steps)
        else:
            emb_prev = ops.embed(tokens[:, :ticks - 1].contiguous(), self.note_embedding_layer.weight, time_major=True)
            prev = torch.cat((ops.broadcast_rows(self.x_0, b)[None], emb_prev), 0)             # (24, B, E), tick-major
            prev = prev.view(nb, steps, b, -1).transpose(0, 1).reshape(steps * nb * b, -1)     # rows (j, beat, b)
            inp = ops.concat_cols(prev, beat_emb[None].expand(steps, -1, -1).reshape(steps * nb * b, hid))
            gi0 = ops.dense(inp, w_ih0, b_ih0, Link.dense(w_ih0.shape[1], w_ih0.shape[0]), ACT_NONE).view(steps, nb * b, -1)
        out1 = self._two_layer_sequence(self.rnn_tick, steps, gi0, h0, m)                      # (6, 4B, H)
        probs = _lin(out1.view(steps * nb * b, hid), self.tick_emb_to_note_emb[0], ACT_RELU)
        weights = probs.view(steps, nb, b, -1).permute(2, 1, 0, 3).reshape(b, ticks, -1)       # tick = 6*beat + j
        return weights, tokens[:, None, :]

    def _free_running_tokens(self, beat_out, h0, beat_emb, mask):
        """argmax-feedback pass (no autograd): the tokens the decoder feeds itself, int64 (B, 24)."""
        nb, b = beat_out.shape[0], beat_out.shape[1]
        hid = self.rnn_hidden_size
        w_ih0, w_hh0, b_ih0, b_hh0 = self.rnn_tick.cell(0)
        if os.environ.get('ARVAE_TICK_STEPWISE', '0') != '1' and ops.tick_free_run_supported(hid, self.num_notes):
            # one launch (csrc/gru_seq.hip tick_free_run_kernel).  W_ih0 acts on [previous-note embedding | beat
            # embedding]: the beat half is applied once per beat, the note half once per vocabulary entry (+ x_0).
            emb_dim = self.note_embedding_dim
            w_note, w_beat = w_ih0.detach()[:, :emb_dim].contiguous(), w_ih0.detach()[:, emb_dim:].contiguous()
            gib = ops.dense(beat_emb.detach(), w_beat, b_ih0.detach(), Link.dense(hid, 3 * hid), ACT_NONE)
            table = torch.cat((self.note_embedding_layer.weight.detach(), self.x_0.detach()[None]), 0)
            ptab = ops.dense(table, w_note, None, Link.dense(emb_dim, 3 * hid), ACT_NONE)
            w_ih1, w_hh1, b_ih1, b_hh1 = self.rnn_tick.cell(1)
            out = self.tick_emb_to_note_emb[0]
            weights = tuple(t.detach() for t in (w_hh0, b_hh0, w_ih1, b_ih1, w_hh1, b_hh1, out.weight, out.bias))
            return ops.tick_free_run(weights, h0[0].detach(), h0[1].detach(), gib, ptab,
                                     None if mask is None else mask.contiguous(), 1.0 / (1.0 - self.dropout), b, nb, 6)
        prev = self.x_0.detach()[None].expand(b, -1).contiguous()
        tokens = []
        for i in range(nb):
            h = [h0[0].detach()[i * b:(i + 1) * b], h0[1].detach()[i * b:(i + 1) * b]]
            be = beat_emb.detach()[i * b:(i + 1) * b]
            for j in range(6):
                t = i * 6 + j
                gi0 = ops.dense(ops.concat_cols(prev, be), w_ih0, b_ih0, Link.dense(w_ih0.shape[1], w_ih0.shape[0]), ACT_NONE)
                h = self._two_layer_step(self.rnn_tick, gi0, h, None if mask is None else mask[t].contiguous())
                idx = ops.row_argmax(_lin(h[1], self.tick_emb_to_note_emb[0], ACT_RELU))
                prev = ops.embed(idx.view(b, 1), self.note_embedding_layer.weight).view(b, -1)
                tokens.append(idx)
        return torch.stack(tokens, 1)

    def forward_beat_rnn(self, z, seq_len, mask=None):
        b = z.size(0)
        h = self.hidden_init(z, 'beat')
        w_ih0, _, b_ih0, _ = self.rnn_beat.cell(0)
        x0 = ops.broadcast_rows(self.b_0, b)                               # constant input b_0 for every beat
        gi0 = ops.dense(x0, w_ih0, b_ih0, Link.dense(1, w_ih0.shape[0]), ACT_NONE)
        out = []
        for i in range(seq_len):
            h = self._two_layer_step(self.rnn_beat, gi0, h, None if mask is None else mask[i].contiguous())
            out.append(h[1])
        return out

    def forward_tick_rnn(self, score_tensor, beat_rnn_out, tick_seq_len, teacher_forced, sampling, mask=None):
        b = score_tensor.size(0)
        w_ih0, _, b_ih0, _ = self.rnn_tick.cell(0)
        prev = ops.broadcast_rows(self.x_0, b)                             # learned start embedding
        weights, samples = [], []
        for i, bo in enumerate(beat_rnn_out):
            h = self.hidden_init(bo, 'tick')                               # hidden is reset per beat
            beat_emb = _lin(bo, self.beat_emb_to_tick_rnn_input[0], ACT_SELU)
            for j in range(tick_seq_len):
                t = i * tick_seq_len + j
                inp = ops.concat_cols(prev, beat_emb)
                gi0 = ops.dense(inp, w_ih0, b_ih0, Link.dense(w_ih0.shape[1], w_ih0.shape[0]), ACT_NONE)
                h = self._two_layer_step(self.rnn_tick, gi0, h, None if mask is None else mask[t].contiguous())
                probs = _lin(h[1], self.tick_emb_to_note_emb[0], ACT_RELU)
                if self.use_teacher_forcing and teacher_forced:
                    idx = score_tensor[:, t].contiguous()
                else:
                    idx = ops.row_argmax(probs.detach())
                prev = ops.embed(idx.view(b, 1), self.note_embedding_layer.weight).view(b, -1)   # carries across beats
                weights.append(probs)
                samples.append(idx)
        return torch.stack(weights, 1), torch.stack(samples, 1)[:, None, :]


_PRIOR_CONSTANTS = {}


def _standard_normal_like(mu):
    """N(0, I) of mu's shape without a launch: loc / scale are expanded views of two cached device scalars (the reference
    fills two tensors per step: measure_vae.py:118-121); `_arvae_standard` lets compute_kld_loss use the closed form."""
    key = (mu.device, mu.dtype)
    if key not in _PRIOR_CONSTANTS:
        _PRIOR_CONSTANTS[key] = (torch.zeros((), device=mu.device, dtype=mu.dtype), torch.ones((), device=mu.device, dtype=mu.dtype))
    zero, one = _PRIOR_CONSTANTS[key]
    prior = distributions.Normal(loc=zero.expand(mu.shape), scale=one.expand(mu.shape), validate_args=False)
    prior._arvae_standard = True
    return prior


class MeasureVAE(Model):
    def __init__(self, dataset, note_embedding_dim=10, metadata_embedding_dim=2, num_encoder_layers=2,
                 encoder_hidden_size=512, encoder_dropout_prob=0.5, latent_space_dim=256, num_decoder_layers=2,
                 decoder_hidden_size=512, decoder_dropout_prob=0.5, has_metadata=True, dataset_type='folk'):
        super().__init__()
        self.dataset_type = dataset_type
        self.num_beats_per_measure, self.num_ticks_per_measure = 4, 24
        self.num_ticks_per_beat = 6
        self.dataset = repr(dataset)
        self.note_embedding_dim, self.metadata_embedding_dim = note_embedding_dim, metadata_embedding_dim
        self.num_encoder_layers, self.encoder_hidden_size = num_encoder_layers, encoder_hidden_size
        self.encoder_dropout_prob, self.latent_space_dim = encoder_dropout_prob, latent_space_dim
        self.num_decoder_layers, self.decoder_hidden_size = num_decoder_layers, decoder_hidden_size
        self.decoder_dropout_prob, self.has_metadata = decoder_dropout_prob, has_metadata
        self.num_notes = len(dataset.note2index_dicts)
        self.encoder = Encoder(note_embedding_dim, encoder_hidden_size, num_encoder_layers, self.num_notes,
                               encoder_dropout_prob, True, latent_space_dim, nn.GRU)
        self.decoder = HierarchicalDecoder(note_embedding_dim, self.num_notes, latent_space_dim, num_decoder_layers,
                                           decoder_hidden_size, decoder_dropout_prob, nn.GRU)
        self.update_filepath()

    def __repr__(self):
        return self.dataset_type + '_MeasureVAE' + self.trainer_config

    def arena_parameters(self):
        """every parameter once, in the order the trainer's flat arena should hold them: per encoder GRU layer the input
        projections of both directions side by side (weight_ih_l*, weight_ih_l*_reverse, then the two bias_ih), so that ONE
        whole-sequence GEMM applies both (Encoder._layer_sequence); everything else in module order.  state_dict keys and
        shapes are untouched (measurevae/encoder.py:27-34)."""
        gru = self.encoder.lstm
        first = []
        for layer in range(gru.num_layers):
            g = lambda n, suf: getattr(gru, f'{n}_l{layer}{suf}')
            first += [g('weight_ih', ''), g('weight_ih', '_reverse'), g('bias_ih', ''), g('bias_ih', '_reverse')]
        # the two heads' first layers read the same hidden vector; the tick RNN's initial state and its beat-embedding input are
        # two layers on the same beat output: each pair as one product (ops.dense_pair)
        enc, dec = self.encoder, self.decoder
        for la, lb in ((enc.linear_mean[0], enc.linear_log_std[0]), (dec.beat_emb_to_tick_rnn_hidden[0], dec.beat_emb_to_tick_rnn_input[0])):
            first += [la.weight, lb.weight, la.bias, lb.bias]
        taken = {id(p) for p in first}
        return first + [p for p in self.parameters() if id(p) not in taken]

    def push_noise(self, eps):
        self.encoder.push_noise(eps)

    def forward(self, measure_score_tensor, measure_metadata_tensor=None, train=True, *, need_prior_sample=True):
        """-> (weights, samples, z_dist, prior_dist, z_tilde, z_prior) as the reference (measure_vae.py:97-131).
        need_prior_sample=False (what this build's trainer passes: it never looks at z_prior, nor does the reference's):
        z_prior is None and the launch that would draw it is skipped."""
        if measure_score_tensor.size(1) != self.num_ticks_per_measure:
            raise AssertionError('a measure has 24 ticks')
        z_dist = self.encoder(measure_score_tensor)
        mu, z_tilde = z_dist.loc, z_dist._arvae_sample
        prior_dist = _standard_normal_like(mu)
        z_prior = ops.normal_noise(mu.shape, mu.device) if need_prior_sample else None     # the reference's second, unused draw
        weights, samples = self.decoder(z=z_tilde, score_tensor=measure_score_tensor, train=train)
        return weights, samples, z_dist, prior_dist, z_tilde, z_prior
