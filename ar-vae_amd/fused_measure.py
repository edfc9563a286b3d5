"""Whole-model training step for MeasureVAE: forward (+ all loss terms) and backward are ONE C call each
(arvae_measure_vae_forward / arvae_measure_vae_backward, csrc/plan_measure.hip), so the host does no per-layer work.

Used by MeasureVAETrainer.loss_and_acc_for_batch when the model's parameters live in the trainer's flat Adam arena in
MeasureVAE.arena_parameters() order.  Autograd sees a single node: its backward accumulates every parameter gradient directly
into the gradient arena and returns nothing for the parameters.  The per-layer Python path (measure_vae.py) stays for the
reference's class API (MeasureVAE.forward returns the weights), data parallelism and configurations the executor is not built
for; tests/test_measure_executor.py holds the two paths against each other.
"""
import ctypes
import weakref

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import _lib, ops
from ._lib import MeasureTables, MeasureVaeDesc

LOSS, RECON, DIST, REG, ACC, KL, NSCALARS = 0, 1, 2, 3, 4, 5, 8


class FusedMeasureVAE:
    """Descriptor + workspace cache binding a MeasureVAE to a FlatAdam arena."""

    def __init__(self, model, optimizer, reg_dims, beta, gamma, delta):
        self.model, self.optimizer = model, optimizer
        self.reg_dims = tuple(int(d) for d in reg_dims)
        self.beta, self.gamma, self.delta = float(beta), float(gamma), float(delta)
        self._desc = None
        self._arena_ptr = None
        self._ws = {}
        self._ws_owner = None
        self._tables = None

    # -- can the executor run this model? -------------------------------------------------------------------------
    @staticmethod
    def supports(model, optimizer, reg_dims):
        """None when the executor can run `model` from `optimizer`'s arena, else the reason it cannot (str)."""
        enc, dec = model.encoder, model.decoder
        if not (ops.gru_sequence_supported(enc.rnn_hidden_size) and ops.gru_sequence_supported(dec.rnn_hidden_size)):
            return 'hidden size not built as a sequence kernel'
        if dec.sampling != 'argmax':
            return 'only argmax sampling'
        if not ops.tick_free_run_supported(dec.rnn_hidden_size, model.num_notes):
            return 'free-running decoder not built for this vocabulary'
        if (model.num_notes + 1) * 1024 + 256 > 65536 or (3 * dec.rnn_hidden_size) % 4:
            return 'vocabulary too large for the segment sums'
        if any(d < 0 or d >= 4 for d in reg_dims) or len(reg_dims) > 16:
            return 'regularised dims must index the four attributes'
        opt = optimizer
        opt.ensure_arena()
        off = {id(p): o for p, o in zip(opt.params, opt._offsets)}
        if any(id(p) not in off for p in model.parameters()):
            return 'a parameter is outside the optimizer arena'

        def back_to_back(a, b):
            return off[id(a)] + (a.numel() + 3) // 4 * 4 == off[id(b)] and a.numel() % 4 == 0
        gru = enc.lstm
        for layer in range(2):
            g = lambda n, suf: getattr(gru, f'{n}_l{layer}{suf}')
            if not (back_to_back(g('weight_ih', ''), g('weight_ih', '_reverse')) and back_to_back(g('bias_ih', ''), g('bias_ih', '_reverse'))):
                return 'encoder input projections are not adjacent in the arena'
        for la, lb in ((enc.linear_mean[0], enc.linear_log_std[0]), (dec.beat_emb_to_tick_rnn_hidden[0], dec.beat_emb_to_tick_rnn_input[0])):
            if not (back_to_back(la.weight, lb.weight) and back_to_back(la.bias, lb.bias)):
                return 'paired Linear layers are not adjacent in the arena'
        return None

    def fits(self, batch):
        """the segment sums over the batch's positions keep their token indices in LDS (csrc/sequence.hip)"""
        m = self.model
        return (m.num_notes + 1) * 1024 + (m.num_ticks_per_measure * batch + 15) // 16 * 8 <= 65536

    def _offset(self, param):
        opt = self.optimizer
        for p, off in zip(opt.params, opt._offsets):
            if p is param:
                return off
        raise KeyError('parameter is not managed by the optimizer arena')

    def descriptor(self):
        arena = self.optimizer.ensure_arena()
        if self._desc is not None and self._arena_ptr == arena.data_ptr():
            return self._desc
        m, o = self.model, self._offset
        enc, dec = m.encoder, m.decoder
        d = MeasureVaeDesc()
        d.vocab, d.emb = m.num_notes, m.note_embedding_dim
        d.enc_hidden, d.dec_hidden, d.zdim = enc.rnn_hidden_size, dec.rnn_hidden_size, m.latent_space_dim
        d.steps, d.beats, d.ticks_per_beat = m.num_ticks_per_measure, m.num_beats_per_measure, m.num_ticks_per_beat
        d.enc_table = o(enc.note_embedding_layer.weight)
        for layer in range(2):
            wf, whf, bf, bhf = enc.lstm.cell(layer, '')
            _, whr, _, bhr = enc.lstm.cell(layer, '_reverse')
            d.enc_w_ih[layer], d.enc_b_ih[layer] = o(wf), o(bf)
            d.enc_w_hh[layer][0], d.enc_w_hh[layer][1] = o(whf), o(whr)
            d.enc_b_hh[layer][0], d.enc_b_hh[layer][1] = o(bhf), o(bhr)
        d.head_w0, d.head_b0 = o(enc.linear_mean[0].weight), o(enc.linear_mean[0].bias)
        d.mean_w2, d.mean_b2 = o(enc.linear_mean[2].weight), o(enc.linear_mean[2].bias)
        d.lstd_w2, d.lstd_b2 = o(enc.linear_log_std[2].weight), o(enc.linear_log_std[2].bias)
        d.dec_table, d.x0, d.b0 = o(dec.note_embedding_layer.weight), o(dec.x_0), o(dec.b_0)
        d.z2beat_w, d.z2beat_b = o(dec.z_to_beat_rnn_input[0].weight), o(dec.z_to_beat_rnn_input[0].bias)
        for layer in range(2):
            for name, rnn in (('beat', dec.rnn_beat), ('tick', dec.rnn_tick)):
                w_ih, w_hh, b_ih, b_hh = rnn.cell(layer)
                getattr(d, f'{name}_w_ih')[layer], getattr(d, f'{name}_w_hh')[layer] = o(w_ih), o(w_hh)
                getattr(d, f'{name}_b_ih')[layer], getattr(d, f'{name}_b_hh')[layer] = o(b_ih), o(b_hh)
        d.tick_init_w, d.tick_init_b = o(dec.beat_emb_to_tick_rnn_hidden[0].weight), o(dec.beat_emb_to_tick_rnn_hidden[0].bias)
        d.out_w, d.out_b = o(dec.tick_emb_to_note_emb[0].weight), o(dec.tick_emb_to_note_emb[0].bias)
        d.enc_dropout, d.dec_dropout = float(enc.dropout), float(dec.dropout)
        d.n_reg = len(self.reg_dims)
        for i, r in enumerate(self.reg_dims):
            d.reg_dims[i] = r
        d.beta, d.gamma, d.delta = self.beta, self.gamma, self.delta
        self._desc, self._arena_ptr = d, arena.data_ptr()
        return d

    def workspace(self, batch, device, ctx=None):
        """The activation workspace of one forward pass: one batch-sized buffer is cached and lent to the pass in flight; a
        forward that starts while an earlier pass still waits for its backward gets a buffer of its own (fused.FusedImageVAE)."""
        key = (batch, str(device))
        owner = self._ws_owner() if self._ws_owner is not None else None
        busy = owner is not None and not getattr(owner, 'ws_released', True)
        ws = self._ws.get(key)
        if ws is None or busy:
            n = _lib.load().arvae_measure_vae_ws_floats(ctypes.byref(self.descriptor()), batch)
            if n < 0:
                _lib.check(-1, 'measure_vae_ws_floats')
            fresh = torch.empty(n, device=device, dtype=torch.float32)
            if busy:
                return fresh
            ws = fresh
            self._ws = {key: ws}
        if ctx is not None:
            ctx.ws_released = False
            self._ws_owner = weakref.ref(ctx)
        return ws

    def tables(self, trainer, device):
        """arvae_measure_tables_t over the trainer's per-vocabulary attribute tables (kept alive here)"""
        if self._tables is None or self._tables[1][0][0].device != device:
            t = trainer._attr_tables(device)
            (midi, is_note, is_dens), w, norm = t
            self._tables = (MeasureTables(ops._ptr(midi), ops._ptr(is_note), ops._ptr(is_dens), ops._ptr(w), float(norm)), t)
        return self._tables[0]

    def run(self, score, train, capacity, tables, eps=None, masks=None, dp=None):
        """-> (loss[1] with grad_fn, scalars[8], acc, z, mu, sigma, tokens (B, 24)).

        dp (arvae_amd.parallel.DataParallel): the regulariser of this rank's rows against the z / label columns gathered from every
        rank, inside the same autograd node: the forward pass stops before the regulariser, one grouped all-gather, then
        arvae_measure_vae_finish; loss = recon + beta |KL| + W * reg_rowblock (parallel.py).

        train: dropout keep-masks are applied (the model is in training mode) and the decoder tosses its teacher-forcing coin.
        eps / masks = (encoder mask, beat mask, tick mask): explicit noise and keep-masks (parity runs); None: drawn by the pass
        itself from the library's Philox streams, in the order the per-layer path draws them."""
        m = self.model
        dec = m.decoder
        if dec.use_teacher_forcing and train:
            teacher_forced = torch.rand(1).item() < dec.teacher_forcing_prob          # host coin (decoder.py:427-428)
        else:
            teacher_forced = False
        dropping = m.training and (m.encoder.dropout > 0 or dec.dropout > 0)
        anchor = self.optimizer.params[0]
        return _MeasureStepFn.apply(anchor, self, score, bool(teacher_forced), bool(dropping), capacity, tables, eps, masks, dp)


class _MeasureStepFn(Function):
    @staticmethod
    def forward(ctx, anchor, fused, score, teacher_forced, dropping, capacity, tables, eps, masks, dp=None):
        ops._dev(score, capacity)
        lib = _lib.load()
        desc = fused.descriptor()
        opt = fused.optimizer
        m = fused.model
        b, steps = score.shape
        dev = score.device
        zd, he, hd = m.latent_space_dim, m.encoder.rnn_hidden_size, m.decoder.rnn_hidden_size
        nb = m.num_beats_per_measure
        ws = ctx.ws = fused.workspace(b, dev, ctx)
        enc, dec = m.encoder, m.decoder
        explicit_masks = dropping and masks is not None
        # all draws by the pass itself, unless the caller fixed some of them (parity runs): then the others come from the same
        # library launches the per-layer path uses, in its order (encoder keep-mask, eps, decoder keep-masks)
        draw = eps is None and not explicit_masks
        enc_mask = dec_mask = None
        if dropping:
            if explicit_masks:
                em, bm, tm = masks
                enc_mask = em.to(dev).contiguous()
                dec_mask = torch.cat((bm.to(dev).reshape(nb, b, hd), tm.to(dev).reshape(steps, b, hd)), 0).contiguous()
            elif draw:
                enc_mask = torch.empty(steps, b, 2 * he, device=dev, dtype=torch.uint8)
                dec_mask = torch.empty(nb + steps, b, hd, device=dev, dtype=torch.uint8)
            else:
                enc_mask = ops.keep_mask((steps, b, 2 * he), enc.dropout, dev)
        if eps is None:
            eps = torch.empty(b, zd, device=dev, dtype=torch.float32) if draw else ops.normal_noise((b, zd), dev)
        else:
            eps = eps.to(dev, torch.float32).contiguous()
        if dropping and dec_mask is None:
            dec_mask = ops.keep_mask((nb + steps, b, hd), dec.dropout, dev)
        desc.rng_draw = int(draw)
        if draw:
            # the per-layer path's draws, in its order: encoder keep-mask, eps, decoder keep-masks (ops.keep_mask / ops.normal_noise)
            desc.rng_seed = ops.rng_seed()
            desc.rng_offset[0] = ops.rng_next_offset() if dropping else 0
            desc.rng_offset[1] = ops.rng_next_offset()
            desc.rng_offset[2] = ops.rng_next_offset() if dropping else 0
            desc.rng_step = 0
            desc.rng_dev_step = ops._ptr(ops.rng_device_step(dev))
        scalars = torch.empty(NSCALARS, device=dev, dtype=torch.float32)
        mu = torch.empty(b, zd, device=dev, dtype=torch.float32)
        sigma, z = torch.empty_like(mu), torch.empty_like(mu)
        tokens = torch.empty(b, steps, device=dev, dtype=torch.int64)
        rowblock = dp is not None and len(fused.reg_dims) > 0
        labels = torch.empty(b, 4, device=dev, dtype=torch.float32) if rowblock else None
        with ops._timed('measure_vae_forward'):
            _lib.check(lib.arvae_measure_vae_forward(
                ctypes.byref(desc), b, ops._ptr(opt.param_arena), ops._ptr(score), ops._ptr(eps), ops._ptr(enc_mask), ops._ptr(dec_mask),
                int(teacher_forced), ops._ptr(capacity), ctypes.byref(tables) if tables is not None else None, ops._ptr(ws),
                ops._ptr(scalars), ops._ptr(mu), ops._ptr(sigma), ops._ptr(z), ops._ptr(tokens), ops._ptr(labels), int(rowblock),
                ops._stream()), 'measure_vae_forward')
        ctx.reg_scale = 1.0
        if rowblock:
            z_all, lab_all = dp.gather_many([z, labels])         # one RCCL launch on the launch stream
            ctx.reg_scale = float(dp.world_size)
            with ops._timed('measure_vae_finish'):
                _lib.check(lib.arvae_measure_vae_finish(
                    ctypes.byref(desc), b, ops._ptr(capacity), ops._ptr(z_all), ops._ptr(lab_all), z_all.shape[0], ctx.reg_scale,
                    ops._ptr(ws), ops._ptr(scalars), ops._ptr(mu), ops._ptr(sigma), ops._ptr(z), ops._ptr(labels), ops._stream()),
                    'measure_vae_finish')
        ctx.fused, ctx.masks = fused, (enc_mask, dec_mask)
        ctx.save_for_backward(score, eps, capacity, mu, sigma, z, tokens, scalars)
        ctx.set_materialize_grads(False)
        loss, acc = scalars[LOSS:LOSS + 1], scalars[ACC]
        ctx.mark_non_differentiable(scalars, acc, z, mu, sigma, tokens)
        return loss, scalars, acc, z, mu, sigma, tokens

    @staticmethod
    @once_differentiable
    def backward(ctx, g_loss, *_unused):
        score, eps, capacity, mu, sigma, z, tokens, scalars = ctx.saved_tensors
        fused = ctx.fused
        lib = _lib.load()
        opt = fused.optimizer
        if g_loss is None:
            return (None,) * 10
        g_loss = g_loss.reshape(1).contiguous()
        enc_mask, dec_mask = ctx.masks
        opt.mark_dirty()                                         # gradients land in the arena without torch's accumulation
        with ops._timed('measure_vae_backward'):
            _lib.check(lib.arvae_measure_vae_backward(
                ctypes.byref(fused.descriptor()), score.shape[0], ops._ptr(opt.param_arena), ops._ptr(opt.grad_arena), ops._ptr(score),
                ops._ptr(eps), ops._ptr(enc_mask), ops._ptr(dec_mask), ops._ptr(capacity), ops._ptr(mu), ops._ptr(sigma), ops._ptr(z),
                ops._ptr(tokens), ops._ptr(scalars), ops._ptr(g_loss), ctx.reg_scale, ops._ptr(ctx.ws), ops._stream()), 'measure_vae_backward')
        ctx.ws_released = True
        return (None,) * 10
