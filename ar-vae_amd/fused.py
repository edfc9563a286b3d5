"""Whole-model training step for the conv AR-VAEs: forward (+ all loss terms) and backward are ONE C call each
(arvae_image_vae_forward / arvae_image_vae_backward), so the host does no per-layer work.

Used by ImageVAETrainer.loss_and_acc_for_batch when the model's parameters live in the trainer's flat
Adam arena.  Autograd sees a single node: its backward accumulates every parameter gradient directly into the
gradient arena and returns nothing for the parameters.
"""
import ctypes
import weakref

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

import os

from . import _lib, ops
from ._lib import ImageVaeDesc, LayerDesc

LOSS, RECON, DIST, REG, ACC, KL, NSCALARS = 0, 1, 2, 3, 4, 5, 8


# Data-parallel overlap: the executors record events where z / the decoder's conv gradients / the
# Linear gradients are final (arvae_image_vae_t.milestones), and the collectives that need them are enqueued on a side
# stream behind those events, so they run under the rest of the pass.  OPT-IN (ARVAE_DP_OVERLAP=1).  Measured on one MI355X
# with one RCCL rank (bench.py --force-dp, B = 512): plain step 0.513 ms, data-parallel step with every collective on the
# launch stream 0.551 ms, with the overlap schedule 0.645 ms -- on this stack every cross-stream dependency (event record on
# the launch stream, wait on the side stream, join before Adam) costs the launch stream 5-25 us of dispatch gap
# (profiles/r3_dp_timeline.txt), more than a one-rank collective takes.  Across GPUs, where an all-reduce of the 1.6 MB
# Linear bucket takes tens of microseconds, the trade may go the other way: no multi-GPU box was available to measure it.
def _dp_overlap():
    return os.environ.get('ARVAE_DP_OVERLAP', '0') == '1'


class _Overlap:
    """the side stream, the three milestone events and their C descriptor for one device"""

    def __init__(self, device):
        self.stream = torch.cuda.Stream(device=device)
        self.events = [torch.cuda.Event() for _ in range(3)]
        for e in self.events:                        # torch creates the HIP event on its first record
            e.record(torch.cuda.current_stream(device))
        self.z_ready, self.dec_grads, self.linear_grads = self.events
        self.struct = _lib.Milestones(*[ctypes.c_void_p(e.cuda_event) for e in self.events])

    def pointer(self):
        return ctypes.cast(ctypes.pointer(self.struct), ctypes.c_void_p)


def _layer_desc(link, is_up, act, dropout, w_off, b_off):
    return LayerDesc(link.desc(0), int(is_up), int(act), int(dropout), 0, int(w_off), int(b_off))


class DeviceStatusError(RuntimeError):
    """a pass reported a device-side failure in its status word; `.skipped` = optimizer updates the device withheld"""

    def __init__(self, message, skipped=0):
        super().__init__(message)
        self.skipped = int(skipped)


class FusedImageVAE:
    """Descriptor + workspace cache binding a MnistVAE/DspritesVAE to a FlatAdam arena."""

    def __init__(self, model, optimizer, reg_dims, beta, gamma, delta, dec_dist):
        self.model, self.optimizer = model, optimizer
        self.reg_dims = tuple(int(d) for d in reg_dims)
        self.beta, self.gamma, self.delta = float(beta), float(gamma), float(delta)
        self.dist = ops.RECON_DIST[dec_dist]
        self._desc = None
        self._arena_ptr = None
        self._ws = {}
        self._ws_owner = None
        self._overlap = None
        self._buckets = None
        self.no_cluster = False          # set once a hand-off has given up: the latent block stays on the row kernels

    def status_word(self, device):
        """the sticky device status word of this model's passes (arvae_image_vae_t.status): it lives in the guard slot behind
        the optimizer's gradient arena, where arvae_adam_step reads it before it touches the weights (optim.py)"""
        return self.optimizer.status_words()

    def check_status(self):
        """Reads the status word (ONE device sync: call it where the host synchronises anyway -- Trainer.loss_and_acc_on_epoch
        does, next to the epoch means).  A hand-off between the workgroups of the clustered latent block that gave up (its
        partners could not all become resident: csrc/midcluster.hip) left the results of that pass undefined.  The update
        kernel saw the same word and skipped that step and every step since (weights, moments and step count are those of
        the last good step): raise DeviceStatusError, and keep every later pass on the kernels without in-launch hand-offs."""
        bits, skipped = self.optimizer.take_status()
        if bits:
            self.no_cluster = True
            code = (bits >> 20) & 7 if (bits & 0xff800000) == 0x40000000 else 0     # exact on the rank that failed
            raise DeviceStatusError(
                f'libarvae_hip: an in-launch hand-off of the clustered latent block gave up (status {bits:#x}: '
                f'{"forward " if code & 1 else ""}{"backward " if code & 2 else ""}{"tickets " if code & 4 else ""}'
                f'{"reported by another rank, " if not code else ""}pass); the results of that training step were undefined and '
                f'{skipped} optimizer update(s) were skipped on the device -- the weights are those of the last good step.  The '
                f'device is probably shared with other processes; later passes of this trainer use the row kernels (no '
                f'hand-offs).', skipped)

    def overlap(self, device):
        if self._overlap is None:
            self._overlap = _Overlap(device)
        return self._overlap

    def grad_buckets(self):
        """float ranges of the gradient arena that become final together: -> (decoder conv layers, Linear layers + heads,
        [the rest]) or None when the arena is not laid out in three contiguous runs (then one all-reduce at the end)."""
        if self._buckets is not None:
            return self._buckets or None
        m, opt = self.model, self.optimizer
        opt.ensure_arena()
        size = {id(p): (p.numel() + 3) // 4 * 4 for p in opt.params}
        off = {id(p): o for p, o in zip(opt.params, opt._offsets)}

        def span(params):
            lo = min(off[id(p)] for p in params)
            hi = max(off[id(p)] + size[id(p)] for p in params)
            return (lo, hi) if hi - lo == sum(size[id(p)] for p in params) else None

        def params_of(layers):
            return [p for lay in layers for p in (lay.weight, lay.bias) if p is not None]
        dec = span(params_of([m.dec_conv[i] for i, _ in m.dec_conv_plan]))
        lin = span(params_of([m.enc_lin[i] for i, _ in m.enc_lin_plan] + [m.enc_mean, m.enc_log_std] +
                             [m.dec_lin[i] for i, _ in m.dec_lin_plan]))
        total = opt.grad_arena.numel()
        if dec is None or lin is None or not (lin[1] <= dec[0] or dec[1] <= lin[0]):
            self._buckets = ()
            return None
        cuts = sorted([dec, lin])
        rest = [(a, b) for a, b in zip([0] + [c[1] for c in cuts], [c[0] for c in cuts] + [total]) if b > a]
        self._buckets = (dec, lin, rest)
        return self._buckets

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
        m = self.model
        d = ImageVaeDesc()
        n_enc_conv = len(m.enc_conv_plan)
        layers = []
        for k, (idx, link) in enumerate(m.enc_conv_plan):
            lay = m.enc_conv[idx]
            layers.append(_layer_desc(link, 0, m.hidden_act, m.dropout_p > 0, self._offset(lay.weight),
                                      self._offset(lay.bias)))
        for idx, link in m.enc_lin_plan:
            lay = m.enc_lin[idx]
            layers.append(_layer_desc(link, 0, m.hidden_act, 0, self._offset(lay.weight), self._offset(lay.bias)))
        d.n_enc = len(layers)
        for i, l in enumerate(layers):
            d.enc[i] = l
        layers = []
        for idx, link in m.dec_lin_plan:
            lay = m.dec_lin[idx]
            layers.append(_layer_desc(link, 0, m.hidden_act, 0, self._offset(lay.weight), self._offset(lay.bias)))
        last = len(m.dec_conv_plan) - 1
        for k, (idx, link) in enumerate(m.dec_conv_plan):
            lay = m.dec_conv[idx]
            hidden = k < last
            layers.append(_layer_desc(link, 1, m.hidden_act if hidden else ops.ACT_NONE,
                                      hidden and m.dropout_p > 0, self._offset(lay.weight), self._offset(lay.bias)))
        d.n_dec = len(layers)
        for i, l in enumerate(layers):
            d.dec[i] = l
        d.head_mu = _layer_desc(m.head_link, 0, ops.ACT_NONE, 0, self._offset(m.enc_mean.weight),
                                self._offset(m.enc_mean.bias))
        d.head_log_std = _layer_desc(m.head_link, 0, ops.ACT_NONE, 0, self._offset(m.enc_log_std.weight),
                                     self._offset(m.enc_log_std.bias))
        d.zdim, d.recon_dist = m.z_dim, self.dist
        d.n_reg = len(self.reg_dims)
        for i, r in enumerate(self.reg_dims):
            d.reg_dims[i] = r
        d.beta, d.gamma, d.delta = self.beta, self.gamma, self.delta
        assert n_enc_conv <= d.n_enc
        self._desc, self._arena_ptr = d, arena.data_ptr()
        return d

    def workspace(self, batch, device, ctx=None):
        """The activation workspace of one forward pass.  One batch-sized buffer is cached and lent to the pass that is in
        flight; a forward that starts while an earlier pass still waits for its backward (two losses summed, a validation
        forward in between, another batch size) gets a buffer of its own, so no pass can overwrite another's activations.
        The buffer travels on the autograd ctx: backward reads exactly what its forward wrote."""
        key = (batch, str(device))
        owner = self._ws_owner() if self._ws_owner is not None else None
        busy = owner is not None and not getattr(owner, 'ws_released', True)
        ws = self._ws.get(key)
        if ws is None or busy:
            n = _lib.load().arvae_image_vae_ws_floats(ctypes.byref(self.descriptor()), batch, 0)
            if n < 0:
                _lib.check(-1, 'image_vae_ws_floats')
            fresh = torch.empty(n, device=device, dtype=torch.float32)
            if busy:
                return fresh                          # not cached: it lives and dies with this pass
            ws = fresh
            self._ws = {key: ws}                      # keep one (batch-sized) workspace alive
        if ctx is not None:
            ctx.ws_released = False
            self._ws_owner = weakref.ref(ctx)
        return ws

    def run(self, x, labels, eps, masks, capacity, external_reg=False, reg_scale=1.0, dp=None, capacity_nonzero=False,
            draw_eps=False, defer_finish=False):
        """-> (loss[1] with grad_fn, scalars[8], acc, z, mu, sigma, logits); loss is scalars[LOSS:LOSS+1].

        dp (arvae_amd.parallel.DataParallel): evaluate the regulariser on this rank's row block against the columns
        gathered from every rank, inside the same autograd node (no torch ops on the hot path): the loss returned is
        recon + beta|KL - c| + W * reg_rowblock and scalars[REG] = W * reg_rowblock.  With capacity_nonzero the KL
        mean is all-reduced first and the term becomes the global beta|KL_global - c| (parallel.py)."""
        d = self.descriptor()                                    # where this pass's eps comes from (csrc/rng.h)
        d.status = ops._ptr(self.status_word(x.device))
        d.flags = 1 if self.no_cluster else 0                    # ARVAE_VAE_NO_CLUSTER
        d.rng_eps = int(bool(draw_eps))
        if draw_eps:
            d.rng_seed, d.rng_offset, d.rng_step = ops.rng_seed(), ops.rng_next_offset(), 0
            d.rng_dev_step = ops._ptr(ops.rng_device_step(x.device))
        # defer_finish (ARVAE_VAE_DEFER_FINISH, include/arvae_hip.h): a training step whose backward() follows at once lets the
        # backward pass's first launch carry the forward pass's finishing step -- the scalars (loss, its split, accuracy) read as
        # NaN until backward() has run.  Under data parallelism: when the library finishes the pass itself behind the gather
        # (arvae_image_vae_finish: a regulariser, capacity 0), whose last launch then parks the step.
        in_lib = dp is not None and len(self.reg_dims) > 0 and not capacity_nonzero
        defer = bool(defer_finish) and (dp is None or in_lib) and not external_reg and torch.is_grad_enabled()
        anchor = self.optimizer.params[0]
        return _FusedStepFn.apply(anchor, self, x, labels, eps, masks, capacity, bool(external_reg), float(reg_scale), dp,
                                  bool(capacity_nonzero), defer)


def _mask_array(masks):
    if not masks or all(m is None for m in masks):
        return None, None
    arr = (ctypes.c_void_p * len(masks))(*[m.data_ptr() for m in masks])
    return arr, masks


class _FusedStepFn(Function):
    @staticmethod
    def forward(ctx, anchor, fused, x, labels, eps, masks, capacity, external_reg, reg_scale, dp=None, capacity_nonzero=False,
                defer_finish=False):
        ops._dev(x, labels, eps, capacity)
        lib = _lib.load()
        desc = fused.descriptor()
        ctx.flags = (desc.flags & ~2) | (2 if defer_finish else 0)      # ARVAE_VAE_DEFER_FINISH: the same flags for this pass's backward
        desc.flags = ctx.flags
        opt = fused.optimizer
        b = x.shape[0]
        dev = x.device
        ws = ctx.ws = fused.workspace(b, dev, ctx)
        zd = fused.model.z_dim
        scalars = torch.empty(NSCALARS, device=dev, dtype=torch.float32)
        mu = torch.empty(b, zd, device=dev, dtype=torch.float32)
        sigma, z = torch.empty_like(mu), torch.empty_like(mu)
        logits = torch.empty_like(x)
        marr, keep = _mask_array(masks)
        rowblock = dp is not None and len(fused.reg_dims) > 0
        # the overlap schedule lives on a side stream with its own events: not inside a stream capture
        ov = fused.overlap(dev) if (dp is not None and _dp_overlap() and not torch.cuda.is_current_stream_capturing()) else None
        desc.milestones = ov.pointer() if ov is not None else None
        ctx.overlap = ov
        if rowblock:
            external_reg = True
            labels = labels.contiguous()
        # data parallel, capacity 0: the library finishes the pass itself once the columns are gathered (arvae_image_vae_finish:
        # row-block regulariser + scalars, two launches) instead of three launches and a torch add from here
        finish_in_lib = rowblock and not (dp is not None and capacity_nonzero)
        with ops._timed('image_vae_forward'):
            _lib.check(lib.arvae_image_vae_forward(
                ctypes.byref(desc), b, ops._ptr(opt.param_arena), ops._ptr(x), ops._ptr(labels),
                labels.shape[1] if labels is not None else 0, ops._ptr(eps), marr, ops._ptr(capacity), None, None,
                -2 if finish_in_lib else (-1 if external_reg else 0), reg_scale, ops._ptr(ws), ops._ptr(scalars), ops._ptr(mu),
                ops._ptr(sigma), ops._ptr(z), ops._ptr(logits), ops._stream()), 'image_vae_forward')
        ctx.dz_unit = None
        if dp is not None and capacity_nonzero:
            # beta*|KL - c| is not shard-linear for c != 0: use the global KL mean (one 4-byte all-reduce).  The backward
            # pass takes sign(KL_local - c_r) with the shifted capacity c_r = c + KL_local - KL_global = sign(KL_global - c).
            cap_r = dp.shifted_capacity(scalars[KL:KL + 1], capacity)
            dist_g = fused.beta * (scalars[KL:KL + 1] - cap_r).abs()
            scalars[LOSS:LOSS + 1].add_(dist_g - scalars[DIST:DIST + 1])
            scalars[DIST:DIST + 1].copy_(dist_g)
            capacity = cap_r
        if rowblock:
            if ov is not None:
                # z is final long before the pass ends (the decoder's launches follow the latent block): its all-gather waits
                # for the executor's event on the side stream and runs under the decoder
                z_all = torch.empty((dp.world_size * b, zd), dtype=z.dtype, device=dev)
                lab_all = torch.empty((dp.world_size * b, labels.shape[1]), dtype=labels.dtype, device=dev)
                with torch.cuda.stream(ov.stream):
                    ov.stream.wait_event(ov.z_ready)
                    _, lab_work = dp.gather_columns(labels, async_op=True, out=lab_all)
                    _, z_work = dp.gather_columns(z, async_op=True, out=z_all)
                lab_work.wait()
                z_work.wait()                                    # the launch stream continues after the collectives
            else:
                z_all, lab_all = dp.gather_many([z, labels])     # one RCCL launch on the launch stream
        if finish_in_lib:
            w = float(dp.world_size)
            with ops._timed('image_vae_finish'):
                _lib.check(lib.arvae_image_vae_finish(
                    ctypes.byref(desc), b, ops._ptr(labels), labels.shape[1], ops._ptr(capacity), ops._ptr(z_all), ops._ptr(lab_all),
                    z_all.shape[0], w, ops._ptr(ws), ops._ptr(scalars), ops._ptr(mu), ops._ptr(sigma), ops._ptr(z), ops._stream()),
                    'image_vae_finish')
            external_reg, reg_scale = False, w                   # backward: the pass's own regulariser gradient (reg_fused 1)
        elif rowblock:
            n, n_all, r = b, z_all.shape[0], len(fused.reg_dims)
            ws_reg = torch.empty(int(lib.arvae_reg_loss_ws_floats(n, r)), device=dev, dtype=torch.float32)
            reg_out = torch.empty(1, device=dev, dtype=torch.float32)
            dz = torch.empty_like(z)
            dims = (ctypes.c_int32 * r)(*fused.reg_dims)
            with ops._timed('reg_loss(row block)'):
                _lib.check(lib.arvae_reg_loss(ops._ptr(z), ops._ptr(labels), n, ops._ptr(z_all), ops._ptr(lab_all), n_all, zd,
                                              labels.shape[1], dims, r, fused.gamma, fused.delta, ops._ptr(ws_reg),
                                              ops._ptr(reg_out), ops._ptr(dz), ops._stream()), 'reg_loss')
            w = float(dp.world_size)
            # scalars[LOSS] += W * reg and scalars[REG] (0 so far) = W * reg: one kernel on the strided pair
            scalars[LOSS:REG + 1:REG - LOSS].add_(reg_out, alpha=w)
            ctx.dz_unit, reg_scale = dz, w
        ctx.fused, ctx.masks, ctx.marr, ctx.dp = fused, keep, marr, dp
        ctx.external_reg, ctx.reg_scale = external_reg, reg_scale
        ctx.save_for_backward(x, eps, capacity, mu, sigma, z, logits)
        # the loss is handed out as its own output (a 1-element view of the scalars) so that backward receives
        # its gradient directly: no slice-backward zeros + copy, and no zero-filled gradients for the outputs
        # nobody differentiates (the logits alone would be an 8 MB fill per step)
        ctx.set_materialize_grads(False)
        loss, acc = scalars[LOSS:LOSS + 1], scalars[ACC]
        ctx.mark_non_differentiable(scalars, acc, mu, sigma, logits)
        if rowblock or not external_reg:
            ctx.mark_non_differentiable(z)
        return loss, scalars, acc, z, mu, sigma, logits

    @staticmethod
    @once_differentiable
    def backward(ctx, g_loss, _g_scalars, _g_acc, g_z, _g_mu, _g_sigma, _g_logits):
        x, eps, capacity, mu, sigma, z, logits = ctx.saved_tensors
        fused = ctx.fused
        lib = _lib.load()
        opt = fused.optimizer
        if g_loss is None:                                       # only z was differentiated (external regulariser)
            g_loss = torch.zeros(1, device=x.device, dtype=torch.float32)
        g_loss = g_loss.reshape(1).contiguous()
        reg_mode = 0 if ctx.external_reg else 1
        if ctx.dz_unit is not None:                              # row-block regulariser evaluated in forward (data parallel)
            reg_mode, g_z = 2, ctx.dz_unit
        dz_extra = g_z.contiguous() if (ctx.external_reg and g_z is not None) else None
        ws = ctx.ws
        opt.mark_dirty()                                         # gradients land in the arena without torch's accumulation
        desc = fused.descriptor()
        desc.flags = ctx.flags                                   # (as this pass's forward call had them: a deferred finishing step)
        with ops._timed('image_vae_backward'):
            _lib.check(lib.arvae_image_vae_backward(
                ctypes.byref(desc), x.shape[0], ops._ptr(opt.param_arena), ops._ptr(opt.grad_arena),
                ops._ptr(x), ops._ptr(eps), ctx.marr, ops._ptr(capacity), ops._ptr(mu), ops._ptr(sigma), ops._ptr(z),
                ops._ptr(logits), ops._ptr(g_loss), ops._ptr(dz_extra), reg_mode, ctx.reg_scale,
                ops._ptr(ws), ops._stream()), 'image_vae_backward')
        ctx.ws_released = True                                   # the cached workspace may serve the next forward
        ov, dp = ctx.overlap, ctx.dp
        buckets = fused.grad_buckets() if (ov is not None and dp is not None) else None
        if buckets is not None:
            # the decoder's conv gradients and the Linear gradients are final well before the pass ends: their all-reduces
            # wait for the executor's events on the side stream and run under the encoder's backward kernels; what is
            # left (the encoder's conv gradients) goes with DataParallel.reduce_gradients
            dec, lin, rest = buckets
            with torch.cuda.stream(ov.stream):
                for event, (lo, hi) in ((ov.dec_grads, dec), (ov.linear_grads, lin)):
                    ov.stream.wait_event(event)
                    dp.start_bucket(opt.grad_arena, lo, hi)
            dp.remaining_buckets = rest
        return (None,) * 12
