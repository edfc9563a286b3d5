"""Forward + backward of a training step replayed from HIP graphs (torch.cuda.CUDAGraph).

The MeasureVAE step is a few hundred small launches issued from Python (whole-sequence GRU kernels, whole-sequence
GEMMs, glue): eager, the host sets the pace (3.9 ms per step at B = 256).  Every launch of the library goes to torch's
current stream and nothing in the step synchronises, so the whole forward + backward can be captured once and replayed:
1.53 ms per step on MI355X.  The optimizer stays outside (its bias corrections are host scalars), as does the host-side
teacher-forcing coin: one graph per control-flow variant.

    graphed = GraphedStep(trainer, example_batch)        # captures (after a few warm-up iterations)
    for batch in loader:
        loss, acc = graphed(batch)                       # zero_grad + loss + backward
        trainer.step()

Data-parallel runs: the step's collectives (the all-gather of the regularised latent / label columns, parallel.py) are RCCL
calls the library enqueues on the launch stream (arvae_comm_*), so the capture RECORDS them like the kernels around them: a
data-parallel MeasureVAE step is ONE graph as well.  There is no torch process group in such a run (no watchdog thread
polling events next to the capture: round 3's chain of graphs cut at eager torch.distributed collectives could be aborted
by exactly that thread).  Over a transport that cannot be captured (parallel.TorchComm) the step stays eager.  Whether a
capture succeeded is decided by ALL ranks together (a MIN all-reduce of a flag after each attempt): either every rank
replays or every rank runs eagerly -- the collective sequences never diverge.  The gradient all-reduce and Adam stay
outside, in trainer.step().

Random numbers drawn on the device inside the step (reparameterisation noise, dropout masks) come from the library's
counter-based Philox stream (csrc/rng.h): the captured launches read their stream position from a device word that the
graph itself advances, so every replay sees fresh values.  Noise / masks pushed through the models' host-side queues are
NOT visible to a captured graph (the queue is consumed at capture time).
"""
import torch


class Capture:
    """forward + backward of one control-flow variant as one HIP graph, captured on a stream of its own"""

    def __init__(self, device=None, thread_local=False):
        self.graph = torch.cuda.CUDAGraph()
        self.stream = torch.cuda.Stream(device=device)
        # 'global' (the default) lets ANY thread's capture-unsafe call invalidate the capture; a data-parallel process has
        # RCCL's helper threads next to us, whose calls are none of the capture's business: only this thread's (and the
        # autograd thread's launches, which target the capturing stream) matter then
        self.mode = 'thread_local' if thread_local else 'global'

    def capture(self, fn):
        torch.cuda.synchronize()
        with torch.cuda.graph(self.graph, stream=self.stream, capture_error_mode=self.mode):
            out = fn()                         # an exception ends the capture (torch.cuda.graph.__exit__) and propagates
        return out

    def replay(self):
        self.graph.replay()


class GraphedStep:
    def __init__(self, trainer, example_batch, warmup=3):
        if not torch.cuda.is_available():
            raise RuntimeError('GraphedStep needs a GPU')
        dp = getattr(trainer, 'data_parallel', None)
        if dp is not None and not dp.capturable:
            raise RuntimeError('the data-parallel transport of this trainer cannot be captured into a HIP graph')
        self.trainer = trainer
        self.hyper = self.hyper_of(trainer)
        self.static = tuple(t.clone() for t in trainer.process_batch_data(example_batch))
        self.decoder = getattr(trainer.model, 'decoder', None)
        coin = self.decoder is not None and getattr(self.decoder, 'use_teacher_forcing', False)
        self.variants = (True, False) if coin else (None,)
        self.prob = self.decoder.teacher_forcing_prob if coin else None
        self.graphs = {}
        trainer.model.train()
        from . import ops
        ops.rng_device_step(self.static[0].device, create=True)   # the captured draws read their stream position from the device
        try:                                                     # the decoder's coin is pinned only while capturing
            for variant in self.variants:
                self._pin(variant)
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    for _ in range(warmup):
                        self._eager()
                torch.cuda.current_stream().wait_stream(side)
                graph = Capture(self.static[0].device, thread_local=dp is not None)
                error = None
                try:
                    out = graph.capture(self._eager)
                except RuntimeError as e:
                    error = e
                if dp is not None and not dp.all_agree(error is None):       # every rank replays, or none does
                    raise RuntimeError(f'graph capture failed on {"this" if error is not None else "another"} rank: {error}')
                if error is not None:
                    raise error
                # the loss terms of THIS variant's step live in its own static buffers (trainer.last_terms is rebound by
                # every capture and by every eager step)
                self.graphs[variant] = (graph, out, dict(getattr(trainer, 'last_terms', {}) or {}))
        finally:
            self._pin(None)

    @staticmethod
    def hyper_of(trainer):
        """the hyper-parameters that are compiled into the captured launches as immediates"""
        reg_dim = getattr(trainer, 'reg_dim', ())
        return (float(getattr(trainer, 'beta', 0.0)), float(getattr(trainer, 'gamma', 0.0)),
                float(getattr(trainer, 'delta', 0.0)), tuple(reg_dim) if isinstance(reg_dim, (tuple, list)) else reg_dim)

    def _pin(self, variant):
        """force the decoder's host coin: 2.0 = always teacher-forced, -1.0 = never, None = restore"""
        if self.prob is not None:
            self.decoder.teacher_forcing_prob = self.prob if variant is None else (2.0 if variant else -1.0)

    def _eager(self):
        from . import ops
        ops.rng_advance_device_step()                            # recorded: every replay moves on in the Philox stream
        self.trainer.zero_grad()
        loss, acc = self.trainer.loss_and_acc_for_batch(self.static, 0, 1, True)
        self.trainer.backward(loss)
        return loss.detach(), None if acc is None else acc.detach()

    def accepts(self, batch):
        """the processed batch when it has the shapes this step was captured for (hand it to __call__ as `data`), else None"""
        data = self.trainer.process_batch_data(batch)
        ok = len(data) == len(self.static) and all(d.shape == s.shape for d, s in zip(data, self.static))
        return data if ok else None

    def __call__(self, batch=None, data=None):
        if data is None:
            data = self.trainer.process_batch_data(batch)
        for dst, src in zip(self.static, data):
            if dst.shape != src.shape:
                raise ValueError(f'GraphedStep was captured for batches of shape {tuple(dst.shape)}, got {tuple(src.shape)}')
            dst.copy_(src, non_blocking=True)
        variant = None if self.prob is None else bool(torch.rand(1).item() < self.prob)
        graph, out, terms = self.graphs[variant]
        graph.replay()
        self.trainer.optimizer.mark_dirty()                      # the replayed backward wrote gradients (no autograd hooks ran)
        if terms:
            self.trainer.last_terms = terms                      # what log_loss_split reads: the replayed variant's terms
        return out
