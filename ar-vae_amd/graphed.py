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

Data-parallel runs: a collective inside the step (the all-gather of the regularised latent / label columns, parallel.py) is
NOT captured.  The capture is cut there instead: the step becomes a chain of graphs with the collectives issued eagerly
between them (`Segments`), on the same static buffers every replay, so that a data-parallel MeasureVAE step also runs at
the replay rate instead of the eager one (1.5 vs 3.9 ms).  The gradient all-reduce and Adam stay outside, in trainer.step().

Random numbers drawn on the device inside the step (reparameterisation noise, dropout masks) come from the library's
counter-based Philox stream (csrc/rng.h): the captured launches read their stream position from a device word that the
graph itself advances, so every replay sees fresh values.  Noise / masks pushed through the models' host-side queues are
NOT visible to a captured graph (the queue is consumed at capture time).
"""
import torch


class Segments:
    """A step captured as a chain of HIP graphs cut at the points where something must run eagerly (a collective).
    While `capture(fn)` runs fn, `split(eager_fn)` ends the graph being captured, runs eager_fn, remembers it and starts the
    next graph from the same memory pool; `replay()` replays graph, eager_fn, graph, ... in that order on the current stream.
    eager_fn must work on tensors it keeps alive (allocated during the capture: their addresses are what the graphs use)."""

    def __init__(self, device=None):
        self.graphs, self.between = [], []
        self.pool = torch.cuda.graph_pool_handle()
        self.stream = torch.cuda.Stream(device=device)
        self.capturing = False

    def _begin(self):
        g = torch.cuda.CUDAGraph()
        # with a process group alive, its watchdog thread polls events while we capture: 'global' mode would turn that into
        # hipErrorStreamCaptureUnsupported; only this thread's (and the autograd thread's launches, which target the capturing
        # stream) matter here
        import torch.distributed as dist
        mode = 'thread_local' if dist.is_available() and dist.is_initialized() else 'global'
        g.capture_begin(pool=self.pool, capture_error_mode=mode)
        self.graphs.append(g)

    def capture(self, fn):
        torch.cuda.synchronize()
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.stream):
            self._begin()
            self.capturing = True
            try:
                out = fn()
            finally:
                self.capturing = False
                self.graphs[-1].capture_end()
        torch.cuda.current_stream().wait_stream(self.stream)
        return out

    def split(self, eager_fn):
        if not self.capturing:
            raise RuntimeError('Segments.split outside capture')
        self.graphs[-1].capture_end()
        result = eager_fn()                    # for real, on the capture stream: its inputs are not computed yet (capture
        self.between.append(eager_fn)          # records, it does not run), only the call sequence and the buffers matter
        self._begin()
        return result

    def replay(self):
        for i, g in enumerate(self.graphs):
            g.replay()
            if i < len(self.between):
                self.between[i]()


class GraphedStep:
    def __init__(self, trainer, example_batch, warmup=3):
        if not torch.cuda.is_available():
            raise RuntimeError('GraphedStep needs a GPU')
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
                graph = Segments(self.static[0].device)
                dp = getattr(trainer, 'data_parallel', None)
                if dp is not None:
                    dp.capture_splitter = graph              # its collectives cut the capture instead of being recorded
                try:
                    out = graph.capture(self._eager)
                finally:
                    if dp is not None:
                        dp.capture_splitter = None
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
