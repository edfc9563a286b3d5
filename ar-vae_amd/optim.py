"""Adam over one flat fp32 arena (one HIP kernel per step; one RCCL all-reduce per step).

Replaces torch.optim.Adam as constructed by the reference's Trainer
(utils/trainer.py:31-34): lr given, betas (0.9, 0.999), eps 1e-8, no weight
decay, no amsgrad.  Parameters stay ordinary nn.Parameters (same state_dict);
their storage is re-pointed into a contiguous arena, and `.grad` of each is a
view into a matching gradient arena, so
  zero_grad()  = one memset -- or nothing: step() clears the gradient arena in the kernel that consumes it
                 (`zero_grads_in_step`, on by default), and the zero_grad() that follows a step has nothing left to do
                 unless a backward pass ran in between (every backward marks the arena dirty: the fused executor
                 through mark_dirty(), torch's own accumulation through a post-accumulate hook on each parameter, the
                 per-layer kernels that add straight into `.grad` through ops.GRAD_WRITE_EPOCH).
                 NOTE for callers that look at gradients AFTER step(): they read zeros, where torch.optim.Adam leaves
                 the last gradient in place (the reference never does: utils/trainer.py:126-147 reads nothing after
                 step()); FlatAdam(..., zero_grads_in_step=False) keeps torch's behaviour.
  all-reduce   = one collective on `grad_arena`,
  step()       = one multi-tensor kernel (arvae_adam_step).

Guard slot.  Eight more floats sit behind the gradient arena in the same allocation.  Word 0 is the sticky device STATUS
word of the passes that write this arena (arvae_image_vae_t.status: an in-launch hand-off that gave up ORs a float-safe
code into it, include/arvae_hip.h ARVAE_STATUS_*); word 4 counts the updates arvae_adam_step skipped because of it.  The
update kernel reads the word itself: a step whose gradients are undefined never reaches the weights, the moments or the
step count, however late the host looks (`take_status`, once per epoch).  Under data parallelism the word travels with
the gradients (`reduce_view`: arena + the first four guard words, ONE SUM all-reduce), so every rank skips the same updates.
"""
import torch

from . import ops


GUARD_FLOATS = 8


class FlatAdam:
    def __init__(self, params, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, zero_grads_in_step=True):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError('optimizer got an empty parameter list')
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.step_count = 0
        self.grad_scale = 1.0          # set to 1/world_size when gradients are SUM all-reduced
        self.zero_grads_in_step = bool(zero_grads_in_step)
        self._arena_clean = False      # the gradient arena is all zeros (set by step(), cleared by any backward pass)
        self._write_epoch = -1         # ops.GRAD_WRITE_EPOCH at the last step(): direct writes since then dirty the arena
        self._hooked = False
        self.param_arena = self.grad_arena = self.exp_avg = self.exp_avg_sq = None
        self._grad_store = self.guard = None
        self._offsets = []

    # -- arena management -----------------------------------------------------------------------
    def _arena_valid(self):
        if self.param_arena is None:
            return False
        base = self.param_arena.data_ptr()
        for p, off in zip(self.params, self._offsets):
            if p.data_ptr() != base + 4 * off or p.device != self.param_arena.device:
                return False
        return True

    def _build_arena(self):
        dev = self.params[0].device
        offsets, total = [], 0
        for p in self.params:
            offsets.append(total)
            total += (p.numel() + 3) // 4 * 4                 # keep every tensor 16-byte aligned
        arena = torch.zeros(total, device=dev, dtype=torch.float32)
        old_guard = self.guard
        store = torch.zeros(total + GUARD_FLOATS, device=dev, dtype=torch.float32)
        grads = store[:total]                                 # what zero_grad() clears and the passes write
        self._grad_store = store
        self.guard = store[total:].view(torch.int32)          # [0] status (reduced with the gradients), [4] skipped updates
        if old_guard is not None:
            self.guard.copy_(old_guard)
        old_m, old_v, old_off = self.exp_avg, self.exp_avg_sq, self._offsets
        self.exp_avg = torch.zeros(total, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(total, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for i, (p, off) in enumerate(zip(self.params, offsets)):
                n = p.numel()
                arena[off:off + n].copy_(p.detach().reshape(-1))
                p.data = arena[off:off + n].view(p.shape)
                p.grad = grads[off:off + n].view(p.shape)
                if old_m is not None:                      # moved device mid-training: carry the moments
                    self.exp_avg[off:off + n].copy_(old_m[old_off[i]:old_off[i] + n])
                    self.exp_avg_sq[off:off + n].copy_(old_v[old_off[i]:old_off[i] + n])
        self.param_arena, self.grad_arena, self._offsets = arena, grads, offsets
        self._arena_clean, self._write_epoch = True, ops.GRAD_WRITE_EPOCH[0]
        if not self._hooked:           # torch's own gradient accumulation (per-layer autograd path) dirties the arena
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._on_accumulate)
            self._hooked = True

    def _on_accumulate(self, _param):
        self._arena_clean = False

    def mark_dirty(self):
        """a backward pass wrote into the gradient arena without going through torch's accumulation (fused.py)"""
        self._arena_clean = False

    def ensure_arena(self):
        if not self._arena_valid():
            self._build_arena()
        return self.param_arena

    # -- the device status word (module docstring: "Guard slot") ---------------------------------
    def status_words(self):
        """int32 view of the guard slot on the arena's device: [0] = sticky status of the passes, [4] = skipped updates"""
        self.ensure_arena()
        return self.guard

    def reduce_view(self):
        """what a data-parallel step SUM all-reduces: the gradient arena and, behind it, the status word"""
        self.ensure_arena()
        return self._grad_store[:self.grad_arena.numel() + 4]

    def take_status(self):
        """-> (status bits, skipped updates) read from the device (ONE sync) and, when set, cleared: the step counter goes
        back by the updates the kernel skipped, so the next applied update continues the bias corrections where the last
        applied one left them."""
        if self.guard is None:
            return 0, 0
        words = self.guard.tolist()
        bits, skipped = words[0] & 0xffffffff, words[4]
        if bits or skipped:
            self.guard.zero_()
            self.step_count -= skipped
        return bits, skipped

    # -- torch.optim.Optimizer surface used by the trainer --------------------------------------
    def zero_grad(self, set_to_none=False):
        self.ensure_arena()
        clean = self._arena_clean and self._write_epoch == ops.GRAD_WRITE_EPOCH[0]
        if not (clean and not torch.cuda.is_current_stream_capturing()):
            self.grad_arena.zero_()    # (a step being captured always records the memset: replays must not depend on what
        self._arena_clean = False      # ran before them; and whoever follows may write gradients without telling us)
        for p, off in zip(self.params, self._offsets):     # re-attach views dropped by zero_grad(None) users
            if p.grad is None or p.grad.data_ptr() != self.grad_arena.data_ptr() + 4 * off:
                p.grad = self.grad_arena[off:off + p.numel()].view(p.shape)

    @torch.no_grad()
    def step(self):
        self.ensure_arena()
        for p, off in zip(self.params, self._offsets):
            if p.grad is not None and p.grad.data_ptr() != self.grad_arena.data_ptr() + 4 * off:
                self.grad_arena[off:off + p.numel()].copy_(p.grad.reshape(-1))
                p.grad = self.grad_arena[off:off + p.numel()].view(p.shape)
        self.step_count += 1
        ops.adam_step(self.param_arena, self.grad_arena, self.exp_avg, self.exp_avg_sq, self.step_count, self.lr,
                      self.betas[0], self.betas[1], self.eps, self.grad_scale, zero_grad=self.zero_grads_in_step,
                      status=self.guard)
        self._arena_clean = self.zero_grads_in_step
        self._write_epoch = ops.GRAD_WRITE_EPOCH[0]

    def state_dict(self):
        return {'step': self.step_count, 'lr': self.lr, 'betas': self.betas, 'eps': self.eps,
                'exp_avg': None if self.exp_avg is None else self.exp_avg.clone(),
                'exp_avg_sq': None if self.exp_avg_sq is None else self.exp_avg_sq.clone()}

    def load_state_dict(self, state):
        self.step_count, self.lr = int(state['step']), float(state['lr'])
        self.betas, self.eps = tuple(state['betas']), float(state['eps'])
        if state.get('exp_avg') is not None:
            self.ensure_arena()
            self.exp_avg.copy_(state['exp_avg'])
            self.exp_avg_sq.copy_(state['exp_avg_sq'])
