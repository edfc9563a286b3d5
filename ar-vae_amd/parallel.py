"""Data-parallel AR-VAE step: one process per GPU, RCCL (torch.distributed 'nccl') over xGMI.

The reference is single-process; this layer is new.  The minibatch is sharded
by rows; the model is replicated.  Per step there are exactly two collectives:

  1. all-gather of the regularised latent/label columns (2 * B_local * R floats per rank, ~20 KB): the
     attribute-regularisation loss averages over ALL N_global^2 pairs (utils/trainer.py:390-401), so
     each rank evaluates its row block against the gathered global columns.  The pair term is symmetric,
     hence the row-block gradient is already the full d(global loss)/d z_i: no gradient exchange.
  2. one SUM all-reduce of the flat gradient arena (2.0 MB for the dSprites model), after which Adam
     scales by 1/world_size.

Loss convention: rank r differentiates  L_r = recon_r + beta*|KL_r - c| + W * reg_rowblock_r ;
mean_r(L_r) equals the single-process loss on the concatenated batch and mean_r(grad L_r) equals its gradient.
For c = 0 (the default, train_image_vae.py:23) that holds as written, because KL_r >= 0.  For c != 0 the term
beta*|mean_r(KL_r) - c| is not shard-linear (utils/trainer.py:354-367): the ranks all-reduce
the scalar KL mean (4 bytes, a third collective) and each uses the shifted capacity c_r = c + KL_r - KL_global, so
that KL_r - c_r = KL_global - c: the value of the term is the global one on every rank and its gradient carries the
global sign (`shifted_capacity`).
"""
import torch
import torch.distributed as dist


def init_from_env():
    """Under torch.distributed.run (WORLD_SIZE > 1): bind this process to its GPU (LOCAL_RANK), join the RCCL process group
    and return a DataParallel; otherwise None.  What the training CLIs call before they build the dataset and the model."""
    import os
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1 and os.environ.get('ARVAE_FORCE_DP', '0') != '1':      # ARVAE_FORCE_DP=1: the same code path on one rank (tests)
        return None
    if world <= 1:
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
    local = int(os.environ.get('LOCAL_RANK', '0'))
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC only on this pool (RCCL needs it)
    device = torch.device('cuda', local)
    torch.cuda.set_device(device)
    if not dist.is_initialized():
        dist.init_process_group('nccl', device_id=device)
    return DataParallel()


class DataParallel:
    def __init__(self, process_group=None, reg_fn=None):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self.group = process_group
        self.world_size = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        from . import ops
        if reg_fn is None:
            reg_fn = ops.reg_loss
        self._reg_fn = reg_fn
        self._dims_cache = {}
        self.capture_splitter = None         # graphed.Segments while a step is being captured: collectives cut the capture
        self._pending = []                   # all-reduces of gradient buckets started during the backward pass (fused.py)
        self.remaining_buckets = None        # float ranges of the arena those do not cover
        ops.rng_set_rank(self.rank)          # per-rank eps / dropout streams ("RNG under DP")

    def attach(self, trainer):
        """make `trainer` data-parallel (its loss step gathers columns, its step() all-reduces); returns self"""
        from . import ops
        trainer.data_parallel = self
        ops.rng_set_rank(self.rank)
        return self

    def finish(self):
        """all ranks are done training: leave the process group (after this, rank 0 may evaluate for as long as it likes --
        nobody waits in a collective that the group's watchdog would time out)"""
        dist.barrier(group=self.group)
        if self.group is None:
            dist.destroy_process_group()

    def broadcast_parameters(self, model, src=0):
        """Make every replica start from rank `src`'s weights."""
        for p in model.parameters():
            dist.broadcast(p.data, src=src, group=self.group)

    def gather_columns(self, local, async_op=None, out=None):
        """(B_local, R) -> (W * B_local, R), rank-major row order.  With async_op the collective is only enqueued:
        -> (out, work); call work.wait() before the first kernel that reads `out` (it overlaps whatever is launched
        in between).  `out`: a caller-allocated result (e.g. allocated on another stream than the one enqueuing)."""
        local = local.contiguous()
        if out is None:
            out = torch.empty((self.world_size * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype,
                              device=local.device)
        if self.capture_splitter is not None and self.capture_splitter.capturing:
            # a step is being captured into HIP graphs (graphed.py): the collective is not recorded; the capture is cut here
            # and the collective runs eagerly between the two graphs, now and on every replay, on these same buffers
            self.capture_splitter.split(lambda: dist.all_gather_into_tensor(out, local, group=self.group))
            return out if async_op is None else (out, None)
        work = dist.all_gather_into_tensor(out, local, group=self.group, async_op=bool(async_op))
        return out if async_op is None else (out, work if async_op else None)

    def reg_loss(self, z, labels, dims, gamma, delta):
        """W * (row-block regularisation loss of this rank's samples against the global batch)."""
        key = (tuple(dims), z.device)
        idx = self._dims_cache.get(key)          # cached: a host -> device copy is not allowed while a step is being captured
        if idx is None:
            idx = self._dims_cache[key] = torch.as_tensor(list(dims), device=z.device, dtype=torch.long)
        z_loc = z.index_select(1, idx)                       # differentiable compaction (B_local, R)
        lab_loc = labels.index_select(1, idx).to(torch.float32)
        packed = self.gather_columns(torch.cat([z_loc.detach(), lab_loc], dim=1))
        r = len(dims)
        z_all, lab_all = packed[:, :r].contiguous(), packed[:, r:].contiguous()
        part = self._reg_fn(z_loc, lab_loc, tuple(range(r)), gamma, delta, z_cols=z_all, lab_cols=lab_all)
        return part * float(self.world_size)

    def shifted_capacity(self, kl_local, capacity):
        """c_r = c + KL_r - mean_r(KL_r) as a detached (1,) tensor: with it beta*|KL_r - c_r| equals the single-process
        beta*|KL_global - c| in value, and its gradient beta*sign(KL_global - c)*dKL_r averages to the global gradient."""
        kl_local = kl_local.detach().reshape(1)
        kl_global = self.mean_scalar(kl_local)
        return (capacity.detach().reshape(1).to(kl_local.dtype) + kl_local - kl_global)

    def start_bucket(self, arena, lo, hi):
        """enqueue the SUM all-reduce of arena[lo:hi] on the CURRENT stream without waiting for it (the fused backward calls
        this on its side stream, behind the event that says the bucket is final); reduce_gradients() joins it"""
        self._pending.append(dist.all_reduce(arena[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def reduce_gradients(self, optimizer):
        """SUM all-reduce of the flat gradient arena; Adam then applies 1/W.  Buckets whose all-reduce was started during
        the backward pass (start_bucket) are joined here and only the ranges they do not cover are reduced now."""
        optimizer.ensure_arena()
        if self._pending:
            for lo, hi in self.remaining_buckets or ():
                dist.all_reduce(optimizer.grad_arena[lo:hi], op=dist.ReduceOp.SUM, group=self.group)
            for work in self._pending:
                work.wait()                                      # the launch stream waits for the side stream's collectives
            self._pending, self.remaining_buckets = [], None
        else:
            dist.all_reduce(optimizer.grad_arena, op=dist.ReduceOp.SUM, group=self.group)
        optimizer.grad_scale = 1.0 / self.world_size

    def mean_scalar(self, value):
        """average a scalar tensor over ranks."""
        v = value.detach().clone().reshape(1)
        if self.capture_splitter is not None and self.capture_splitter.capturing:       # see gather_columns
            self.capture_splitter.split(lambda: dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group))
        else:
            dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group)
        return v / self.world_size
