"""Data-parallel AR-VAE step: one process per GPU, RCCL over xGMI, the collectives owned by libarvae_hip.so.

The reference is single-process; this layer is new.  The minibatch is sharded
by rows; the model is replicated.  Per step there are exactly two exchanges:

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

Transport.  `LibraryComm` (the default on GPUs) issues every collective through the library's C-ABI (`arvae_comm_*`,
include/arvae_hip.h): plain RCCL calls on the launch stream, like the kernels around them.  There is no torch process
group in such a run, hence no watchdog thread next to a HIP-graph capture, no Work objects to poll, and the collectives
are RECORDED by a capture (ar-vae_amd/graphed.py: a data-parallel MeasureVAE step is one graph).  Ranks find each other
through the launcher's TCP store (MASTER_ADDR / MASTER_PORT, `torch.distributed.rendezvous('env://')`), used once to hand
out RCCL's unique id.  `TorchComm` wraps an initialised torch.distributed group instead: what the CPU tests run on (gloo),
and an alternative on GPUs (`ARVAE_DP_TRANSPORT=torch`); its collectives cannot be captured, so steps stay eager there.
`StagedComm` moves device tensors through a gloo group on the host (`ARVAE_DP_TRANSPORT=staged`): several ranks can then share
one GPU, which is how the world-2 tests of the HIP path run on one-GPU boxes.
"""
import ctypes
import os
import time

import torch

from . import _lib

def _seconds(var, default):
    try:
        return float(os.environ.get(var, default))
    except ValueError:
        return float(default)


# Deadlines of the multi-rank path (seconds; environment overrides for slow launchers).  Nothing in a data-parallel run waits
# without one: the ranks' rendezvous at the store, ncclCommInitRank (arvae_comm_init's timeout), and every host wait for the
# stream behind a collective (LibraryComm.wait_idle) -- a peer that died shows as RuntimeError on the survivors, which abort
# their communicator and exit non-zero, so that the launcher (torch.distributed.run, bench.py's spawn_ranks: both end the
# job when one rank fails) does not keep a hung job alive.
JOIN_TIMEOUT_S = _seconds('ARVAE_DP_JOIN_TIMEOUT', 180)          # all ranks present at the store / inside ncclCommInitRank
IDLE_TIMEOUT_S = _seconds('ARVAE_DP_TIMEOUT', 300)               # the stream behind a collective drains

_DTYPES = {torch.float32: 0, torch.float64: 1, torch.int64: 2, torch.uint8: 3}
_OPS = {'sum': 0, 'max': 1, 'min': 2}


class _StreamJoin:
    """what an asynchronous collective of LibraryComm returns: work enqueued on the stream that was current at the call;
    wait() makes the stream current THEN wait for it (same meaning as torch's Work.wait() for a device collective)"""

    def __init__(self):
        self.event = torch.cuda.Event()
        self.event.record(torch.cuda.current_stream())

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


def store_get(store, key, rank, timeout):
    """store.get(key) with a deadline: RuntimeError when nobody has set the key within `timeout` seconds"""
    deadline = time.monotonic() + timeout
    while True:
        try:
            if store.check([key]):
                return store.get(key)
        except RuntimeError:
            pass
        if time.monotonic() > deadline:
            raise RuntimeError(f'rank {rank}: rank 0 did not publish the RCCL id within {timeout:.0f} s')
        time.sleep(0.01)


def all_present(store, key, rank, world, timeout):
    """every rank adds itself to the counter `key` and waits until all `world` have: RuntimeError after `timeout` seconds"""
    store.add(key, 1)
    deadline = time.monotonic() + timeout
    while True:
        here = int(store.add(key, 0))
        if here >= world:
            return
        if time.monotonic() > deadline:
            raise RuntimeError(f'rank {rank}: only {here} of {world} ranks reached the communicator set-up within {timeout:.0f} s')
        time.sleep(0.01)


class CommInitError(RuntimeError):
    """arvae_comm_init failed or timed out.  After a TIME-OUT its helper thread is still inside ncclCommInitRank: the process must
    not go through a normal interpreter shutdown (HIP / RCCL static destructors beside that live thread can hang or crash) --
    `leave_after_comm_failure()` is how the CLIs and bench.py go."""


def leave_after_comm_failure(exc, code=5):
    """print the error, flush, and leave the process without running destructors (see CommInitError)"""
    import sys
    print(f'libarvae_hip: {exc}; leaving the process', file=sys.stderr, flush=True)
    sys.stdout.flush()
    os._exit(code)


_KEYS_USED = set()          # store keys this process has built a communicator under (a key's rendezvous counter is never reset)


class LibraryComm:
    """RCCL communicator owned through libarvae_hip.so (arvae_comm_*).  Collectives take contiguous device tensors and are
    enqueued on torch's CURRENT stream; nothing here synchronises except barrier()."""
    capturable = True

    def __init__(self, rank, world, store=None, device=None, key='arvae/comm/0'):
        if not torch.cuda.is_available():
            raise RuntimeError('LibraryComm needs a GPU: the collectives of the HIP path have no CPU fallback')
        self.lib = _lib.load()
        version = self.lib.arvae_comm_available()
        if version < 0:
            _lib.check(version, 'comm_available')
        self.rccl_version = version
        self.rank, self.world_size = int(rank), int(world)
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        torch.cuda.set_device(self.device)
        if self.world_size > 1:
            # one communicator per store key: the key's unique id and its 'present' counter stay in the store, so a second
            # rendezvous under the same key would pass the gate at once and read the previous id
            if key in _KEYS_USED:
                raise ValueError(f'store key {key!r} has been used for a communicator already: give every communicator of a job its own key')
            _KEYS_USED.add(key)
        nbytes = 128
        self.handle = None
        if self.rank == 0:
            buf = (ctypes.c_char * nbytes)()
            _lib.check(self.lib.arvae_comm_unique_id(buf), 'comm_unique_id')
            ident = bytes(buf)
            if self.world_size > 1:
                store.set(key, ident)
        else:
            ident = bytes(self._store_get(store, key))          # waits (with a deadline) until rank 0 has published it
        if len(ident) != nbytes:
            raise RuntimeError('bad RCCL unique id from the store')
        if self.world_size > 1:
            # every rank confirms that it holds the id and is about to enter ncclCommInitRank; nobody enters before all have
            # confirmed -- a rank that died on the way here fails the others at THIS deadline, with a message, instead of leaving
            # them inside RCCL
            self._all_present(store, key + '/present')
        handle = ctypes.c_void_p()
        try:
            _lib.check(self.lib.arvae_comm_init(ident, self.rank, self.world_size, int(JOIN_TIMEOUT_S * 1000), ctypes.byref(handle)),
                       'comm_init')
        except RuntimeError as e:
            raise CommInitError(str(e)) from e
        self.handle = handle
        self.store = store                                      # rank 0 hosts it: alive as long as the communicator
        self._self_test()

    def _store_get(self, store, key):
        return store_get(store, key, self.rank, JOIN_TIMEOUT_S)

    def _all_present(self, store, key):
        all_present(store, key, self.rank, self.world_size, JOIN_TIMEOUT_S)

    def _self_test(self):
        """three tiny collectives with known answers, once, right after the communicator is built: a mis-wired job (ranks on
        the wrong devices, a transport that moves nothing) fails HERE with a message, not as a wrong gradient later"""
        w, r = self.world_size, self.rank
        one = torch.full((4,), float(r + 1), device=self.device)
        self.all_reduce(one)
        gathered = torch.empty(4 * w, device=self.device)
        self.all_gather(gathered, torch.full((4,), float(r), device=self.device))
        sent = torch.full((4,), float(r + 3), device=self.device)
        self.broadcast(sent, src=0)
        self.wait_idle()
        want = torch.arange(w, dtype=torch.float32).repeat_interleave(4)
        if not (bool((one == w * (w + 1) / 2).all()) and torch.equal(gathered.cpu(), want) and bool((sent == 3.0).all())):
            raise RuntimeError(f'RCCL self-test failed on rank {r} of {w}: all_reduce {one.tolist()}, all_gather {gathered.tolist()}, '
                               f'broadcast {sent.tolist()}')
        self.check()

    # -- collectives ---------------------------------------------------------------------------------------------
    @staticmethod
    def _stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _typed(self, t):
        if not t.is_cuda or not t.is_contiguous():
            raise ValueError('collectives take contiguous device tensors')
        code = _DTYPES.get(t.dtype)
        if code is None:                                        # moved as bytes (all-gather / broadcast only)
            return t.view(torch.uint8), 3, False
        return t, code, True

    def all_gather(self, out, local):
        """out[W * n, ...] (rank-major) <- every rank's local[n, ...]"""
        if out.numel() != self.world_size * local.numel() or out.dtype != local.dtype:
            raise ValueError('all_gather: out must hold world_size x local')
        src, code, _ = self._typed(local)
        dst, _, _ = self._typed(out)
        _lib.check(self.lib.arvae_comm_all_gather(self.handle, src.data_ptr(), dst.data_ptr(), src.numel(), code,
                                                  self._stream()), 'comm_all_gather')
        return out

    def all_gather_many(self, pairs):
        """[(out, local), ...] as ONE RCCL launch"""
        _lib.check(self.lib.arvae_comm_group_begin(), 'comm_group_begin')
        try:
            for out, local in pairs:
                self.all_gather(out, local)
        finally:
            _lib.check(self.lib.arvae_comm_group_end(), 'comm_group_end')

    def all_reduce(self, t, op='sum'):
        buf, code, arithmetic = self._typed(t)
        if not arithmetic:
            raise ValueError(f'all_reduce: unsupported dtype {t.dtype}')
        _lib.check(self.lib.arvae_comm_all_reduce(self.handle, buf.data_ptr(), buf.numel(), code, _OPS[op], self._stream()),
                   'comm_all_reduce')
        return t

    def all_reduce_async(self, t, op='sum'):
        self.all_reduce(t, op)
        return _StreamJoin()

    def all_gather_async(self, out, local):
        self.all_gather(out, local)
        return _StreamJoin()

    def broadcast(self, t, src=0):
        buf, code, _ = self._typed(t)
        _lib.check(self.lib.arvae_comm_broadcast(self.handle, buf.data_ptr(), buf.numel(), code, int(src), self._stream()),
                   'comm_broadcast')
        return t

    def barrier(self):
        """every rank has reached this point and this device is idle"""
        token = torch.zeros(1, device=self.device, dtype=torch.float32)
        self.all_reduce(token)
        self.wait_idle()

    def wait_idle(self, timeout=None):
        """Host wait for everything enqueued on the current stream so far -- what torch.cuda.synchronize() does, except that it
        cannot wait forever behind a collective whose peer has died: an event is recorded and polled, the communicator's
        asynchronous error state is read BETWEEN the polls (RCCL reports a lost peer there), and a deadline bounds the rest.
        On either, the communicator is aborted (its kernels leave the stream) and RuntimeError is raised."""
        timeout = IDLE_TIMEOUT_S if timeout is None else timeout
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(self.device))
        deadline = time.monotonic() + timeout
        pause = 0.0
        while not done.query():
            try:
                self.check()
            except RuntimeError:
                self.abort()
                raise
            if time.monotonic() > deadline:
                self.abort()
                raise RuntimeError(f'rank {self.rank}: the stream did not drain within {timeout:.0f} s behind a collective '
                                   f'(a peer rank is gone or stuck); communicator aborted')
            time.sleep(pause)
            pause = min(0.002, pause + 0.0001)                  # the first polls spin: a step is a fraction of a millisecond
        self.check()

    def check(self):
        """raises once the communicator has failed asynchronously (a peer died mid-collective)"""
        if self.handle is not None:
            _lib.check(self.lib.arvae_comm_async_error(self.handle), 'communicator')

    def abort(self):
        """tear the communicator down without waiting for its peers (ncclCommAbort): after an error, before the process exits"""
        handle, self.handle = self.handle, None
        if handle is not None:
            self.lib.arvae_comm_abort(handle)

    def close(self):
        if self.handle is not None:
            self.wait_idle()
            handle, self.handle = self.handle, None
            _lib.check(self.lib.arvae_comm_destroy(handle), 'comm_destroy')


class TorchComm:
    """the same collectives over an initialised torch.distributed process group (gloo on CPU in tests/test_parallel_gloo.py,
    'nccl' = RCCL as an alternative on GPUs).  Not capturable: a step over this transport runs eagerly."""
    capturable = False

    def __init__(self, process_group=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed is not initialised')
        self.dist, self.group = dist, process_group
        self.world_size, self.rank = dist.get_world_size(process_group), dist.get_rank(process_group)
        on_gpu = dist.get_backend(process_group) == 'nccl'
        self.device = torch.device('cuda', torch.cuda.current_device()) if on_gpu else torch.device('cpu')
        self._owns_group = False

    def _op(self, op):
        return {'sum': self.dist.ReduceOp.SUM, 'max': self.dist.ReduceOp.MAX, 'min': self.dist.ReduceOp.MIN}[op]

    def all_gather(self, out, local):
        self.dist.all_gather_into_tensor(out, local, group=self.group)
        return out

    def all_gather_async(self, out, local):
        return self.dist.all_gather_into_tensor(out, local, group=self.group, async_op=True)

    def all_gather_many(self, pairs):
        for out, local in pairs:
            self.all_gather(out, local)

    def all_reduce(self, t, op='sum'):
        self.dist.all_reduce(t, op=self._op(op), group=self.group)
        return t

    def all_reduce_async(self, t, op='sum'):
        return self.dist.all_reduce(t, op=self._op(op), group=self.group, async_op=True)

    def broadcast(self, t, src=0):
        self.dist.broadcast(t, src=self.dist.get_global_rank(self.group, src) if self.group is not None else src, group=self.group)
        return t

    def barrier(self):
        self.dist.barrier(group=self.group)
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def check(self):
        pass

    def wait_idle(self, timeout=None):
        if torch.cuda.is_available() and self.device.type == 'cuda':
            torch.cuda.synchronize()

    def abort(self):
        pass

    def close(self):
        if self._owns_group and self.group is None and self.dist.is_initialized():
            self.dist.destroy_process_group()


class StagedComm(TorchComm):
    """Device tensors through a CPU process group (gloo): every collective copies to host memory, runs there and copies back,
    synchronising the stream.  Slow by construction and never the default; it exists so that the MULTI-RANK logic of the HIP
    path (row-block regulariser against gathered columns, gradient all-reduce + 1/W, shifted capacity, per-rank random
    streams, sharded loaders) can run with more ranks than the box has GPUs -- several ranks on one device --, which RCCL
    refuses.  tests/test_parallel_gpu.py runs its world-2 cases over it on one-GPU boxes."""
    capturable = False

    def __init__(self, process_group=None, device=None):
        super().__init__(process_group)
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)

    def _host(self, t):
        return t.detach().to('cpu')

    def all_gather(self, out, local):
        host = torch.empty(out.shape, dtype=out.dtype)
        self.dist.all_gather_into_tensor(host, self._host(local), group=self.group)
        out.copy_(host)
        return out

    def all_gather_async(self, out, local):
        self.all_gather(out, local)
        return _Done()

    def all_reduce(self, t, op='sum'):
        host = self._host(t)
        self.dist.all_reduce(host, op=self._op(op), group=self.group)
        t.copy_(host)
        return t

    def all_reduce_async(self, t, op='sum'):
        self.all_reduce(t, op)
        return _Done()

    def broadcast(self, t, src=0):
        host = self._host(t)
        self.dist.broadcast(host, src=src, group=self.group)
        t.copy_(host)
        return t


class _Done:
    def wait(self):
        pass


def _rendezvous_store(rank, world):
    """the launcher's key-value store: torch.distributed.run's agent store when there is one (TORCHELASTIC_USE_AGENT_STORE),
    else a TCP store rank 0 hosts at MASTER_ADDR:MASTER_PORT.  No process group is created."""
    from torch.distributed import rendezvous
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    store, got_rank, got_world = next(iter(rendezvous('env://', rank=rank, world_size=world)))
    return store


def connect(rank=None, world=None, device=None, transport=None, key='arvae/comm/0'):
    """-> LibraryComm (default) or TorchComm for this process's rank; RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the
    environment when not given.  Binds the process to its GPU first."""
    rank = int(os.environ.get('RANK', '0')) if rank is None else int(rank)
    world = int(os.environ.get('WORLD_SIZE', '1')) if world is None else int(world)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC only on this pool (RCCL needs it)
    if device is None:
        device = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
    torch.cuda.set_device(device)
    transport = transport or os.environ.get('ARVAE_DP_TRANSPORT', 'library')
    if transport == 'torch':
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if not dist.is_initialized():
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device(device))
        comm = TorchComm()
        comm._owns_group = True
        return comm
    if transport == 'staged':                                       # several ranks may share one device (tests)
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if not dist.is_initialized():
            dist.init_process_group('gloo', rank=rank, world_size=world)
        comm = StagedComm(device=device)
        comm._owns_group = True
        return comm
    if transport != 'library':
        raise ValueError(f'unknown data-parallel transport {transport!r}')
    store = _rendezvous_store(rank, world) if world > 1 else None
    return LibraryComm(rank, world, store, device, key=key)


def init_from_env():
    """Under torch.distributed.run (WORLD_SIZE > 1): bind this process to its GPU (LOCAL_RANK), join the job's RCCL
    communicator and return a DataParallel; otherwise None.  What the training CLIs call before they build the dataset and
    the model."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1 and os.environ.get('ARVAE_FORCE_DP', '0') != '1':      # ARVAE_FORCE_DP=1: the same code path on one rank (tests)
        return None
    if world <= 1:
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
    try:
        return DataParallel(comm=connect())
    except CommInitError as e:                                  # (never returns: see CommInitError)
        leave_after_comm_failure(e)


class DataParallel:
    def __init__(self, process_group=None, reg_fn=None, comm=None):
        """comm: a LibraryComm / TorchComm (`connect()`); without one, the initialised torch.distributed group
        `process_group` (None = the default group) is wrapped in a TorchComm."""
        self.comm = comm if comm is not None else TorchComm(process_group)
        self.world_size, self.rank = self.comm.world_size, self.comm.rank
        from . import ops
        if reg_fn is None:
            reg_fn = ops.reg_loss
        self._reg_fn = reg_fn
        self._dims_cache = {}
        self._pending = []                   # all-reduces of gradient buckets started during the backward pass (fused.py)
        self.remaining_buckets = None        # float ranges of the arena those do not cover
        ops.rng_set_rank(self.rank)          # per-rank eps / dropout streams ("RNG under DP")

    @property
    def capturable(self):
        """may a step with this object's collectives be captured into a HIP graph?"""
        return self.comm.capturable

    def attach(self, trainer):
        """make `trainer` data-parallel (its loss step gathers columns, its step() all-reduces); returns self"""
        from . import ops
        trainer.data_parallel = self
        ops.rng_set_rank(self.rank)
        return self

    def finish(self):
        """all ranks are done training: leave the job (after this, rank 0 may evaluate for as long as it likes -- nobody
        waits in a collective).  The communicator's error state is read first and while the last barrier drains: a rank
        that lost a peer raises here instead of waiting."""
        self.comm.check()
        self.comm.barrier()
        self.comm.close()

    def abort(self):
        """after an exception in the training loop: give up the communicator without waiting for the other ranks; the caller
        then exits non-zero and the launcher ends the job"""
        self.comm.abort()

    def broadcast_parameters(self, model, src=0):
        """Make every replica start from rank `src`'s weights."""
        for p in model.parameters():
            self.comm.broadcast(p.data, src=src)

    def all_agree(self, ok):
        """True iff `ok` holds on EVERY rank (one MIN all-reduce of a flag): how ranks take a decision together"""
        flag = torch.full((1,), 1.0 if ok else 0.0, dtype=torch.float32, device=self.comm.device)
        self.comm.all_reduce(flag, 'min')
        self.comm.wait_idle()                                    # bounded: a dead peer raises here instead of hanging .item()
        return bool(flag.item() > 0.5)

    def gather_columns(self, local, async_op=None, out=None):
        """(B_local, R) -> (W * B_local, R), rank-major row order.  With async_op the collective is only enqueued:
        -> (out, work); call work.wait() before the first kernel that reads `out` (it overlaps whatever is launched
        in between).  `out`: a caller-allocated result (e.g. allocated on another stream than the one enqueuing)."""
        local = local.contiguous()
        if out is None:
            out = torch.empty((self.world_size * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype,
                              device=local.device)
        if async_op:
            return out, self.comm.all_gather_async(out, local)
        self.comm.all_gather(out, local)
        return out if async_op is None else (out, None)

    def gather_many(self, locals_):
        """several (B_local, *) tensors -> their (W * B_local, *) gathers, one launch where the transport can"""
        locals_ = [t.contiguous() for t in locals_]
        outs = [torch.empty((self.world_size * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device) for t in locals_]
        self.comm.all_gather_many(list(zip(outs, locals_)))
        return outs

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
        self._pending.append(self.comm.all_reduce_async(arena[lo:hi]))

    def reduce_gradients(self, optimizer):
        """SUM all-reduce of the flat gradient arena; Adam then applies 1/W.  Buckets whose all-reduce was started during
        the backward pass (start_bucket) are joined here and only the ranges they do not cover are reduced now."""
        optimizer.ensure_arena()
        if self._pending:
            for lo, hi in self.remaining_buckets or ():
                self.comm.all_reduce(optimizer.grad_arena[lo:hi])
            self.comm.all_reduce(optimizer.reduce_view()[optimizer.grad_arena.numel():])    # the status word (optim.py)
            for work in self._pending:
                work.wait()                                      # the launch stream waits for the side stream's collectives
            self._pending, self.remaining_buckets = [], None
        else:
            # gradients + the sticky device status word behind them (optim.py "Guard slot"): a pass that failed on ONE rank
            # makes EVERY rank's update kernel skip this step, so the replicas stay identical
            self.comm.all_reduce(optimizer.reduce_view())
        optimizer.grad_scale = 1.0 / self.world_size

    def mean_scalar(self, value):
        """average a scalar tensor over ranks."""
        v = value.detach().clone().reshape(1)
        self.comm.all_reduce(v)
        return v / self.world_size

    def mean_stats(self, values, device):
        """host floats -> their means over ranks (epoch statistics: one float64 all-reduce)"""
        stats = torch.tensor(list(values), dtype=torch.float64, device=device)
        self.comm.all_reduce(stats)
        self.comm.wait_idle()                                    # once per epoch: the bounded wait + the communicator's error state
        return (stats / self.world_size).tolist()
