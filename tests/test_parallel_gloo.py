"""Data-parallel logic on CPU: two gloo ranks, each holding half of a batch, must reproduce the single-process
loss and gradient.  The DP layer (ar-vae_amd/parallel.py) is compute-agnostic; here the CPU oracle supplies the
math (tests may do that -- the product path never does), so this covers the N>1 collectives and scaling rules:
row-major all-gather of the regularised columns, W x row-block reg loss, SUM all-reduce + 1/W in Adam."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIMS = (1, 2, 3, 4, 5)
BETA, GAMMA, DELTA = 4.0, 10.0, 1.0


def _cpu_arena(n):
    """a real FlatAdam over one CPU parameter: the arena, its guard slot and reduce_view() are plain torch (only step() needs
    the HIP library)"""
    from arvae_amd.optim import FlatAdam
    opt = FlatAdam([torch.nn.Parameter(torch.zeros(n))])
    opt.ensure_arena()
    return opt


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, b_total, out_path, capacity):
    import sys
    sys.path.insert(0, ROOT)
    from arvae_amd import synthetic as syn
    from arvae_amd.parallel import DataParallel
    from oracle import image_vae as o_vae
    from oracle import losses as o_losses
    torch.set_num_threads(1)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        state = syn.synth_state(o_vae.DSPRITES_SHAPES, 1, 1.6)
        x, lab = syn.dsprites_batch(b_total, seed=1234)
        eps = syn.normal_noise((b_total, 10), seed=12)
        bl = b_total // world
        sl = slice(rank * bl, (rank + 1) * bl)
        p = {k: torch.from_numpy(v).clone().requires_grad_(True) for k, v in state.items()}
        xt, lt, et = torch.from_numpy(x[sl]), torch.from_numpy(lab[sl]), torch.from_numpy(eps[sl])
        dp = DataParallel(reg_fn=o_losses.reg_loss_row_block)
        logits, mu, sigma, z = o_vae.forward('dsprites', p, xt, et)
        recon = o_losses.bce_with_logits_per_batch(logits, xt)
        cap = 0.0
        if capacity != 0.0:                                    # |KL - c| needs the global KL mean (parallel.py)
            cap = dp.shifted_capacity(o_losses.kld_loss(mu, sigma, 1.0, 0.0), torch.tensor([capacity]))
        kld = o_losses.kld_loss(mu, sigma, BETA, cap).reshape(())
        reg = dp.reg_loss(z, lt, DIMS, GAMMA, DELTA)           # = W * row-block share
        loss = recon + kld + reg
        loss.backward()
        flat = torch.cat([p[k].grad.reshape(-1) for k in state])
        opt = _cpu_arena(flat.numel())
        opt.grad_arena[:flat.numel()].copy_(flat)
        dp.reduce_gradients(opt)
        reduced = opt.grad_arena.clone()
        clean = int(opt.status_words()[0]) == 0                # nobody failed: the status word behind the gradients stays 0
        # a pass that fails on ONE rank (here: the last) must stop EVERY rank's update: the sticky word rides behind the
        # gradients through the same SUM all-reduce (optim.py "Guard slot") and stays set through later steps
        if rank == world - 1:
            opt.status_words()[0] = 0x40000000 | (2 << 20)      # ARVAE_STATUS_HANDOFF_BWD
        dp.reduce_gradients(opt)
        dp.reduce_gradients(opt)
        flagged = clean and dp.all_agree(int(opt.status_words()[0]) != 0) and int(opt.status_words()[4]) == 0
        bits, skipped = opt.take_status()
        flagged = flagged and bits != 0 and skipped == 0 and int(opt.status_words()[0]) == 0
        opt.grad_arena.copy_(reduced)
        mean_loss = float(dp.mean_scalar(loss))
        gathered = dp.gather_columns(torch.full((bl, 2), float(rank)))
        # the rest of the transport's surface (what the HIP path calls on its communicator: several gathers as one call, a
        # decision taken by all ranks, epoch statistics, a broadcast)
        many = dp.gather_many([torch.full((bl, 3), float(rank)), torch.full((bl, 1), float(10 + rank))])
        agree = (dp.all_agree(True), dp.all_agree(rank == 0))
        stats = dp.mean_stats([float(rank), 2.0 * rank], 'cpu')
        sent = torch.full((5,), float(rank + 7))
        dp.comm.broadcast(sent, src=0)
        if rank == 0:
            np.savez(out_path, grad=(opt.grad_arena[:flat.numel()] * opt.grad_scale).numpy(), loss=mean_loss,
                     gathered=gathered.numpy(), scale=opt.grad_scale, many0=many[0].numpy(), many1=many[1].numpy(),
                     agree=np.array(agree), stats=np.array(stats), sent=sent.numpy(), capturable=dp.capturable,
                     flagged=flagged)
    finally:
        dist.destroy_process_group()


# KL mean of this batch: 3.686 (shards 3.564 / 3.807): c = 3.7 lies BETWEEN the shard means, so the local signs disagree
# and only the all-reduced mean gives the single-process gradient; c = 30 flips the sign on every rank
@pytest.mark.parametrize('world,capacity', [(2, 0.0), (2, 3.7), (2, 30.0), (4, 0.0), (4, 3.7), (8, 0.0)])
def test_ranks_step_equals_single_process(tmp_path, world, capacity):
    from arvae_amd import synthetic as syn
    from oracle import image_vae as o_vae
    from oracle import step as o_step
    b_total = 16
    out = str(tmp_path / 'dp.npz')
    mp.spawn(_worker, args=(world, _free_port(), b_total, out, capacity), nprocs=world, join=True)
    got = np.load(out)
    state = syn.synth_state(o_vae.DSPRITES_SHAPES, 1, 1.6)
    x, lab = syn.dsprites_batch(b_total, seed=1234)
    eps = syn.normal_noise((b_total, 10), seed=12)
    ref = o_step.image_step('dsprites', state, x, lab, eps, DIMS, BETA, GAMMA, DELTA, capacity=capacity)
    want = np.concatenate([ref['grads'][k].ravel() for k in state])
    assert got['scale'] == pytest.approx(1.0 / world)
    np.testing.assert_allclose(got['loss'], ref['terms']['loss'], rtol=1e-5)
    assert np.linalg.norm(got['grad'] - want) <= 1e-4 * np.linalg.norm(want)
    # all_gather_into_tensor is rank-major: rows of rank 0 first
    bl = b_total // world
    ranks = np.repeat(np.arange(world), bl)
    assert (got['gathered'] == ranks[:, None]).all()
    assert (got['many0'] == ranks[:, None]).all() and (got['many1'] == 10 + ranks[:, None]).all()
    assert got['agree'].tolist() == [True, False]               # one rank saying no is everybody's no
    np.testing.assert_allclose(got['stats'], [(world - 1) / 2, world - 1.0])
    assert bool(got['flagged'])                                 # one rank's device status reaches every rank's update kernel
    assert (got['sent'] == 7).all() and not bool(got['capturable'])   # torch.distributed collectives are not captured into graphs


def _rng_worker(rank, world, port, out_path):
    import sys
    sys.path.insert(0, ROOT)
    from arvae_amd import ops
    from arvae_amd.parallel import DataParallel
    from oracle import philox
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        ops.rng_reseed(7)                                       # every rank seeds alike, as the trainers' constructors do
        DataParallel(reg_fn=lambda *a, **k: None)
        seed = ops.rng_seed()
        eps = torch.from_numpy(philox.normal(640, seed, offset=ops.rng_next_offset()))
        mask = torch.from_numpy(philox.keep_mask(640, 0.5, seed, offset=ops.rng_next_offset()).astype(np.float32))
        rows = [torch.zeros(2, 640) for _ in range(world)]
        dist.all_gather(rows, torch.stack([eps, mask]))
        seeds = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(seeds, torch.tensor([seed & 0x7FFFFFFFFFFFFFFF]))
        if rank == 0:
            np.savez(out_path, rows=torch.stack(rows).numpy(), seeds=torch.cat(seeds).numpy())
    finally:
        dist.destroy_process_group()


def test_ranks_draw_different_noise_and_masks(tmp_path):
    """advisor finding, round 2: all ranks seed torch alike (torch.manual_seed(rand)), so the library's Philox key must carry
    the rank or a global batch of W*B rows holds only B distinct eps / keep-mask rows.  Two gloo ranks select their stream
    as a data-parallel run does and draw (with the oracle's restatement of the device generator, keyed by ops.rng_seed()):
    rank 0 keeps the single-process stream, rank 1 draws something else."""
    from oracle import philox
    out = str(tmp_path / 'rng.npz')
    mp.spawn(_rng_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    assert got['seeds'][0] == 7 and got['seeds'][1] != 7
    np.testing.assert_array_equal(got['rows'][0, 0], philox.normal(640, 7, offset=0))
    assert np.abs(got['rows'][0, 0] - got['rows'][1, 0]).max() > 0.5              # eps differ
    assert 0.3 < (got['rows'][0, 1] != got['rows'][1, 1]).mean() < 0.7            # keep-masks differ like independent coins


@pytest.mark.parametrize('launcher', ['torchrun', 'env'])
def test_ranks_meet_at_the_launchers_store(launcher):
    """The HIP path's data-parallel runs have no torch process group: the ranks meet once at the launcher's key-value store to
    hand out RCCL's unique id (parallel.connect -> parallel._rendezvous_store).  'torchrun': under
    `python -m torch.distributed.run --master-addr 127.0.0.1` (what the driver's multi-GPU bench uses) the store is the
    agent's (TORCHELASTIC_USE_AGENT_STORE); 'env': with RANK / WORLD_SIZE / MASTER_* alone (bench.py's own rank launcher) rank 0
    hosts it."""
    import subprocess
    import sys
    worker = os.path.join(ROOT, 'tests', 'rdzv_worker.py')
    port = _free_port()
    if launcher == 'torchrun':
        r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr',
                            '127.0.0.1', '--master-port', str(port), worker], capture_output=True, text=True, timeout=300)
        out = r.stdout + r.stderr
        assert r.returncode == 0, out[-3000:]
        assert 'rank 0 of 2 ok agent_store=True' in out and 'rank 1 of 2 ok agent_store=True' in out
    else:
        procs = [subprocess.Popen([sys.executable, worker], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                                  env=dict(os.environ, RANK=str(rk), WORLD_SIZE='2', LOCAL_RANK=str(rk), MASTER_ADDR='127.0.0.1',
                                           MASTER_PORT=str(port))) for rk in range(2)]
        outs = [p.communicate(timeout=300)[0] for p in procs]
        for rk, (p, o) in enumerate(zip(procs, outs)):
            assert p.returncode == 0 and f'rank {rk} of 2 ok' in o, o[-3000:]


def test_rendezvous_waits_are_bounded():
    """The two waits in front of ncclCommInitRank (parallel.LibraryComm.__init__) end by themselves: a rank whose peers never
    arrive gets RuntimeError after the deadline instead of waiting in the store (or, later, inside RCCL) forever."""
    import socket
    from datetime import timedelta
    from torch.distributed import TCPStore
    from arvae_amd import parallel
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    store = TCPStore('127.0.0.1', port, 1, True, timeout=timedelta(seconds=30), wait_for_workers=False)
    import time
    t0 = time.monotonic()
    with pytest.raises(RuntimeError, match='only 1 of 2 ranks'):
        parallel.all_present(store, 'k/present', 0, 2, 0.5)
    with pytest.raises(RuntimeError, match='did not publish'):
        parallel.store_get(store, 'k/id', 1, 0.5)
    assert time.monotonic() - t0 < 10
    parallel.all_present(store, 'k/alone', 0, 1, 0.5)                 # complete groups pass
    store.set('k/id', b'x' * 128)
    assert bytes(parallel.store_get(store, 'k/id', 1, 0.5)) == b'x' * 128
