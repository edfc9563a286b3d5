"""One rank of the data-parallel MeasureVAE graph-replay test (tests/test_parallel_gpu.py): the rank's rows of a fixed batch go
through ONE training step twice -- eagerly and replayed from a HIP graph that holds the step's collective (the library's RCCL
all-gather, recorded like a kernel) -- with arvae_amd.parallel attached; rank 0 saves both results.

    python tests/dp_measure_worker.py <rank> <world> <port> <out.npz> <batch_total> [<repeats> [<path>]]

repeats > 1: capture + replay that many times in this process (the stress loop of tools/dp_replay_loop.sh).
path: 'executor' (default: the whole-model executor, arvae_measure_vae_forward / _finish / _backward around one grouped all-gather)
or 'layers' (the per-layer autograd path with the all-gather between encoder and regulariser).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class FolkDataset:
    class_name = '4by4_FolkNBarDataset_1_'
    n_bars = 1

    def __init__(self):
        from arvae_amd import synthetic as syn
        self.index2note_dicts, self.note2index_dicts = syn.measure_vocabulary()


def main():
    rank, world, port, out, b_total = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
    repeats = int(sys.argv[6]) if len(sys.argv) > 6 else 1
    path = sys.argv[7] if len(sys.argv) > 7 else 'executor'
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import numpy as np
    import torch
    from arvae_amd import parallel
    from arvae_amd import synthetic as syn
    from arvae_amd.graphed import GraphedStep
    from arvae_amd.measure_vae import MeasureVAE
    from arvae_amd.measure_vae_trainer import MeasureVAETrainer
    from arvae_amd.parallel import DataParallel
    from oracle import measure_vae as o_mvae
    dev = torch.device('cuda', rank % torch.cuda.device_count())       # (transport 'staged': ranks may share a device)
    torch.cuda.set_device(dev)
    comm = parallel.connect(rank, world, dev)
    try:
        ds = FolkDataset()
        state = syn.synth_state(o_mvae.shapes(), 4)
        model = MeasureVAE(ds, 10, 2, 2, 128, 0.0, 32, 2, 128, 0.0, False, 'folk')
        model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
        trainer = MeasureVAETrainer(ds, model, lr=1e-4, reg_type=('all',), reg_dim=(0, 1, 2, 3), beta=0.001, gamma=1.0,
                                    capacity=0.0, rand=0, delta=10.0)
        trainer.cuda()
        trainer.use_fused_step = path == 'executor'
        dp = DataParallel(comm=comm).attach(trainer)
        assert (trainer._fused_binding() is not None) == (path == 'executor')
        model.train()
        model.decoder.teacher_forcing_prob = 2.0
        bl = b_total // world
        sl = slice(rank * bl, (rank + 1) * bl)
        score = torch.from_numpy(syn.measure_batch(b_total, seed=5)[sl]).to(dev)
        eps = torch.from_numpy(syn.normal_noise((b_total, 32), seed=1)[sl]).to(dev)
        res = {}
        try:
            model.encoder.static_eps = eps
            # eager data-parallel step
            trainer.zero_grad()
            loss, acc = trainer.loss_and_acc_for_batch((score, score), 0, 0, True)
            trainer.backward(loss)
            dp.reduce_gradients(trainer.optimizer)
            res['eager'] = (float(dp.mean_scalar(loss.detach())), trainer.optimizer.grad_arena.clone() * trainer.optimizer.grad_scale)
            # the same step replayed from graphs (twice: a replay must not depend on what the capture left behind)
            variants = 0
            if dp.capturable:
                for _ in range(repeats):
                    graphed = GraphedStep(trainer, (score, score))
                    for _ in range(2):
                        loss_g, _ = graphed((score, score))
                        dp.reduce_gradients(trainer.optimizer)
                variants = len(graphed.graphs)
                res['replay'] = (float(dp.mean_scalar(loss_g.detach())), trainer.optimizer.grad_arena.clone() * trainer.optimizer.grad_scale)
            else:                                                    # a transport that cannot be captured: the eager step only
                res['replay'] = res['eager']
        finally:
            type(model.encoder).static_eps = None
            model.encoder.static_eps = None
        torch.cuda.synchronize()
        if rank == 0:
            # (the arena's order is the optimizer's business -- arena_parameters() -- : spans are looked up by parameter)
            where = {id(p): off for p, off in zip(trainer.optimizer.params, trainer.optimizer._offsets)}
            names = [k for k, _ in model.named_parameters()]
            sizes = [(where[id(p)], p.numel()) for _, p in model.named_parameters()]
            np.savez(out, world=dp.world_size, variants=variants, transport=type(comm).__name__, loss_eager=res['eager'][0], loss_replay=res['replay'][0],
                     grad_eager=res['eager'][1].cpu().numpy(), grad_replay=res['replay'][1].cpu().numpy(),
                     names=np.array(names), spans=np.array(sizes))
        comm.barrier()
    finally:
        comm.close()


if __name__ == '__main__':
    main()
