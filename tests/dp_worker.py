"""One rank of the RCCL data-parallel parity tests (tests/test_parallel_gpu.py): runs ONE dSprites AR-VAE training step
on this rank's rows of a fixed global batch through the HIP path (fused or per-layer) with arvae_amd.parallel attached,
and lets rank 0 save the rank-averaged loss, the all-reduced gradient arena (times 1/W) and the updated weights.

    python tests/dp_worker.py <rank> <world> <port> <out.npz> <capacity> <fused 0|1> <batch_total>
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class DspritesDataset:
    pass


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    out, capacity, fused, b_total = sys.argv[4], float(sys.argv[5]), bool(int(sys.argv[6])), int(sys.argv[7])
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import numpy as np
    import torch
    from arvae_amd import parallel
    from arvae_amd import synthetic as syn
    from arvae_amd.image_vae import DspritesVAE
    from arvae_amd.image_vae_trainer import ImageVAETrainer
    from arvae_amd.parallel import DataParallel
    # 'staged' (a gloo group on the host under device tensors): ranks may share a device -- world 2 on a one-GPU box
    dev = torch.device('cuda', rank % torch.cuda.device_count())
    # ARVAE_DP_TRANSPORT: 'library' (default: RCCL through libarvae_hip.so, no torch process group) | 'torch'
    comm = parallel.connect(rank, world, dev)
    try:
        model = DspritesVAE()
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        state = syn.synth_state(shapes, 1, 1.6)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
        trainer = ImageVAETrainer(DspritesDataset(), model, lr=1e-4, reg_type=('all',), reg_dim=(1, 2, 3, 4, 5), beta=4.0,
                                  gamma=10.0, capacity=capacity, rand=0, delta=1.0)
        trainer.cuda()
        trainer.use_fused = fused
        dp = DataParallel(comm=comm).attach(trainer)
        dp.broadcast_parameters(model)
        model.train()
        x, lab = syn.dsprites_batch(b_total, seed=1234)
        eps = syn.normal_noise((b_total, 10), seed=12)
        bl = b_total // world
        sl = slice(rank * bl, (rank + 1) * bl)
        model.push_noise(torch.from_numpy(eps[sl]))
        trainer.zero_grad()
        loss, acc = trainer.loss_and_acc_for_batch((torch.from_numpy(x[sl]).to(dev), torch.from_numpy(lab[sl]).to(dev)), 0, 0, True)
        loss.backward()
        dp.reduce_gradients(trainer.optimizer)                 # what Trainer.step() does before Adam
        grads = {k: (p.grad.detach() * trainer.optimizer.grad_scale).cpu().numpy() for k, p in model.named_parameters()}
        trainer.optimizer.step()
        mean_loss = float(dp.mean_scalar(loss.detach()))
        mean_acc = float(dp.mean_scalar(acc.detach()))
        terms = {k: float(dp.mean_scalar(v)) for k, v in trainer.last_terms.items() if v is not None}
        torch.cuda.synchronize()
        trainer.check_device_status()                          # (a hand-off of the clustered latent block that gave up raises here)
        if rank == 0:
            np.savez(out, loss=mean_loss, acc=mean_acc, world=dp.world_size, transport=type(comm).__name__,
                     **{'term/' + k: v for k, v in terms.items()}, **{'grad/' + k: v for k, v in grads.items()},
                     **{'param/' + k: p.detach().cpu().numpy() for k, p in model.named_parameters()})
        comm.barrier()
    finally:
        comm.close()


if __name__ == '__main__':
    main()
