"""One of several processes that train on the SAME GPU at once (tests/test_hip_parity.py::test_two_processes_share_the_device,
::test_handoff_that_never_completes_raises_instead_of_hanging): dSprites AR-VAE at the headline batch through the fused HIP
path, `steps` times zero_grad + loss + backward on fixed inputs (no Adam step: every repetition must reproduce the first one
bit for bit), then the trainer's device status check.

    python tests/shared_device_worker.py <out.npz> <steps> <sync_dir | -> <expect_failure 0|1>

sync_dir: the process writes ready_<pid> there after its warm-up step and starts the timed loop when the file `go` appears
(so that the parent can line several of them up).  expect_failure 1: the process runs with ARVAE_MIDC_DROP_ARRIVAL in the
diagnostic library -- the first pass must come back (bounded poll), check_device_status() must raise, and the steps after
that (row kernels) must be right.  A failure exits non-zero; nothing here re-executes the process.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class DspritesDataset:
    pass


def epoch_mode(out, steps):
    """expect_failure 2: Trainer.loss_and_acc_on_epoch over `steps` batches of 512.  Under ARVAE_MIDC_DROP_ARRIVAL the first
    pass of the epoch fails; the update kernel withholds every update of that attempt, the trainer repeats the epoch on the
    row kernels: the weights and the epoch mean written to `out` must be those of an undisturbed run."""
    import numpy as np
    import torch
    from arvae_amd import synthetic as syn
    from arvae_amd.image_vae import DspritesVAE
    from arvae_amd.image_vae_trainer import ImageVAETrainer
    dev = torch.device('cuda:0')
    b = 512
    model = DspritesVAE()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    state = syn.synth_state(shapes, 1, 1.6)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    trainer = ImageVAETrainer(DspritesDataset(), model, lr=1e-3, reg_type=('all',), reg_dim=(1, 2, 3, 4, 5), beta=4.0,
                              gamma=10.0, capacity=0.0, rand=0, delta=1.0)
    trainer.cuda()
    model.train()
    batches = []
    for i in range(steps):
        x, lab = syn.dsprites_batch(b, seed=100 + i)
        batches.append((torch.from_numpy(x).to(dev), torch.from_numpy(lab).to(dev)))
    noise = [torch.from_numpy(syn.normal_noise((b, 10), seed=7 + i)) for i in range(steps)]

    class Loader(list):
        def __iter__(self):                                   # every attempt at the epoch draws the same eps
            for i, item in enumerate(list.__iter__(self)):
                model.push_noise(noise[i])
                yield item
    mean_loss, mean_acc = trainer.loss_and_acc_on_epoch(Loader(batches), epoch_num=0, train=True)
    torch.cuda.synchronize()
    np.savez(out, mean_loss=mean_loss, mean_acc=mean_acc, step_count=trainer.optimizer.step_count,
             no_cluster=bool(trainer._fused.no_cluster), params=trainer.optimizer.param_arena.cpu().numpy())


def main():
    out, steps, sync_dir, expect_failure = sys.argv[1], int(sys.argv[2]), sys.argv[3], bool(int(sys.argv[4]))
    if int(sys.argv[4]) == 2:
        return epoch_mode(out, steps)
    import numpy as np
    import torch
    from arvae_amd import synthetic as syn
    from arvae_amd.image_vae import DspritesVAE
    from arvae_amd.image_vae_trainer import ImageVAETrainer
    dev = torch.device('cuda:0')
    b = 512
    model = DspritesVAE()
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    state = syn.synth_state(shapes, 1, 1.6)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    trainer = ImageVAETrainer(DspritesDataset(), model, lr=1e-4, reg_type=('all',), reg_dim=(1, 2, 3, 4, 5), beta=4.0,
                              gamma=10.0, capacity=0.0, rand=0, delta=1.0)
    trainer.cuda()
    model.train()
    x, lab = syn.dsprites_batch(b, seed=1234)
    eps = syn.normal_noise((b, 10), seed=1)
    xt, lt, et = torch.from_numpy(x).to(dev), torch.from_numpy(lab).to(dev), torch.from_numpy(eps)

    def one():
        model.push_noise(et)
        trainer.zero_grad()
        loss, acc = trainer.loss_and_acc_for_batch((xt, lt), 0, 0, True)
        loss.backward()
        one.acc = acc
        return loss

    raised = ''
    opt = trainer.optimizer
    opt.ensure_arena()
    before = opt.param_arena.clone()
    t0 = time.time()
    first = one()
    torch.cuda.synchronize()
    first_seconds = time.time() - t0
    if expect_failure:
        trainer.step()                                        # Adam on the undefined gradients of the failed pass: must be a no-op
        trainer.step()                                        # ... for as long as the host has not looked (sticky word)
    skipped = 0
    try:
        trainer.check_device_status()
    except RuntimeError as e:
        raised = str(e)
        skipped = getattr(e, 'skipped', -1)
    if expect_failure:
        if not raised:
            print('the dropped arrival was not reported', file=sys.stderr)
            sys.exit(3)
        # the update kernel read the status word itself: weights bit-identical to before the dropped pass, moments untouched,
        # the gradient arena cleared for the next pass, both withheld updates counted and taken back off the step counter
        if not (torch.equal(opt.param_arena, before) and float(opt.exp_avg.abs().max()) == 0.0 and float(opt.exp_avg_sq.abs().max()) == 0.0
                and float(opt.grad_arena.abs().max()) == 0.0 and skipped == 2 and opt.step_count == 0
                and opt.status_words().tolist() == [0] * 8):
            print(f'the failed pass reached the optimizer state (skipped {skipped}, step_count {opt.step_count})', file=sys.stderr)
            sys.exit(7)
        if first_seconds > 30.0:
            print(f'the failing pass took {first_seconds:.1f} s', file=sys.stderr)
            sys.exit(4)
        first = one()                                         # the row kernels from here on
        torch.cuda.synchronize()
        trainer.check_device_status()
    elif raised:
        print(raised, file=sys.stderr)
        sys.exit(5)
    ref_loss = first.detach().clone()
    ref_grad = trainer.optimizer.grad_arena.clone()
    if sync_dir != '-':
        open(os.path.join(sync_dir, f'ready_{os.getpid()}'), 'w').close()
        deadline = time.time() + 120
        while not os.path.exists(os.path.join(sync_dir, 'go')):
            if time.time() > deadline:
                sys.exit(6)
            time.sleep(0.005)
    same = torch.ones((), dtype=torch.bool, device=dev)
    t0 = time.time()
    for _ in range(steps):
        loss = one()
        same &= (loss.detach() == ref_loss).all() & (trainer.optimizer.grad_arena == ref_grad).all()
    torch.cuda.synchronize()
    seconds = time.time() - t0
    if os.environ.get('ARVAE_MIDC_DUMP'):                     # diagnostic library: what the workgroups that gave up saw
        import ctypes
        from arvae_amd import _lib
        buf = (ctypes.c_uint32 * (1 + 64 * 8))()
        if _lib.load().arvae_debug_midc_failures(buf) == 0 and buf[0]:
            rows = [list(buf[1 + 8 * i:1 + 8 * i + 7]) for i in range(min(int(buf[0]), 64))]
            t0 = min(r[5] for r in rows)
            print(f'{buf[0]} give-ups; (code, block, xcc, target, seen, t - t0 [10 ns], ctr index):', file=sys.stderr)
            for r in sorted(rows, key=lambda r: (r[5] - t0) & 0xffffffff):
                print('   ', r[0], r[1], r[2] & 0xf, r[3], r[4], (r[5] - t0) & 0xffffffff, r[6] % 4096, file=sys.stderr)
    trainer.check_device_status()                             # raises (exit 1) if a hand-off gave up
    grads = {k: p.grad.detach().cpu().numpy() for k, p in model.named_parameters()}
    if expect_failure:                                        # and the next good gradient IS applied, as update number one
        one()
        trainer.step()
        torch.cuda.synchronize()
        trainer.check_device_status()
        if torch.equal(opt.param_arena, before) or opt.step_count != 1 or float(opt.exp_avg.abs().max()) == 0.0:
            print('the update after the recovery was withheld too', file=sys.stderr)
            sys.exit(8)
    np.savez(out, loss=float(ref_loss), acc=float(one.acc), same=bool(same), seconds=seconds, first_seconds=first_seconds, raised=raised,
             terms_keys=np.array(list(trainer.last_terms.keys())),
             terms_vals=np.array([np.nan if v is None else float(v) for v in trainer.last_terms.values()]),
             **{f'g/{k}': v for k, v in grads.items()})


if __name__ == '__main__':
    main()
