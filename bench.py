#!/usr/bin/env python3
"""Throughput of the AR-VAE training step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1]): dSprites AR-VAE, per-GPU batch 512, fp32,
beta=4, gamma=10, delta=1, reg_dim=(1..5).  One step = zero_grad -> loss_and_acc_for_batch -> backward ->
Adam, on synthetic dSprites-shaped inputs that are resident in HBM before the timed region; the
reparameterisation noise is drawn on the device each step.  Prints ONE JSON line (rank 0).

Extra objects in the line:
  roofline      the dominant kernel of the step, timed live with HIP events on the launch stream in a
                separate instrumented pass (algorithmic FLOP / average launch duration vs the fp32 MFMA
                peak, or layer-boundary bytes vs the HBM peak for a memory-bound kernel)
  cpu_baseline  the CPU oracle (oracle/step.py, a port: the reference's Python cannot travel) timed on
                this box's host cores on a bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# roofline constants: /opt/skills/guides/MI355X_MICROARCH.md (chip-level parameters)
PEAK_F32_MFMA_TFLOPS = 157.3
PEAK_BF16_MFMA_TFLOPS = 2500.0
PEAK_HBM_GBS = 8000.0
# The 32-channel conv kernels run at fp32 accuracy on the bf16 MFMA: every operand is three bf16 terms and every
# multiply-add six partial products (conv32.hip), so their MFMA work is 6x the algorithmic FLOP at the bf16 rate.
BF16X3_PRODUCTS = 6
# algorithmic work per image per step, SURVEY.md section 8(d) (dSprites)
FLOP_PER_IMAGE = 73_708_544
BYTES_PER_IMAGE = 1_862_936
PARAM_BYTES_PER_STEP = 20_080_200

# algorithmic work of the step's link kernels, per image and launch: MACs (SURVEY.md section 8(d) per-layer
# table) and layer-boundary bytes (operand tensors read once + result written once, fp32)
KERNEL_WORK = {
    'down32_kernel<16>': (4_194_304, 4 * (32768 + 8192)), 'up32_kernel<16>': (4_194_304, 4 * (8192 + 32768)),
    'wgrad32_kernel<16>': (4_194_304, 4 * (8192 + 32768)),
    'down32_kernel<8>': (1_048_576, 4 * (8192 + 2048)), 'up32_kernel<8>': (1_048_576, 4 * (2048 + 8192)),
    'wgrad32_kernel<8>': (1_048_576, 4 * (2048 + 8192)),
    'down32_kernel<4>': (262_144, 4 * (2048 + 512)), 'up32_kernel<4>': (262_144, 4 * (512 + 2048)),
    'wgrad32_kernel<4>': (262_144, 4 * (512 + 2048)),
    'down_c1_kernel': (524_288, 4 * (4096 + 32768)), 'wgrad_c1_kernel': (524_288, 4 * (32768 + 4096)),
    # last decoder layer with the reconstruction term fused in: lo in; logits, d/dlogits out; image in
    'up_c1_kernel(recon)': (524_288, 4 * (32768 + 3 * 4096)), 'up_c1_kernel': (524_288, 4 * (32768 + 4096)),
}

REG_DIMS = (1, 2, 3, 4, 5)
BETA, GAMMA, DELTA = 4.0, 10.0, 1.0


class DspritesDataset:          # ImageVAETrainer sniffs the dataset's class name (reference image_vae_trainer.py:81-86)
    pass


def dsprites_shapes():
    from arvae_amd.image_vae import DspritesVAE
    return {k: tuple(v.shape) for k, v in DspritesVAE().state_dict().items()}


class MorphoMnistDataset:
    pass


class FolkDataset:              # what MeasureVAE / MeasureVAETrainer read from the reference's FolkNBarDataset
    class_name = '4by4_FolkNBarDataset_1_'
    n_bars = 1

    def __init__(self):
        from arvae_amd import synthetic as syn
        self.index2note_dicts, self.note2index_dicts = syn.measure_vocabulary()

    def __repr__(self):
        return self.class_name


def build_side_workload(kind, device, batch, graphs=False):
    """(step function, unit) for the secondary workloads (not the headline metric): BASELINE.json configs[2], [4]."""
    from arvae_amd import synthetic as syn
    if kind == 'mnist':
        from arvae_amd.image_vae import MnistVAE
        from arvae_amd.image_vae_trainer import ImageVAETrainer
        model = MnistVAE()
        state = syn.synth_state({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=3, gain=0.7)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
        trainer = ImageVAETrainer(MorphoMnistDataset(), model, lr=1e-4, reg_type=('all',), reg_dim=(1, 2, 3, 4, 5, 6),
                                  beta=1.0, gamma=10.0, capacity=0.0, rand=0, delta=1.0)
        trainer.cuda()
        x, lab = syn.mnist_batch(batch, seed=4321)
        data = (torch.from_numpy(x).to(device), torch.from_numpy(lab).to(device))
    else:
        from arvae_amd.measure_vae import MeasureVAE
        from arvae_amd.measure_vae_trainer import MeasureVAETrainer
        ds = FolkDataset()
        model = MeasureVAE(ds, 10, 2, 2, 128, 0.5, 32, 2, 128, 0.5, False, 'folk')
        trainer = MeasureVAETrainer(ds, model, lr=1e-4, reg_type=('all',), reg_dim=(0, 1, 2, 3), beta=0.001, gamma=1.0,
                                    capacity=0.0, rand=0, delta=10.0)
        trainer.cuda()
        score = torch.from_numpy(syn.measure_batch(batch, seed=5)).to(device)
        data = (score, score)
        model.train()
        if graphs:                                               # forward + backward replayed from HIP graphs (graphed.py)
            from arvae_amd.graphed import GraphedStep
            graphed = GraphedStep(trainer, data)

            def gstep(i):
                loss, _ = graphed(data)
                trainer.step()
                return loss
            return gstep, 'measures/s'
    model.train()

    def step(i):
        trainer.zero_grad()
        loss, _ = trainer.loss_and_acc_for_batch(data, 0, i, True)
        loss.backward()
        trainer.step()
        return loss
    return step, ('images/s' if kind == 'mnist' else 'measures/s')


def build_trainer(device, world):
    from arvae_amd import synthetic as syn
    from arvae_amd.image_vae import DspritesVAE
    from arvae_amd.image_vae_trainer import ImageVAETrainer
    model = DspritesVAE()
    state = syn.synth_state({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=1)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()})
    trainer = ImageVAETrainer(DspritesDataset(), model, lr=1e-4, reg_type=('all',), reg_dim=REG_DIMS, beta=BETA,
                              gamma=GAMMA, capacity=0.0, rand=0, delta=DELTA)
    model.to(device)
    trainer.capacity = trainer.capacity.to(device)
    if world > 1:
        from arvae_amd.parallel import DataParallel
        DataParallel().attach(trainer)
    model.train()
    return trainer, state


# algorithmic FLOP and layer-boundary bytes per image / measure and training step
SIDE_WORK = {'mnist': (431466496.0, 3276816.0), 'measure': (93.7e6, 0.7e6)}


def side_cpu_baseline(kind, batch, budget_s=20.0):
    """the CPU oracle's training step on the secondary workloads, bounded to ~budget_s of CPU work."""
    from arvae_amd import synthetic as syn
    from oracle import step as o_step
    cores = torch.get_num_threads()
    if kind == 'mnist':
        from arvae_amd.image_vae import MnistVAE
        state = syn.synth_state({k: tuple(v.shape) for k, v in MnistVAE().state_dict().items()}, seed=3, gain=0.7)
        x, lab = syn.mnist_batch(batch, seed=4321)
        eps = syn.normal_noise((batch, 16), seed=1)
        run = lambda cur, adam, n: o_step.image_step('mnist', cur, x, lab, eps, (1, 2, 3, 4, 5, 6), 1.0, 10.0, 1.0,
                                                     adam_state=adam, step_no=n)
        unit, what = 'images/s', 'Morpho-MNIST AR-VAE'
    else:
        from oracle import attributes as o_attr
        from oracle import measure_vae as o_mvae
        state = syn.synth_state(o_mvae.shapes(), 4)
        score = syn.measure_batch(batch, seed=5)
        eps = syn.normal_noise((batch, 32), seed=1)
        attr = o_attr.attribute_labels(score, *syn.measure_tables())
        run = lambda cur, adam, n: o_step.measure_step(cur, score, eps, attr, (0, 1, 2, 3), 0.001, 1.0, 10.0, n % 2 == 0,
                                                       adam_state=adam, step_no=n)
        unit, what = 'measures/s', 'MeasureVAE (teacher forcing on alternate steps)'
    cur, adam, times, n = state, None, [], 0
    t_start = time.perf_counter()
    while True:
        t0 = time.perf_counter()
        res = run(cur, adam, n + 1)
        cur, adam = res['params'], res['adam']
        times.append(time.perf_counter() - t0)
        n += 1
        if n >= 3 and (time.perf_counter() - t_start > budget_s or n >= 40):
            break
    steady = sorted(times[1:])
    med = steady[len(steady) // 2]
    return {'value': batch / med, 'unit': unit, 'cores': cores, 'kind': 'port',
            'sample': f'{n} full training steps (first discarded) of the {what} at batch {batch}, fp32, PyTorch-CPU oracle, '
                      f'{cores} threads, median step {med * 1e3:.1f} ms'}


def cpu_baseline(batch, state, budget_s=20.0):
    """CPU oracle on the same workload, bounded to ~budget_s of CPU work."""
    from arvae_amd import synthetic as syn
    from oracle import step as o_step
    x, lab = syn.dsprites_batch(batch, seed=1234)
    eps = syn.normal_noise((batch, 10), seed=1)
    cores = torch.get_num_threads()
    cur, adam = state, None
    times = []
    t_start = time.perf_counter()
    n = 0
    while True:
        t0 = time.perf_counter()
        res = o_step.image_step('dsprites', cur, x, lab, eps, REG_DIMS, BETA, GAMMA, DELTA, adam_state=adam,
                                step_no=n + 1)
        cur, adam = res['params'], res['adam']
        times.append(time.perf_counter() - t0)
        n += 1
        if n >= 3 and (time.perf_counter() - t_start > budget_s or n >= 40):
            break
    steady = sorted(times[1:])
    med = steady[len(steady) // 2]
    return {'value': batch / med, 'unit': 'images/s', 'cores': cores, 'kind': 'port',
            'sample': f'{n} full training steps (first discarded) of the dSprites AR-VAE at batch {batch}, fp32, '
                      f'PyTorch-CPU oracle, {cores} threads, median step {med * 1e3:.1f} ms'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--batch', type=int, default=512, help='per-GPU batch')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--breakdown', action='store_true', help='print the per-kernel-family table to stderr')
    ap.add_argument('--force-dp', action='store_true',
                    help='run the data-parallel code path (RCCL all-gather + all-reduce) even with one rank')
    ap.add_argument('--no-graphs', action='store_true', help='measure workload: eager launches instead of HIP-graph replay')
    ap.add_argument('--workload', default='dsprites', choices=['dsprites', 'mnist', 'measure'],
                    help='dsprites = the headline metric (default); mnist / measure = secondary single-GPU timings')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} '
                         f'(WORLD_SIZE={world})')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the AR-VAE hot path has no CPU fallback')
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)
    use_dp = world > 1 or args.force_dp
    if use_dp:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
        dist.init_process_group('nccl', device_id=device)

    from arvae_amd import synthetic as syn

    if args.workload != 'dsprites':
        bsz = args.batch if args.batch != 512 else (1024 if args.workload == 'mnist' else 256)
        side_step, unit = build_side_workload(args.workload, device, bsz, graphs=args.workload == 'measure' and not args.no_graphs)
        for i in range(args.warmup):
            side_step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            loss = side_step(i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rate = bsz * args.steps / dt
        flop, byts = SIDE_WORK[args.workload]
        line = {'metric': f'training {unit} ({args.workload} AR-VAE, batch {bsz}) -- secondary workload',
                'value': rate, 'unit': unit, 'n_gpus': 1, 'steps': args.steps,
                'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps, 'higher_is_better': True,
                'dtype': 'f32' if args.workload == 'mnist' else 'f32 (GRU MFMAs: 3-term bf16 split, fp32-accurate)',
                'data': 'synthetic', 'final_loss': float(loss.detach()),
                'config': {'workload': args.workload, 'batch': bsz,
                           'launch': 'hip-graph replay of fwd+bwd' if args.workload == 'measure' and not args.no_graphs else 'eager'},
                # whole-step fractions of the datasheet roofs (SURVEY 8(d) algorithmic FLOP / layer-boundary bytes per unit)
                'step_roofline': {'flop_per_unit': flop, 'bytes_per_unit': byts,
                                  'flop_frac_fp32': rate * flop / (PEAK_F32_MFMA_TFLOPS * 1e12),
                                  'hbm_frac': rate * byts / (PEAK_HBM_GBS * 1e9)}}
        if not args.no_cpu_baseline:
            line['cpu_baseline'] = side_cpu_baseline(args.workload, bsz)
        print(json.dumps(line))
        return

    trainer, state = build_trainer(device, 2 if use_dp else 1)
    if use_dp:
        trainer.data_parallel.broadcast_parameters(trainer.model)
    b = args.batch
    x_np, lab_np = syn.dsprites_batch(b, seed=1234 + rank)
    x = torch.from_numpy(x_np).to(device)
    lab = torch.from_numpy(lab_np).to(device)

    def step(i):
        trainer.zero_grad()
        loss, _ = trainer.loss_and_acc_for_batch((x, lab), 0, i, True)
        loss.backward()
        trainer.step()
        return loss

    def fence():
        torch.cuda.synchronize()
        if use_dp:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dp:
        import torch.distributed as dist
        tmax = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax)
    final_loss = float(loss.detach())
    if not np.isfinite(final_loss):
        raise SystemExit(f'non-finite loss {final_loss}')

    # ---- instrumented pass: per-kernel device time from HIP events recorded by the library on the launch
    # stream after every kernel (arvae_profile_begin/_end); a separate pass so the timed region is untouched
    import ctypes
    from arvae_amd import _lib
    lib = _lib.load()
    prof_steps = 10
    lib.arvae_profile_begin(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    for i in range(prof_steps):
        step(i)
    buf = ctypes.create_string_buffer(1 << 16)
    lib.arvae_profile_end(buf, len(buf))
    prof = {}
    for line in buf.value.decode().splitlines():
        name, calls, ms = line.split('\t')
        macs, nbytes = KERNEL_WORK.get(name, (0, 0))
        prof[name] = dict(calls=int(calls), ms=float(ms), flop=2.0 * macs * b * int(calls),
                          bytes=float(nbytes) * b * int(calls))
    fence()

    if rank != 0:
        dist.destroy_process_group()
        return
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * b * args.steps / elapsed
    prof.pop('(gap)', None)          # launch gaps the library marks separately so that kernel times exclude them
    dom_name, dom = max(((k, v) for k, v in prof.items() if v['flop'] > 0), key=lambda kv: kv[1]['ms'])
    avg_ms = dom['ms'] / dom['calls']
    split = dom_name.startswith(('down32', 'up32', 'wgrad32')) and not os.environ.get('ARVAE_CONV32_FP32')
    mfma_peak = PEAK_BF16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
    mfma_work = dom['flop'] * (BF16X3_PRODUCTS if split else 1)
    mfma_tf = mfma_work / dom['calls'] / (avg_ms * 1e-3) / 1e12
    hbm_gbs = dom['bytes'] / dom['calls'] / (avg_ms * 1e-3) / 1e9
    # the roof this kernel sits closer to binds it
    if mfma_tf / mfma_peak >= hbm_gbs / PEAK_HBM_GBS:
        roof = {'bound': 'mfma', 'achieved': mfma_tf, 'peak': mfma_peak, 'unit': 'TFLOP/s', 'frac': mfma_tf / mfma_peak,
                'traffic': None}
        if split:
            roof['mfma_work'] = 'executed bf16 MFMA FLOP = 6 partial products x algorithmic FLOP (fp32-accurate split)'
            roof['fp32_equivalent_tflops'] = mfma_tf / BF16X3_PRODUCTS
    else:
        roof = {'bound': 'hbm', 'achieved': hbm_gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': hbm_gbs / PEAK_HBM_GBS,
                'traffic': None}
    roof['other_roof_frac'] = {'mfma': mfma_tf / mfma_peak, 'hbm_algorithmic_bytes': hbm_gbs / PEAK_HBM_GBS}
    # HBM traffic of that kernel from the PMC counters: collected in separate rocprofv3 --pmc passes of this
    # same command (FETCH_SIZE / WRITE_SIZE cannot share a pass) and committed under profiles/
    try:
        with open(os.path.join(ROOT, 'profiles', 'r1_pmc_traffic.json')) as f:
            roof['traffic'] = json.load(f)['kernels'][dom_name]['hbm_bytes_per_launch']
        roof['traffic_source'] = 'profiles/r1_pmc_traffic.json (rocprofv3 --pmc, 2*FETCH_SIZE + WRITE_SIZE, B=512)'
    except (OSError, KeyError, ValueError):
        roof['traffic'] = None
    if b != 512:
        roof['traffic'] = None
    roof.update({'kernel': dom_name, 'launches_per_step': dom['calls'] / prof_steps, 'avg_launch_us': avg_ms * 1e3,
                 'share_of_device_time': dom['ms'] / sum(v['ms'] for v in prof.values())})
    per_gpu = value / world
    line = {
        'metric': 'training images/sec (dSprites beta-VAE+AR, per-GPU batch 512)', 'value': value,
        'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32' if os.environ.get('ARVAE_CONV32_FP32') else 'f32 (conv MFMAs: 3-term bf16 split, 6 products, fp32-accurate)',
        'data': 'synthetic',
        'config': {'workload': 'dSprites AR-VAE full training step (fwd + bwd + Adam), 1x64x64 inputs, z=10, '
                               'reg_dim=(1,2,3,4,5), beta=4 gamma=10 delta=1',
                   'per_gpu_batch': b, 'global_batch': b * world, 'parallelism': f'dp{world}' + (' (forced DP path)' if args.force_dp and world == 1 else ''),
                   'images_per_sec_per_gpu': per_gpu, 'final_loss': final_loss},
        'roofline': roof,
        'step_roofline': {
            'flop_frac_fp32': per_gpu * FLOP_PER_IMAGE / (PEAK_F32_MFMA_TFLOPS * 1e12),
            'hbm_frac': per_gpu * (BYTES_PER_IMAGE + PARAM_BYTES_PER_STEP / b) / (PEAK_HBM_GBS * 1e9),
            'binding': 'HBM roof 4.2 M img/s; fp32-MFMA roof 2.13 M img/s (no longer binding: the conv layers run on '
                       'the bf16 MFMA at 6 products per multiply-add, a 5.6 M img/s roof)'},
    }
    if args.breakdown:
        tot = sum(v['ms'] for v in prof.values())
        for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['ms']):
            print(f"  {k:24s} {v['calls'] / prof_steps:6.1f} calls/step {v['ms'] / prof_steps * 1e3:9.1f} us/step "
                  f"{100 * v['ms'] / tot:5.1f}%  {v['flop'] / max(v['ms'], 1e-9) / 1e9:8.2f} TFLOP/s "
                  f"{v['bytes'] / max(v['ms'], 1e-9) / 1e6:8.1f} GB/s", file=sys.stderr)
        print(f'  device-time sum {tot / prof_steps * 1e3:.1f} us/step vs wall {ms_per_step * 1e3:.1f} us/step',
              file=sys.stderr)
    if world == 1 and not args.no_cpu_baseline:
        line['cpu_baseline'] = cpu_baseline(b, state)
    print(json.dumps(line), flush=True)
    if use_dp:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
