#!/usr/bin/env python3
"""Throughput of the AR-VAE training step on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU over RCCL -- the library's own communicator (arvae_comm_*: RCCL calls on the launch stream; the
ranks meet once at the launcher's TCP store to hand out RCCL's unique id; no torch process group).  Launched under
torch.distributed.run the ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment; launched plainly (`python bench.py --gpus 8`) this process starts the N
ranks itself as fresh child processes -- before anything here touches the GPU -- and prints rank 0's line.

Workload (BASELINE.json configs[1]): dSprites AR-VAE, per-GPU batch 512, fp32, beta=4, gamma=10,
delta=1, reg_dim=(1..5).  One step = zero_grad -> loss_and_acc_for_batch -> backward -> (gradient all-reduce) -> Adam,
on synthetic dSprites-shaped inputs resident in HBM before the timed region; the reparameterisation noise is drawn on
the device each step.  `--workload mnist|measure` times BASELINE.json configs[2] / configs[4] the same way.

Timing: W warm-up steps, then regions of EXACTLY K steps, each bracketed by barrier + torch.cuda.synchronize() on both
sides and timed by the host clock (max over ranks) and by HIP events on the launch stream.  Regions repeat until
>= 1 s has been timed (>= 3 regions); `ms_per_step` / `value` are the MEDIAN region, the spread is under `timing`.

Prints ONE JSON line (rank 0).  Extra objects in the line:
  roofline      the dominant kernel of the step, timed live with HIP events on the launch stream in a separate
                instrumented pass (arvae_profile_begin/_end): algorithmic bytes (or executed MFMA FLOP) per launch /
                average launch duration against the HBM (or MFMA) peak; `rocprof_names` = the kernel names a
                rocprofv3 --kernel-trace of this command shows for that label
  cpu_baseline  the CPU oracle (oracle/step.py, a port: the reference's Python cannot travel) timed on this box's host
                cores on a bounded sample of the same workload (rank 0, N = 1 only), all cores and 8 threads
  secondary     (default run, N = 1) the Morpho-MNIST B=1024 and MeasureVAE B=256 steps, same timing rules; N > 1: the MeasureVAE
                step, 256 measures per rank (BASELINE.json configs[4], weak scaling)
  dp            (N > 1 or --force-dp) HIP-event cost of the step's gradient all-reduce and grouped all-gather in isolation,
                and (N > 1) ms per step with the collectives on the launch stream vs the overlap schedule
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
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
# The 32-channel conv kernels (dSprites) run at fp32 accuracy on the fp16 MFMA: every operand is two scaled fp16 terms and every
# multiply-add three partial products (conv32_common.h), so their MFMA work is 3x the algorithmic FLOP at the fp16 rate (the
# same dense peak as bf16).  The wide conv / GRU / rows-GEMM kernels of the secondary workloads still run the three-term bf16
# split: six partial products.
F16X2_PRODUCTS = 3
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
    # round 6: conv2 and conv3 of the forward pass as one launch (a workgroup runs the 8x8 layer on the images whose 16x16 layer it
    # has just stored): the layer-boundary bytes of both layers (SURVEY 8(d) counts the 16x16 tensor written and read)
    'chain(down32<16> + down32<8>)': (4_194_304 + 1_048_576, 4 * (32768 + 8192) + 4 * (8192 + 2048)),
    'wgrad32_kernel<8>': (1_048_576, 4 * (2048 + 8192)),
    'down32_kernel<4>': (262_144, 4 * (2048 + 512)), 'up32_kernel<4>': (262_144, 4 * (512 + 2048)),
    'wgrad32_kernel<4>': (262_144, 4 * (512 + 2048)),
    'down_c1_kernel': (524_288, 4 * (4096 + 32768)), 'wgrad_c1_kernel': (524_288, 4 * (32768 + 4096)),
    # last decoder layer with the reconstruction term fused in: lo in; logits, d/dlogits out; image in
    'up_c1_kernel(recon)': (524_288, 4 * (32768 + 3 * 4096)), 'up_c1_kernel': (524_288, 4 * (32768 + 4096)),
    # A layer's two backward products in one launch: SURVEY 8(d)'s layer-boundary bytes of a layer's BACKWARD -- the upstream
    # gradient read ONCE, the saved input once, the input gradient written once.  (Both halves of a paired launch stream the
    # upstream gradient on disjoint CUs: the second read is TRAFFIC -- roofline.traffic, profiles/*_pmc_traffic.json --, not work.)
    'pair4(down32 + wgrad32)': (2 * 262_144, 4 * (2048 + 512 + 512)), 'pair4(up32 + wgrad32)': (2 * 262_144, 4 * (512 + 2048 + 2048)),
    'pair_c1(down_c1 + wgrad_c1)': (2 * 524_288, 4 * (4096 + 32768 + 32768)),
    # the first layer's weight gradient with the grouped Linear weight gradients riding in its grid
    'pair(wgrad_c1 + dense_wgrad_batch)': (524_288 + 400_896, 4 * (32768 + 4096) + 4 * 3102, 4 * 400_896),
    'pair(down32<16> + wgrad32<16>)': (2 * 4_194_304, 4 * (32768 + 8192 + 8192)), 'pair(up32<16> + wgrad32<16>)': (2 * 4_194_304, 4 * (8192 + 32768 + 32768)),
    'pair(down32<8> + wgrad32<8>)': (2 * 1_048_576, 4 * (8192 + 2048 + 2048)), 'pair(up32<8> + wgrad32<8>)': (2 * 1_048_576, 4 * (2048 + 8192 + 8192)),
    # round 5: conv2's backward with conv1's weight gradient in the Up half's store waves (the data gradient in between is never
    # stored): the layer-boundary bytes of BOTH layers' backward passes (SURVEY 8(d) counts the gradient between them written and read)
    'pair(up32<16> + wgrad32<16> + wgrad_c1)': (2 * 4_194_304 + 524_288, 4 * (8192 + 32768 + 32768) + 4 * (32768 + 4096)),
    # the first encoder layer with the step's weight preparation riding in its grid; the decoder's first convolution with
    # the regulariser's workgroups riding in its grid
    'down_c1_kernel(+ weight prep)': (524_288, 4 * (4096 + 32768), 8_000_000), 'up32_kernel<4>(+ reg_loss)': (262_144, 4 * (512 + 2048)),
    # the latent block (six Linear layers + heads, one launch per pass): MACs of SURVEY 8(d)'s per-layer table; bytes =
    # every layer's input and output once (forward), upstream gradient + activation derivative + input gradient (backward),
    # upstream gradient + saved input (weight gradients); the third number is bytes per LAUNCH that do not scale with the
    # batch: the 1.6 MB of fp32 matrices read (forward, backward) or written (weight gradients) once
    'mid_forward_kernel': (400_896, 4 * 3102, 4 * 400_896), 'mid_backward_kernel': (400_896, 4 * (3102 + 1556), 4 * 400_896),
    # the same block on clusters of workgroups (midcluster.hip: the default for this model since round 4)
    'midc_forward_kernel': (400_896, 4 * 3102, 4 * 400_896), 'midc_backward_kernel': (400_896, 4 * (3102 + 1556), 4 * 400_896),
    # ... with the 4x4 conv layers on either side of the block inside the launch (round 5): + conv4 and deconv1 forward (262 144 MACs
    # and 2048 + 512 elements each) / their data and weight gradients (4 x 262 144 MACs; upstream gradient, saved input, input gradient)
    'midc_forward_kernel(+ conv4, deconv1)': (400_896 + 2 * 262_144, 4 * (3102 + 2 * (2048 + 512)), 4 * (400_896 + 2 * 16_416)),
    'midc_backward_kernel(+ conv4, deconv1)': (400_896 + 4 * 262_144, 4 * (3102 + 1556 + (2048 + 512 + 512) + (512 + 2048 + 2048)), 4 * (400_896 + 2 * 16_416)),
    'up32_kernel<8>(+ reg_loss)': (1_048_576, 4 * (2048 + 8192)),
    'dense_wgrad_batch_kernel': (400_896, 4 * 3102, 4 * 400_896),
    # fixed-order sum of the conv layers' weight-gradient slabs: 4 x 256 slabs of 64 KB (16x16 and 8x8 layers), 2 x 128
    # (4x4 layers), 2 x 256 x 2 KB (single-channel layers) + bias partials read, 2 MB of gradients written
    'slab_reduce_batch_kernel': (0, 0, 4 * 256 * 65664 + 2 * 128 * 65664 + 2 * 256 * 2176 + 2_000_000),
    # round 6: the two closing launches as one grid (dense.hip dense_wgrad_slab_kernel): the Linear weight gradients' work and
    # bytes + the slabs as the paired launches of rounds 3-5 leave them (4 x ~130 slabs of 64 KB, 2 x 16 tap slabs of 66 KB, the
    # single-channel layers' 2 KB slabs) read once, 2 MB of gradients written
    'pair(dense_wgrad_batch + slab_reduce_batch)': (400_896, 4 * 3102, 4 * 400_896 + 4 * 130 * 65664 + 2 * 16 * 67584 + 2 * 256 * 2176 + 2_000_000),
}
# the library's timeline labels one kernel FAMILY; these are the instantiations a rocprofv3 --kernel-trace of the
# default build lists for it at B = 512 (profiles/*_kernel_stats.csv)
ROCPROF_NAMES = {
    'wgrad32_kernel<16>': ['arvae::wgrad32r_kernel<16, 1>', 'arvae::wgrad32r_kernel<16, 2>'],
    'up32_kernel<16>': ['arvae::up32p_kernel<1>', 'arvae::up32p_kernel<3>'],
    'down32_kernel<16>': ['arvae::down32p_kernel<16, 1>', 'arvae::down32p_kernel<16, 3>'],
    'wgrad32_kernel<8>': ['arvae::wgrad32r_kernel<8, 1>', 'arvae::wgrad32r_kernel<8, 2>'],
    'up32_kernel<8>': ['arvae::up32x_kernel<8, 1, 32>', 'arvae::up32x_kernel<8, 3, 32>'],
    'down32_kernel<8>': ['arvae::down32p_kernel<8, 1>', 'arvae::down32p_kernel<8, 3>'],
    'chain(down32<16> + down32<8>)': ['arvae::chain_down_kernel<1>'],
    'wgrad32_kernel<4>': ['arvae::wgrad32x_kernel<4, 1>', 'arvae::wgrad32x_kernel<4, 2>'],
    'up32_kernel<4>': ['arvae::up32x_kernel<4, 1, 32>', 'arvae::up32x_kernel<4, 3, 32>'],
    'down32_kernel<4>': ['arvae::down32s_kernel<4, 1>', 'arvae::down32s_kernel<4, 2>'],
    'pair4(down32 + wgrad32)': ['arvae::pair4_down_kernel<2, 2>'], 'pair4(up32 + wgrad32)': ['arvae::pair4_up_kernel<3, 1>'],
    'pair_c1(down_c1 + wgrad_c1)': ['arvae::pair_c1_kernel<1>'], 'pair(wgrad_c1 + dense_wgrad_batch)': ['arvae::dense_wgrad_c1_kernel'],
    'pair(dense_wgrad_batch + slab_reduce_batch)': ['arvae::dense_wgrad_slab_kernel'],
    'pair(down32<16> + wgrad32<16>)': ['arvae::pair_down_wgrad_kernel<16, 3, 2>'], 'pair(down32<8> + wgrad32<8>)': ['arvae::pair_down_wgrad_kernel<8, 3, 2>'],
    'pair(up32<16> + wgrad32<16>)': ['arvae::pair_up16_wgrad_kernel<3, 1>'], 'pair(up32<8> + wgrad32<8>)': ['arvae::pair_up8_wgrad_kernel<3, 1>'],
    'pair(up32<16> + wgrad32<16> + wgrad_c1)': ['arvae::pair_up16_wgrad_kernel<3, 1, true>'],
    'down_c1_kernel(+ weight prep)': ['arvae::down_c1s_prep_kernel'], 'up32_kernel<4>(+ reg_loss)': ['arvae::up32x_reg_kernel<4, 1>'],
    'up32_kernel<8>(+ reg_loss)': ['arvae::up32x_reg_kernel<8, 1>'],
    'midc_forward_kernel(+ conv4, deconv1)': ['midc_forward_kernel'], 'midc_backward_kernel(+ conv4, deconv1)': ['midc_backward_kernel'],
    'down_c1_kernel': ['arvae::down_c1s_kernel<0>', 'arvae::down_c1s_kernel<1>'], 'wgrad_c1_kernel': ['arvae::wgrad_c1s_kernel'],
    'up_c1_kernel(recon)': ['arvae::up_c1_kernel<0, true>'],
    'conv64_down(wide)': ['arvae::conv64s_kernel<3, 2, 0>'], 'conv64_down(narrow)': ['arvae::conv64s_kernel<3, 1, 0>'],
    'conv64_up(wide)': ['arvae::conv64s_kernel<4, 2, 0>'], 'conv64_up(narrow)': ['arvae::conv_rows_x3_kernel<true>'],
    'conv64_wgrad(pairs, wide)': ['arvae::conv_wgrad_pairs_h2_kernel'], 'conv64_wgrad(pairs, narrow)': ['arvae::conv_wgrad_pairs_h2_kernel'],
    'conv64_wgrad(rows)': ['arvae::conv_wgrad_rows_x3_kernel'],
    # (round 5: the recurrences own 4 / 8 / 16 batch rows per workgroup -- the instantiation a B = 256 step runs is the 4-row one)
    'gru_seq_fwd_kernel': ['arvae::gru_seq_fwd_h2_kernel<128, 4>'], 'gru_seq_bwd_kernel': ['arvae::gru_seq_bwd_h2_kernel<128, 4>'],
    'tick_free_run_x3_kernel': ['arvae::tick_free_run_x3_kernel<128>'],
    'tick_free_run_h2_kernel': ['arvae::tick_free_run_h2_kernel<128, true, 4>', 'arvae::tick_free_run_h2_kernel<128, false, 4>'],
    'rows_gemm_kernel<fwd>': ['arvae::rows_gemm_x3_kernel'], 'rows_gemm_kernel<dgrad>': ['arvae::rows_gemm_x3_kernel'],
    'rows_gemm_kernel<wgrad>': ['arvae::rows_gemm_x3_kernel'],
}



# ONE peak per arithmetic (MI355X_MICROARCH.md), against ALGORITHMIC FLOP (fp32-equivalent) and layer-boundary bytes:
#   'f16x2'  the 32-channel conv kernels: fp16 MFMA on scaled two-term operands, 3 partial products per multiply-add
#            -> 2500 / 3 = 833.3 TFLOP/s of fp32-equivalent work;
#   'fp32'   v_mfma_f32_* kernels (single-channel layers, the latent block): 157.3 TFLOP/s;
#   HBM      8.0 TB/s for every kernel.
ARITH_PEAK_TFLOPS = {'f16x2': PEAK_BF16_MFMA_TFLOPS / F16X2_PRODUCTS, 'fp32': PEAK_F32_MFMA_TFLOPS}


def arithmetic_of(label):
    return 'f16x2' if label.startswith(('down32', 'up32', 'wgrad32', 'pair4', 'pair(down32', 'pair(up32', 'chain(down32')) else 'fp32'


def kernel_rooflines(prof, prof_steps, pmc_kernels=None):
    """every launch site of the step: launches per step, average launch duration (HIP events, this run), algorithmic FLOP and
    bytes per launch, and its fractions of the two roofs -- each arithmetic against its ONE stated peak (ARITH_PEAK_TFLOPS), so
    a kernel's `frac` does not depend on which kernel happens to be the step's longest.  Sorted by device time."""
    rows = []
    total = sum(v['ms'] for v in prof.values())
    for label, v in prof.items():
        calls = max(v['calls'], 1)
        avg_s = v['ms'] / calls * 1e-3
        flop, nbytes = v.get('flop', 0.0) / calls, v.get('bytes', 0.0) / calls
        arith = arithmetic_of(label)
        row = {'kernel': label, 'launches_per_step': v['calls'] / prof_steps, 'avg_launch_us': avg_s * 1e6,
               'us_per_step': 1e3 * v['ms'] / prof_steps, 'share_of_device_time': v['ms'] / total if total else 0.0}
        if nbytes > 0 and avg_s > 0:
            hbm = nbytes / avg_s / 1e9 / PEAK_HBM_GBS
            mf = flop / avg_s / 1e12 / ARITH_PEAK_TFLOPS[arith] if flop else 0.0
            row.update({'algorithmic_bytes_per_launch': nbytes, 'algorithmic_flop_per_launch': flop, 'arithmetic': arith if flop else None,
                        'hbm_frac': hbm, 'mfma_frac': mf if flop else None, 'bound': 'mfma' if mf >= hbm else 'hbm', 'frac': max(mf, hbm)})
            if pmc_kernels and label in pmc_kernels:
                row['traffic'] = pmc_kernels[label]['hbm_bytes_per_launch']
        rows.append(row)
    rows.sort(key=lambda r: -r['us_per_step'])
    return rows


def rocprof_names(label):
    """kernel names of a rocprofv3 trace behind a timeline label (labels that are not in the table are kernel names)"""
    return ROCPROF_NAMES.get(label, ['arvae::' + label] if label.endswith('_kernel') else None)


def sq_counters(workload, names):
    """rocprofv3 SQ counters of a kernel from the committed summary (profiles/r5_sq_counters.json, tools/pmc_sq_round.sh): the
    counter-derived share of SIMD-cycles with the matrix pipe busy, beside the FLOP / time arithmetic of this run."""
    table = src = None
    for tag in ('r6', 'r5'):
        try:
            with open(os.path.join(ROOT, 'profiles', f'{tag}_sq_counters.json')) as f:
                table, src = json.load(f).get(workload, {}), f'profiles/{tag}_sq_counters.json'
            break
        except (OSError, ValueError):
            continue
    if table is None:
        return None
    for nm in names or []:
        key = nm.replace('arvae::', '')
        for k, v in table.items():
            if k.startswith(key) or key in k:
                return {'kernel': k, 'mfma_busy_frac': v['mfma_busy_frac'], 'valu_per_mfma': v.get('valu_per_mfma'),
                        'avg_launch_us_profiled': v['avg_launch_us_profiled'],
                        'source': src + ' (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES / (1024 x GRBM_GUI_ACTIVE / 8))'}
    return None


REG_DIMS = (1, 2, 3, 4, 5)
BETA, GAMMA, DELTA = 4.0, 10.0, 1.0

# algorithmic FLOP and layer-boundary bytes per image / measure and training step
SIDE_WORK = {'mnist': (431466496.0, 3276816.0), 'measure': (93.7e6, 0.7e6)}
SIDE_BATCH = {'mnist': 1024, 'measure': 256}
# Label families of the secondary workloads whose algorithmic work per training step is known (MACs per unit summed over
# the family's launches in one step; SURVEY 8(d) per-layer tables): the wide Morpho-MNIST layers run conv2 / conv3 /
# deconv1 / deconv2 once per direction in each of the three products (forward-type, data gradient, weight gradient);
# MeasureVAE's recurrent products are 3*H*H MACs per row and time step over 152 row-steps per measure.
SIDE_KERNEL_MACS = {
    'mnist': {'conv64_down(wide)': 2 * 31_719_424, 'conv64_down(narrow)': 2 * 2_957_312,
              'conv64_up(wide)': 2 * 31_719_424, 'conv64_up(narrow)': 2 * 2_957_312,
              'conv64_wgrad(pairs, wide)': 2 * 31_719_424, 'conv64_wgrad(pairs, narrow)': 2 * 2_957_312,
              'conv64_wgrad(rows)': 2 * (31_719_424 + 2_957_312)},
    'measure': {'gru_seq_fwd_kernel': 152 * 3 * 128 * 128, 'gru_seq_bwd_kernel': 2 * 152 * 3 * 128 * 128},
}
# families that do not launch in every step: MACs per unit and LAUNCH (the free-running decoder pass runs on the steps
# whose teacher-forcing coin says no: 24 ticks x (W_hh0, W_ih1, W_hh1: 9 H^2; note projection H V))
SIDE_KERNEL_MACS_PER_LAUNCH = {'measure': {'tick_free_run_x3_kernel': 24 * (9 * 128 * 128 + 128 * 35),
                                           'tick_free_run_h2_kernel': 24 * (9 * 128 * 128 + 128 * 35)}}


class DspritesDataset:          # ImageVAETrainer sniffs the dataset's class name (reference image_vae_trainer.py:81-86)
    pass


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


# ---- N > 1 without a launcher: start the ranks here ------------------------------------------------------------------
def visible_gpu_count():
    """GPUs this process's children would see, counted WITHOUT touching the HIP runtime: a *_VISIBLE_DEVICES list when one is
    set, otherwise the KFD topology nodes that have SIMDs (CPU nodes report simd_count 0).  None when neither is readable."""
    for var in ('HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES'):
        val = os.environ.get(var)
        if val is not None:
            return len([v for v in val.split(',') if v.strip() != ''])
    nodes = '/sys/class/kfd/kfd/topology/nodes'
    try:
        count = 0
        for node in os.listdir(nodes):
            with open(os.path.join(nodes, node, 'properties')) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            count += int(props.get('simd_count', '0')) > 0
        return count
    except (OSError, ValueError):
        return None


def spawn_ranks(n, argv, script=None, gpu_count=visible_gpu_count, out=None):
    """Run this script (or `script`) as n fresh child processes (one rank per GPU) and relay rank 0's stdout.  This process
    never touches the GPU: the device count comes from sysfs / the environment (visible_gpu_count), so every child starts
    clean.  A rank that exits non-zero ends the others and its code is re-raised here."""
    have = gpu_count() if callable(gpu_count) else gpu_count
    if have is not None and have < n:
        raise SystemExit(f'--gpus {n}: this node exposes {have} GPU(s)')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    threads = max(1, (os.cpu_count() or n) // n)
    procs = []
    out0 = tempfile.TemporaryFile(mode='w+')
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')         # dmabuf IPC only on this pool (RCCL needs it)
        env.setdefault('OMP_NUM_THREADS', str(threads))
        procs.append(subprocess.Popen([sys.executable, script or os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    rc = 0
    pending = list(procs)
    while pending:
        time.sleep(0.05)
        for p in list(pending):
            code = p.poll()
            if code is None:
                continue
            pending.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in pending:                                  # a rank failed: the others would wait for it forever
                    q.terminate()
    out0.seek(0)
    (out or sys.stdout).write(out0.read())
    (out or sys.stdout).flush()
    if rc != 0:
        raise SystemExit(f'a rank exited with code {rc}')


# ---- timing -------------------------------------------------------------------------------------------------------------
class Fence:
    """barrier + torch.cuda.synchronize() on both sides; max over ranks of a host-measured duration."""

    def __init__(self, device, comm):
        self.device, self.comm = device, comm            # comm: arvae_amd.parallel.LibraryComm / TorchComm, None = one process

    @property
    def use_dp(self):
        return self.comm is not None

    def __call__(self):
        if self.comm is not None and hasattr(self.comm, 'wait_idle'):
            self.comm.wait_idle()                        # = synchronize, but bounded and watching the communicator's error state
            self.comm.barrier()                          # a one-element all-reduce + (bounded) wait: every rank is here and idle
            return
        torch.cuda.synchronize()
        if self.comm is not None:
            self.comm.barrier()

    def max_over_ranks(self, seconds):
        if self.comm is None:
            return seconds
        t = torch.tensor([seconds], device=self.device, dtype=torch.float64)
        self.comm.all_reduce(t, 'max')
        return float(t)


def timed_regions(step, steps, warmup, fence, min_total_s=1.0, min_regions=3, max_regions=400):
    """-> (median region seconds, timing dict, last loss).  Every region is exactly `steps` steps between two fences."""
    for i in range(warmup):
        step(i)
    host, dev = [], []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    loss = None
    while True:
        fence()
        e0.record()                                   # torch's current stream = the stream every library launch goes to
        t0 = time.perf_counter()
        for i in range(steps):
            loss = step(i)
        e1.record()
        fence()
        host.append(fence.max_over_ranks(time.perf_counter() - t0))
        dev.append(e0.elapsed_time(e1) * 1e-3)
        # the stopping rule reads only rank-reduced numbers, so every rank runs the same number of regions
        if (len(host) >= min_regions and sum(host) >= min_total_s) or len(host) >= max_regions:
            break
    med = float(np.median(host))
    k = 1e3 / steps
    timing = {'regions': len(host), 'steps_per_region': steps, 'ms_per_step_median': med * k,
              'ms_per_step_min': min(host) * k, 'ms_per_step_max': max(host) * k, 'ms_per_step_first_region': host[0] * k,
              'hip_event_ms_per_step_median': float(np.median(dev)) * k, 'timed_seconds_total': sum(host),
              # the regions in order, thinned to <= 32 entries: a box's first second runs slower than its steady state
              'ms_per_step_regions': [round(h * k, 4) for h in host[::max(1, (len(host) + 31) // 32)]],
              'rule': 'regions of exactly K steps, barrier + synchronize on both sides, repeated until >= 1 s is timed; '
                      'value and ms_per_step are the median region (host clock, max over ranks)'}
    return med, timing, loss


def kernel_profile(step, n_steps):
    """per-label device time from the library's HIP-event timeline (one event after every kernel on the launch stream)"""
    import ctypes
    from arvae_amd import _lib
    lib = _lib.load()
    lib.arvae_profile_begin(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    for i in range(n_steps):
        step(i)
    buf = ctypes.create_string_buffer(1 << 16)
    lib.arvae_profile_end(buf, len(buf))
    prof = {}
    for line in buf.value.decode().splitlines():
        name, calls, ms = line.split('\t')
        prof[name] = {'calls': int(calls), 'ms': float(ms)}
    prof.pop('(gap)', None)          # launch gaps the library marks separately so that kernel times exclude them
    return prof


# ---- workloads ----------------------------------------------------------------------------------------------------------
COMM = None            # the job's communicator (main() connects it when there is more than one rank or --force-dp)


def build_trainer(device, use_dp):
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
    if use_dp:
        from arvae_amd.parallel import DataParallel
        DataParallel(comm=COMM).attach(trainer)
    model.train()
    return trainer, state


def build_side_workload(kind, device, batch, rank=0, use_dp=False, graphs=False):
    """-> (timed step, eager step for the instrumented pass, unit) for BASELINE.json configs[2] / configs[4]."""
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
        x, lab = syn.mnist_batch(batch, seed=4321 + rank)
        data = (torch.from_numpy(x).to(device), torch.from_numpy(lab).to(device))
    else:
        from arvae_amd.measure_vae import MeasureVAE
        from arvae_amd.measure_vae_trainer import MeasureVAETrainer
        ds = FolkDataset()
        model = MeasureVAE(ds, 10, 2, 2, 128, 0.5, 32, 2, 128, 0.5, False, 'folk')
        trainer = MeasureVAETrainer(ds, model, lr=1e-4, reg_type=('all',), reg_dim=(0, 1, 2, 3), beta=0.001, gamma=1.0,
                                    capacity=0.0, rand=0, delta=10.0)
        trainer.cuda()
        score = torch.from_numpy(syn.measure_batch(batch, seed=5 + rank)).to(device)
        data = (score, score)
    if use_dp:
        from arvae_amd.parallel import DataParallel
        dp = DataParallel(comm=COMM)
        dp.attach(trainer)
        dp.broadcast_parameters(model)
    model.train()

    def eager(i):
        trainer.zero_grad()
        loss, _ = trainer.loss_and_acc_for_batch(data, 0, i, True)
        trainer.backward(loss)
        trainer.step()
        return loss

    unit = 'images/s' if kind == 'mnist' else 'measures/s'
    if kind == 'measure' and graphs:                             # forward + backward replayed from HIP graphs (graphed.py)
        from arvae_amd.graphed import GraphedStep
        graphed = GraphedStep(trainer, data)

        def gstep(i):
            loss, _ = graphed(data)
            trainer.step()
            return loss
        return gstep, eager, unit
    return eager, eager, unit


def side_roofline(kind, prof, prof_steps, batch):
    """dominant library kernel of a secondary workload: share of the step and, where the label's algorithmic work per
    step is known, its achieved rate against the MFMA peak it runs on (three-term bf16 split: 6 products per MAC)."""
    if not prof:
        return None
    total = sum(v['ms'] for v in prof.values())
    # the timeline labels one launch SITE; a rocprofv3 trace lists kernels: labels that are instantiations of one kernel are
    # one family here (e.g. the wide and the narrow paired-row weight gradients), and the family with the largest TOTAL time
    # in a step is the dominant one -- the same ranking a --kernel-trace --stats summary gives
    fams = {}
    for label, v in prof.items():
        key = tuple(rocprof_names(label) or [label])
        f = fams.setdefault(key, {'labels': [], 'ms': 0.0, 'calls': 0})
        f['labels'].append(label)
        f['ms'] += v['ms']
        f['calls'] += v['calls']
    key, dom = max(fams.items(), key=lambda kv: kv[1]['ms'])
    name = ' + '.join(sorted(dom['labels']))
    out = {'kernel': name, 'rocprof_names': list(key), 'launches_per_step': dom['calls'] / prof_steps,
           'avg_launch_us': 1e3 * dom['ms'] / dom['calls'], 'us_per_step': 1e3 * dom['ms'] / prof_steps,
           'share_of_device_time': dom['ms'] / total, 'device_time_us_per_step': 1e3 * total / prof_steps}
    for tag in ('r6', 'r5', 'r4', 'r3', 'r2'):      # the top kernel of the committed rocprofv3 --kernel-trace --stats summary of this workload
        try:
            import csv
            with open(os.path.join(ROOT, 'profiles', f'{tag}_{kind}_kernel_stats.csv')) as f:
                top = max(csv.DictReader(f), key=lambda r: float(r['TotalDurationNs']))
            out['rocprof_top_kernel'] = {'name': top['Name'].split('(')[0].replace('void ', ''), 'avg_launch_us': float(top['AverageNs']) / 1e3,
                                         'share_of_kernel_time': float(top['Percentage']) / 100.0,
                                         'source': f'profiles/{tag}_{kind}_kernel_stats.csv'}
            break
        except (OSError, KeyError, ValueError):
            continue
    out['sq_counters'] = sq_counters(kind, list(key))
    macs = sum(SIDE_KERNEL_MACS.get(kind, {}).get(lb, 0) for lb in dom['labels'])
    for lb in dom['labels']:
        per_launch = SIDE_KERNEL_MACS_PER_LAUNCH.get(kind, {}).get(lb)
        if per_launch:
            macs += per_launch * prof[lb]['calls'] / prof_steps
    if macs:
        # (the wide 64-channel convolutions and their weight gradient moved to the two-term fp16 arithmetic in round 3, the GRU
        # recurrences in rounds 4 (forward, free-running) and 5 (backward))
        products = F16X2_PRODUCTS if ((kind == 'mnist' and ('wide' in name or 'pairs' in name)) or
                                      (kind == 'measure' and ('gru_seq' in name or 'tick_free_run_h2' in name))) else BF16X3_PRODUCTS
        tf = 2.0 * macs * batch * products / (dom['ms'] / prof_steps * 1e-3) / 1e12
        out.update({'bound': 'mfma', 'achieved': tf, 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': tf / PEAK_BF16_MFMA_TFLOPS, 'traffic': None, 'fp32_equivalent_tflops': tf / products,
                    'mfma_work': 'executed 16-bit MFMA FLOP = %d partial products x algorithmic FLOP of all the label\'s '
                                 'launches in one step' % products})
    return out


def run_side(kind, device, args, fence, rank, world, use_dp, with_cpu):
    """one secondary workload -> its result dict (the main line when selected with --workload)"""
    bsz = args.batch if (args.workload == kind and args.batch != 512) else SIDE_BATCH[kind]
    # MeasureVAE: a step is two library calls (the whole-model executor; three and one grouped all-gather under data parallelism):
    # eager.  --graphs replays it from a HIP graph, collectives recorded with the kernels
    graphs = kind == 'measure' and not args.no_graphs and args.graphs
    step, eager, unit = build_side_workload(kind, device, bsz, rank, use_dp, graphs)
    steps = args.steps if args.workload == kind else max(10, min(args.steps, 50))
    med, timing, loss = timed_regions(step, steps, args.warmup, fence, args.min_seconds)
    prof_steps = 4
    prof = kernel_profile(eager, prof_steps)
    fence()
    rate = world * bsz * steps / med
    flop, byts = SIDE_WORK[kind]
    per_gpu = rate / world
    res = {'metric': f'training {unit} ({kind} AR-VAE, per-GPU batch {bsz})', 'value': rate, 'unit': unit, 'n_gpus': world,
           'steps': steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * med / steps, 'higher_is_better': True,
           'scaling': 'weak', 'vs_baseline': None,
           'dtype': ('f32 (wide conv MFMAs: scaled 2-term fp16 split, 3 products; narrow conv / Linear: 3-term bf16 split, 6 products; fp32-accurate)'
                     if kind == 'mnist' else
                     'f32 (recurrences: scaled 2-term fp16 split, 3 products, scales taken from the data; Linear layers: 3-term bf16 split, 6 products; fp32-accurate)'),
           'data': 'synthetic',
           'config': {'workload': ('Morpho-MNIST AR-VAE full training step, 1x28x28 inputs, z=16, reg_dim=(1..6), dropout 0.5'
                                   if kind == 'mnist' else
                                   'FolkNBar MeasureVAE full training step, 24-tick measures, V=35, z=32, reg_dim=(0..3), '
                                   'dropout 0.5, teacher forcing p=0.5'),
                      'per_gpu_batch': bsz, 'global_batch': bsz * world, 'parallelism': f'dp{world}',
                      'launch': 'hip-graph replay of fwd+bwd' if graphs else ('eager' if kind == 'mnist' else 'eager, whole-model executor (two library calls per step)'),
                      'final_loss': float(loss.detach())},
           'timing': timing,
           # whole-step fractions of the datasheet roofs (SURVEY 8(d) algorithmic FLOP / layer-boundary bytes per unit)
           # (round 5: against the roof the arithmetic runs under -- the 16-bit MFMA peak divided by the partial products of the split
           # that keeps fp32 accuracy, 3 for the scaled two-term fp16 kernels -- not against the fp32-MFMA peak this path left in round 3)
           'step_roofline': {'flop_per_unit': flop, 'bytes_per_unit': byts,
                             'flop_frac_mfma16_3products': per_gpu * flop / (PEAK_BF16_MFMA_TFLOPS / F16X2_PRODUCTS * 1e12),
                             'roof_tflops_fp32_equivalent': PEAK_BF16_MFMA_TFLOPS / F16X2_PRODUCTS,
                             'hbm_frac': per_gpu * byts / (PEAK_HBM_GBS * 1e9)},
           'roofline': side_roofline(kind, prof, prof_steps, bsz)}
    if with_cpu:
        res['cpu_baseline'] = side_cpu_baseline(kind, bsz)
    return res


# ---- CPU oracle baselines (test infrastructure used as the reported baseline; never on the product path) ---------------
def _cpu_steps(run, state, budget_s, threads):
    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
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
    finally:
        torch.set_num_threads(old)
    steady = sorted(times[1:])
    return steady[len(steady) // 2], n


def side_cpu_baseline(kind, batch, budget_s=10.0):
    """the CPU oracle's training step on a secondary workload, ~budget_s of CPU work on 8 threads (the thread count of
    SURVEY.md section 6's reference timings; more threads are slower for these step sizes)."""
    from arvae_amd import synthetic as syn
    from oracle import step as o_step
    threads = min(8, os.cpu_count() or 8)
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
    med, n = _cpu_steps(run, state, budget_s, threads)
    return {'value': batch / med, 'unit': unit, 'cores': threads, 'kind': 'port',
            'sample': f'{n} full training steps (first discarded) of the {what} at batch {batch}, fp32, PyTorch-CPU oracle, '
                      f'{threads} threads, median step {med * 1e3:.1f} ms'}


def cpu_baseline(batch, state, budget_s=18.0):
    """CPU oracle on the headline workload: all host cores (~budget_s of CPU work), then 8 threads (~budget_s / 2)."""
    from arvae_amd import synthetic as syn
    from oracle import step as o_step
    x, lab = syn.dsprites_batch(batch, seed=1234)
    eps = syn.normal_noise((batch, 10), seed=1)
    run = lambda cur, adam, n: o_step.image_step('dsprites', cur, x, lab, eps, REG_DIMS, BETA, GAMMA, DELTA,
                                                 adam_state=adam, step_no=n)
    cores = torch.get_num_threads()
    med, n = _cpu_steps(run, state, budget_s, cores)
    out = {'value': batch / med, 'unit': 'images/s', 'cores': cores, 'kind': 'port',
           'sample': f'{n} full training steps (first discarded) of the dSprites AR-VAE at batch {batch}, fp32, '
                     f'PyTorch-CPU oracle, {cores} threads, median step {med * 1e3:.1f} ms'}
    if cores != 8:
        med8, n8 = _cpu_steps(run, state, budget_s / 2, 8)
        out['value_8_threads'] = batch / med8
        out['sample_8_threads'] = f'{n8} steps, 8 threads, median step {med8 * 1e3:.1f} ms'
    return out


# ---- data-parallel extras (N > 1, or --force-dp): where a multi-rank step's time goes ---------------------------------
def dp_collective_costs(trainer, batch, device, reps=20):
    """HIP-event time of the step's two exchanges in isolation, back to back on the launch stream: the SUM all-reduce of a
    gradient-arena-sized buffer and the grouped all-gather of z and the labels (us per call, max over ranks)."""
    dp = trainer.data_parallel
    arena = torch.zeros_like(trainer.optimizer.grad_arena)
    z = torch.zeros(batch, trainer.model.z_dim, device=device)
    lab = torch.zeros(batch, 6, device=device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = {'grad_arena_bytes': arena.numel() * 4, 'gathered_bytes_per_rank': (z.numel() + lab.numel()) * 4}
    for name, call in (('all_reduce_us', lambda: COMM.all_reduce(arena)), ('all_gather_us', lambda: dp.gather_many([z, lab]))):
        for _ in range(3):
            call()
        COMM.barrier()
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        COMM.wait_idle()
        t = torch.tensor([e0.elapsed_time(e1) * 1e3 / reps], device=device, dtype=torch.float64)
        COMM.all_reduce(t, 'max')
        out[name] = float(t)
    return out


def dp_overlap_trial(step, fence, steps=60, warm=10):
    """ms per step with the gradient all-reduce / column all-gather on the launch stream (the default) and with the overlap
    schedule (ARVAE_DP_OVERLAP=1: per-bucket all-reduces on a side stream behind the executor's milestone events), same
    ranks, same inputs, a few dozen steps each way -- the numbers that decide the default at W >= 2."""
    res = {}
    old = os.environ.get('ARVAE_DP_OVERLAP')
    try:
        for tag, val in (('launch_stream', '0'), ('overlap', '1'), ('launch_stream_again', '0')):
            os.environ['ARVAE_DP_OVERLAP'] = val
            for i in range(warm):
                step(i)
            fence()
            t0 = time.perf_counter()
            for i in range(steps):
                step(i)
            fence()
            res[tag + '_ms_per_step'] = 1e3 * fence.max_over_ranks(time.perf_counter() - t0) / steps
    finally:
        if old is None:
            os.environ.pop('ARVAE_DP_OVERLAP', None)
        else:
            os.environ['ARVAE_DP_OVERLAP'] = old
    base = min(res['launch_stream_ms_per_step'], res['launch_stream_again_ms_per_step'])
    res['overlap_faster'] = bool(res['overlap_ms_per_step'] < 0.98 * base)
    return res


# ---- the headline workload ---------------------------------------------------------------------------------------------
def run_dsprites(device, args, fence, rank, world, use_dp):
    from arvae_amd import synthetic as syn
    trainer, state = build_trainer(device, use_dp)
    if use_dp:
        trainer.data_parallel.broadcast_parameters(trainer.model)
    b = args.batch
    x_np, lab_np = syn.dsprites_batch(b, seed=1234 + rank)
    x = torch.from_numpy(x_np).to(device)
    lab = torch.from_numpy(lab_np).to(device)

    def step(i):
        trainer.zero_grad()
        loss, _ = trainer.loss_and_acc_for_batch((x, lab), 0, i, True)
        trainer.backward(loss)                        # = loss.backward() with a cached seed gradient (Trainer.backward)
        trainer.step()
        return loss

    med, timing, loss = timed_regions(step, args.steps, args.warmup, fence, args.min_seconds)
    final_loss = float(loss.detach())
    if not np.isfinite(final_loss):
        raise SystemExit(f'non-finite loss {final_loss}')

    # ---- instrumented pass: per-kernel device time from HIP events recorded by the library on the launch stream after
    # every kernel (arvae_profile_begin/_end); a separate pass so the timed regions are untouched
    prof_steps = 10
    prof = kernel_profile(step, prof_steps)
    for name, v in prof.items():
        macs, nbytes, fixed = (KERNEL_WORK.get(name, (0, 0)) + (0,))[:3]
        v['flop'] = 2.0 * macs * b * v['calls']
        v['bytes'] = (float(nbytes) * b + fixed) * v['calls']
    fence()
    dp_info = None
    if use_dp and hasattr(COMM, 'wait_idle') and b == 512:
        # (every rank runs these: they are collectives)  The trial comes LAST and under a short deadline: the overlap schedule has
        # never met a second rank on hardware available to this build, and a failure there must not cost the headline line
        dp_info = {'world': world, 'collectives_on': 'launch stream (ARVAE_DP_OVERLAP unset)' if os.environ.get('ARVAE_DP_OVERLAP', '0') != '1' else 'side stream (ARVAE_DP_OVERLAP=1)'}
        try:
            dp_info.update(dp_collective_costs(trainer, b, device))
            dp_info['share_of_step'] = (dp_info['all_reduce_us'] + dp_info['all_gather_us']) / (1e6 * med / args.steps)
        except Exception as e:
            dp_info['collective_costs_error'] = f'{type(e).__name__}: {e}'
    run_dsprites.trial = (lambda: dp_overlap_trial(step, fence)) if dp_info is not None and world > 1 else None
    if rank != 0:
        return None

    ms_per_step = 1e3 * med / args.steps
    value = world * b * args.steps / med
    # the dominant kernel = the label with the most device time among ALL labels whose algorithmic work is known
    # (every kernel of the step above 2 % of its device time has a KERNEL_WORK entry; `unaccounted_labels` lists the rest)
    dom_name, dom = max(((k, v) for k, v in prof.items() if v['bytes'] > 0), key=lambda kv: kv[1]['ms'])
    avg_ms = dom['ms'] / dom['calls']
    arith = arithmetic_of(dom_name)
    split = arith == 'f16x2'                                                                              # (fp16 two-term MFMA kernels)
    # ONE peak per arithmetic, against algorithmic (fp32-equivalent) FLOP: 2500 / 3 products for the two-term fp16 kernels,
    # 157.3 for fp32-MFMA kernels (ARITH_PEAK_TFLOPS) -- the same pricing as every row of roofline.kernels
    mfma_peak = ARITH_PEAK_TFLOPS[arith]
    mfma_work = dom['flop']
    mfma_tf = mfma_work / dom['calls'] / (avg_ms * 1e-3) / 1e12
    hbm_gbs = dom['bytes'] / dom['calls'] / (avg_ms * 1e-3) / 1e9
    # the roof this kernel sits closer to binds it
    if mfma_tf / mfma_peak >= hbm_gbs / PEAK_HBM_GBS:
        roof = {'bound': 'mfma', 'achieved': mfma_tf, 'peak': mfma_peak, 'unit': 'TFLOP/s', 'frac': mfma_tf / mfma_peak,
                'traffic': None, 'arithmetic': arith}
        if split:
            roof['mfma_work'] = ('algorithmic fp32-equivalent FLOP against the fp16 MFMA peak over the 3 partial products the scaled '
                                 'two-term split issues per multiply-add (2500 / 3 TFLOP/s)')
            roof['executed_fp16_tflops'] = mfma_tf * F16X2_PRODUCTS
    else:
        roof = {'bound': 'hbm', 'achieved': hbm_gbs, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': hbm_gbs / PEAK_HBM_GBS,
                'traffic': None, 'arithmetic': arith}
    roof['other_roof_frac'] = {'mfma': mfma_tf / mfma_peak, 'hbm_algorithmic_bytes': hbm_gbs / PEAK_HBM_GBS}
    # HBM traffic of that kernel from the PMC counters: collected in separate rocprofv3 --pmc passes of this
    # same command (FETCH_SIZE / WRITE_SIZE cannot share a pass) and committed under profiles/
    step_traffic = pmc_kernels = None
    for tag in ('r6', 'r5', 'r4', 'r3', 'r2', 'r1'):
        try:
            with open(os.path.join(ROOT, 'profiles', f'{tag}_pmc_traffic.json')) as f:
                pmc = json.load(f)
            pmc_kernels = pmc['kernels']
            roof['traffic'] = pmc['kernels'][dom_name]['hbm_bytes_per_launch']
            roof['traffic_source'] = f'profiles/{tag}_pmc_traffic.json (rocprofv3 --pmc, 2*FETCH_SIZE + WRITE_SIZE, B=512)'
            # PMC bytes of one whole step: the file's own sum over every kernel it traced, or (older files, which list
            # only the big kernels) the per-launch figures times this run's launches per step
            step_traffic = pmc.get('step_hbm_bytes') or sum(
                pmc['kernels'][k]['hbm_bytes_per_launch'] * v['calls'] / prof_steps for k, v in prof.items() if k in pmc['kernels'])
            break
        except (OSError, KeyError, ValueError):
            roof['traffic'] = None
    if b != 512:
        roof['traffic'] = step_traffic = None
    roof['step_traffic_bytes'] = step_traffic
    roof['step_algorithmic_bytes'] = BYTES_PER_IMAGE * b + PARAM_BYTES_PER_STEP
    # the whole step against the same roofs (the figure that is comparable from round to round: it does not depend on which
    # kernel is the longest): algorithmic bytes / FLOP of one step over the timed step
    step_s = med / args.steps
    roof['step'] = {'ms_per_step': 1e3 * step_s,
                    'hbm_frac': (BYTES_PER_IMAGE * b + PARAM_BYTES_PER_STEP) / step_s / (PEAK_HBM_GBS * 1e9),
                    'flop_frac_fp32_equivalent': FLOP_PER_IMAGE * b / step_s / (ARITH_PEAK_TFLOPS['f16x2'] * 1e12),
                    'flop_peak_tflops': ARITH_PEAK_TFLOPS['f16x2'], 'hbm_peak_gbs': PEAK_HBM_GBS,
                    'traffic_over_algorithmic_bytes': (step_traffic / (BYTES_PER_IMAGE * b + PARAM_BYTES_PER_STEP)) if step_traffic else None,
                    'north_star_target_hbm_frac': 0.40}
    roof['kernels'] = kernel_rooflines(prof, prof_steps, pmc_kernels if b == 512 else None)
    # the same kernels in the committed rocprofv3 --kernel-trace --stats summary (launches there run back to back; the live
    # figure above brackets every launch with its own events, which costs each kernel the overlap with its neighbours' tails)
    try:
        import csv
        stats_csv = next(t for t in ('r6', 'r5', 'r4', 'r3', 'r2') if os.path.exists(os.path.join(ROOT, 'profiles', f'{t}_dsprites_kernel_stats.csv')))
        with open(os.path.join(ROOT, 'profiles', f'{stats_csv}_dsprites_kernel_stats.csv')) as f:
            rows = [r for r in csv.DictReader(f) if any(nm in r['Name'] for nm in (rocprof_names(dom_name) or []))]
        calls = sum(int(r['Calls']) for r in rows)
        if calls and b == 512:
            roof['rocprof_avg_launch_us'] = sum(float(r['TotalDurationNs']) for r in rows) / calls / 1e3
            roof['rocprof_source'] = f'profiles/{stats_csv}_dsprites_kernel_stats.csv'
            # the same algorithmic work over the trace's duration (launches back to back, no event between them)
            work = (mfma_work if roof['bound'] == 'mfma' else dom['bytes']) / dom['calls']
            roof['achieved_rocprof'] = work / (roof['rocprof_avg_launch_us'] * 1e-6) / (1e12 if roof['bound'] == 'mfma' else 1e9)
            roof['frac_rocprof'] = roof['achieved_rocprof'] / roof['peak']
    except (OSError, KeyError, ValueError, StopIteration):
        pass
    roof['timing_source'] = ('achieved / frac: HIP events around every launch of the kernel, this run (avg_launch_us); achieved_rocprof / '
                             'frac_rocprof: the committed rocprofv3 --kernel-trace --stats average of the same kernel (rocprof_avg_launch_us)')
    roof['sq_counters'] = sq_counters('dsprites', rocprof_names(dom_name))
    roof.update({'kernel': dom_name, 'rocprof_names': rocprof_names(dom_name),
                 'launches_per_step': dom['calls'] / prof_steps, 'avg_launch_us': avg_ms * 1e3,
                 'algorithmic_bytes_per_launch': dom['bytes'] / dom['calls'],
                 'share_of_device_time': dom['ms'] / sum(v['ms'] for v in prof.values()),
                 'unaccounted_labels': sorted(k for k, v in prof.items() if v['bytes'] <= 0)})
    per_gpu = value / world
    line = {
        'metric': 'training images/sec (dSprites beta-VAE+AR, per-GPU batch 512)', 'value': value,
        'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32 (conv MFMAs: scaled 2-term fp16 split, 3 products, fp32-accurate)',
        'data': 'synthetic',
        'config': {'workload': 'dSprites AR-VAE full training step (fwd + bwd + Adam), 1x64x64 inputs, z=10, '
                               'reg_dim=(1,2,3,4,5), beta=4 gamma=10 delta=1',
                   'per_gpu_batch': b, 'global_batch': b * world,
                   'parallelism': f'dp{world}' + (' (forced DP path)' if args.force_dp and world == 1 else ''),
                   'rccl_world_size': world if use_dp else None,
                   'collectives': None if COMM is None else type(COMM).__name__ + (
                       ' (RCCL %d via libarvae_hip.so, launch stream)' % COMM.rccl_version if hasattr(COMM, 'rccl_version') else
                       ' (gloo on the host under device tensors: tests)' if type(COMM).__name__ == 'StagedComm' else ' (torch.distributed nccl group)'),
                   'images_per_sec_per_gpu': per_gpu, 'final_loss': final_loss},
        'dp': dp_info,
        'timing': timing,
        'roofline': roof,
        'step_roofline': {
            'flop_frac_mfma16_3products': per_gpu * FLOP_PER_IMAGE / (PEAK_BF16_MFMA_TFLOPS / F16X2_PRODUCTS * 1e12),
            'hbm_frac': per_gpu * (BYTES_PER_IMAGE + PARAM_BYTES_PER_STEP / b) / (PEAK_HBM_GBS * 1e9),
            'binding': 'HBM roof 4.2 M img/s; fp32-MFMA roof 2.13 M img/s (no longer binding: the conv layers run on '
                       'the fp16 MFMA at 3 products per multiply-add, an 11 M img/s roof)'},
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
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=30)
    ap.add_argument('--batch', type=int, default=512, help='per-GPU batch')
    ap.add_argument('--min-seconds', type=float, default=1.0,
                    help='keep timing K-step regions until this much time is covered (at least 3 regions)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the Morpho-MNIST / MeasureVAE timings of the default run')
    ap.add_argument('--breakdown', action='store_true', help='print the per-kernel-family table to stderr')
    ap.add_argument('--force-dp', action='store_true',
                    help='run the data-parallel code path (RCCL all-gather + all-reduce) even with one rank')
    ap.add_argument('--no-graphs', action='store_true', help='(accepted for older scripts: the measure workload runs eagerly by default)')
    ap.add_argument('--graphs', action='store_true', help='measure workload: HIP-graph replay of the step')
    ap.add_argument('--workload', default='dsprites', choices=['dsprites', 'mnist', 'measure'],
                    help='dsprites = the headline metric (default); mnist / measure = BASELINE.json configs[2] / configs[4]')
    args = ap.parse_args()
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # before anything touches the GPU: dmabuf IPC only on this pool
                                                                 # (RCCL under `torch.distributed.run bench.py` needs it too)
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:             # no launcher: start the ranks (nothing has touched the GPU)
        return spawn_ranks(args.gpus, sys.argv[1:])
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world and not (args.gpus == 1 and world == 1):
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the AR-VAE hot path has no CPU fallback')
    if os.environ.get('ARVAE_DP_TRANSPORT') == 'staged':        # (tests: host-staged collectives, several ranks may share a device)
        local_rank %= torch.cuda.device_count()
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)
    use_dp = world > 1 or args.force_dp
    out_fd = None
    if use_dp:
        # RCCL prints a version banner on stdout; the contract is ONE JSON line there.  Everything this process (and the
        # libraries it loads) writes to fd 1 goes to stderr from here on; the line itself is written to the saved descriptor.
        sys.stdout.flush()
        out_fd = os.dup(1)
        os.dup2(2, 1)
        # the job's RCCL communicator, owned through libarvae_hip.so (arvae_comm_*): the ranks meet at the launcher's TCP
        # store (MASTER_ADDR / MASTER_PORT) to hand out RCCL's unique id; no torch process group (ARVAE_DP_TRANSPORT=torch
        # selects torch.distributed's 'nccl' group instead)
        global COMM
        from arvae_amd import parallel
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29511')
        try:
            COMM = parallel.connect(rank, world, device)
        except parallel.CommInitError as e:                      # the init helper thread may still be inside RCCL: no destructors
            parallel.leave_after_comm_failure(e)
    fence = Fence(device, COMM)

    if args.workload != 'dsprites':
        line = run_side(args.workload, device, args, fence, rank, world, use_dp, with_cpu=world == 1 and not args.no_cpu_baseline)
    else:
        line = run_dsprites(device, args, fence, rank, world, use_dp)
        if line is not None and world == 1 and not args.force_dp and not args.no_secondary and args.batch == 512:
            sec = {}
            for kind in ('mnist', 'measure'):
                try:
                    r = run_side(kind, device, args, fence, 0, 1, False, with_cpu=not args.no_cpu_baseline)
                    sec[kind] = {k: r[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'config', 'timing',
                                                   'step_roofline', 'roofline', 'cpu_baseline') if k in r}
                except Exception as e:                              # the headline line must not depend on a side workload
                    sec[kind] = {'error': f'{type(e).__name__}: {e}'}
            line['secondary'] = sec
            # early in the line (a log tail keeps the front of a long JSON line's keys last: these three numbers are the round's)
            head = {k: line[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step')}
            head['secondary_ms_per_step'] = {k: v.get('ms_per_step') for k, v in sec.items()}
            line = {**head, **{k: v for k, v in line.items() if k not in head}}
    comm_ok = True
    if args.workload == 'dsprites' and use_dp and world > 1 and not args.no_secondary and args.batch == 512:
        # BASELINE.json configs[4] ("MeasureVAE batch = 256, 1 -> 8 MI355X DP"): weak scaling, 256 measures per rank, the executor's
        # data-parallel path (forward to z, grouped all-gather, arvae_measure_vae_finish, backward, all-reduce, Adam) -- every
        # rank runs it, rank 0 reports it beside the headline
        try:
            r = run_side('measure', device, args, fence, rank, world, True, with_cpu=False)
            if line is not None:
                line['secondary'] = {'measure': {k: r[k] for k in ('metric', 'value', 'unit', 'n_gpus', 'ms_per_step', 'steps', 'config', 'timing',
                                                                  'step_roofline', 'roofline') if k in r}}
        except Exception as e:
            comm_ok = COMM is not None and getattr(COMM, 'handle', True) is not None
            if line is not None:
                line['secondary'] = {'measure': {'error': f'{type(e).__name__}: {e}'}}
    trial = getattr(run_dsprites, 'trial', None) if args.workload == 'dsprites' else None
    if trial is not None and comm_ok:
        from arvae_amd import parallel
        parallel.IDLE_TIMEOUT_S = min(parallel.IDLE_TIMEOUT_S, 60.0)
        try:
            res = trial()
        except Exception as e:                                       # (wait_idle has aborted the communicator by now)
            res = {'error': f'{type(e).__name__}: {e}'}
            comm_ok = False
        if line is not None and line.get('dp') is not None:
            line['dp']['overlap_trial'] = res
            # the decision the numbers support: the timed regions above ran the schedule named in `collectives_on`
            line['dp']['overlap_decision'] = ('trial failed: keep the collectives on the launch stream' if 'error' in res else
                                              'ARVAE_DP_OVERLAP=1 is faster here' if res.get('overlap_faster') else
                                              'collectives on the launch stream (the default) are as fast or faster here')
    if rank == 0 and line is not None:
        if out_fd is None:
            print(json.dumps(line), flush=True)
        else:
            os.write(out_fd, (json.dumps(line) + '\n').encode())
    if COMM is not None and comm_ok:
        COMM.barrier()
        COMM.close()
    elif COMM is not None:
        sys.stdout.flush()
        sys.stderr.flush()
        # The communicator is gone: do not wait for anybody on the way out.  Exit code: only OPTIONAL extras run after the
        # headline measurement (the secondary workload, the overlap trial) can bring us here -- a failure of the timed
        # data-parallel step itself raises out of main() and exits non-zero -- so the headline line above is valid and says
        # which extra failed (`secondary.measure.error` / `dp.overlap_trial.error`); the ranks leave with 0 so that a
        # launcher does not discard a good measurement, and with ARVAE_BENCH_STRICT=1 with 3.
        os._exit(3 if os.environ.get('ARVAE_BENCH_STRICT') == '1' else 0)


if __name__ == '__main__':
    main()
