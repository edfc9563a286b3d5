"""Data-parallel AR-VAE step over RCCL on real GPUs (golden G8): W ranks, each holding B/W rows
of a fixed batch, must reproduce the single-process step on the whole batch -- loss terms, the gradient every rank holds
after the all-reduce (times 1/W), and the weights after Adam.  world = 1 exercises the collectives' code path on a
one-GPU box; world = 2 needs two GPUs (skipped otherwise) and is the first test that runs RCCL between ranks."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from arvae_amd import synthetic as syn
from oracle import image_vae as o_vae
from oracle import step as o_step

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B_TOTAL = 64


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _run_ranks(world, out, capacity, fused):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'dp_worker.py'), str(r), str(world), str(port), out,
                               str(capacity), str(int(fused)), str(B_TOTAL)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True) for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    for p, o in zip(procs, logs):
        assert p.returncode == 0, o[-3000:]
    return np.load(out)


# capacity 3.7 lies between the two shards' KL means of this batch (test_parallel_gloo.py): beta*|KL - c| then needs the
# all-reduced KL mean; 25 is the golden's value (dsprites_step_b8_cap_gauss uses it at B = 8)
@pytest.mark.parametrize('fused', [True, False], ids=['fused', 'per_layer'])
@pytest.mark.parametrize('capacity', [0.0, 3.7])
@pytest.mark.parametrize('world', [1, 2])
def test_rccl_ranks_equal_single_process(tmp_path, golden_dir, world, capacity, fused):
    if torch.cuda.device_count() < world:
        pytest.skip(f'needs {world} GPUs, this box has {torch.cuda.device_count()}')
    got = _run_ranks(world, str(tmp_path / 'dp.npz'), capacity, fused)
    assert int(got['world']) == world
    state = syn.synth_state(o_vae.DSPRITES_SHAPES, 1, 1.6)
    x, lab = syn.dsprites_batch(B_TOTAL, seed=1234)
    eps = syn.normal_noise((B_TOTAL, 10), seed=12)
    ref = o_step.image_step('dsprites', state, x, lab, eps, (1, 2, 3, 4, 5), 4.0, 10.0, 1.0, capacity=capacity)
    srcs = [ref['terms']]
    if capacity == 0.0:                                        # the reference's own output for this batch (golden G4 / G8)
        srcs.append(np.load(os.path.join(golden_dir, 'dsprites_step_b64.npz')))
    for src in srcs:
        np.testing.assert_allclose(got['loss'], float(src['loss']), rtol=1e-4)
        np.testing.assert_allclose(got['acc'], float(src['acc']), rtol=1e-4)
        for k in ('recons', 'dist', 'reg'):
            np.testing.assert_allclose(got['term/' + k], float(src[k]), rtol=1e-4)
    for name in state:
        gr = got['grad/' + name].astype(np.float64).ravel()
        want = ref['grads'][name].astype(np.float64).ravel()
        assert np.linalg.norm(gr - want) <= 2e-3 * np.linalg.norm(want) + 1e-9, name
        # Adam's first step is -lr * g / (|g| + eps): an entry whose gradient is ~0 turns summation noise into a full-size
        # step, so the update is compared by its norm (as the golden tests do) and by the share of entries that moved alike
        d_got = got['param/' + name].astype(np.float64).ravel() - state[name].astype(np.float64).ravel()
        d_ref = ref['params'][name].astype(np.float64).ravel() - state[name].astype(np.float64).ravel()
        np.testing.assert_allclose(np.linalg.norm(d_got), np.linalg.norm(d_ref), rtol=2e-3, err_msg=name)
        assert (np.abs(d_got - d_ref) > 2e-5).mean() <= 2e-3, name


def _bench(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(args), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize('workload', ['dsprites', 'measure'])
def test_bench_starts_its_own_ranks(workload):
    """`python bench.py --gpus 2` without a launcher: the ranks are started by bench.py itself and rank 0's line says 2"""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs 2 GPUs')
    line = _bench('--gpus', '2', '--steps', '3', '--warmup', '1', '--min-seconds', '0', '--no-cpu-baseline', '--workload', workload)
    assert line['n_gpus'] == 2 and line['config']['global_batch'] == 2 * line['config']['per_gpu_batch']
    assert line['scaling'] == 'weak' and np.isfinite(line['value']) and line['value'] > 0


def test_bench_refuses_more_ranks_than_gpus():
    have = torch.cuda.device_count()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', str(have + 1), '--steps', '1', '--warmup', '0'],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and f'exposes {have} GPU' in (r.stderr + r.stdout)


def test_bench_line_has_the_contract_fields():
    """default-shaped run (short): roofline, timing, secondary workloads with their own rooflines"""
    line = _bench('--steps', '5', '--warmup', '2', '--min-seconds', '0', '--no-cpu-baseline')
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'timing', 'secondary'):
        assert k in line, k
    assert line['steps'] == 5 and line['warmup'] == 2 and line['n_gpus'] == 1 and line['vs_baseline'] is None
    rf = line['roofline']
    assert rf['bound'] in ('hbm', 'mfma') and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-9 and rf['rocprof_names']
    assert line['timing']['regions'] >= 3
    for kind in ('mnist', 'measure'):
        sec = line['secondary'][kind]
        assert 'error' not in sec, sec
        assert sec['value'] > 0 and sec['roofline']['kernel'] and 0 < sec['step_roofline']['flop_frac_fp32'] < 1
