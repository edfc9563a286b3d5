"""Diagnostic (dense.hip built with -DRG_STAMPS into tools/bin/lib_rgst.so): phase timeline of a rows-GEMM weight gradient
(6144 rows, 128 -> 384 features: the MeasureVAE's W_ih gradient) and of the forward product of the same layer."""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arvae_amd import ops, _lib
lib = _lib.load()
dev = torch.device('cuda:0')
rows, n_in, n_out = 6144, 128, 384
link = ops.Link.dense(n_in, n_out)
x = torch.randn(rows, n_in, device=dev)
g = torch.randn(rows, n_out, device=dev)
w = torch.randn(n_out, n_in, device=dev) * 0.05
b = torch.zeros(n_out, device=dev)
dw, db = torch.zeros_like(w), torch.zeros_like(b)
fn = ctypes.CDLL(os.environ.get('ARVAE_LIB') or _lib.LIB_PATH).arvae_debug_rg_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]


def report(tag, launch, chunks):
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (512 * 32))()
    assert fn(buf, 512 * 32) == 0
    st = np.array(buf, dtype=np.uint64).reshape(512, 32).astype(np.int64)
    t0 = st[:, 0].min()
    print('== %s: workgroup start %.2f .. %.2f us after the first, end %.2f .. %.2f' % (
        tag, (st[:, 0].min() - t0) / 100, (st[:, 0].max() - t0) / 100, (st[:, 31].min() - t0) / 100, (st[:, 31].max() - t0) / 100))
    print('   entry -> loads issued %.2f us' % ((st[:, 1] - st[:, 0]).mean() / 100))
    names = ['barrier', 'wait loads', 'split + LDS writes', 'barrier', 'next loads + MFMAs']
    prev = st[:, 1]
    for c in range(chunks):
        s = 2 + 5 * c
        d = [(st[:, s] - prev)] + [st[:, s + i + 1] - st[:, s + i] for i in range(4)]
        print('   chunk %d: ' % c + ', '.join('%s %.2f' % (n, v.mean() / 100) for n, v in zip(names, d)))
        prev = st[:, s + 4]
    print('   loop end -> stores issued + drained %.2f us ; workgroup lifetime %.2f us (mean)' % (
        (st[:, 31] - st[:, 30]).mean() / 100, (st[:, 31] - st[:, 0]).mean() / 100))


report('weight gradient (K x rows operands, 128-row slices)',
       lambda: ops.link_wgrad(link, rows, ops._operand(g), ops._operand(x), dw, db, 1), 4)
report('forward (rows x K operands, K = 128)', lambda: ops.link_down(link, rows, ops._operand(x), w, b, 0, None), 4)
