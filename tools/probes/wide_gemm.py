"""Timing probe of the wide tile GEMM (csrc/dense.hip wide_gemm_x3_kernel) through arvae_debug_wide_gemm: Morpho-MNIST's four
products at B = 1024, and what is left of each with the MFMAs (1) / the LDS commits (2) / the result stores (4) switched off.
    python tools/probes/wide_gemm.py"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from arvae_amd import _lib  # noqa: E402

dll = ctypes.CDLL(os.environ.get('ARVAE_LIB') or os.path.join(ROOT, 'ar-vae_amd', 'libarvae_hip_diag.so'))      # (arvae_debug_* live in the diagnostic build)
fn = dll.arvae_debug_wide_gemm
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_int] * 6 + [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device('cuda:0')


def timeit(f, n=40):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1000


M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cases = [('F1 x0 . W_e0 (fp32 A, split K)', 0, 0, 1, M, 256, 2888, 8), ('F2 y_d0 . W_d1 (planes A)', 1, 0, 0, M, 2888, 256, 1),
         ('B1 g . W_d1 (fp32 A, K x rows B, split K)', 0, 1, 1, M, 256, 2888, 8), ('B2 gpre . W_e0 (planes A, K x rows B)', 1, 1, 0, M, 2888, 256, 1)]
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, a_planes, b_krows, partial, m, n, k, slices in cases:
    kpad, npad = (k + 31) // 32 * 32, (n + 31) // 32 * 32
    a = torch.randn(3 * m * k if a_planes else m * k, device=dev).to(torch.bfloat16 if a_planes else torch.float32)
    b = torch.randn(3 * npad * kpad, device=dev).to(torch.bfloat16)
    out = torch.empty(slices * m * n, device=dev)
    row = []
    for dbg in (0, 1, 2, 3, 4, 7):
        def call():
            rc = fn(a_planes, b_krows, partial, m, n, k, a.data_ptr(), b.data_ptr(), out.data_ptr(), slices, dbg, stream)
            assert rc == 0, rc
        row.append('%d: %5.1f' % (dbg, timeit(call)))
    print('%-46s M %d N %d K %d x%d   us by dbg  %s' % (name, m, n, k, slices, '  '.join(row)), flush=True)
print('--- F2 shape, K sweep (dbg 0 / 7) ---')
for k in (32, 64, 128, 256, 512, 1024):
    m, n = M, 2888
    kpad, npad = (k + 31) // 32 * 32, (n + 31) // 32 * 32
    a = torch.randn(3 * m * k, device=dev).to(torch.bfloat16)
    b = torch.randn(3 * npad * kpad, device=dev).to(torch.bfloat16)
    out = torch.empty(m * n, device=dev)
    res = []
    for dbg in (0, 7):
        res.append(timeit(lambda: fn(1, 0, 0, m, n, k, a.data_ptr(), b.data_ptr(), out.data_ptr(), 1, dbg, stream)))
    print('K %4d: %5.1f us, loads only %5.1f us' % (k, res[0], res[1]), flush=True)
