"""Diagnostic (build conv64s.hip with -DC64S_STAMPS into a separate library): phase timeline of one conv64s launch
(64 -> 64 channels, 25x25 -> 22x22, n = 1024: the Morpho-MNIST layer)."""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arvae_amd import ops, _lib
lib = _lib.load()
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
link = ops.Link(25, 25, 64, 22, 22, 64, 4, 4, 1, 0)
hi = torch.randn(n, 25, 25, 64, device=dev)
w = torch.randn(64, 64, 4, 4, device=dev) * 0.05
b = torch.zeros(64, device=dev)
for _ in range(3):
    ops.link_down(link, n, ops._operand(hi), w, b, 2, None)
torch.cuda.synchronize()
fn = ctypes.CDLL(os.environ.get('ARVAE_LIB') or _lib.LIB_PATH).arvae_debug_c64s_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * (64 * 64))()
assert fn(buf, 64 * 64) == 0
st = np.array(buf, dtype=np.uint64).reshape(64, 64).astype(np.int64)
print('entry -> loop %.2f us' % ((st[:, 1] - st[:, 0]).mean() / 100))
names = ['k-loop', 'w-loads + barrier', 'exchange 0', 'epilogue 0', 'exchange 1', 'epilogue 1', 'to next tile']
for t in range(8):
    s = 2 + 6 * t
    d = np.diff(st[:, s:s + 7], axis=1).mean(0) / 100
    print('tile %d: ' % t + ', '.join('%s %.2f' % (nm, v) for nm, v in zip(names, d)) + '  | tile total %.2f' % ((st[:, s + 6] - st[:, s]).mean() / 100))
