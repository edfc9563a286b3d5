"""Diagnostic (needs a build with ARVAE_HIPCC_FLAGS=-DARVAE_STAMPS): phase timeline of down32_kernel<16>."""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arvae_amd import ops, _lib
lib = _lib.load()
dev = torch.device('cuda:0')
n = 512
link = ops.Link(32, 32, 32, 16, 16, 32, 4, 4, 2, 1)
x = torch.randn(n, 32, 32, 32, device=dev)
w = torch.randn(32, 32, 4, 4, device=dev) * 0.1
b = torch.zeros(32, device=dev)
for _ in range(3):
    y = ops.link_down(link, n, ops._operand(x), w, b, 1, None)
torch.cuda.synchronize()
cnt = 512 * 8 * 6
buf = (ctypes.c_ulonglong * cnt)()
lib._handle  # noqa
fn = ctypes.CDLL(_lib.LIB_PATH).arvae_debug_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert fn(buf, cnt) == 0
st = np.array(buf, dtype=np.uint64).reshape(512, 8, 6).astype(np.int64)
names = ['top->committed(barriers+commit)', 'issue', 'mfma', 'barrier+redwrite+barrier', 'epilogue', 'loop back']
t0 = st[:, 0, 0].min()
for wg in (0, 1, 256, 257, 511):
    print('wg', wg, 'start', st[wg, 0, 0] - t0, 'tiles:')
    for it in range(4):
        d = np.diff(st[wg, it])
        nxt = st[wg, it + 1, 0] - st[wg, it, 5] if it < 3 else 0
        print('   ', it, dict(zip(['commit', 'issue', 'mfma', 'redwr', 'epi'], d.tolist())), 'gap', int(nxt))
d = np.diff(st[:, :4, :], axis=2)
print('mean over WGs/tiles:', dict(zip(['commit', 'issue', 'mfma', 'redwr', 'epi'], d.reshape(-1, 5).mean(0).round().tolist())))
print('tile period mean', (st[:, 1:4, 0] - st[:, 0:3, 0]).mean(), 'total', (st[:, 3, 5] - st[:, 0, 0]).mean())
