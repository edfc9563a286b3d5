"""Diagnostic (needs a build with ARVAE_HIPCC_FLAGS=-DARVAE_STAMPS): phase timeline of the <16> conv32 kernels."""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arvae_amd import ops, _lib
lib = _lib.load()
dev = torch.device('cuda:0')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
link = ops.Link(32, 32, 32, 16, 16, 32, 4, 4, 2, 1)
hi = torch.randn(n, 32, 32, 32, device=dev)
lo = torch.randn(n, 16, 16, 32, device=dev)
w = torch.randn(32, 32, 4, 4, device=dev) * 0.1
b = torch.zeros(32, device=dev)
fn = ctypes.CDLL(_lib.LIB_PATH).arvae_debug_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ['sync', 'commit+sync', 'set_tile', 'mfma(+issue)', 'epilogue']


def report(tag, launch, wgs_per_cu=1):
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    cnt = 512 * 64 * 2
    buf = (ctypes.c_ulonglong * cnt)()
    assert fn(buf, cnt) == 0
    st = np.array(buf, dtype=np.uint64).reshape(512, 64, 2).astype(np.int64)
    tiles = n * 2
    nwg = min(256 * wgs_per_cu, tiles)
    per_wg = max(1, min(8, -(-tiles // nwg)))
    st = st[:nwg]
    t0 = st[:, 0, 1].min()
    print(f'== {tag}: {nwg} WGs x {per_wg} tiles; whole kernel {(st[:, 63, 1].max() - t0) / 100:.1f} us (wall clock stamps)')
    for lo_, hi_ in ((0, min(256, nwg)), (256, nwg)):
        if hi_ <= lo_:
            continue
        cyc, wall = st[lo_:hi_, :, 0], st[lo_:hi_, :, 1]
        tot_c = (cyc[:, 63] - cyc[:, 0]).mean()
        tot_w = (wall[:, 63] - wall[:, 0]).mean()
        print(f'  WGs {lo_}..{hi_ - 1}: start {(wall[:, 0].mean() - t0) / 100:.1f} us, end {(wall[:, 63].mean() - t0) / 100:.1f} us, '
              f'span {tot_c:.0f} ticks = {tot_w / 100:.1f} us -> {tot_c / tot_w * 100:.0f} MHz')
        print('   weights', (cyc[:, 1] - cyc[:, 0]).mean().round(), ' init+first issue', (cyc[:, 2] - cyc[:, 1]).mean().round())
        for t in range(per_wg):
            s = 3 + 6 * t
            d = np.diff(cyc[:, s:s + 6], axis=1).mean(0).round()
            print('   tile', t, dict(zip(names, d.tolist())))
        print('   drain', (cyc[:, 63] - cyc[:, 3 + 6 * per_wg - 1]).mean().round())


report('down32<16>', lambda: ops.link_down(link, n, ops._operand(hi), w, b, 1, None))
report('up32<16>', lambda: ops.link_up(link, n, ops._operand(lo), w, b, 1, None))

g_lo = torch.randn(n, 16, 16, 32, device=dev)
dw = torch.zeros_like(w)
db = torch.zeros_like(b)
report('wgrad32<16>', lambda: ops.link_wgrad(link, n, ops._operand(g_lo), ops._operand(hi), dw, db, 1))
