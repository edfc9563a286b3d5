"""Diagnostic (tools/build_diag.sh lib_wgrst conv32.hip -DWGR_STAMPS; ARVAE_LIB=tools/bin/lib_wgrst.so): per-step timeline of wgrad32r_kernel<16>."""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arvae_amd import ops, _lib
lib = _lib.load()
dev = torch.device('cuda:0')
lo_sz = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
link = ops.Link(2 * lo_sz, 2 * lo_sz, 32, lo_sz, lo_sz, 32, 4, 4, 2, 1)
hi = torch.randn(n, 2 * lo_sz, 2 * lo_sz, 32, device=dev)
lo = torch.randn(n, lo_sz, lo_sz, 32, device=dev)
dw = torch.zeros(32, 32, 4, 4, device=dev); db = torch.zeros(32, device=dev)
fn = ctypes.CDLL(os.environ.get('ARVAE_LIB') or _lib.LIB_PATH).arvae_debug_wgr_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
for _ in range(20):
    ops.link_wgrad(link, n, ops._operand(lo), ops._operand(hi), dw, db, 1)
torch.cuda.synchronize()
cnt = 256 * 2 * 64 * 2
buf = (ctypes.c_ulonglong * cnt)()
assert fn(buf, cnt) == 0
st = np.array(buf, dtype=np.uint64).reshape(256, 2, 64, 2).astype(np.int64)
steps = int(os.environ.get('ARVAE_WGR_SPW', n * lo_sz * lo_sz // 32 // 256))
nwg = min(256, n * lo_sz * lo_sz // 32 // steps)
st = st[:nwg]
c, w = st[:, 0, :, 0], st[:, 0, :, 1]
p, pw = st[:, 1, :, 0], st[:, 1, :, 1]
t0 = min(w[:, 0].min(), pw[:, 0].min())
print(f'kernel span (wall stamps) {(max(w[:, 63].max(), pw[:, 63].max()) - t0) / 100:.1f} us; consumer start {(w[:,0].mean()-t0)/100:.2f} us, first barrier passed {(w[:,1].mean()-t0)/100:.2f} us, loop end {(w[:,62].mean()-t0)/100:.2f}, slab written {(w[:,63].mean()-t0)/100:.2f}')
span_c = (c[:, 62] - c[:, 1]).mean(); span_w = (w[:, 62] - w[:, 1]).mean()
print(f'consumer loop: {span_c:.0f} ticks = {span_w / 100:.2f} us -> {span_c / span_w * 100:.0f} MHz; {span_c / steps:.0f} ticks per step ({steps} steps)')
for i in range(min(steps, 16)):
    arrive, passed = c[:, 2 + 2 * i], c[:, 3 + 2 * i]
    prev = c[:, 1] if i == 0 else c[:, 3 + 2 * (i - 1)]
    pa, pb = p[:, 2 + 2 * i], p[:, 3 + 2 * i]
    print(f' step {i:2d}: consumer work {(arrive - prev).mean():6.0f}  wait at barrier {(passed - arrive).mean():6.0f} | producer issue+commit {(pb - pa).mean():6.0f}')
