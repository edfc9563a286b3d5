"""Diagnostic: one row-staged 64 -> 64 launch (25x25 -> 22x22, n = 1024: the Morpho-MNIST layer) timed with HIP events,
with and without a keep-mask, for the library named by ARVAE_LIB."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arvae_amd import ops, _lib
lib = _lib.load()
dev = torch.device('cuda:0')
n = 1024
link = ops.Link(25, 25, 64, 22, 22, 64, 4, 4, 1, 0)
hi = torch.randn(n, 25, 25, 64, device=dev)
w = torch.randn(64, 64, 4, 4, device=dev) * 0.05
b = torch.zeros(64, device=dev)
mask = (torch.rand(n, 22, 22, 64, device=dev) > 0.5).to(torch.uint8)
out = torch.empty(n, 22, 22, 64, device=dev)
for name, m in (('no mask', None), ('keep-mask', mask)):
    for _ in range(3):
        ops.link_down(link, n, ops._operand(hi), w, b, 2, m, out=out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.link_down(link, n, ops._operand(hi), w, b, 2, m, out=out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 100)
    print('%-10s %.1f us per call (weight prep + amax + conv)' % (name, min(ts)))
