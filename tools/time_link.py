"""Time one link kernel family in isolation (HIP events, B = 512 by default):  python tools/time_link.py wgrad 16"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from arvae_amd import ops
what, lo = sys.argv[1], int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 512
dev = torch.device('cuda:0')
hi_t = torch.randn(n, 2 * lo, 2 * lo, 32, device=dev)
lo_t = torch.randn(n, lo, lo, 32, device=dev)
w = torch.randn(32, 32, 4, 4, device=dev) * 0.1
b = torch.randn(32, device=dev)
link = ops.Link(2 * lo, 2 * lo, 32, lo, lo, 32, 4, 4, 2, 1)
dw, db = torch.zeros_like(w), torch.zeros_like(b)
def run():
    if what == 'wgrad':
        ops.link_wgrad(link, n, ops._operand(lo_t), ops._operand(hi_t), dw, db, 1)
    elif what == 'down':
        ops.link_down(link, n, ops._operand(hi_t), w, b, ops.ACT_RELU, None)
    else:
        ops.link_up(link, n, ops._operand(lo_t), w, b, ops.ACT_RELU, None)
for _ in range(20): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 200
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
print(f'{what}<{lo}> n={n}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per call (includes its slab reduce / launch gaps)')
