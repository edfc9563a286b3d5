"""Diagnostic (needs a build with ARVAE_HIPCC_FLAGS=-DARVAE_GRU_STAMPS): cycles per phase of a GRU sequence step."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arvae_amd import ops, _lib
lib = _lib.load()
dev = torch.device('cuda:0')
T, R, H = 24, 256, 128
fn = ctypes.CDLL(os.environ.get('ARVAE_LIB') or _lib.LIB_PATH).arvae_debug_gru_stamps
fn.argtypes = [ctypes.c_void_p]
gi = [torch.randn(T, R, 3 * H, device=dev) for _ in range(2)]
w = [torch.randn(3 * H, H, device=dev) * 0.05 for _ in range(2)]
b = [torch.zeros(3 * H, device=dev) for _ in range(2)]
for _ in range(3):
    with torch.no_grad():
        ops.gru_sequence(T, [(gi[0], w[0], b[0], None, False), (gi[1], w[1], b[1], None, True)])
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)()
assert fn(buf) == 0
names = ['prefetch issue + deferred stores', 'LDS operand reads + MFMAs', 'gate math + LDS writes', 'barrier']
steps = buf[4]
tot = sum(buf[k] for k in range(4))
for k in range(4):
    print(f'{names[k]:36s} {buf[k] / steps:8.0f} cycles/step {100 * buf[k] / tot:5.1f}%')
print(f'total {tot / steps:.0f} cycles/step (s_memtime ticks)')
