"""Diagnostic (tools/build_diag.sh lib_s8st conv64.hip -DS8_STAMPS; ARVAE_LIB=tools/bin/lib_s8st.so): cycles per phase of a tile of
conv_s8_h2_kernel (the Morpho-MNIST 8 -> 64 products), thread 0 of workgroup 0, the LAST launch of a training step."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda:0')
step, eager, unit = bench.build_side_workload('mnist', dev, 1024, 0, False, False)
for i in range(3):
    eager(i)
torch.cuda.synchronize()
from arvae_amd import _lib
fn = ctypes.CDLL(os.environ.get('ARVAE_LIB') or _lib.LIB_PATH).arvae_debug_s8_stamps
fn.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 8)()
assert fn(buf) == 0
names = ['loop top (geometry)', 'barrier 1 (previous reads) + split + LDS writes', 'barrier 2', 'epilogue operands + next source requested',
         'operand reads + 24 MFMAs + result tile -> LDS', 'barrier 3', 'epilogue: LDS reads, activation, gate, stores']
tiles = buf[7]
tot = sum(buf[k] for k in range(7))
for k in range(7):
    print(f'{names[k]:56s} {buf[k] / tiles:8.0f} cycles/tile {100 * buf[k] / tot:5.1f}%')
print(f'total {tot / tiles:.0f} cycles/tile over {tiles} tiles (s_memtime ticks: 100 MHz -> x10 ns)')
