"""Diagnostic (a build of gru_seq.hip with -DARVAE_GRU_STAMPS [-DGRU_STAMP_WAVE=w], loaded through ARVAE_LIB): cycles per phase of a
tick of the free-running decoder (tick_free_run_h2_kernel), B = 256, H = 128, 4 beats x 6 ticks, vocabulary 35, dropout 0.5."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from arvae_amd import ops, _lib
lib = _lib.load()
dev = torch.device('cuda:0')
B, H, V, beats, tpb = 256, 128, 35, 4, 6
g = torch.Generator(device='cpu').manual_seed(1)
def rnd(*shape, s=0.1): return (torch.randn(*shape, generator=g) * s).to(dev)
weights = (rnd(3 * H, H), rnd(3 * H), rnd(3 * H, H), rnd(3 * H), rnd(3 * H, H), rnd(3 * H), rnd(V, H, s=0.5), rnd(V))
h0a, h0b = torch.tanh(rnd(beats * B, H, s=1.0)), torch.tanh(rnd(beats * B, H, s=1.0))
gib, ptab = rnd(beats * B, 3 * H, s=0.5), rnd(V + 1, 3 * H, s=0.5)
mask = (torch.rand(beats * tpb, B, H, generator=g) >= 0.5).to(torch.uint8).to(dev)
for _ in range(3):
    ops.tick_free_run(weights, h0a, h0b, gib, ptab, mask, 2.0, B, beats, tpb)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    ops.tick_free_run(weights, h0a, h0b, gib, ptab, mask, 2.0, B, beats, tpb)
e1.record(); torch.cuda.synchronize()
print('launch (weight prep + decoder): %.1f us' % (e0.elapsed_time(e1) * 100))
fn = ctypes.CDLL(os.environ.get('ARVAE_LIB') or _lib.LIB_PATH).arvae_debug_tick_stamps
fn.argtypes = [ctypes.c_void_p]
buf = (ctypes.c_ulonglong * 9)()
assert fn(buf) == 0
names = ['tick top: beat state / token projections requested', "layer 0 at the top (a beat's first tick only)",
         'layer 0 gates (wait for the projections) + LDS writes', 'barrier', 'layer 1: operand reads + MFMAs behind the weight stream',
         'layer 1 gates + LDS writes', "barrier + logits / argmax + the next tick's layer 0", 'barrier + candidates -> token']
ticks = buf[8]
tot = sum(buf[k] for k in range(8))
for k in range(8):
    print(f'{names[k]:58s} {buf[k] / ticks:8.0f} cycles/tick {100 * buf[k] / tot:5.1f}%')
print(f'total {tot / ticks:.0f} cycles/tick')
