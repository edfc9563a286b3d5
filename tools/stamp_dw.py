"""Diagnostic (build dense.hip with ARVAE_HIPCC_FLAGS=-DDW_STAMPS): phase timeline of dense_wgrad_batch_kernel tiles at B = 512."""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from arvae_amd import _lib, synthetic as syn
dev = torch.device('cuda:0')
trainer, _ = bench.build_trainer(dev, False)
x, lab = syn.dsprites_batch(512, seed=1)
x, lab = torch.from_numpy(x).to(dev), torch.from_numpy(lab).to(dev)
for i in range(5):
    trainer.zero_grad(); loss, _ = trainer.loss_and_acc_for_batch((x, lab), 0, i, True); loss.backward(); trainer.step()
torch.cuda.synchronize()
fn = ctypes.CDLL(_lib.LIB_PATH).arvae_debug_dw_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * (512 * 8))()
assert fn(buf, 512 * 8) == 0
st = np.array(buf, dtype=np.uint64).reshape(512, 8).astype(np.int64)
st = st[st[:, 0] > 0]
t0 = st[:, 0].min()
names = ['start -> first round done', 'first round -> loop done', 'loop done -> barrier', 'partials to LDS', 'sum + stores']
cols = [(0, 2), (2, 3), (3, 4), (4, 5), (5, 6)]
print(len(st), 'tiles; start spread', (st[:, 0].max() - t0) / 100.0, 'us; last end', (st[:, 6].max() - t0) / 100.0, 'us')
for n, (a, b) in zip(names, cols):
    d = (st[:, b] - st[:, a]) / 100.0
    print(f'{n:28s} mean {d.mean():6.2f} us   max {d.max():6.2f} us')
