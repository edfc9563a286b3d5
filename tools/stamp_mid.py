"""Diagnostic (build midblock.hip with ARVAE_HIPCC_FLAGS=-DMID_STAMPS): phase timeline of mid_forward_kernel at B = 512."""
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
fn = ctypes.CDLL(os.environ.get('ARVAE_LIB') or _lib.LIB_PATH).arvae_debug_mid_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * (128 * 16))()
assert fn(buf, 128 * 16) == 0
st = np.array(buf, dtype=np.uint64).reshape(128, 16).astype(np.int64)
names = ['touch + x0 load', 'enc fc1', 'enc fc2', 'heads + z', 'dec fc3', 'dec fc4', 'dec fc5']
d = np.diff(st[:, :8], axis=1) / 100.0
for n, m, mx in zip(names, d.mean(0), d.max(0)):
    print(f'{n:18s} mean {m:6.2f} us   max {mx:6.2f} us')
print('total', (st[:, 7] - st[:, 0]).mean() / 100.0, 'us; spread of start', (st[:, 0].max() - st[:, 0].min()) / 100.0)
