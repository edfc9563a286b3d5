"""Diagnostic (build conv32k.hip with ARVAE_HIPCC_FLAGS=-DD32K_STAMPS): phase timeline of the LAST down32k launch of a step."""
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
    trainer.zero_grad(); loss, _ = trainer.loss_and_acc_for_batch((x, lab), 0, i, True); 
torch.cuda.synchronize()
fn = ctypes.CDLL(_lib.LIB_PATH).arvae_debug_d32k_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * (64 * 64))()
assert fn(buf, 64 * 64) == 0
st = np.array(buf, dtype=np.uint64).reshape(64, 64).astype(np.int64)
# forward only: the last down32k launch is the 8x8 layer (2 tiles per workgroup); the 16x16 one is overwritten -> run with
# ARVAE_ONLY16=1 handled below by reading tiles count
t0 = st[:, 0].min()
print('kernel entry -> loop: %.2f us (mean)' % ((st[:, 1] - st[:, 0]).mean() / 100))
nt = 0
while 8 + 5 * nt < 64 and st[0, 8 + 5 * nt] > st[0, 0]:
    nt += 1
print('tiles per workgroup seen:', nt)
names = ['k-loop', 'wait barrier 1', 'write partials + barrier 2', 'sum + epilogue']
for k in range(nt):
    d = [(st[:, 5 + 5 * k] - st[:, 4 + 5 * k]), (st[:, 6 + 5 * k] - st[:, 5 + 5 * k]), (st[:, 7 + 5 * k] - st[:, 6 + 5 * k]), (st[:, 8 + 5 * k] - st[:, 7 + 5 * k])]
    print('tile %d: ' % k + ', '.join('%s %.2f' % (n, v.mean() / 100) for n, v in zip(names, d)), ' | start at %.2f us' % ((st[:, 4 + 5 * k] - st[:, 0]).mean() / 100))
print('total %.2f us' % ((st[:, 8 + 5 * (nt - 1)] - st[:, 0]).mean() / 100))
