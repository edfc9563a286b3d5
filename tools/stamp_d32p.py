"""Diagnostic (tools/build_diag.sh lib_d32pst conv32.hip -DD32K_STAMPS; ARVAE_LIB=tools/bin/lib_d32pst.so): phase timeline of the
LAST down32p launch of a forward pass -- consumers (thread 0) and producers (thread 256) of workgroups 0..31."""
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
    trainer.zero_grad(); loss, _ = trainer.loss_and_acc_for_batch((x, lab), 0, i, True)
torch.cuda.synchronize()
fn = ctypes.CDLL(os.environ.get('ARVAE_LIB') or _lib.LIB_PATH).arvae_debug_d32k_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * (64 * 64))()
assert fn(buf, 64 * 64) == 0
st = np.array(buf, dtype=np.uint64).reshape(64, 64).astype(np.int64)
con, pro = st[:32], st[32:]
print('consumer entry -> loop start: %.2f us; producer entry is %.2f us after consumer entry' %
      ((con[:, 1] - con[:, 0]).mean() / 100, (pro[:, 0] - con[:, 0]).mean() / 100))
nt = 0
while 8 + 5 * nt < 64 and con[0, 8 + 5 * nt] > con[0, 0]:
    nt += 1
print('tiles per workgroup seen:', nt)
for k in range(nt):
    c = [con[:, 5 + 5 * k] - con[:, 4 + 5 * k], con[:, 6 + 5 * k] - con[:, 5 + 5 * k], con[:, 7 + 5 * k] - con[:, 6 + 5 * k], con[:, 8 + 5 * k] - con[:, 7 + 5 * k]]
    p = [pro[:, 5 + 5 * k] - pro[:, 4 + 5 * k], pro[:, 6 + 5 * k] - pro[:, 5 + 5 * k], pro[:, 7 + 5 * k] - pro[:, 6 + 5 * k]]
    print('tile %d  consumer: k-loop %.2f, wait A %.2f, exchange + wait B %.2f, epilogue %.2f | producer: commit + loads %.2f, wait A %.2f, wait B %.2f | start %.2f us'
          % ((k,) + tuple(v.mean() / 100 for v in c) + tuple(v.mean() / 100 for v in p) + ((con[:, 4 + 5 * k] - con[:, 0]).mean() / 100,)))
print('total %.2f us' % ((con[:, 8 + 5 * (nt - 1)] - con[:, 0]).mean() / 100))
