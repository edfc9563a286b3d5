"""Diagnostic (tools/build_diag.sh lib_up32pst conv32.hip -DARVAE_STAMPS; ARVAE_LIB=tools/bin/lib_up32pst.so): phase timeline of
the forward up32p launch of a fused dSprites step -- consumers (thread 0) and producers (thread 256) of every workgroup."""
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
backward = len(sys.argv) > 1 and sys.argv[1] == 'bwd'      # then the last up32p launch is the gated data gradient of conv2
for i in range(5):
    trainer.zero_grad(); loss, _ = trainer.loss_and_acc_for_batch((x, lab), 0, i, True)
    if backward:
        loss.backward()
torch.cuda.synchronize()
print('last up32p launch:', 'backward (EP_GATE_B)' if backward else 'forward (EP_RELU)')
fn = ctypes.CDLL(os.environ.get('ARVAE_LIB') or _lib.LIB_PATH).arvae_debug_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
cnt = 512 * 64 * 2
buf = (ctypes.c_ulonglong * cnt)()
assert fn(buf, cnt) == 0
st = np.array(buf, dtype=np.uint64).reshape(512, 64, 2).astype(np.int64)
con, pro = st[:256, :, 1], st[256:, :, 1]          # 100 MHz wall clock
us = lambda v: v.mean() / 100
print('compute entry -> loop start %.2f us ; whole kernel (compute waves) %.2f us, (store waves incl. last epilogue) %.2f us' %
      (us(con[:, 1] - con[:, 0]), us(con[:, 63] - con[:, 0]), us(pro[:, 63] - con[:, 0])))
for k in range(4):
    s = 3 + 6 * k
    print('tile %d compute: request + k-loop %.2f, split next patch %.2f, wait A %.2f, handoff %.2f, wait B %.2f | store: wait A %.2f, wait B %.2f, '
          'gate request %.2f, epilogue %.2f | start %.2f us'
          % (k, us(con[:, s + 1] - con[:, s]), us(con[:, s + 2] - con[:, s + 1]), us(con[:, s + 3] - con[:, s + 2]), us(con[:, s + 4] - con[:, s + 3]),
             us(con[:, s + 5] - con[:, s + 4]), us(pro[:, s + 1] - pro[:, s]), us(pro[:, s + 2] - pro[:, s + 1]), us(pro[:, s + 3] - pro[:, s + 2]),
             us(pro[:, s + 4] - pro[:, s + 3]), us(con[:, s] - con[:, 0])))
