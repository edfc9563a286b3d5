"""Diagnostic (tools/bin/lib_midcst.so = midcluster.hip built with -DMIDC_STAMPS): phase timeline of the clustered latent block
(midc_forward_kernel / midc_backward_kernel) at B = 512: per phase the mean / max over the 256 workgroups, wall-clock stamps."""
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
fn = ctypes.CDLL(os.environ.get('ARVAE_LIB') or _lib.LIB_PATH).arvae_debug_midc_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
buf = (ctypes.c_ulonglong * (2 * 256 * 16))()
assert fn(buf, 2 * 256 * 16) == 0
st = np.array(buf, dtype=np.uint64).reshape(2, 256, 16).astype(np.int64)
names = [['weights issued + x0 -> LDS', 'enc0 product + slice store', 'arrive + poll (1)', 'gather (1)', 'enc1 product + store',
          'arrive + poll (2)', 'gather (2)', 'heads + z', 'dec0', 'dec1 product + store + arrive + poll (3)', 'gather (3)',
          'dec2 product', 'store + amax'],
         ['weights issued + g -> LDS', 'dec2^T product + store', 'arrive + poll (1)', 'gather (1)', 'dec1^T product + store',
          'arrive + poll (2)', 'gather (2)', 'dec0^T + d(mu, log_std)', 'heads^T', 'enc1^T product + store + arrive + poll (3)',
          'gather (3)', 'enc0^T product', 'store + amax']]
for pas, title in enumerate(('forward', 'backward')):
    s = st[pas]
    d = np.diff(s[:, :14], axis=1) / 100.0
    print(f'--- midc_{title}_kernel: first start -> last end {(s[:, 13].max() - s[:, 0].min()) / 100.0:.2f} us; per workgroup '
          f'{(s[:, 13] - s[:, 0]).mean() / 100.0:.2f} us; spread of start {(s[:, 0].max() - s[:, 0].min()) / 100.0:.2f} us')
    for n, m, mx in zip(names[pas], d.mean(0), d.max(0)):
        print(f'{n:44s} mean {m:6.2f} us   max {mx:6.2f} us')
    if pas == 0 and s[:, 14].any():
        print(f'   (dec0: product {(s[:, 14] - s[:, 8]).mean() / 100.0:.2f}, epilogue {(s[:, 15] - s[:, 14]).mean() / 100.0:.2f}, '
              f'next weights issued + barrier {(s[:, 9] - s[:, 15]).mean() / 100.0:.2f} us)')
