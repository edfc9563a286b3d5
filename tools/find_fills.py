"""Diagnostic: which torch ops in one training step launch fill / copy / elementwise kernels (host-side stacks)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device('cuda:0')
trainer, _ = bench.build_trainer(dev, 1)
from arvae_amd import synthetic as syn
x, lab = syn.dsprites_batch(512, seed=1234)
x, lab = torch.from_numpy(x).to(dev), torch.from_numpy(lab).to(dev)


def step(i):
    trainer.zero_grad()
    loss, _ = trainer.loss_and_acc_for_batch((x, lab), 0, i, True)
    loss.backward()
    trainer.step()


for i in range(3):
    step(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(3)
    torch.cuda.synchronize()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith('aten::') and ev.name in (
            'aten::fill_', 'aten::zero_', 'aten::copy_', 'aten::normal_', 'aten::add', 'aten::mul', 'aten::ones_like',
            'aten::zeros', 'aten::zeros_like', 'aten::contiguous', 'aten::clone', 'aten::_to_copy'):
        st = [s for s in (ev.stack or []) if 'site-packages' not in s and 'tools/find_fills' not in s][:3]
        print(f'{ev.name:18s} {str(ev.input_shapes)[:40]:40s}', ' <- '.join(s.split('/')[-1] for s in st))
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=12, max_name_column_width=60))
