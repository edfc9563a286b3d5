"""Experiment: can the MeasureVAE forward + backward be captured in a HIP graph (torch.cuda.CUDAGraph)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from arvae_amd import synthetic as syn

dev = torch.device('cuda:0')
from arvae_amd.measure_vae import MeasureVAE
from arvae_amd.measure_vae_trainer import MeasureVAETrainer
ds = bench.FolkDataset()
model = MeasureVAE(ds, 10, 2, 2, 128, 0.5, 32, 2, 128, 0.5, False, 'folk')
trainer = MeasureVAETrainer(ds, model, lr=1e-4, reg_type=('all',), reg_dim=(0, 1, 2, 3), beta=0.001, gamma=1.0, capacity=0.0,
                            rand=0, delta=10.0)
trainer.cuda()
model.train()
score = torch.from_numpy(syn.measure_batch(256, seed=5)).to(dev)


def fwd_bwd():
    trainer.zero_grad()
    loss, _ = trainer.loss_and_acc_for_batch((score, score), 0, 0, True)
    loss.backward()
    return loss


for _ in range(3):
    fwd_bwd(); trainer.step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    fwd_bwd(); trainer.step()
torch.cuda.synchronize()
print('eager ms/step', (time.perf_counter() - t) / 10 * 1e3)

model.decoder.use_teacher_forcing = False          # one control-flow variant for the experiment
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        fwd_bwd()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        static_loss = fwd_bwd()
except Exception as e:                                 # noqa: BLE001
    print('capture failed:', type(e).__name__, str(e)[:400])
    sys.exit(0)
torch.cuda.synchronize()
for _ in range(3):
    g.replay(); trainer.step()
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    g.replay(); trainer.step()
torch.cuda.synchronize()
print('graph ms/step', (time.perf_counter() - t) / 20 * 1e3, 'loss', float(static_loss))
