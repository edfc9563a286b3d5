"""Experiment: dSprites step with fwd+bwd replayed from a HIP graph vs eager."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from arvae_amd import synthetic as syn
from arvae_amd.graphed import GraphedStep
dev = torch.device('cuda:0')
trainer, _ = bench.build_trainer(dev, 1)
x, lab = syn.dsprites_batch(512, seed=1234)
x, lab = torch.from_numpy(x).to(dev), torch.from_numpy(lab).to(dev)


def eager(i):
    trainer.zero_grad()
    loss, _ = trainer.loss_and_acc_for_batch((x, lab), 0, i, True)
    loss.backward()
    trainer.step()


for i in range(30): eager(i)
torch.cuda.synchronize(); t = time.perf_counter()
for i in range(200): eager(i)
torch.cuda.synchronize(); print('eager ms/step', (time.perf_counter() - t) / 200 * 1e3)
g = GraphedStep(trainer, (x, lab))
for i in range(30):
    g((x, lab)); trainer.step()
torch.cuda.synchronize(); t = time.perf_counter()
for i in range(200):
    g((x, lab)); trainer.step()
torch.cuda.synchronize(); print('graph ms/step', (time.perf_counter() - t) / 200 * 1e3)
