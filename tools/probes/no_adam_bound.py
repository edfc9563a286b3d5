# upper bound of what fusing the Adam launch away could buy: the same step with the launch skipped (no launch in its place)
import sys, json, subprocess, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
which = sys.argv[1]
import torch
sys.argv = ['bench.py', '--no-cpu-baseline', '--no-secondary']
import bench
from arvae_amd import ops
if which == 'skip':
    def fake(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0, zero_grad=False, status=None):
        return                                               # (no launch at all: the arena keeps accumulating, the timing does not care)
    ops.adam_step = fake
    import arvae_amd.optim as optim
    optim.ops.adam_step = fake
bench.main()
