"""Error of the whole-sequence GRU kernels against torch.nn.GRU (fp32 CPU and float64) over random parameter draws:
how much room the tolerances of tests/test_hip_parity.py::test_gru_sequence_vs_torch leave."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from arvae_amd import ops
dev = torch.device('cuda:0')
rows, hid, steps, fin = 37, 128, 24, 10
worst = {}
for seed in range(12):
    torch.manual_seed(seed)
    rs = np.random.RandomState(15 + seed)
    gru = torch.nn.GRU(fin, hid, 1, bidirectional=True)
    x = torch.from_numpy(rs.standard_normal((steps, rows, fin)).astype(np.float32)).requires_grad_(True)
    h0 = torch.from_numpy(rs.standard_normal((2, rows, hid)).astype(np.float32)).requires_grad_(True)
    gy = torch.from_numpy(rs.standard_normal((steps, rows, 2 * hid)).astype(np.float32))
    y, hn = gru(x, h0)
    (y * gy).sum().backward()
    g64 = torch.nn.GRU(fin, hid, 1, bidirectional=True).double()
    g64.load_state_dict({k: v.double() for k, v in gru.state_dict().items()})
    y64, _ = g64(x.detach().double(), h0.detach().double())
    prm = {k: v.detach().clone().to(dev).requires_grad_(True) for k, v in gru.named_parameters()}
    xd = x.detach().to(dev).requires_grad_(True)
    hd = h0.detach().to(dev).requires_grad_(True)
    dirs = []
    for d, suf in enumerate(('', '_reverse')):
        gi = ops.dense(xd.view(steps * rows, fin), prm['weight_ih_l0' + suf], prm['bias_ih_l0' + suf], ops.Link.dense(fin, 3 * hid), 0).view(steps, rows, 3 * hid)
        dirs.append((gi, prm['weight_hh_l0' + suf], prm['bias_hh_l0' + suf], hd[d], d == 1))
    yd, _ = ops.gru_sequence(steps, dirs)
    (yd * gy.to(dev)).sum().backward()
    e_hip = float((yd.cpu().double() - y64).abs().max()); e_cpu = float((y.double() - y64).abs().max())
    e_vs = float((yd.cpu() - y).abs().max())
    gx = float((xd.grad.cpu() - x.grad).abs().max() / x.grad.abs().max())
    gw = max(float((prm[k].grad.cpu() - v.grad).abs().max() / v.grad.abs().max()) for k, v in gru.named_parameters())
    print(f'seed {seed}: |y_hip - y64| {e_hip:.2e}  |y_cpu32 - y64| {e_cpu:.2e}  |y_hip - y_cpu32| {e_vs:.2e}   dx rel-max {gx:.2e}  dW rel-max {gw:.2e}')
