import sys, torch
sys.path.insert(0, '/root/repo')
from arvae_amd import ops
dev = torch.device('cuda:0')
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1000
for rows, fin, fout in ((2048, 2888, 256), (2048, 256, 2888), (1024, 2888, 256), (1024, 256, 2888)):
    x = torch.randn(rows, fin, device=dev); w = torch.randn(fout, fin, device=dev) * 0.02; b = torch.zeros(fout, device=dev)
    with torch.no_grad():
        t = timeit(lambda: ops.dense(x, w, b, ops.Link.dense(fin, fout), 2))
    print(rows, fin, fout, 'fwd %.1f us' % t)
