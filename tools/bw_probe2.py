"""HBM streaming rates of plain torch kernels at the c1 kernels' sizes (what a write- or read-dominated kernel can reach)."""
import torch, time
dev = torch.device('cuda:0')
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for mb in (67, 134, 268):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
    t_fill = timeit(lambda: a.fill_(1.0))
    t_copy = timeit(lambda: b.copy_(a))
    t_read = timeit(lambda: a.sum())
    print(f'{mb} MB: fill (write only) {t_fill:.1f} us = {mb * 1.048576 / t_fill:.2f} TB/s; copy {t_copy:.1f} us = {2 * mb * 1.048576 / t_copy:.2f} TB/s; sum (read only) {t_read:.1f} us = {mb * 1.048576 / t_read:.2f} TB/s')
