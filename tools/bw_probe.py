import torch, time
d=torch.device('cuda')
for mb in (64, 256, 1024):
    n=mb*1024*1024//4
    a=torch.empty(n,device=d); b=torch.empty(n,device=d)
    for name,fn,bytes_ in (('fill',lambda: a.fill_(1.0),n*4),('copy',lambda: b.copy_(a),n*8),('read(sum)',lambda: a.sum(),n*4)):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t=time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize(); dt=(time.perf_counter()-t)/20
        print(f'{mb:5d} MB {name:10s} {dt*1e6:8.1f} us  {bytes_/dt/1e12:6.2f} TB/s')
