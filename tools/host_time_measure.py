"""Diagnostic: host time per training step (enqueue only, no synchronisation inside the loop) against the device's step time -- how far
the Python side runs ahead of the stream.  python tools/host_time_measure.py [measure|mnist]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
kind = sys.argv[1] if len(sys.argv) > 1 else 'measure'
dev = torch.device('cuda:0')
step, _, unit = bench.build_side_workload(kind, dev, bench.SIDE_BATCH[kind])
for i in range(20):
    step(i)
torch.cuda.synchronize()
for n in (50, 200):
    t0 = time.perf_counter()
    for i in range(n):
        step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('%s, %d steps: host enqueue %.3f ms/step, until the stream drained %.3f ms/step' % (kind, n, 1e3 * (t1 - t0) / n, 1e3 * (t2 - t0) / n))
