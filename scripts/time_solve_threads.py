import os, sys, time
sys.path.insert(0, 'kbbq-py_amd')
import numpy as np, torch
from kbbq import _device as dev, _solve
n = 5_000_000
b = dev.ReadBatch.synthetic(0, n, n, seed=1)
t = dev.Tables(1, 300); dev.accumulate(b, t); torch.cuda.synchronize()
def T(f, reps=50):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for th in (1, 2, 4, 6, 8, 12, 16):
    _solve.COMBILN_THREADS = th
    print('threads %2d: dev.solve %.3f ms' % (th, T(lambda: dev.solve(t))))
