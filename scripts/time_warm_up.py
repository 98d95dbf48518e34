#!/usr/bin/env python3
"""What the command line's device warm-up (kbbq._device.warm_up: ~0.2 s of a 0.95 s command, and on its critical path -- the
file scan hides behind it, the fill waits for it) consists of, in a fresh torch-free process.  usage (GPU box): python scripts/time_warm_up.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd'))
t = [time.perf_counter()]
def lap(what):
    t.append(time.perf_counter()); print('%-60s %7.1f ms' % (what, (t[-1] - t[-2]) * 1e3), flush=True)
import numpy as np
lap('import numpy')
from kbbq import _device as dev, _native as N
lap('import kbbq._device')
N.load()
lap('dlopen libkbbq_hip (+ the HIP runtime)')
dev.use_native_memory()
lap('use_native_memory (kbbq_device_count: runtime initialisation)')
ctx = dev.context(0)
lap('kbbq_ctx_create (context, stream, status words)')
b = dev.ReadBatch.synthetic(0, 128, 128, seed=1)
ctx.sync()
lap('first kernel (ks_synth: code object load)')
tables = dev.Tables(1, 300)
dev.accumulate(b, tables); ctx.sync()
lap('K1 on 128 reads')
tab = dev.device_logtab(0)
lap('device gammaln proven against the host (1.3 M arguments)')
lut, shape = dev.solve_lut(tables); ctx.sync()
lap('solve')
out = dev.apply(b, lut, shape).cpu()
lap('K2 + download')
p = dev.pinned('ingest', 0, 80 << 20)
lap('80 MB page-locked buffer')
p2 = dev.pinned('ingest', 1, 80 << 20)
lap('another')
torch = dev._torch()
x = torch.empty((8_000_000 * 152,), dtype=torch.uint8, device='cuda')
lap('1.2 GB of device memory (hipMalloc)')
y = torch.empty((8_000_000 * 152,), dtype=torch.uint8, device='cuda')
lap('another 1.2 GB')
print('total %.1f ms' % ((t[-1] - t[0]) * 1e3))
os._exit(0)
