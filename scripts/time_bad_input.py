#!/usr/bin/env python3
"""What a batch costs in which EVERY chunk is flagged (qualities above 42 everywhere: another encoding): K1 and K2 on 50 M such reads -- the
status words take one same-address atomic per flagged chunk unless a lane first looks whether the word can still change (flag(), kbbq_kernels.h).
usage (GPU box): python scripts/time_bad_input.py [reads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd')); sys.path.insert(0, ROOT)
import torch, bench
from kbbq import _device as dev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
ctx = dev.context()
for layout in ('packed', 'reads'):
    res = bench.Resident(dev, torch, 0, n, 1, 1, layout); res.free_rows(0)
    dev.accumulate(res.batch, res.tables); lut, shape = dev.solve_lut(res.tables)
    res.batch.qual.add_(31)                      # phred+64: every counted quality above 42
    for name, fn in (('K1', lambda: dev.accumulate(res.batch, res.tables, check=False)), ('K2', lambda: dev.apply(res.batch, lut, shape, out=res.out, check=False))):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        try:
            ctx.status(); what = 'no status?'
        except Exception as e:
            what = type(e).__name__
        print('%s rows, %s on %d M reads whose every chunk is flagged: %.1f ms (%s)' % (res.name.split(',')[0], name, n // 1_000_000, dt * 1e3, what), flush=True)
    del res
    torch.cuda.empty_cache()
