#!/usr/bin/env python3
"""K1 / K2 on the headline layout when the qualities are distributed as instruments emit them (a few distinct values, one of them dominant)
instead of uniformly over Q0-41 as BASELINE's synthetic reads have them: timing only (the corrected plane is kept, so the 'errors' no longer
follow the qualities -- irrelevant for the kernels' work)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd'))
import torch
from kbbq import _device as dev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
ctx = dev.context()
for name, values, probs in (('uniform Q0-41 (BASELINE)', None, None),
                            ('4 bins: Q2 2 %, Q11 5 %, Q25 13 %, Q37 80 %', (2, 11, 25, 37), (0.02, 0.05, 0.13, 0.80)),
                            ('8 bins, Q37-41 dominant', (2, 8, 14, 22, 27, 33, 37, 41), (0.01, 0.02, 0.03, 0.05, 0.09, 0.15, 0.35, 0.30)),
                            ('one value: Q30 everywhere', (30,), (1.0,))):
    b = dev.ReadBatch.synthetic(0, n, n, seed=1)
    if values is not None:
        step = 5_000_000
        edges = torch.tensor(probs, device='cuda').cumsum(0)
        vals = torch.tensor(values, dtype=torch.uint8, device='cuda') + 33
        for lo in range(0, n, step):
            q = b.qual[lo:lo + step]
            u = torch.rand(q.shape, device='cuda')
            q.copy_(torch.where(q != 0, vals[torch.bucketize(u, edges).clamp_(max=len(values) - 1)], q))
            del u
    laid = dev.lay_out(b, 1, 150, packed=True)
    del b
    torch.cuda.empty_cache()
    t = dev.Tables(1, 300)
    out = torch.empty_like(laid.qual)
    for rep in range(2):
        ctx.kernel_ms(0, reset=True); ctx.kernel_ms(1, reset=True); ctx.timing(True)
        for _ in range(5):
            t.buf.zero_()
            dev.accumulate(laid, t, check=False)
            lut, shape = dev.solve_lut(t, check=False, reuse=True)
            dev.apply(laid, lut, shape, out=out, check=False)
        ctx.timing(False)
        k1, n1 = ctx.kernel_ms(0); k2, n2 = ctx.kernel_ms(1)
    try:
        ctx.status()
    except Exception as e:
        print('   status:', type(e).__name__)
    print('%-48s K1 %.3f ms  K2 %.3f ms' % (name, k1 / n1, k2 / n2), flush=True)
    del laid, out
    torch.cuda.empty_cache()
