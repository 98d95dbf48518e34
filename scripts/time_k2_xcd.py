#!/usr/bin/env python3
"""K2 storing through the permutation (rows grouped by read group going back into input order, BASELINE config 3's layout-inclusive leg):
workgroups of one relative position on the same XCD (KBBQ_K2_XCD=8, the default) against ranks as they are (=0), alternating in one
process; outputs compared.  usage (GPU box): python scripts/time_k2_xcd.py [reads] [read groups]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd')); sys.path.insert(0, ROOT)
import torch, bench
from kbbq import _device as dev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
R = int(sys.argv[2]) if len(sys.argv) > 2 else 8
res = bench.Resident(dev, torch, 0, n, 1, R, 'packed')
res.free_rows(0)
ctx = dev.context()
dev.accumulate(res.batch, res.tables)
lut, shape = dev.solve_lut(res.tables)
outs = {}
for rep in range(4):
    for xcd in ('0', '8', '4', '16'):
        os.environ['KBBQ_K2_XCD'] = xcd
        dev.apply(res.batch, lut, shape, out=res.out, check=False, restore_order=True)
        ctx.kernel_ms(1, reset=True); ctx.timing(True)
        for _ in range(5):
            dev.apply(res.batch, lut, shape, out=res.out, check=False, restore_order=True)
        ctx.timing(False)
        ms = ctx.kernel_ms(1)[0] / 5
        if rep == 0:
            outs[xcd] = res.out.clone()
        print('rep %d KBBQ_K2_XCD=%-2s K2 through the permutation %.3f ms' % (rep, xcd, ms), flush=True)
print('same bytes:', all(torch.equal(outs['0'], v) for v in outs.values()))
os.environ.pop('KBBQ_K2_XCD')
ctx.kernel_ms(1, reset=True); ctx.timing(True)
for _ in range(5):
    dev.apply(res.batch, lut, shape, out=res.out, check=False)
ctx.timing(False)
print('grouped order (no permutation): %.3f ms' % (ctx.kernel_ms(1)[0] / 5))
