#!/usr/bin/env python3
"""The persistent K2 (character planes, one read per row: what rows a caller already holds get) with its workgroups walking the rows as several
sequential fronts (KBBQ_K2_PARTS) instead of one; outputs compared.  usage (GPU box): python scripts/time_k2_parts_rows.py [reads]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd')); sys.path.insert(0, ROOT)
import torch, bench
from kbbq import _device as dev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
res = bench.Resident(dev, torch, 0, n, 1, 1, 'reads'); res.free_rows(0)
ctx = dev.context()
dev.accumulate(res.batch, res.tables); lut, shape = dev.solve_lut(res.tables)
os.environ['KBBQ_K2_PARTS'] = '0'
dev.apply(res.batch, lut, shape, out=res.out, check=False); want = res.out.clone()
for rep in range(3):
    for P in ('0', '4', '8', '16', '32'):
        os.environ['KBBQ_K2_PARTS'] = P
        res.out.zero_()
        dev.apply(res.batch, lut, shape, out=res.out, check=False)
        ok = torch.equal(res.out, want)
        ctx.kernel_ms(1, reset=True); ctx.timing(True)
        for _ in range(5): dev.apply(res.batch, lut, shape, out=res.out, check=False)
        ctx.timing(False)
        print('rep %d parts %-3s K2 (persistent, character rows) %.3f ms  same bytes %s' % (rep, P, ctx.kernel_ms(1)[0] / 5, ok), flush=True)
