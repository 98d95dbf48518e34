#!/usr/bin/env python3
"""The short-lived K2 with its tiles cut into P parts (KBBQ_K2_XCD_TILES = P: workgroup b walks part b % P; 1 = 8, 0 = tile b for workgroup b), headline
layout, three alternating rounds.  usage (GPU box): python scripts/time_k2_parts.py"""
import os, sys
sys.path.insert(0, 'kbbq-py_amd'); sys.path.insert(0, '.')
import torch, bench
from kbbq import _device as dev
res = bench.Resident(dev, torch, 0, 50_000_000, 1, 1, 'packed'); res.free_rows(0)
ctx = dev.context()
dev.accumulate(res.batch, res.tables); lut, shape = dev.solve_lut(res.tables)
for rep in range(3):
    for P in ('0', '1', '16', '32', '64', '256', '2', '4'):
        os.environ['KBBQ_K2_XCD_TILES'] = P
        dev.apply(res.batch, lut, shape, out=res.out, check=False)
        ctx.kernel_ms(1, reset=True); ctx.timing(True)
        for _ in range(5): dev.apply(res.batch, lut, shape, out=res.out, check=False)
        ctx.timing(False)
        print('rep %d parts %-3s K2 %.3f ms' % (rep, P, ctx.kernel_ms(1)[0] / 5), flush=True)
