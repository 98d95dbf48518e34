#!/usr/bin/env python3
"""Do the streaming kernels care which XCD reads which part of the planes?  Headline layout (50 M x 2 x 150 bp, mate-pair rows, 4-bit planes):
K2 with every XCD taking a contiguous eighth of the tiles (KBBQ_K2_XCD_TILES=1, the default) against tile b for workgroup b (=0), alternating;
results compared.  (The same for K1 -- every XCD's workgroups walking a contiguous eighth of the rows -- was measured with this script's first
form and a switch that has since been removed: 3.40-3.44 against 3.36-3.43 ms, nothing.)
usage (GPU box): python scripts/time_xcd_streams.py [reads]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd')); sys.path.insert(0, ROOT)
import torch, bench
from kbbq import _device as dev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
res = bench.Resident(dev, torch, 0, n, 1, 1, 'packed')
res.free_rows(0)
ctx = dev.context()
dev.accumulate(res.batch, res.tables)
lut, shape = dev.solve_lut(res.tables)
want_t = res.tables.buf.clone()
dev.apply(res.batch, lut, shape, out=res.out, check=False)
want_o = res.out.clone()
for rep in range(4):
    for k1, k2 in (('0', '0'), ('1', '1')):
        os.environ['KBBQ_K2_XCD_TILES'] = k2
        ctx.kernel_ms(0, reset=True); ctx.kernel_ms(1, reset=True); ctx.timing(True)
        for _ in range(5):
            res.tables.buf.zero_()
            dev.accumulate(res.batch, res.tables, check=False)
            dev.apply(res.batch, lut, shape, out=res.out, check=False)
        ctx.timing(False)
        ok = torch.equal(res.tables.buf, want_t) and torch.equal(res.out, want_o)
        print('rep %d XCD-contiguous %s: K1 %.3f ms  K2 %.3f ms  same results %s' % (rep, 'yes' if k1 == '1' else 'no ', ctx.kernel_ms(0)[0] / 5, ctx.kernel_ms(1)[0] / 5, ok), flush=True)
