#!/usr/bin/env python3
"""k7_lay_out (input-order character rows -> mate-pair rows, 4-bit planes, gathered by read group) with its workgroups walking the destination
rows as several sequential fronts (KBBQ_K7_PARTS), 1 and 8 read groups; planes compared.  usage (GPU box): python scripts/time_layout_parts.py [reads]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd'))
import torch
from kbbq import _device as dev
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
for R in (1, 8):
    b = dev.ReadBatch.synthetic(0, n, n, seed=1, nrg=R)
    os.environ['KBBQ_K7_PARTS'] = '0'
    want = dev.lay_out(b, R, 150, packed=True)
    for rep in range(3):
        for P in ('0', '8', '16', '4'):
            os.environ['KBBQ_K7_PARTS'] = P
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            got = dev.lay_out(b, R, 150, packed=True); del got
            a.record(); got = dev.lay_out(b, R, 150, packed=True); e.record(); torch.cuda.synchronize()
            ok = all(torch.equal(getattr(got, k), getattr(want, k)) for k in ('seq', 'cseq', 'qual', 'meta'))
            print('%d read group(s) rep %d parts %-2s lay_out %.3f ms  same planes %s' % (R, rep, P, a.elapsed_time(e), ok), flush=True)
            del got
    del b, want
    torch.cuda.empty_cache()
