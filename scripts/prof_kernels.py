#!/usr/bin/env python3
"""Profiling driver: K1 -> K3 -> K2 on a device-resident synthetic batch, a few repetitions.
Run under rocprofv3 (kernel trace or one --pmc set per run)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd'))

ap = argparse.ArgumentParser()
ap.add_argument('--reads', type=int, default=20_000_000)
ap.add_argument('--rgs', type=int, default=1)
ap.add_argument('--reps', type=int, default=3)
ap.add_argument('--len', type=int, default=150)
ap.add_argument('--pairs', action='store_true', help='mate-pair rows instead of one read per row')
ap.add_argument('--packed', action='store_true', help='the bench layout: dev.lay_out (mate-pair rows, 4-bit sequence planes, grouped by read group)')
args = ap.parse_args()

import torch
from kbbq import _device as dev

b = dev.ReadBatch.synthetic(0, args.reads, args.reads, seed=1, nrg=args.rgs, len_lo=args.len, len_hi=args.len)
if args.packed:
    b = dev.lay_out(b, args.rgs, args.len, packed=True)
else:
    if args.pairs:
        b = dev.PairBatch.from_reads(b)
    if args.rgs > 1:
        b = dev.group_by_rg(b, args.rgs)
out = torch.empty_like(b.qual)
t = dev.Tables(args.rgs, 2 * args.len)
for _ in range(args.reps):
    t.buf.zero_()
    dev.accumulate(b, t, check=False)
    lut, shape, _, _ = dev.solve(t)
    dev.apply(b, lut, shape, out=out, check=False)
torch.cuda.synchronize()
dev.context().status()
print('done', args.reads, 'reads x', args.reps)
