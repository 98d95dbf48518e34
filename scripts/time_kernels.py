#!/usr/bin/env python3
"""Kernel timing (HIP events) of K1/K2 on a resident synthetic batch; honours KBBQ_ABLATE_* (timing-only builds)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd'))
ap = argparse.ArgumentParser()
ap.add_argument('--reads', type=int, default=20_000_000)
ap.add_argument('--rgs', type=int, default=1)
ap.add_argument('--reps', type=int, default=5)
ap.add_argument('--len', type=int, default=150)
ap.add_argument('--pairs', action='store_true', help='mate-pair rows instead of one read per row')
ap.add_argument('--group', action='store_true', help='rows grouped by read group')
ap.add_argument('--packed', action='store_true', help='the bench layout: dev.lay_out (mate-pair rows, 4-bit sequence planes, grouped by read group)')
ap.add_argument('--single', action='store_true', help='with --packed: one read per row on 4-bit planes (what single-end and mixed-length inputs get)')
ap.add_argument('--restore', action='store_true', help='K2 stores through the permutation (rows grouped by read group go back into input order)')
args = ap.parse_args()
import torch
from kbbq import _device as dev
b = dev.ReadBatch.synthetic(0, args.reads, args.reads, seed=1, nrg=args.rgs, len_lo=args.len, len_hi=args.len)
if args.packed:
    b = dev.lay_out(b, args.rgs, args.len, packed=True, pairs=False if args.single else None)
else:
    if args.pairs:
        b = dev.PairBatch.from_reads(b)
    if args.group:
        b = dev.group_by_rg(b, args.rgs)
out = torch.empty_like(b.qual)
t = dev.Tables(args.rgs, 2 * args.len)
ctx = dev.context()
dev.accumulate(b, t, check=False)
lut, shape, _, _ = dev.solve(t)
dev.apply(b, lut, shape, out=out, check=False, restore_order=args.restore)
torch.cuda.synchronize()
ctx.kernel_ms(0, reset=True); ctx.kernel_ms(1, reset=True); ctx.timing(True)
for _ in range(args.reps):
    dev.accumulate(b, t, check=False)
    dev.apply(b, lut, shape, out=out, check=False, restore_order=args.restore)
torch.cuda.synchronize()
k1, n1 = ctx.kernel_ms(0); k2, n2 = ctx.kernel_ms(1)
bases = args.reads * args.len
print('ABLATE K1=%s K2=%s reads=%d rgs=%d len=%d: K1 %.3f ms (%.0f GB/s alg)  K2 %.3f ms (%.0f GB/s alg)' % (
    os.environ.get('KBBQ_ABLATE_K1', '0'), os.environ.get('KBBQ_ABLATE_K2', '0'), args.reads, args.rgs, args.len,
    k1 / n1, 3 * bases / (k1 / n1) / 1e6, k2 / n2, 3 * bases / (k2 / n2) / 1e6))
