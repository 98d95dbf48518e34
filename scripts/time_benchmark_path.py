#!/usr/bin/env python3
"""Kernel timing of the benchmark path (K4 find_errors, K5 count_q) on synthetic aligned reads
built directly as arrays (no SAM text): n reads x L bases against a random genome."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd'))
ap = argparse.ArgumentParser(); ap.add_argument('--reads', type=int, default=4_000_000); ap.add_argument('--len', type=int, default=150)
ap.add_argument('--genome', type=int, default=200_000_000)
a = ap.parse_args()
import numpy as np, torch
from kbbq import _device as dev, _native as N
n, L, G = a.reads, a.len, a.genome
pitch = (L + 15) // 16 * 16
g = torch.randint(0, 4, (G,), dtype=torch.uint8, device='cuda')
genome = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device='cuda')[g.long()]
mask = (torch.rand(G, device='cuda') < 0.01).to(torch.uint8)
start = torch.randint(0, G - 2 * L - 64, (n,), dtype=torch.int64, device='cuda')
idx = start[:, None] + torch.arange(pitch, device='cuda')[None, :]
seq = genome[idx]                                    # reads = reference windows ...
err_at = torch.rand((n, pitch), device='cuda') < 0.01
seq = torch.where(err_at, torch.tensor(65, dtype=torch.uint8, device='cuda'), seq)   # ... with 1 % substitutions
seq = torch.cat([seq, torch.zeros((1, pitch), dtype=torch.uint8, device='cuda')])     # slack row
lens = torch.full((n,), L, dtype=torch.int32, device='cuda')
# CIGAR: 80 % "LM", 20 % "50M2I(L-52)M" (ref window L-2)
ins = torch.rand(n, device='cuda') < 0.2
ref_len = torch.where(ins, L - 2, L).to(torch.int32)
cig_n = torch.where(ins, 3, 1).to(torch.int32)
cig_off = torch.cumsum(cig_n, 0).to(torch.int32) - cig_n
ncig = int(cig_n.sum())
cigar = torch.zeros(ncig, dtype=torch.int32, device='cuda')
o = cig_off.long()
cigar[o[~ins]] = (L << 4) | 0
cigar[o[ins]] = (50 << 4) | 0; cigar[o[ins] + 1] = (2 << 4) | 1; cigar[o[ins] + 2] = ((L - 52) << 4) | 0
flip = (torch.rand(n, device='cuda') < 0.5).to(torch.uint8)
err = torch.empty((n, pitch), dtype=torch.uint8, device='cuda'); skip = torch.empty_like(err)
qual = torch.randint(2, 42, (n, pitch), dtype=torch.uint8, device='cuda')
counts = torch.zeros(512, dtype=torch.int64, device='cuda')
ctx = dev.context(); lib = N.load()
def k4():
    N.check(lib.kbbq_find_errors_dev(ctx.handle, N.ptr(seq), N.ptr(lens), n, pitch, N.ptr(start), N.ptr(ref_len),
                                     N.ptr(cig_off), N.ptr(cig_n), N.ptr(cigar), N.ptr(genome), N.ptr(mask), N.ptr(flip),
                                     N.ptr(err), N.ptr(skip)))
def k5():
    N.check(lib.kbbq_count_q_dev(ctx.handle, N.ptr(qual), N.ptr(err), N.ptr(skip), N.ptr(lens), n, pitch, 0, N.ptr(counts)))
for f, name, bpb in ((k4, 'K4 find_errors', 5), (k5, 'K5 count_q', 3)):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print('%s: %.3f ms for %d reads x %d = %.1f Gbases/s, %.0f GB/s algorithmic (%d B/base)' % (
        name, dt * 1e3, n, L, n * L / dt / 1e9, bpb * n * L / dt / 1e9, bpb), flush=True)
ctx.status()
print('errors flagged %.4f, skipped %.4f' % (float(err[:, :L].float().mean()), float(skip[:, :L].float().mean())))
