#!/usr/bin/env python3
"""Kernel timing of the benchmark path (K4 find_errors, K5 count_q) and of the BAM-sourced tally
(K4 -> K6 canonical reads -> K1) on synthetic aligned reads built directly as arrays (no SAM
text): n reads x L bases against a random genome."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd'))
ap = argparse.ArgumentParser(); ap.add_argument('--reads', type=int, default=4_000_000); ap.add_argument('--len', type=int, default=150)
ap.add_argument('--genome', type=int, default=200_000_000)
ap.add_argument('--sorted', action='store_true', help='reads in coordinate order (a sorted BAM) instead of random order')
ap.add_argument('--separate', action='store_true', help='reference and site mask as two arrays (default: mask in bit 7 of the reference bytes)')
ap.add_argument('--ins', type=float, default=0.2, help='fraction of reads with a 2-base insertion (3 CIGAR operations)')
ap.add_argument('--flip', type=float, default=0.5, help='fraction of reverse-strand reads')
a = ap.parse_args()
import numpy as np, torch
from kbbq import _device as dev, _native as N
n, L, G = a.reads, a.len, a.genome
pitch = (L + 15) // 16 * 16
g = torch.randint(0, 4, (G,), dtype=torch.uint8, device='cuda')
genome = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device='cuda')[g.long()]
mask = (torch.rand(G, device='cuda') < 0.01).to(torch.uint8)
start = torch.randint(0, G - L, (n,), dtype=torch.int64, device='cuda')
start[:64] = G - L                                   # some windows end exactly at the genome's last byte
if a.sorted:
    start = torch.sort(start).values
idx = (start[:, None] + torch.arange(pitch, device="cuda")[None, :]).clamp_(max=G - 1)
seq = genome[idx]                                    # reads = reference windows ...
err_at = torch.rand((n, pitch), device='cuda') < 0.01
seq = torch.where(err_at, torch.tensor(65, dtype=torch.uint8, device='cuda'), seq)   # ... with 1 % substitutions
lens = torch.full((n,), L, dtype=torch.int32, device='cuda')
# CIGAR: 80 % "LM", 20 % "50M2I(L-52)M" (ref window L-2)
ins = torch.rand(n, device='cuda') < a.ins
ref_len = torch.where(ins, L - 2, L).to(torch.int32)
cig_n = torch.where(ins, 3, 1).to(torch.int32)
cig_off = torch.cumsum(cig_n, 0).to(torch.int32) - cig_n
ncig = int(cig_n.sum())
cigar = torch.zeros(ncig, dtype=torch.int32, device='cuda')
o = cig_off.long()
cigar[o[~ins]] = (L << 4) | 0
cigar[o[ins]] = (50 << 4) | 0; cigar[o[ins] + 1] = (2 << 4) | 1; cigar[o[ins] + 2] = ((L - 52) << 4) | 0
flip = (torch.rand(n, device='cuda') < a.flip).to(torch.uint8)
err = torch.zeros((n, pitch), dtype=torch.uint8, device='cuda'); skip = torch.zeros_like(err)
qual = torch.randint(2, 42, (n, pitch), dtype=torch.uint8, device='cuda')
counts = torch.zeros(512, dtype=torch.int64, device='cuda')
ctx = dev.context(); lib = N.load()
fused = genome | (mask << 7)
gptr, mptr = (N.ptr(genome), N.ptr(mask)) if a.separate else (N.ptr(fused), None)
def k4():
    N.check(lib.kbbq_find_errors_dev(ctx.handle, N.ptr(seq), N.ptr(lens), n, pitch, N.ptr(start), N.ptr(ref_len),
                                     N.ptr(cig_off), N.ptr(cig_n), N.ptr(cigar), gptr, mptr, G, N.ptr(flip),
                                     N.ptr(err), N.ptr(skip)))
def k5():
    N.check(lib.kbbq_count_q_dev(ctx.handle, N.ptr(qual), N.ptr(err), N.ptr(skip), N.ptr(lens), n, pitch, 0, N.ptr(counts)))
# BAM-sourced tally: K4 flags (not flipped) -> K6 -> K1
oq = qual + 33
noflip = torch.zeros_like(flip)
clip = torch.full((n,), L << 16, dtype=torch.int32, device='cuda')
soft = torch.rand(n, device='cuda') < 0.2                      # 20 % of the reads carry 5-base soft clips at both ends
clip[soft] = 5 | ((L - 5) << 16)
trim = torch.zeros(n, dtype=torch.int32, device='cuda')
trimmed = torch.rand(n, device='cuda') < 0.05
trim[trimmed] = (L - 20) | (L << 16)
flags = (torch.randint(0, 4, (n,), device='cuda', dtype=torch.int32))      # strand and mate bits, read group 0
batch = dev.ReadBatch(n, pitch, with_corrected=True)
tables = dev.Tables(1, 2 * L)
def k4n():
    N.check(lib.kbbq_find_errors_dev(ctx.handle, N.ptr(seq), N.ptr(lens), n, pitch, N.ptr(start), N.ptr(ref_len),
                                     N.ptr(cig_off), N.ptr(cig_n), N.ptr(cigar), gptr, mptr, G, N.ptr(noflip),
                                     N.ptr(err), N.ptr(skip)))
def k6():
    N.check(lib.kbbq_canonical_reads_dev(ctx.handle, N.ptr(seq), N.ptr(oq), N.ptr(err), N.ptr(skip), N.ptr(lens),
                                         N.ptr(clip), N.ptr(trim), N.ptr(flags), n, pitch, L, 6, 6,
                                         N.ptr(batch.seq), N.ptr(batch.cseq), N.ptr(batch.qual), N.ptr(batch.meta)))
def k1():
    dev.accumulate(batch, tables, 6, check=False, dinuc_minscore=6)
def k4f():
    N.check(lib.kbbq_find_errors_dev(ctx.handle, N.ptr(seq), N.ptr(lens), n, pitch, N.ptr(start), N.ptr(ref_len),
                                     N.ptr(cig_off), N.ptr(cig_n), N.ptr(cigar), gptr, mptr, G, N.ptr(flip),
                                     N.ptr(err), None))
def k5f():
    N.check(lib.kbbq_count_q_dev(ctx.handle, N.ptr(qual), N.ptr(err), None, N.ptr(lens), n, pitch, 0, N.ptr(counts)))
def k4nf():
    N.check(lib.kbbq_find_errors_dev(ctx.handle, N.ptr(seq), N.ptr(lens), n, pitch, N.ptr(start), N.ptr(ref_len),
                                     N.ptr(cig_off), N.ptr(cig_n), N.ptr(cigar), gptr, mptr, G, N.ptr(noflip),
                                     N.ptr(err), None))
def k6f():
    N.check(lib.kbbq_canonical_reads_dev(ctx.handle, N.ptr(seq), N.ptr(oq), N.ptr(err), None, N.ptr(lens),
                                         N.ptr(clip), N.ptr(trim), N.ptr(flags), n, pitch, L, 6, 6,
                                         N.ptr(batch.seq), N.ptr(batch.cseq), N.ptr(batch.qual), N.ptr(batch.meta)))
packed = dev.ReadBatch(n, pitch, with_corrected=True, nib=True)
def k6p():
    N.check(lib.kbbq_canonical_reads_rows_dev(ctx.handle, N.ptr(seq), N.ptr(oq), N.ptr(err), None, N.ptr(lens),
                                              N.ptr(clip), N.ptr(trim), N.ptr(flags), n, pitch, L, 6, 6, N.ROWS_NIBBLES,
                                              N.ptr(packed.seq), N.ptr(packed.cseq), N.ptr(packed.qual), N.ptr(packed.meta)))
def k1p():
    dev.accumulate(packed, tables, 6, check=False, dinuc_minscore=6)
for f, name, bpb in ((k4, 'K4 find_errors', 5), (k5, 'K5 count_q', 3), (k4n, 'K4 (no flip)', 5),
                     (k6, 'K6 canonical_reads', 7), (k1, 'K1 on canonical reads', 3),
                     (k4f, 'K4 find_errors -> flags plane', 4), (k5f, 'K5 count_q <- flags plane', 2),
                     (k4nf, 'K4 (no flip) -> flags plane', 4), (k6f, 'K6 canonical_reads <- flags plane', 6),
                     (k6p, 'K6 <- flags plane -> 4-bit planes', 5), (k1p, 'K1 on 4-bit canonical reads', 2)):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print('%s: %.3f ms for %d reads x %d = %.1f Gbases/s, %.0f GB/s algorithmic (%d B/base)' % (
        name, dt * 1e3, n, L, n * L / dt / 1e9, bpb * n * L / dt / 1e9, bpb), flush=True)
ctx.status()
print('errors flagged %.4f, skipped %.4f' % (float(err[:, :L].float().mean()), float(skip[:, :L].float().mean())))
