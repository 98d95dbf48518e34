#!/usr/bin/env python3
"""Mixed-length reads (36..300 bases, ascending): one batch at the widest pitch vs length bands at their own pitch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd'))
import numpy as np, torch
from kbbq import _device as dev, fastx
n, R, S = 10_000_000, 1, 300
b = dev.ReadBatch.synthetic(0, n, n, seed=5, len_lo=36, len_hi=S, nrg=R)
lens = b.lengths_host()
bases = int(lens.sum())
def timeit(f, reps=3):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
t = dev.Tables(R, 2 * S)
dev.accumulate(b, t); lut, shape, _, _ = dev.solve(t)
out = torch.empty_like(b.qual)
k1 = timeit(lambda: dev.accumulate(b, t, check=False)); k2 = timeit(lambda: dev.apply(b, lut, shape, out=out, check=False))
print('one batch, pitch %d: K1 %.3f ms  K2 %.3f ms  (%d reads, %.2f Gbases)' % (b.pitch, k1 * 1e3, k2 * 1e3, n, bases / 1e9))
bands = []
for lo, hi, longest, shortest in fastx.length_bands(lens):
    pitch = fastx.pitch_for(longest)
    bb = dev.ReadBatch(hi - lo, pitch)
    bb.seq.copy_(b.seq[lo:hi, :pitch]); bb.cseq.copy_(b.cseq[lo:hi, :pitch]); bb.qual.copy_(b.qual[lo:hi, :pitch]); bb.meta.copy_(b.meta[lo:hi])
    bands.append((bb, longest, torch.empty_like(bb.qual), shortest))
def k1b(hint=True):
    for bb, longest, _, shortest in bands: dev.accumulate(bb, t, check=False, s_band=longest, s_min=shortest if hint else 0)
def k2b():
    for bb, _, o, _ in bands: dev.apply(bb, lut, shape, out=o, check=False)
k1 = timeit(k1b); k1n = timeit(lambda: k1b(False)); k2 = timeit(k2b)
print('%d bands (pitches %s): K1 %.3f ms (%.3f ms without the shortest-read promise)  K2 %.3f ms'
      % (len(bands), [x[0].pitch for x in bands], k1 * 1e3, k1n * 1e3, k2 * 1e3))
for bb, longest, _, shortest in bands:
    nb = int((bb.meta[:bb.n].cpu().numpy().view(np.uint32) & 0xFFFF).sum())
    a = timeit(lambda: dev.accumulate(bb, t, check=False, s_band=longest, s_min=shortest))
    c = timeit(lambda: dev.accumulate(bb, t, check=False, s_band=longest))
    print('  band %3d..%3d pitch %3d: %7.1f Gbases/s with the promise, %7.1f without' % (shortest, longest, bb.pitch, nb / a / 1e9, nb / c / 1e9))
dev.context().status()
