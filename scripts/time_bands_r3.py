#!/usr/bin/env python3
"""K1 / K2 per length band of BASELINE config 5 (a launch per band), with the bytes each band's rows hold: where the merged launch's time goes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd'))
import torch
from kbbq import _device as dev, fastx
lo, hi, n = 36, 300, 20_000_000
bands, at = [], lo
for c in fastx.BAND_CLASSES:
    if c >= at:
        bands.append((at, min(c, hi))); at = min(c, hi) + 1
    if at > hi:
        break
per = n // len(bands) & ~1
ctx = dev.context()
t = dev.Tables(1, 2 * hi)
tot1 = tot2 = 0.0
for k, (blo, bhi) in enumerate(bands):
    b = dev.ReadBatch.synthetic(k * per, per, per * len(bands), seed=1, len_lo=blo, len_hi=bhi)
    st = dev.meta_stats(b)
    rows = dev.lay_out(b, 1, st['longest'], packed=True, pairs=False, stats=st)
    bases = int(b.lengths_host().sum())
    out = torch.empty_like(rows.qual)
    dev.accumulate(rows, t, s_band=st['longest'], s_min=st['shortest'])
    lut, shape = dev.solve_lut(t)
    dev.apply(rows, lut, shape, out=out)
    ctx.kernel_ms(0, reset=True); ctx.kernel_ms(1, reset=True); ctx.timing(True)
    for _ in range(10):
        dev.accumulate(rows, t, check=False, s_band=st['longest'], s_min=st['shortest'])
        dev.apply(rows, lut, shape, out=out, check=False)
    ctx.timing(False)
    k1 = ctx.kernel_ms(0)[0] / 10; k2 = ctx.kernel_ms(1)[0] / 10
    nbytes = rows.n * rows.pitch
    print('%3d-%3d pitch %3d: K1 %.1f us = %.2f TB/s of its 2 B per padded byte (%.0f Gbases/s)   K2 %.1f us = %.2f TB/s of 2.5 B' %
          (blo, bhi, rows.pitch, k1 * 1e3, 2 * nbytes / k1 / 1e9, bases / k1 / 1e6, k2 * 1e3, 2.5 * nbytes / k2 / 1e9), flush=True)
    tot1 += k1; tot2 += k2
    del b, rows, out
print('sum K1 %.3f ms  K2 %.3f ms' % (tot1, tot2))
