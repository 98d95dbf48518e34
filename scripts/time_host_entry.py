#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry points (kbbq_accumulate / kbbq_apply: a caller's NumPy planes in pageable memory,
rows staged slab by slab through page-locked buffers) beside the same rows uploaded whole (torch copies of pageable arrays) and
then tallied / applied by the device-plane entry points.  usage (GPU box): python scripts/time_host_entry.py [reads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd'))
import numpy as np
import torch
from kbbq import _device as dev, _native as N
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
b = dev.ReadBatch.synthetic(0, n, n, seed=1)
seq, cseq, qual, meta = (x.cpu().numpy() for x in (b.seq, b.cseq, b.qual, b.meta))
meta = meta.view(np.uint32)
pitch, R, S2 = seq.shape[1], 1, 300
bases = n * 150
t = dev.Tables(R, S2)
dev.accumulate(b, t)
lut, shape = dev.solve_lut(t)
want_out = dev.apply(b, lut, shape).cpu().numpy()
want_tabs = [x.copy() for x in t.to_host()]
del b
torch.cuda.empty_cache()
ctx, lib = dev.context(), N.load()
from kbbq.gatk import applybqsr
from kbbq import _solve
vec = _solve.vectors_from_tables(*want_tabs, 42)
dqs = applybqsr.get_delta_qs(*vec)
a = [np.ascontiguousarray(x, dtype=np.int64) for x in (vec[0],) + tuple(dqs)]
for rep in range(3):
    tabs = [np.zeros_like(x) for x in want_tabs]
    t0 = time.perf_counter()
    N.check(lib.kbbq_accumulate(ctx.handle, N.ptr(seq), N.ptr(cseq), N.ptr(qual), N.ptr(meta), n, pitch, R, S2, 6, *[N.ptr(x) for x in tabs]))
    t1 = time.perf_counter()
    out = np.empty_like(qual)
    t2 = time.perf_counter()
    N.check(lib.kbbq_apply(ctx.handle, N.ptr(seq), N.ptr(qual), N.ptr(meta), n, pitch, R, 43, S2, 17, 6, *[N.ptr(x) for x in a], N.ptr(out)))
    t3 = time.perf_counter()
    ok = all(np.array_equal(g, w) for g, w in zip(tabs, want_tabs)) and np.array_equal(out[:, :150], want_out[:, :150])
    print('slabs:  kbbq_accumulate %.3f s = %.1f Gbases/s (%.1f GB/s of planes in)   kbbq_apply %.3f s = %.1f Gbases/s (%.1f GB/s in, %.1f out)   same results %s'
          % (t1 - t0, bases / (t1 - t0) / 1e9, 3 * seq.nbytes / (t1 - t0) / 1e9, t3 - t2, bases / (t3 - t2) / 1e9, 2 * seq.nbytes / (t3 - t2) / 1e9,
             seq.nbytes / (t3 - t2) / 1e9, ok), flush=True)
    t0 = time.perf_counter()
    bb = dev.ReadBatch.from_host(seq, qual, meta.view(np.int32), cseq=cseq)
    tt = dev.Tables(R, S2)
    dev.accumulate(bb, tt)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    o = dev.apply(bb, lut, shape).cpu().numpy()
    t2 = time.perf_counter()
    print('whole:  upload + accumulate %.3f s = %.1f Gbases/s   apply + download %.3f s (planes already on the device)' % (t1 - t0, bases / (t1 - t0) / 1e9, t2 - t1), flush=True)
    del bb, o
