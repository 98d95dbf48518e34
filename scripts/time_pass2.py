#!/usr/bin/env python3
"""Where the in-process file path's time goes beyond its stage sums: bench.py's extra_file_path input, recalibrate_fastq under
KBBQ_TIMING=2's timeline plus wall-clock marks around pass 2's pieces.  usage (GPU box): KBBQ_TIMING=2 python scripts/time_pass2.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd')); sys.path.insert(0, ROOT)
import torch, bench
from kbbq import _device as dev, _trace, recalibrate, _egress
marks = []
orig_emit = _egress.emit_records
def emit(*a, **k):
    marks.append(('emit start', time.time()))
    r = orig_emit(*a, **k)
    marks.append(('emit end', time.time()))
    return r
_egress.emit_records = emit
orig_rel = dev.release_pinned
def rel(tag):
    t = time.time(); orig_rel(tag); marks.append(('release_pinned %s %.4f' % (tag, time.time() - t), time.time()))
dev.release_pinned = rel
for rep in range(3):
    del marks[:]
    t0 = time.time()
    out = bench.extra_file_path(torch, dev, n=8_000_000)
    print('rep %d wall %.3f  %.2f Gbases/s  stages %s' % (rep, out['wall_s'], out['value'] / 1e9, out['stages_s']))
    base = marks[0][1] if marks else t0
    for name, t in marks:
        print('   %-40s %+.4f' % (name, t - base))
