#!/usr/bin/env python3
"""Timing of the native layout pass (dev.lay_out: sidecar statistics, read-group counting sort, k7_lay_out) on a resident
synthetic batch, with the destination planes allocated beforehand (second call) -- HIP events of the whole call."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd'))
ap = argparse.ArgumentParser(); ap.add_argument('--reads', type=int, default=50_000_000); ap.add_argument('--rgs', type=int, default=1)
a = ap.parse_args()
import torch
from kbbq import _device as dev
b = dev.ReadBatch.synthetic(0, a.reads, a.reads, seed=1, nrg=a.rgs)
for packed in (True, False):
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); laid = dev.lay_out(b, a.rgs, 150, packed=packed); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        src = 3 * a.reads * b.pitch; dst = laid.n * laid.pitch * (2 if laid.nib else 3)
        print('lay_out packed=%s rgs=%d: %.2f ms (%s): %.0f GB/s of source + destination bytes' % (packed, a.rgs, ms, laid.describe(), (src + dst) / ms / 1e6), flush=True)
        del laid
