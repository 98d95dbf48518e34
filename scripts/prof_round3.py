#!/usr/bin/env python3
"""Profiling driver for round 3's new kernels: the merged length-band launches (k1v3_bands, k2t_bands) on BASELINE config 5's bands and the fused
aligned-read tally (k1v3_aligned) beside K4 / K6 / K1.  Run under rocprofv3 (kernel trace or one --pmc set per run)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd')); sys.path.insert(0, ROOT)
import torch
import bench
from kbbq import _device as dev
r = bench.extra_mixed_lengths(torch, dev, 20_000_000, 2, 1)
print('config 5: verified', r['verified'], '%.0f Gbases/s' % (r['value'] / 1e9))
r = bench.extra_aligned(torch, dev, n=16_000_000, reps=2)
print('aligned: fused verified', r['k61_fused_tally']['verified'], r['whole_tally_ms'])
