#!/usr/bin/env python3
"""The aligned-read kernels of bench.py (extra_aligned) once more, for `rocprofv3 --kernel-trace --stats -- python3 scripts/prof_tally.py`:
per-kernel times of the one-pass BAM-sourced tally (kbbq_tally_aligned_dev: k4_read_records, k4v2_find_errors over the listed reads,
k1v3_aligned_ref) beside K4 + the fused kernel."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd')); sys.path.insert(0, ROOT)
import torch, bench
from kbbq import _device as dev
out = bench.extra_aligned(torch, dev, n=int(os.environ.get('N', 16_000_000)), G=200_000_000, reps=3)
print(json.dumps(out['whole_tally_ms']))
