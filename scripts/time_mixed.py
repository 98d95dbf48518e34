#!/usr/bin/env python3
"""bench.py's config5_mixed_lengths entry on its own (for rocprofv3 --kernel-trace: one K1 / K2 launch per length band)."""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd'))
ap = argparse.ArgumentParser(); ap.add_argument('--reads', type=int, default=20_000_000); ap.add_argument('--steps', type=int, default=5)
a = ap.parse_args()
import torch
import bench
from kbbq import _device as dev
dev.warm_up()
r = bench.extra_mixed_lengths(torch, dev, a.reads, a.steps, 1)
print(json.dumps({k: v for k, v in r.items() if k != 'layout'}))
