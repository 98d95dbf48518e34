#!/usr/bin/env python3
"""BASELINE config 5 (36-300 bp, merged launches) with other length-band classes than kbbq.fastx.BAND_CLASSES:
what do narrower bands (less padding per row) buy?  usage: python scripts/time_mixed_bands.py [--reads N] 48,64,96,... [more lists]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd'))
ap = argparse.ArgumentParser()
ap.add_argument('--reads', type=int, default=20_000_000)
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('classes', nargs='*')
args = ap.parse_args()
import torch
import bench
from kbbq import _device as dev, fastx
dev.warm_up(0)
shipped = fastx.BAND_CLASSES
for spec in ['shipped'] + args.classes + ['shipped']:
    fastx.BAND_CLASSES = shipped if spec == 'shipped' else tuple(int(x) for x in spec.split(',')) + tuple(c for c in shipped if c > 320)
    r = bench.extra_mixed_lengths(torch, dev, args.reads, args.steps, 2)
    print('%-60s bands %2d  padded/bases %.3f  step %.3f ms = %.3f Tbases/s   K1 %.3f ms  K2 %.3f ms   verified %s' % (
        spec[:60], r['workload'].count('') and len(r['layout'].split('; ')), r['padded_row_bytes_per_plane'] / r['bases_per_step'], r['ms_per_step'],
        r['value'] / 1e12, r['k1_accumulate_all_bands']['avg_ms'], r['k2_apply_all_bands']['avg_ms'], r['verified']), flush=True)
    torch.cuda.empty_cache()
