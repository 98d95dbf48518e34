#!/usr/bin/env python3
"""BASELINE config 5 as FILES: a pair of FASTQ files whose reads grow from 36 to 300 bases (non-decreasing, as the reference requires:
recalibrate.py:89-101) through the whole command -- does the file path keep its rate when every length band has its own pitch?
usage (GPU box): python scripts/time_mixed_file.py [reads]"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
import numpy as np
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
tmp = os.environ.get('TMPDIR', '/tmp')
fa, fb, fo = (os.path.join(tmp, 'kbbq_mixed_%d_%s.fq' % (os.getpid(), x)) for x in 'abo')
rng = np.random.default_rng(1)
lengths = np.arange(36, 301)
per = max(2, (n // len(lengths)) & ~1)
bases = 0
with open(fa, 'wb') as A, open(fb, 'wb') as B:
    first = 0
    for L in lengths:
        L = int(L)
        seq = np.frombuffer(b'ACGT', dtype=np.uint8)[rng.integers(0, 4, (per, L))]
        cor = np.where(rng.random((per, L)) < 0.01, np.frombuffer(b'ACGT', dtype=np.uint8)[rng.integers(0, 4, (per, L))], seq)
        qual = rng.integers(33, 75, (per, L)).astype(np.uint8)
        ids = first + np.arange(per)
        digits = ((ids >> 1)[:, None] // 10 ** np.arange(8, -1, -1)[None, :] % 10 + 48).astype(np.uint8)
        for fh, s in ((A, seq), (B, cor)):
            rec = np.empty((per, 14 + L + 3 + L + 1), dtype=np.uint8)
            rec[:, 0] = ord('@'); rec[:, 1] = ord('r'); rec[:, 2:11] = digits; rec[:, 11] = ord('/'); rec[:, 12] = 49 + (ids & 1); rec[:, 13] = 10
            rec[:, 14:14 + L] = s; rec[:, 14 + L] = 10; rec[:, 15 + L] = ord('+'); rec[:, 16 + L] = 10
            rec[:, 17 + L:17 + 2 * L] = qual; rec[:, 17 + 2 * L] = 10
            fh.write(rec.tobytes())
        first += per; bases += per * L
print('%d reads of 36-300 bases, %.2f G bases, %.2f GB per file' % (first, bases / 1e9, os.path.getsize(fa) / 1e9), flush=True)
env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, 'kbbq-py_amd'), KBBQ_TIMING='1')
for rep in range(3):
    if os.path.exists(fo):
        os.remove(fo)
    t0 = time.perf_counter()
    with open(fo, 'wb') as out:
        r = subprocess.run([sys.executable, '-m', 'kbbq.main', 'recalibrate', '-f', fa, fb], env=env, stdout=out, stderr=subprocess.PIPE)
    dt = time.perf_counter() - t0
    stages = [l for l in r.stderr.decode().splitlines() if l.startswith('kbbq stages')]
    print('rep %d: rc %d, %.3f s wall = %.2f Gbases/s; %s' % (rep, r.returncode, dt, bases / dt / 1e9, stages[-1] if stages else r.stderr.decode()[-300:]), flush=True)
for p in (fa, fb, fo):
    os.remove(p)
