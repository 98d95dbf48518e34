#!/usr/bin/env python3
"""kbbq recalibrate -f A B > out, as a user runs it, on a synthetic pair: wall time by stage (KBBQ_TIMING=1)."""
import argparse, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
ap = argparse.ArgumentParser(); ap.add_argument('--reads', type=int, default=4_000_000); ap.add_argument('--dir', default='/tmp')
ap.add_argument('--reps', type=int, default=2); ap.add_argument('--keep', action='store_true')
a = ap.parse_args()
import numpy as np, oracle as O
n = a.reads
fa, fb, fo = (os.path.join(a.dir, x) for x in ('e2e_a.fq', 'e2e_b.fq', 'e2e_out.fq'))
step = 1_000_000
with open(fa, 'wb') as A, open(fb, 'wb') as B:
    for first in range(0, n, step):
        m = min(step, n - first)
        seq, cseq, qual, meta = O.synth(first, m, n, 1)
        idx = np.arange(first, first + m)
        names = np.char.add(np.char.zfill((idx >> 1).astype(str), 9), np.where(idx & 1, '/2', '/1'))
        nm = np.frombuffer(''.join(names.tolist()).encode(), dtype=np.uint8).reshape(m, 11)
        for f, plane in ((A, seq), (B, cseq)):
            rec = np.empty((m, 318), dtype=np.uint8)
            rec[:, 0] = ord('@'); rec[:, 1] = ord('r'); rec[:, 2:13] = nm; rec[:, 13] = 10
            rec[:, 14:164] = plane[:, :150]; rec[:, 164] = 10; rec[:, 165] = ord('+'); rec[:, 166] = 10
            rec[:, 167:317] = qual[:, :150]; rec[:, 317] = 10
            rec.tofile(f)
env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, 'kbbq-py_amd'), KBBQ_TIMING='1')
for rep in range(a.reps):
    t0 = time.perf_counter()
    with open(fo, 'wb') as out:
        subprocess.run([sys.executable, '-m', 'kbbq.main', 'recalibrate', '-f', fa, fb], env=env, stdout=out, check=True)
    dt = time.perf_counter() - t0
    print('rep %d: %d reads, %.3f s wall incl. interpreter start = %.2f Gbases/s; output %d bytes'
          % (rep, n, dt, n * 150 / dt / 1e9, os.path.getsize(fo)), flush=True)
import hashlib
h = hashlib.sha256()
with open(fo, 'rb') as f:
    for blk in iter(lambda: f.read(1 << 24), b''):
        h.update(blk)
print('output sha256', h.hexdigest()[:16])
for p in (fa, fb, fo):
    if not a.keep:
        os.remove(p)
