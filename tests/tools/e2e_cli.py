#!/usr/bin/env python3
"""kbbq recalibrate -f A B > out, as a user runs it, on a synthetic pair: wall time by stage (KBBQ_TIMING=1)."""
import argparse, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
ap = argparse.ArgumentParser(); ap.add_argument('--reads', type=int, default=4_000_000); ap.add_argument('--dir', default='/tmp')
ap.add_argument('--reps', type=int, default=2); ap.add_argument('--keep', action='store_true')
ap.add_argument('--modes', default='resident', help='comma-separated: resident, budget=SIZE (KBBQ_DEVICE_BUDGET), sequential (KBBQ_SEQUENTIAL=1), '
                                                     'pipes (-f <(cat A) <(cat B))')
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
import hashlib
def sha(path):
    h = hashlib.sha256()
    with open(path, 'rb') as f:
        for blk in iter(lambda: f.read(1 << 24), b''):
            h.update(blk)
    return h.hexdigest()[:16]
base_env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, 'kbbq-py_amd'), KBBQ_TIMING=os.environ.get('KBBQ_TIMING', '1'))     # 2: the stages' timeline since the process started
shas = {}
for mode in a.modes.split(','):
    env = dict(base_env)
    cmd = [sys.executable, '-m', 'kbbq.main', 'recalibrate', '-f', fa, fb]
    shell = False
    if mode.startswith('budget='):
        env['KBBQ_DEVICE_BUDGET'] = mode.split('=', 1)[1]
    elif mode == 'sequential':
        env['KBBQ_SEQUENTIAL'] = '1'
    elif mode == 'pipes':
        cmd, shell = '%s -m kbbq.main recalibrate -f <(cat %s) <(cat %s)' % (sys.executable, fa, fb), True
    for rep in range(a.reps):
        if os.path.exists(fo):
            os.remove(fo)                    # a fresh output file: truncating 2.5 GB of page cache is the kernel's 0.4 s, not the command's
        t0 = time.perf_counter()
        with open(fo, 'wb') as out:
            subprocess.run(cmd, env=env, stdout=out, check=True, shell=shell, executable='/bin/bash' if shell else None)
        dt = time.perf_counter() - t0
        print('%s rep %d: %d reads, %.3f s wall incl. interpreter start = %.2f Gbases/s; output %d bytes'
              % (mode, rep, n, dt, n * 150 / dt / 1e9, os.path.getsize(fo)), flush=True)
    shas[mode] = sha(fo)
    print('%s output sha256 %s' % (mode, shas[mode]), flush=True)
print('all modes wrote the same bytes:', len(set(shas.values())) == 1)
for p in (fa, fb, fo):
    if not a.keep:
        os.remove(p)
