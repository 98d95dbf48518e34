#!/usr/bin/env python3
"""gzip-compressed FASTQ through the readers: one member inflated on all threads (csrc/parallel_gunzip.cpp) against libdeflate
and zlib on one thread -- the mapped reader (whole file), the sequential reader (segment by segment), and the whole command on a
.fq.gz pair.  usage: python scripts/time_gunzip.py [reads]   (host only up to the last step, which needs the GPU)"""
import os, subprocess, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd'))
import numpy as np
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
tmp = os.environ.get('TMPDIR', '/tmp')
rng = np.random.default_rng(1)
paths = []
for tag, err in (('a', 0.01), ('b', 0.0)):
    rec = np.empty((n, 318), dtype=np.uint8)
    if tag == 'a':
        seq = np.frombuffer(b'ACGT', dtype=np.uint8)[rng.integers(0, 4, (n, 150))]
        qual = np.frombuffer(b'FFFFFFFFFF:::,,#', dtype=np.uint8)[rng.integers(0, 16, (n, 150))]     # instrument-binned qualities
    else:
        flip = rng.random((n, 150)) < 0.01
        seq = np.where(flip, np.frombuffer(b'ACGT', dtype=np.uint8)[rng.integers(0, 4, (n, 150))], seq)
    ids = np.arange(n)
    digits = ((ids >> 1)[:, None] // 10 ** np.arange(8, -1, -1)[None, :] % 10 + 48).astype(np.uint8)
    rec[:, 0] = ord('@'); rec[:, 1] = ord('r'); rec[:, 2:11] = digits; rec[:, 11] = ord('/'); rec[:, 12] = 49 + (ids & 1); rec[:, 13] = 10
    rec[:, 14:164] = seq; rec[:, 164] = 10; rec[:, 165] = ord('+'); rec[:, 166] = 10; rec[:, 167:317] = qual; rec[:, 317] = 10
    p = os.path.join(tmp, 'kbbq_gz_%d_%s.fq' % (os.getpid(), tag))
    rec.tofile(p); del rec
    t0 = time.perf_counter()
    subprocess.check_call(['gzip', '-6', '-k', '-f', p])
    print('%s: %.0f MB of text -> %.0f MB of gzip -6 in %.0f s' % (tag, os.path.getsize(p) / 1e6, os.path.getsize(p + '.gz') / 1e6, time.perf_counter() - t0), flush=True)
    paths.append(p)
text_bytes = os.path.getsize(paths[0])
code = ('import sys, time; sys.path.insert(0, %r)\n'
        'from kbbq import fastx\n'
        'p, mode = sys.argv[1], sys.argv[2]\n'
        'best = 9e9\n'
        'for rep in range(3):\n'
        '    t0 = time.perf_counter()\n'
        '    if mode == "mapped":\n'
        '        r = fastx.NativeFastq(p); nrec = r.n; r.close()\n'
        '    else:\n'
        '        s = fastx.FastqStream(p); nrec = 0\n'
        '        while True:\n'
        '            seg, end = s.next(256 << 20)\n'
        '            if seg is None: break\n'
        '            nrec += seg.n; seg.close()\n'
        '        s.close()\n'
        '    best = min(best, time.perf_counter() - t0)\n'
        'print("%%d records, best of three %%.3f s" %% (nrec, best))\n' % os.path.join(ROOT, 'kbbq-py_amd'))
variants = (('all threads (parallel_gunzip)', {}), ('all threads, no huge pages asked', {'KBBQ_HUGE_PAGES': '0'}), ('libdeflate, one thread', {'KBBQ_PGZ_MIN_BYTES': str(1 << 60)}),
            ('zlib, one thread', {'KBBQ_PGZ_MIN_BYTES': str(1 << 60), 'KBBQ_LIBDEFLATE': '0'}))
for mode in ('mapped', 'stream'):
    for label, env in variants:
        if mode == 'stream' and label.startswith('libdeflate'):
            continue                       # (the sequential reader's one-thread path is zlib)
        r = subprocess.run([sys.executable, '-c', code, paths[0] + '.gz', mode], env=dict(os.environ, **env), capture_output=True, timeout=1200)
        out = r.stdout.decode().strip()
        secs = float(out.split()[-2]) if r.returncode == 0 and out else float('nan')
        print('%-7s %-32s %s = %.2f GB/s of text' % (mode, label, out or r.stderr.decode()[-300:], text_bytes / secs / 1e9), flush=True)
        if os.environ.get('KBBQ_PGZ_TRACE'):                     # the decoder's own account of its windows (the last repetition's)
            lines = [l for l in r.stderr.decode().splitlines() if l.startswith('[pgz]')]
            for l in lines[-6:]:
                print('        ' + l, flush=True)
if '--cli' in sys.argv:
    out = os.path.join(tmp, 'kbbq_gz_%d_out.fq' % os.getpid())
    shas = []
    for label, fa, fb, env in (('plain text', paths[0], paths[1], {}), ('.fq.gz, all threads', paths[0] + '.gz', paths[1] + '.gz', {}),
                               ('.fq.gz, zlib on one thread per file', paths[0] + '.gz', paths[1] + '.gz', {'KBBQ_PGZ_MIN_BYTES': str(1 << 60), 'KBBQ_LIBDEFLATE': '0'}),
                               ('.fq.gz read sequentially, all threads', paths[0] + '.gz', paths[1] + '.gz', {'KBBQ_SEQUENTIAL': '1'}),
                               ('.fq.gz read sequentially, zlib', paths[0] + '.gz', paths[1] + '.gz', {'KBBQ_SEQUENTIAL': '1', 'KBBQ_PGZ_MIN_BYTES': str(1 << 60)})):
        for rep in range(2):
            if os.path.exists(out):
                os.remove(out)
            t0 = time.perf_counter()
            with open(out, 'wb') as fh:
                subprocess.run([sys.executable, '-m', 'kbbq.main', 'recalibrate', '-f', fa, fb], stdout=fh, check=True,
                               env=dict(os.environ, PYTHONPATH=os.path.join(ROOT, 'kbbq-py_amd'), **env))
            dt = time.perf_counter() - t0
        import hashlib
        h = hashlib.sha256()
        with open(out, 'rb') as fh:
            for blk in iter(lambda: fh.read(1 << 24), b''):
                h.update(blk)
        shas.append(h.hexdigest()[:16])
        print('kbbq recalibrate -f  %-44s %.2f s (second run) = %.2f Gbases/s   sha %s' % (label, dt, n * 150 / dt / 1e9, shas[-1]), flush=True)
    print('same bytes in every form:', len(set(shas)) == 1)
    os.remove(out)
for p in paths:
    os.remove(p); os.remove(p + '.gz')
