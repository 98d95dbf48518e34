#!/usr/bin/env python3
"""Where the in-process file path spends its time, and what bounds its host stages: recalibrate_fastq on 8 M synthetic reads
(bench.py's extra.file_path) with 4 / 8 / 16 host threads, the parked workers on and off, resident and streamed within a
device budget.  A stage that takes the same time with half the threads is not bound by the cores (page faults, memory, PCIe)
and gains from running beside another stage.  usage (GPU box): python scripts/time_file_stages.py [reads]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd'))
import numpy as np
NATIVE = '--native' in sys.argv          # device memory from the library's own C ABI (kbbq/_hipmem.py), as the command line without torch
TIMELINE = '--timeline' in sys.argv       # the default configuration only, every stage's start and end since recalibrate_fastq was called
if TIMELINE:
    sys.argv.remove('--timeline')
if NATIVE:
    sys.argv.remove('--native')
    from kbbq import _device as dev
    dev.use_native_memory()
else:
    import torch
from kbbq import _device as dev, recalibrate, _trace
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
tmp = os.environ.get('TMPDIR', '/tmp')
fa, fb, fo = (os.path.join(tmp, 'kbbq_stages_%d_%s.fq' % (os.getpid(), x)) for x in 'abo')
batch = dev.ReadBatch.synthetic(0, n, n, seed=1)
seq, cseq, qual = (getattr(batch, p).cpu().numpy()[:n, :150] for p in ('seq', 'cseq', 'qual'))
del batch
for path, plane in ((fa, seq), (fb, cseq)):
    rec = np.empty((n, 318), dtype=np.uint8)
    ids = np.arange(n)
    digits = ((ids >> 1)[:, None] // 10 ** np.arange(8, -1, -1)[None, :] % 10 + 48).astype(np.uint8)
    rec[:, 0] = ord('@'); rec[:, 1] = ord('r'); rec[:, 2:11] = digits; rec[:, 11] = ord('/')
    rec[:, 12] = 49 + (ids & 1); rec[:, 13] = 10
    rec[:, 14:164] = plane; rec[:, 164] = 10; rec[:, 165] = ord('+'); rec[:, 166] = 10
    rec[:, 167:317] = qual; rec[:, 317] = 10
    rec.tofile(path)
    del rec
dev.warm_up()


def run(label, env, reps=3):
    saved_env = {k: os.environ.get(k) for k in env}
    os.environ.update({k: v for k, v in env.items() if v is not None})
    for k, v in env.items():
        if v is None:
            os.environ.pop(k, None)
    best = None
    try:
        for _ in range(reps):
            if os.path.exists(fo):
                os.remove(fo)                # a fresh output file: truncating 2.5 GB of page cache is the kernel's 0.4 s, not the path's
            _trace.collect(True)
            t0 = time.perf_counter()
            recalibrate.recalibrate_fastq([fa, fb], output=fo)
            wall = time.perf_counter() - t0
            st = _trace.collect(False)
            if best is None or wall < best[0]:
                best = (wall, st)
    finally:
        for k, v in saved_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    wall, st = best
    if NATIVE:
        label += ' [native memory]'
    print('%-34s wall %.3f s = %.2f Gbases/s   %s' % (label, wall, n * 150 / wall / 1e9, '  '.join('%s %.3f' % kv for kv in st.items())), flush=True)
    return wall


try:
    if TIMELINE:
        _trace.TIMELINE = True
        for rep in range(3):
            if os.path.exists(fo):
                os.remove(fo)
            _trace.collect(True); del _trace._events[:]
            w0 = time.time(); t0 = time.perf_counter()
            recalibrate.recalibrate_fastq([fa, fb], output=fo)
            wall = time.perf_counter() - t0
            _trace.collect(False)
            print('repetition %d: wall %.3f s' % (rep, wall))
            for a, b, name in sorted(_trace._events):
                if b - a >= 0.001:
                    print('  %7.3f .. %7.3f  %s' % (a - w0, b - w0, name))
            sys.stdout.flush()
        raise SystemExit(0)
    if NATIVE:
        for rep in range(3):             # every repetition on its own line: is `apply` (40 ms in the command line) a first-call cost?
            run('repetition %d' % rep, {}, reps=1)
        raise SystemExit(0)
    base = run('16 threads (default)', {})
    for t in ('8', '4'):
        run('%s threads' % t, {'KBBQ_HOST_THREADS': t})
    run('fresh threads per region', {'KBBQ_THREAD_POOL': '0'})       # read when the pool is first made: only differs in a fresh process
    s = run('device budget 256M', {'KBBQ_DEVICE_BUDGET': '256M'})
    print('streamed / resident wall: %.3f' % (s / base), recalibrate.LAST_RUN.get('streamed'))
    run('device budget 1G', {'KBBQ_DEVICE_BUDGET': '1G'})
    run('sequential, 256M segments', {'KBBQ_SEQUENTIAL': '1', 'KBBQ_DEVICE_BUDGET': '256M'})
finally:
    for p in (fa, fb, fo):
        if os.path.exists(p):
            os.remove(p)
