#!/usr/bin/env python3
"""End-to-end file timing of kbbq.recalibrate.recalibrate_fastq (FASTQ text in -> FASTQ text out),
stage by stage.  Synthetic pair written with the oracle's generator (test infrastructure)."""
import argparse, contextlib, io, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
ap = argparse.ArgumentParser(); ap.add_argument('--reads', type=int, default=2_000_000); ap.add_argument('--dir', default='/tmp')
a = ap.parse_args()
import numpy as np, torch, oracle as O
from kbbq import fastx, recalibrate, _device as dev
n = a.reads
seq, cseq, qual, meta = O.synth(0, n, n, 1)
def write(path, plane):
    rec = np.empty((n, 1 + 12 + 1 + 150 + 3 + 150 + 1), dtype=np.uint8)     # fixed-width names r%09d/1
    names = np.char.add(np.char.zfill((np.arange(n) >> 1).astype(str), 9), np.where(np.arange(n) & 1, '/2', '/1'))
    nm = np.frombuffer(''.join(names.tolist()).encode(), dtype=np.uint8).reshape(n, 11)
    rec[:, 0] = ord('@'); rec[:, 1] = ord('r'); rec[:, 2:13] = nm; rec[:, 13] = 10
    rec[:, 14:164] = plane[:, :150]; rec[:, 164] = 10; rec[:, 165] = ord('+'); rec[:, 166] = 10
    rec[:, 167:317] = qual[:, :150]; rec[:, 317] = 10
    rec.tofile(path)
fa, fb = os.path.join(a.dir, 'e2e_a.fq'), os.path.join(a.dir, 'e2e_b.fq')
write(fa, seq); write(fb, cseq)
bases = n * 150
dev.context()
for rep in range(2):
    ta = time.perf_counter(); A_, B_ = fastx.NativeFastq(fa), fastx.NativeFastq(fb); tb = time.perf_counter(); sc = A_.scan(B_, False); tc = time.perf_counter()
    print('  open+index %.3fs  scan %.3fs' % (tb - ta, tc - tb)); del A_, B_
    t0 = time.perf_counter(); packed = fastx.pack_pair(fa, fb, False); t1 = time.perf_counter()
    b = dev.ReadBatch.from_host(packed['seq'], packed['qual'], packed['meta'], cseq=packed['cseq']); torch.cuda.synchronize(); t2 = time.perf_counter()
    t = dev.Tables(1, 300); dev.accumulate(b, t); lut, shape, _, _ = dev.solve(t); out = dev.apply(b, lut, shape); torch.cuda.synchronize(); t3 = time.perf_counter()
    newq = out[:n].cpu().numpy(); t4 = time.perf_counter()
    txt = packed['text'].format_array(0, n, newq); t5 = time.perf_counter()
    print('rep %d: pack %.3fs (%.2f Gbases/s)  H2D %.3fs  kernels+solve %.3fs  D2H %.3fs  format %.3fs (%.1f MB)  total %.3fs = %.2f Gbases/s'
          % (rep, t1 - t0, bases / (t1 - t0) / 1e9, t2 - t1, t3 - t2, t4 - t3, t5 - t4, len(txt) / 1e6, t5 - t0, bases / (t5 - t0) / 1e9), flush=True)
buf = io.StringIO()
t0 = time.perf_counter()
with contextlib.redirect_stdout(buf):
    recalibrate.recalibrate_fastq([fa, fb])
t1 = time.perf_counter()
print('recalibrate_fastq() end to end incl. print into StringIO: %.3fs = %.2f Gbases/s; output %d chars, sha %s'
      % (t1 - t0, bases / (t1 - t0) / 1e9, len(buf.getvalue()), O.sha256(buf.getvalue())[:12]))
# through a binary stdout (a file): the writer's bytes go out without a decode
outp = os.path.join(a.dir, 'e2e_out.fq')
sys.stdout.flush()
saved = os.dup(1); fd = os.open(outp, os.O_WRONLY | os.O_CREAT | os.O_TRUNC); os.dup2(fd, 1)
t0 = time.perf_counter()
recalibrate.recalibrate_fastq([fa, fb])
sys.stdout.flush()
t1 = time.perf_counter()
os.dup2(saved, 1); os.close(fd); os.close(saved)
print('recalibrate_fastq() end to end into a file: %.3fs = %.2f Gbases/s; %d bytes, sha %s'
      % (t1 - t0, bases / (t1 - t0) / 1e9, os.path.getsize(outp), O.sha256(open(outp, 'rb').read())[:12]))
os.remove(outp)
os.remove(fa); os.remove(fb)
