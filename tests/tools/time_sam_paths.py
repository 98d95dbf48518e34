#!/usr/bin/env python3
"""End-to-end timing of the SAM-fed paths (truth-set benchmark, BAM-sourced tally) on a synthetic alignment file:
native SAM reader -> arrays -> K4 (-> K5 | K6 -> K1).  The generator is test infrastructure (oracle/)."""
import argparse, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
ap = argparse.ArgumentParser(); ap.add_argument('--pairs', type=int, default=50000); ap.add_argument('--len', type=int, default=150)
ap.add_argument('--bam', action='store_true', help='also: the same alignments as BAM (written by tests/bamwriter.py) -> arrays')
a = ap.parse_args()
import numpy as np, torch
import oracle_bqsr as OQ
from kbbq import aln, benchmark
from kbbq.gatk import bqsr
d = tempfile.mkdtemp()
t0 = time.perf_counter()
paths = OQ.synth_bqsr_set(d, seed=1, npairs=a.pairs, S=a.len, contigs=(('chr1', 3_000_000), ('chr2', 1_000_000)))
print('generated %d alignments in %.1f s (%.1f MB of SAM)' % (2 * a.pairs, time.perf_counter() - t0, os.path.getsize(paths['sam']) / 1e6), flush=True)
bases = 2 * a.pairs * a.len
var = benchmark.get_var_sites(paths['vcf'])
ref = benchmark.get_ref_dict(paths['fa'])
for rep in range(2):
    t0 = time.perf_counter(); bam = aln.AlignmentFile(paths['sam']); t1 = time.perf_counter()
    aq, nb = benchmark.benchmark_bam(bam, ref, var, use_oq=True); torch.cuda.synchronize(); t2 = time.perf_counter()
    vec = bqsr.bam_to_bqsr_covariates(aln.AlignmentFile(paths['sam']), paths['fa'], var); torch.cuda.synchronize(); t3 = time.perf_counter()
    print('rep %d: SAM -> arrays %.3f s (%.0f Mbases/s); benchmark_bam %.3f s; open + bam_to_bqsr_covariates %.3f s (%.0f Mbases/s); counted %d bases'
          % (rep, t1 - t0, bases / (t1 - t0) / 1e6, t2 - t1, t3 - t2, bases / (t3 - t2) / 1e6, int(vec[2].sum())), flush=True)
if a.bam:
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import bamwriter
    t0 = time.perf_counter(); bamwriter.write_bam(os.path.join(d, 'x.bam'), open(paths['sam']).read())
    print('BAM written in %.1f s (%.1f MB)' % (time.perf_counter() - t0, os.path.getsize(os.path.join(d, 'x.bam')) / 1e6), flush=True)
    for rep in range(3):
        t0 = time.perf_counter(); b = aln.AlignmentFile(os.path.join(d, 'x.bam')).batch(); t1 = time.perf_counter()
        print('rep %d: BAM -> arrays %.3f s (%.0f Mbases/s), %d alignments' % (rep, t1 - t0, bases / (t1 - t0) / 1e6, b.n), flush=True)
t0 = time.perf_counter(); objs = list(aln.AlignmentFile(paths['sam'])); t1 = time.perf_counter()
print('for comparison, one Python object per alignment (the pysam-style iteration): %.2f s = %.1f Mbases/s' % (t1 - t0, bases / (t1 - t0) / 1e6))
