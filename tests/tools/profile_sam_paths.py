#!/usr/bin/env python3
"""cProfile of the SAM-fed paths on a synthetic alignment file (test infrastructure: the generator is the oracle's): where do
benchmark_bam and bam_to_bqsr_covariates spend their time besides the reader and the kernels?"""
import argparse, cProfile, os, pstats, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'kbbq-py_amd')); sys.path.insert(0, os.path.join(ROOT, 'oracle'))
ap = argparse.ArgumentParser(); ap.add_argument('--pairs', type=int, default=100000); ap.add_argument('--len', type=int, default=150)
a = ap.parse_args()
import numpy as np, torch
import oracle_bqsr as OQ
from kbbq import aln, benchmark
from kbbq.gatk import bqsr
d = tempfile.mkdtemp()
paths = OQ.synth_bqsr_set(d, seed=1, npairs=a.pairs, S=a.len, contigs=(('chr1', 3_000_000), ('chr2', 1_000_000)))
var = benchmark.get_var_sites(paths['vcf'])
ref = benchmark.get_ref_dict(paths['fa'])
bqsr.bam_to_bqsr_covariates(aln.AlignmentFile(paths['sam']), paths['fa'], var)          # warm
benchmark.benchmark_bam(aln.AlignmentFile(paths['sam']), ref, var, use_oq=True)
for name, fn in (('bam_to_bqsr_covariates', lambda: bqsr.bam_to_bqsr_covariates(aln.AlignmentFile(paths['sam']), paths['fa'], var)),
                 ('benchmark_bam', lambda: benchmark.benchmark_bam(aln.AlignmentFile(paths['sam']), ref, var, use_oq=True))):
    pr = cProfile.Profile()
    t0 = time.perf_counter(); pr.enable(); fn(); torch.cuda.synchronize(); pr.disable(); dt = time.perf_counter() - t0
    print('==== %s: %.3f s for %d alignments' % (name, dt, 2 * a.pairs), flush=True)
    pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
