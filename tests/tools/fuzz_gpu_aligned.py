#!/usr/bin/env python3
"""Randomised differential campaign of the aligned-read paths against the CPU oracles, for --seconds of random synthetic
sets: truth-set benchmark (K4 flags, K5 counts; SAM text and a BAM image of it) and the BAM-sourced tally (K4 -> K6 ->
K1).  Exit code 1 on any mismatch.  Test infrastructure (uses oracle/ and tests/bamwriter.py)."""
import argparse, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for d in ('kbbq-py_amd', 'oracle', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, d))
ap = argparse.ArgumentParser(); ap.add_argument('--seconds', type=float, default=300); ap.add_argument('--seed', type=int, default=1)
a = ap.parse_args()
import numpy as np
import _shim, bamwriter
import oracle_benchmark as OB, oracle_bqsr as OQ
from kbbq import aln, benchmark as bm
from kbbq.gatk import bqsr
rng = np.random.default_rng(a.seed)
t_end = time.time() + a.seconds
cases = bad = 0
def check(tag, cond, info):
    global bad
    if not cond:
        bad += 1
        print('MISMATCH %s %s' % (tag, info), flush=True)
def contigs(shortest):
    return tuple(('c%d' % i, int(rng.integers(shortest, shortest + 4000))) for i in range(int(rng.integers(1, 4))))
while time.time() < t_end:
    cases += 1
    tmp = tempfile.mkdtemp()
    try:
        if cases % 2:
            lo = int(rng.integers(5, 140)); info = dict(kind='bench', seed=cases, npairs=int(rng.integers(20, 500)), readlen=(lo, lo + int(rng.integers(1, 60))), contigs=contigs(3000))
            p = OB.synth_truthset(tmp, **{k: v for k, v in info.items() if k != 'kind'})
            ref, var = bm.get_ref_dict(p['fa']), bm.get_var_sites(p['vcf'])
            with open(p['bed']) as fh:
                full = bm.get_full_skips(ref, var, fh)
            oref = OB.get_ref_dict(p['fa'])
            want = OB.get_error_dict(list(_shim.AlignmentFile(p['sam'])), oref, full)
            bam = bamwriter.write_bam(os.path.join(tmp, 't.bam'), open(p['sam']).read())
            for src in (p['sam'], bam):
                got = bm.get_error_dict(aln.AlignmentFile(src), ref, full)
                check('error dict keys', list(got) == list(want), info)
                check('error dict', all(np.array_equal(got[k][0], want[k][0]) and np.array_equal(got[k][1], want[k][1]) for k in want), info)
            for use_oq in (False, True):
                q, t = bm.benchmark_bam(aln.AlignmentFile(bam), ref, var, use_oq=use_oq, bedfh=open(p['bed']))
                oq, ot = OB.benchmark_bam(list(_shim.AlignmentFile(p['sam'])), oref, OB.get_var_sites(p['vcf']), use_oq=use_oq, bed_path=p['bed'])
                check('benchmark_bam oq=%s' % use_oq, np.array_equal(q, oq) and np.array_equal(t, ot), info)
            q, t = bm.benchmark_fastq(p['fq'], aln.AlignmentFile(p['sam']), ref, var, open(p['bed']))
            oq, ot = OB.benchmark_fastq(p['fq'], list(_shim.AlignmentFile(p['sam'])), oref, OB.get_var_sites(p['vcf']), p['bed'])
            check('benchmark_fastq', np.array_equal(q, oq) and np.array_equal(t, ot), info)
        else:
            S = int(rng.integers(16, 200))
            info = dict(kind='tally', seed=cases, npairs=int(rng.integers(20, 400)), S=S, contigs=contigs(14 * S + 200), nrg=int(rng.integers(1, 6)))
            minscore = int(rng.choice([6, 6, 2, 10, 15]))
            p = OQ.synth_bqsr_set(tmp, **{k: v for k, v in info.items() if k != 'kind'})
            sb, fa = _shim.AlignmentFile(p['sam']), _shim.FastaFile(p['fa'])
            oref = {c: fa.fetch(c) for c in fa.references}
            var = bm.get_var_sites(p['vcf'])
            want = OQ.bam_to_bqsr_covariates(list(sb), [rg['ID'] for rg in sb.as_dict()['RG']], oref, var, minscore=minscore)
            bam = bamwriter.write_bam(os.path.join(tmp, 't.bam'), open(p['sam']).read())
            for src in (p['sam'], bam):
                got = bqsr.bam_to_bqsr_covariates(aln.AlignmentFile(src), p['fa'], var, minscore=minscore)
                check('tally minscore=%d' % minscore, all(np.array_equal(g, w) for g, w in zip(got, want)), info)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    if cases % 20 == 0:
        print('%d cases, %d mismatches, %.0f s left' % (cases, bad, t_end - time.time()), flush=True)
print('done: %d cases, %d mismatches' % (cases, bad))
sys.exit(1 if bad else 0)
