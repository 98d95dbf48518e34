#!/usr/bin/env python3
"""
gen_golden.py -- produce tests/golden/* by running the UNMODIFIED reference
(/root/reference/kbbq) in the build container.  TEST INFRASTRUCTURE ONLY.

    python oracle/gen_golden.py            # all cases (about two minutes)

The reference cannot travel to the GPU box, so its outputs are committed as
small fixtures (data only): the 9 covariate vectors (recalibrate.py:121), the 4
delta-Q tables (applybqsr.py:103), SHA-256 of the FASTQ text it prints plus the
first/last records, and a few numeric tables (prior, q<->p, a gatk_delta_q
grid).  Inputs are NOT stored: they are regenerated from (seed, shape) by
oracle.synth(), and their SHA-256 is stored so a test can prove it feeds the
oracle the very bytes the reference saw.

Timing of the reference's own passes on this container's CPU (1 core; the
reference is single threaded) is recorded as well: BASELINE.md section 3(1).
"""
import contextlib
import io
import json
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import oracle as O      # noqa: E402  (own code: synthetic inputs + FASTQ writer only)
import _shim            # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')

# name -> synthetic shape.  C1 is BASELINE.json configs[0]; the others are cuts of
# configs[2] (8 RGs via --infer-rg) and configs[4] (mixed, ascending lengths).
CASES = {
    'c1_10k_1rg':       dict(n=10000, seed=1, len_lo=150, len_hi=150, nrg=1, infer_rg=False, qlo=0, qhi=41),
    'c3cut_2k_8rg':     dict(n=2000, seed=3, len_lo=150, len_hi=150, nrg=8, infer_rg=True, qlo=0, qhi=41),
    'c5cut_2k_mixed':   dict(n=2000, seed=5, len_lo=36, len_hi=300, nrg=2, infer_rg=True, qlo=0, qhi=41),
    'q42_500_3rg':      dict(n=500, seed=7, len_lo=100, len_hi=100, nrg=3, infer_rg=True, qlo=2, qhi=42),
    'short_64_1rg':     dict(n=64, seed=9, len_lo=1, len_hi=17, nrg=1, infer_rg=False, qlo=0, qhi=41),
}


def run_case(name, c, recal, applybqsr, tmp):
    seq, cseq, qual, meta = O.synth(0, c['n'], c['n'], c['seed'], c['len_lo'], c['len_hi'],
                                    c['nrg'], c['qlo'], c['qhi'])
    names = O.synth_names(0, c['n'], c['nrg'], with_rg=c['infer_rg'])
    fa = os.path.join(tmp, name + '.fq')
    fb = os.path.join(tmp, name + '.cor.fq')
    O.write_fastq(fa, names, seq, qual, meta)
    O.write_fastq(fb, names, cseq, qual, meta)
    in_sha = [O.sha256(open(f, 'rb').read()) for f in (fa, fb)]

    t0 = time.perf_counter()
    vectors = recal.fastq_to_covariate_arrays([fa, fb], infer_rg=c['infer_rg'])
    t1 = time.perf_counter()
    dqs = applybqsr.get_delta_qs(*vectors)
    t2 = time.perf_counter()
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        recal.recalibrate_fastq([fa, fb], infer_rg=c['infer_rg'])
    t3 = time.perf_counter()
    text = buf.getvalue()
    recs = text.split('\n')
    bases = int((meta & 0xFFFF).sum())
    info = dict(case=c, input_sha256=in_sha, output_sha256=O.sha256(text),
                output_len=len(text), first_records='\n'.join(recs[:32]),
                last_records='\n'.join(recs[-33:]), bases=bases,
                reference_timing=dict(pass1_s=t1 - t0, solve_s=t2 - t1, end_to_end_s=t3 - t2,
                                      pass1_bases_per_s=bases / (t1 - t0),
                                      end_to_end_bases_per_s=bases / (t3 - t2),
                                      cores_used=1, host_cores=os.cpu_count()))
    keys = ['meanq', 'rg_errs', 'rg_total', 'q_errs', 'q_total', 'pos_errs', 'pos_total',
            'dinuc_errs', 'dinuc_total']
    arrs = {k: np.asarray(v).astype(np.int64) for k, v in zip(keys, vectors)}
    for k, v in zip(['rgdq', 'qdq', 'posdq', 'dinucdq'], dqs):
        arrs[k] = np.asarray(v).astype(np.int64)
    np.savez_compressed(os.path.join(GOLD, name + '.npz'), **arrs)
    with open(os.path.join(GOLD, name + '.json'), 'w') as fh:
        json.dump(info, fh, indent=1)
    print('%-18s bases=%d pass1 %.2fs solve %.2fs e2e %.2fs  out=%s' % (
        name, bases, t1 - t0, t2 - t1, t3 - t2, info['output_sha256'][:12]), flush=True)
    run_report_case(name, c, vectors, names)


def run_report_case(name, c, vectors, names):
    """Model file: the reference's vectors_to_report on the vectors it just
    produced, printed through its RecalibrationReport, re-read through its fromfile."""
    import types
    from kbbq import compare_reads as utils
    from kbbq import recaltable
    from kbbq.gatk import bqsr
    if c['infer_rg']:
        rg_order = []
        for nm in names:
            rg = utils.fastq_infer_rg(types.SimpleNamespace(name=nm))
            if rg not in rg_order:
                rg_order.append(rg)
    else:
        rg_order = ['0']
    rep = bqsr.vectors_to_report(*[np.asarray(v) for v in vectors], rg_order)
    text = str(rep)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, 'r.txt')
        rep.write(path)
        assert open(path).read() == text
        again = recaltable.RecalibrationReport.fromfile(path)
        reread = str(again) == text
    lines = text.split('\n')
    info = dict(rg_order=rg_order, sha256=O.sha256(text), length=len(text), nlines=len(lines),
                head='\n'.join(lines[:40]), sampled={str(i): lines[i] for i in range(0, len(lines), 499)},
                table_heads=[ln for ln in lines if ln.startswith('#:GATKTable:')],
                reference_reread_is_identical=bool(reread))
    with open(os.path.join(GOLD, 'report_' + name + '.json'), 'w') as fh:
        json.dump(info, fh, indent=1)
    if len(text) < 200000:
        with open(os.path.join(GOLD, 'report_' + name + '.txt'), 'w') as fh:
            fh.write(text)
    print('%-18s report %d lines sha=%s reread_identical=%s' % (name, len(lines), info['sha256'][:12], reread), flush=True)


def numeric_tables(utils):
    """Small numeric goldens pinning the model's floating point (hazards H4, H5)."""
    out = {}
    out['prior_dist_hex'] = [float(x).hex() if np.isfinite(x) else '-inf'
                             for x in utils.RescaledNormal.prior_dist]
    out['prior_dist_is_float64_exact'] = bool(all(
        (not np.isfinite(x)) or np.longdouble(float(x)) == x for x in utils.RescaledNormal.prior_dist))
    q = np.arange(43)
    out['q_to_p_hex'] = [float(x).hex() for x in utils.q_to_p(q)]
    out['p_to_q_of_q_to_p'] = [int(x) for x in utils.p_to_q(utils.q_to_p(q))]
    out['p_to_q_samples'] = dict(p=[.2, .3, .4, .1, .01, .001, 0.0, 1.0, 1e-9],
                                 q=[int(x) for x in utils.p_to_q(np.array([.2, .3, .4, .1, .01, .001, 0.0, 1.0, 1e-9]))])
    # gatk_delta_q grid: every prior 0..42 against a list of (errs, total) cells
    rng = np.random.default_rng(12345)
    cells = [(0, 0), (0, 1), (1, 1), (0, 10), (10, 10), (5, 1000), (0, 1000), (1000, 1000),
             (10, 1000), (200, 1000), (0, 50000), (123456, 10 ** 7), (0, 10 ** 10),
             (10 ** 10, 10 ** 10), (7 * 10 ** 8, 7 * 10 ** 9), (3, 7 * 10 ** 9),
             (2 ** 31, 2 ** 33), (999999999, 10 ** 10)]
    for _ in range(400):
        t = int(10 ** rng.uniform(0, 10.3))
        e = int(t * 10 ** (-rng.uniform(0, 4.5)))
        cells.append((min(e, t), t))
    errs = np.array([c[0] for c in cells], dtype=np.int64)
    tot = np.array([c[1] for c in cells], dtype=np.int64)
    prior = np.broadcast_to(np.arange(43)[:, None], (43, len(cells))).copy()
    dq = utils.gatk_delta_q(prior, np.broadcast_to(errs, prior.shape).copy(),
                            np.broadcast_to(tot, prior.shape).copy())
    return out, dict(grid_errs=errs, grid_total=tot, grid_dq=np.asarray(dq).astype(np.int64))


BENCH_CASES = {'bench_a': dict(seed=11, npairs=150), 'bench_b': dict(seed=12, npairs=60, readlen=(20, 80))}


def run_bench_case(name, c, tmp):
    """Benchmark path: the reference's own functions on a synthetic truth set."""
    import contextlib
    import io
    import oracle_benchmark as OB
    import pysam                                        # the shim's stand-in
    from kbbq import benchmark as rb
    d = os.path.join(tmp, name); os.makedirs(d)
    paths = OB.synth_truthset(d, **c)
    sha = {k: O.sha256(open(v, 'rb').read()) for k, v in paths.items()}
    ref = rb.get_ref_dict(paths['fa'])
    var = rb.get_var_sites(paths['vcf'])
    with open(paths['bed']) as bedfh:
        fullskips = rb.get_full_skips(ref, var, bedfh)
    edict = rb.get_error_dict(pysam.AlignmentFile(paths['sam']), ref, fullskips)
    keys = list(edict)
    arrs = dict(errors=np.concatenate([edict[k][0] for k in keys]).astype(np.uint8),
                skips=np.concatenate([edict[k][1] for k in keys]).astype(np.uint8),
                lens=np.array([len(edict[k][0]) for k in keys], dtype=np.int64))
    for tag, kw in (('bam', dict()), ('bam_oq', dict(use_oq=True))):
        with open(paths['bed']) as bedfh:
            a, t = rb.benchmark_bam(pysam.AlignmentFile(paths['sam']), ref, var, bedfh=bedfh, **kw)
        arrs[tag + '_q'], arrs[tag + '_n'] = np.asarray(a).astype(np.int64), np.asarray(t).astype(np.int64)
    with open(paths['bed']) as bedfh:
        a, t = rb.benchmark_fastq(paths['fq'], pysam.AlignmentFile(paths['sam']), ref, var, bedfh)
    arrs['fastq_q'], arrs['fastq_n'] = np.asarray(a).astype(np.int64), np.asarray(t).astype(np.int64)
    # without a BED as well
    a, t = rb.benchmark_bam(pysam.AlignmentFile(paths['sam']), ref, var)
    arrs['nobed_q'], arrs['nobed_n'] = np.asarray(a).astype(np.int64), np.asarray(t).astype(np.int64)
    texts = {}
    for tag, kw in (('bam', dict()), ('fastq', dict(fastqfile=paths['fq']))):
        buf = io.StringIO()
        with open(paths['bed']) as bedfh, contextlib.redirect_stdout(buf):
            rb.benchmark(paths['sam'], paths['fa'], paths['vcf'], label='lbl', bedfh=bedfh, **kw)
        texts[tag] = buf.getvalue()
    np.savez_compressed(os.path.join(GOLD, name + '.npz'), **arrs)
    with open(os.path.join(GOLD, name + '.json'), 'w') as fh:
        json.dump(dict(case=c, input_sha256=sha, read_keys=keys, printed=texts), fh, indent=1)
    print('%-18s reads=%d bases=%d errors=%d skips=%d' % (name, len(keys), arrs['lens'].sum(),
                                                          arrs['errors'].sum(), arrs['skips'].sum()), flush=True)


BQSR_CASES = {'bqsr_a': dict(seed=21, npairs=150, S=60), 'bqsr_b': dict(seed=22, npairs=80, S=37, nrg=2)}


def run_bqsr_case(name, c, tmp):
    """BAM-sourced tally: the reference's bam_to_bqsr_covariates and its
    per-read covariate / trimming functions on synthetic alignments (SAM text through the
    stand-in reader)."""
    import oracle_bqsr as OQ
    import pysam
    from kbbq import compare_reads as utils
    from kbbq.gatk import bqsr
    d = os.path.join(tmp, name); os.makedirs(d)
    paths = OQ.synth_bqsr_set(d, **c)
    sha = {k: O.sha256(open(v, 'rb').read()) for k, v in paths.items()}
    var_pos = utils.get_var_sites(paths['vcf'])
    vectors = bqsr.bam_to_bqsr_covariates(pysam.AlignmentFile(paths['sam']), paths['fa'], var_pos)
    keys = ['meanq', 'rg_errs', 'rg_total', 'q_errs', 'q_total', 'pos_errs', 'pos_total',
            'dinuc_errs', 'dinuc_total']
    arrs = {k: np.asarray(v).astype(np.int64) for k, v in zip(keys, vectors)}
    reads = list(pysam.AlignmentFile(paths['sam']))
    arrs['cycle'] = np.concatenate([bqsr.bamread_bqsr_cycle(r) for r in reads]).astype(np.int64)
    arrs['dinuc'] = np.concatenate([bqsr.bamread_bqsr_dinuc(r) for r in reads]).astype(np.int64)
    arrs['trim'] = np.concatenate([bqsr.trim_bamread(r) for r in reads]).astype(np.uint8)
    arrs['boundary'] = np.array([-(2 ** 40) if bqsr.bamread_adaptor_boundary(r) is None
                                 else bqsr.bamread_adaptor_boundary(r) for r in reads], dtype=np.int64)
    bam = pysam.AlignmentFile(paths['sam'])
    rep = str(bqsr.bam_to_report(bam, paths['fa'], var_pos))
    # the per-read ApplyBQSR emulation (gatk/applybqsr.py:46-78) with the model solved from the vectors above
    from kbbq.gatk import applybqsr
    dqs = applybqsr.get_delta_qs(*vectors)
    rg_to_int = {rg: i for i, rg in enumerate(utils.get_rg_to_pu(bam))}
    arrs['ab_cycle'] = np.concatenate([applybqsr.bamread_cycle_covariates(r) for r in reads]).astype(np.int64)
    arrs['ab_dinuc'] = np.concatenate([applybqsr.bamread_dinuc_covariates(r) for r in reads]).astype(np.int64)
    arrs['ab_recal'] = np.concatenate([applybqsr.recalibrate_bamread(r, vectors[0], *dqs, rg_to_int) for r in reads]).astype(np.int64)
    np.savez_compressed(os.path.join(GOLD, name + '.npz'), **arrs)
    with open(os.path.join(GOLD, name + '.json'), 'w') as fh:
        json.dump(dict(case=c, input_sha256=sha, rg_to_pu=utils.get_rg_to_pu(bam),
                       report_sha256=O.sha256(rep), report_len=len(rep)), fh, indent=1)
    print('%-18s reads=%d counted=%d errors=%d trimmed=%d meanq=%s' % (
        name, len(reads), arrs['rg_total'].sum(), arrs['rg_errs'].sum(), arrs['trim'].sum(),
        arrs['meanq'].tolist()), flush=True)


def main():
    os.makedirs(GOLD, exist_ok=True)
    _shim.install()
    import scipy
    from kbbq import recalibrate as recal            # the reference, unmodified
    from kbbq import compare_reads as utils
    from kbbq.gatk import applybqsr
    assert recal.__file__.startswith('/root/reference/')
    only = sys.argv[1:]
    with tempfile.TemporaryDirectory() as tmp:
        for name, c in CASES.items():
            if only and name not in only:
                continue
            run_case(name, c, recal, applybqsr, tmp)
        for name, c in BENCH_CASES.items():
            if only and name not in only:
                continue
            run_bench_case(name, c, tmp)
        for name, c in BQSR_CASES.items():
            if only and name not in only:
                continue
            run_bqsr_case(name, c, tmp)
    if not only:
        info, arrs = numeric_tables(utils)
        info['versions'] = dict(numpy=np.__version__, scipy=scipy.__version__,
                                python=sys.version.split()[0])
        np.savez_compressed(os.path.join(GOLD, 'numeric.npz'), **arrs)
        with open(os.path.join(GOLD, 'numeric.json'), 'w') as fh:
            json.dump(info, fh, indent=1)
        print('numeric tables written')


if __name__ == '__main__':
    main()
