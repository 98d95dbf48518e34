"""
GPU parity tests of the benchmark path: K4 (find_read_errors) and K5
(calculate_q) through the kbbq.benchmark / kbbq.compare_reads API, against the reference's
known answers on the SAM-spec example, the goldens from the unmodified reference on synthetic
truth sets, and the CPU oracle.
"""
import sys

import numpy as np
import pytest

from conftest import load_golden
from test_oracle_benchmark import CORRECT_BENCHMARK, simple, OB   # noqa: F401  (fixtures)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def bm():
    import torch
    assert torch.cuda.is_available()
    from kbbq import benchmark
    return benchmark


def test_simple_known_answers(bm, simple, capfd, monkeypatch):
    """reference tests/test_compare_reads.py:87-122 and tests/test_benchmark.py:7-171."""
    import kbbq.main
    from kbbq import aln, compare_reads
    ref = bm.get_ref_dict(simple['fa'])
    assert ''.join(ref['ref']) == 'AGCATGTTAGATAAGATAGCTGTGCTAGTAGGCAGTCAGCGCCAT' and ref['ref'].dtype == np.dtype('U1')
    var = bm.get_var_sites(simple['vcf'])
    assert var == {'ref': [9]}
    with open(simple['bed']) as fh:
        bed = bm.get_bed_dict(ref, fh)
    want = np.zeros(45, dtype=bool); want[8:45] = True
    assert np.array_equal(bed['ref'], want)
    with open(simple['bed']) as fh:
        full = bm.get_full_skips(ref, var, fh)
    want = np.zeros(45, dtype=bool); want[0:8] = True; want[9] = True
    assert np.array_equal(full['ref'], want)
    reads = list(aln.AlignmentFile(simple['sam']))
    assert bm.get_bam_readname(reads[0]) == 'r001/1' and bm.get_bam_readname(reads[1]) == 'r001/2'
    r1skips = np.zeros(17, dtype=bool); r1skips[3] = True; r1skips[0:2] = True
    r2errs = np.zeros(9, dtype=bool); r2errs[5] = True
    e, s = compare_reads.find_read_errors(reads[0], ref, full)
    assert e.dtype == bool and not e.any() and np.array_equal(s, r1skips)
    e, s = compare_reads.find_read_errors(reads[1], ref, full)
    assert np.array_equal(e, r2errs) and not s.any()
    clipped = aln.AlignedRead.fromstring('clipped\t0\tref\t9\t255\t1M9H\t*\t0\t0\tA\t)', None)
    e, s = compare_reads.find_read_errors(clipped, ref, full)
    assert list(e) == [False] and list(s) == [False]
    clipped.cigartuples = [('L', 9)]
    with pytest.raises(ValueError):
        compare_reads.find_read_errors(clipped, ref, full)
    ed = bm.get_error_dict(aln.AlignmentFile(simple['sam']), ref, full)
    assert np.array_equal(ed['r001/1'][0], np.zeros(17, dtype=bool)) and np.array_equal(ed['r001/1'][1], r1skips)
    assert np.array_equal(ed['r001/2'][0], np.flip(r2errs)) and not ed['r001/2'][1].any()
    a, t = bm.calculate_q(np.array([False, True, True] + [False] * 100), np.array([3, 2, 1] + [1] * 100))
    assert list(a) == [0, 20, 0, 42] and list(t) == [0, 101, 1, 1]
    assert np.array_equal(bm.get_bamread_quals(reads[1]), [29, 27, 29, 30, 30, 30, 29, 29, 29])
    capfd.readouterr()
    bm.print_benchmark(a, 'test', t)
    assert capfd.readouterr().out == "1\t20\ttest\t101\n2\t0\ttest\t1\n3\t42\ttest\t1\n"
    with open(simple['bed']) as fh:
        bm.benchmark(simple['sam'], simple['fa'], simple['vcf'], label='test', bedfh=fh)
    assert capfd.readouterr().out == CORRECT_BENCHMARK
    with open(simple['bed']) as fh:
        bm.benchmark(simple['sam'], simple['fa'], simple['vcf'], fastqfile=simple['fq'], label='test', bedfh=fh)
    assert capfd.readouterr().out == CORRECT_BENCHMARK
    for extra in ([], ['-f', simple['fq']]):
        with monkeypatch.context() as m:
            m.setattr(sys, 'argv', [sys.argv[0], 'benchmark', '-b', simple['sam'], '-r', simple['fa'],
                                    '-v', simple['vcf'], '-d', simple['bed'], '--label=test'] + extra)
            kbbq.main.main()
        assert capfd.readouterr().out == CORRECT_BENCHMARK


@pytest.mark.parametrize('name', ['bench_a', 'bench_b'])
def test_matches_reference_goldens_and_oracle(bm, OB, oracle, name, tmp_path, capfd):
    import _shim
    from kbbq import aln, compare_reads
    info, gold = load_golden(name)
    paths = OB.synth_truthset(str(tmp_path), **info['case'])
    assert {k: oracle.sha256(open(v, 'rb').read()) for k, v in paths.items()} == info['input_sha256']
    ref, var = bm.get_ref_dict(paths['fa']), bm.get_var_sites(paths['vcf'])
    with open(paths['bed']) as fh:
        full = bm.get_full_skips(ref, var, fh)
    oref = OB.get_ref_dict(paths['fa'])
    assert all(np.array_equal(full[c], OB.get_full_skips(oref, OB.get_var_sites(paths['vcf']), paths['bed'])[c]) for c in full)
    ed = bm.get_error_dict(aln.AlignmentFile(paths['sam']), ref, full)
    assert list(ed) == info['read_keys']
    assert np.array_equal(np.concatenate([ed[k][0] for k in ed]).astype(np.uint8), gold['errors'])
    assert np.array_equal(np.concatenate([ed[k][1] for k in ed]).astype(np.uint8), gold['skips'])
    # per-read API on a few reads against the oracle
    oreads = list(_shim.AlignmentFile(paths['sam']))
    ofull = {c: full[c] for c in full}
    for r, o in list(zip(aln.AlignmentFile(paths['sam']), oreads))[:25]:
        e, s = compare_reads.find_read_errors(r, ref, full)
        oe, os_ = OB.find_read_errors(o, oref, ofull)
        assert np.array_equal(e, oe) and np.array_equal(s, os_)
    for tag, kw in (('bam', dict()), ('bam_oq', dict(use_oq=True))):
        with open(paths['bed']) as fh:
            a, t = bm.benchmark_bam(aln.AlignmentFile(paths['sam']), ref, var, bedfh=fh, **kw)
        assert np.array_equal(a, gold[tag + '_q']) and np.array_equal(t, gold[tag + '_n']), tag
    with open(paths['bed']) as fh:
        a, t = bm.benchmark_fastq(paths['fq'], aln.AlignmentFile(paths['sam']), ref, var, fh)
    assert np.array_equal(a, gold['fastq_q']) and np.array_equal(t, gold['fastq_n'])
    a, t = bm.benchmark_bam(aln.AlignmentFile(paths['sam']), ref, var)
    assert np.array_equal(a, gold['nobed_q']) and np.array_equal(t, gold['nobed_n'])
    capfd.readouterr()
    for tag, kw in (('bam', dict()), ('fastq', dict(fastqfile=paths['fq']))):
        with open(paths['bed']) as fh:
            bm.benchmark(paths['sam'], paths['fa'], paths['vcf'], label='lbl', bedfh=fh, **kw)
        assert capfd.readouterr().out == info['printed'][tag]


def test_python_index_wraps_and_errors(bm, OB):
    """The reference's negative-index behaviour and its exceptions on odd CIGARs."""
    import _shim
    from kbbq import aln, compare_reads
    ref = {'c': aln.chars('ACGTACGTACGTACGTACGT')}
    mask = np.zeros(20, dtype=bool); mask[[4, 5, 11]] = True
    full = {'c': mask}
    oref = {'c': aln.codes(ref['c'])}
    lines = ['a\t0\tc\t3\t60\t2D4M1I3M\t*\t0\t0\tTACGTTAC\tIIIIIIII',      # leading deletion: ORs into skips[-1]
             'b\t0\tc\t5\t60\t3I4M\t*\t0\t0\tGGGACGT\tIIIIIII',          # leading insertion: subset[-1] & subset[0]
             'c\t16\tc\t1\t60\t2S3M2N2M1S2H\t*\t0\t0\tTTACGTAG\tIIIIIIII']
    for ln in lines:
        r, o = aln.AlignedRead(ln), _shim.AlignedSegment(ln)
        e, s = compare_reads.find_read_errors(r, ref, full)
        oe, os_ = OB.find_read_errors(o, oref, full)
        assert np.array_equal(e, oe) and np.array_equal(s, os_), ln
    with pytest.raises(IndexError):                                   # insertion as the last reference-consuming op
        compare_reads.find_read_errors(aln.AlignedRead('d\t0\tc\t1\t60\t4M2I\t*\t0\t0\tACGTGG\tIIIIII'), ref, full)
    with pytest.raises(ValueError):                                   # read runs off the end of the contig
        compare_reads.find_read_errors(aln.AlignedRead('e\t0\tc\t18\t60\t6M\t*\t0\t0\tCGTACG\tIIIIII'), ref, full)
    with pytest.raises(ValueError):
        bm.calculate_q(np.array([True]), np.array([-1]))


@pytest.mark.parametrize('case', [dict(seed=101, npairs=700), dict(seed=102, npairs=500, readlen=(10, 40)),
                                  dict(seed=103, npairs=400, readlen=(140, 160), contigs=(('a', 4000), ('b', 2500), ('c', 2600)))])
def test_random_truth_sets_against_the_oracle(bm, OB, case, tmp_path):
    """More random truth sets (substitutions, indels, N-skips, clips, both strands): K4 flags and K5 counts against the
    oracle's scalar walk, through the native SAM reader and through pysam-style read objects."""
    import _shim
    from kbbq import aln
    paths = OB.synth_truthset(str(tmp_path), **case)
    ref, var = bm.get_ref_dict(paths['fa']), bm.get_var_sites(paths['vcf'])
    with open(paths['bed']) as fh:
        full = bm.get_full_skips(ref, var, fh)
    oref = OB.get_ref_dict(paths['fa'])
    want = OB.get_error_dict(list(_shim.AlignmentFile(paths['sam'])), oref, full)
    for source in (aln.AlignmentFile(paths['sam']), list(aln.AlignmentFile(paths['sam']))):
        got = bm.get_error_dict(source, ref, full)
        assert list(got) == list(want)
        for k in want:
            assert np.array_equal(got[k][0], want[k][0]) and np.array_equal(got[k][1], want[k][1]), k
    for use_oq in (False, True):
        a, t = bm.benchmark_bam(aln.AlignmentFile(paths['sam']), ref, var, use_oq=use_oq, bedfh=open(paths['bed']))
        oa, ot = OB.benchmark_bam(list(_shim.AlignmentFile(paths['sam'])), oref, OB.get_var_sites(paths['vcf']), use_oq=use_oq,
                                  bed_path=paths['bed'])
        assert np.array_equal(a, oa) and np.array_equal(t, ot)
    a, t = bm.benchmark_fastq(paths['fq'], aln.AlignmentFile(paths['sam']), ref, var, open(paths['bed']))
    oa, ot = OB.benchmark_fastq(paths['fq'], list(_shim.AlignmentFile(paths['sam'])), oref, OB.get_var_sites(paths['vcf']),
                                paths['bed'])
    assert np.array_equal(a, oa) and np.array_equal(t, ot)


@pytest.mark.parametrize('name', ['bench_a'])
def test_bam_input_prints_the_reference_tables(bm, OB, name, tmp_path, capfd):
    """The same truth set as a BAM file (what the reference's users have): `kbbq benchmark` prints the golden tables,
    and the BAM-sourced tally gives the golden vectors from a BAM."""
    import bamwriter
    info, gold = load_golden(name)
    paths = OB.synth_truthset(str(tmp_path), **info['case'])
    bam = bamwriter.write_bam(tmp_path / 'truth.bam', open(paths['sam']).read())
    capfd.readouterr()
    for tag, kw in (('bam', dict()), ('fastq', dict(fastqfile=paths['fq']))):
        with open(paths['bed']) as fh:
            bm.benchmark(bam, paths['fa'], paths['vcf'], label='lbl', bedfh=fh, **kw)
        assert capfd.readouterr().out == info['printed'][tag]
    from kbbq import aln
    from kbbq.gatk import bqsr
    from test_oracle_bqsr import VEC, _inputs
    import oracle as O
    binfo, bgold, bpaths = _inputs('bqsr_a', tmp_path, O)
    bbam = bamwriter.write_bam(tmp_path / 'tally.bam', open(bpaths['sam']).read())
    got = bqsr.bam_to_bqsr_covariates(aln.AlignmentFile(bbam), bpaths['fa'], bm.get_var_sites(bpaths['vcf']))
    for k, g in zip(VEC, got):
        assert np.array_equal(g, bgold[k]), k


def test_separate_mask_array_gives_the_same_flags(bm, OB, tmp_path, monkeypatch):
    """kbbq_find_errors_dev with the site mask as its own array (the C ABI's general form; the Python layer normally
    fuses it into bit 7 of the reference bytes): same flags, same counts."""
    import _shim
    from kbbq import aln
    paths = OB.synth_truthset(str(tmp_path), seed=77, npairs=600)
    ref, var = bm.get_ref_dict(paths['fa']), bm.get_var_sites(paths['vcf'])
    with open(paths['bed']) as fh:
        full = bm.get_full_skips(ref, var, fh)
    fused = bm.get_error_dict(aln.AlignmentFile(paths['sam']), ref, full)
    want = OB.get_error_dict(list(_shim.AlignmentFile(paths['sam'])), OB.get_ref_dict(paths['fa']), full)
    monkeypatch.setenv('KBBQ_REFERENCE_MASK', 'separate')
    g = bm._Genome(ref, full)
    assert g.mask is not None
    separate = bm.get_error_dict(aln.AlignmentFile(paths['sam']), ref, full)
    assert list(separate) == list(fused) == list(want)
    for k in want:
        for got in (separate[k], fused[k]):
            assert np.array_equal(got[0], want[k][0]) and np.array_equal(got[1], want[k][1]), k
    a, t = bm.benchmark_bam(aln.AlignmentFile(paths['sam']), ref, var, bedfh=open(paths['bed']))
    monkeypatch.delenv('KBBQ_REFERENCE_MASK')
    assert bm._Genome(ref, full).mask is None
    a2, t2 = bm.benchmark_bam(aln.AlignmentFile(paths['sam']), ref, var, bedfh=open(paths['bed']))
    assert np.array_equal(a, a2) and np.array_equal(t, t2)
