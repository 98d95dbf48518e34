"""
BAM-sourced covariate tally, CPU only: the oracle
(oracle/oracle_bqsr.py) against the reference's own known answers
(tests/test_gatk_bqsr.py:9-122, restated as data over the SAM-spec example of its
tests/conftest.py:46-69,86-101) and against goldens from the UNMODIFIED reference on
synthetic alignments (tests/golden/bqsr_*.npz); and the product's host-side pieces
(kbbq.aln read attributes, adaptor boundary / trimming in kbbq.gatk.bqsr).
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLD

VEC = ['meanq', 'rg_errs', 'rg_total', 'q_errs', 'q_total', 'pos_errs', 'pos_total',
       'dinuc_errs', 'dinuc_total']

SIMPLE_FASTA = '>ref\nAGCATGTTAGATAAGATAGCTGTGCTAGTAGGCAGTCAGCGCCAT\n'
SIMPLE_SAM = ('@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:ref\tLN:45\n'
              'r001\t99\tref\t7\t30\t8M2I4M1D3M\t=\t37\t39\tTTAGATAAAGGATACTG\t==99=?<*+/5:@A99:\n'
              'r001\t147\tref\t37\t30\t9M\t=\t7\t-39\tCAGCGGCAT\t><>???>>>\tNM:i:1\n')
DINUCS = [a + b for a in 'ATGC' for b in 'ATGC']


def readers():
    """(name, AlignmentFile class, cigar setter) for the oracle's stand-ins and the product's reader."""
    import _shim
    from kbbq import aln
    return [('shim', _shim.AlignmentFile), ('product', aln.AlignmentFile)]


@pytest.fixture(params=['shim', 'product'])
def simple_reads(request, tmp_path):
    p = tmp_path / 'simple.sam'
    p.write_text(SIMPLE_SAM)
    cls = dict(readers())[request.param]
    return lambda: list(cls(str(p)))


def test_known_answers_cycle_and_dinuc(oracle, simple_reads):
    import oracle_bqsr as OQ
    r = simple_reads()
    assert np.array_equal(OQ.bqsr_cycle(r[0]), np.arange(17))
    correct = np.flip(-(np.arange(9) + 1))
    assert np.array_equal(OQ.bqsr_cycle(r[1]), correct)
    r[1].cigartuples = [(4, 2), (0, 7)]                 # 9M -> 2S7M
    correct[0:2] = 0
    assert np.array_equal(OQ.bqsr_cycle(r[1]), correct)

    r = simple_reads()
    d = ['TT', 'TA', 'AG', 'GA', 'AT', 'TA', 'AA', 'AA', 'AG', 'GG', 'GA', 'AT', 'TA', 'AC', 'CT', 'TG']
    assert np.array_equal(OQ.bqsr_dinuc(r[0], use_oq=False), [-1] + [DINUCS.index(x) for x in d])
    d = ['AT', 'TG', 'GC', 'CC', 'CG', 'GC', 'CT', 'TG']
    correct = np.flip(np.array([-1] + [DINUCS.index(x) for x in d]))
    assert np.array_equal(OQ.bqsr_dinuc(r[1], use_oq=False), correct)
    r[1].cigartuples = [(4, 2), (0, 7)]
    correct[0:2] = 0
    assert np.array_equal(OQ.bqsr_dinuc(r[1], use_oq=False), correct)


def test_known_answers_adaptor_boundary_and_trim(oracle, simple_reads):
    import oracle_bqsr as OQ
    r = simple_reads()
    assert OQ.adaptor_boundary(r[0]) == 45 and OQ.adaptor_boundary(r[1]) == 5
    r[0].tlen = 0
    assert OQ.adaptor_boundary(r[0]) is None
    r[1].next_reference_start = 1000
    assert OQ.adaptor_boundary(r[1]) is None
    r = simple_reads()
    r[0].reference_start = 1000
    assert OQ.adaptor_boundary(r[0]) is None

    r = simple_reads()
    assert not OQ.trim(r[0]).any() and not OQ.trim(r[1]).any() and OQ.trim(r[0]).shape == (17,)
    want = np.zeros(9, dtype=bool); want[0] = True
    assert np.array_equal(OQ.trim(r[1], 36), want)                       # start of a reverse read
    want = np.zeros(17, dtype=bool); want[-1] = True
    assert np.array_equal(OQ.trim(r[0], 21), want)                       # end of a forward read
    want = np.zeros(17, dtype=bool); want[7:] = True
    assert np.array_equal(OQ.trim(r[0], 13), want)                       # left of an insertion
    want = np.zeros(17, dtype=bool); want[10:] = True
    assert np.array_equal(OQ.trim(r[0], 14), want)                       # right of an insertion
    want = np.zeros(17, dtype=bool); want[-3:] = True
    assert np.array_equal(OQ.trim(r[0], 18), want)                       # inside a deletion
    r[0].cigartuples = [(0, 8), (1, 2), (0, 4), (2, 4)]                  # deletion to the end
    assert not OQ.trim(r[0], 18).any()
    r[1].cigartuples = [(2, 1), (0, 8)]
    assert not OQ.trim(r[1], 36).any()


def test_known_answer_tally(oracle, tmp_path):
    """tests/test_gatk_bqsr.py:38-72: one hard-clipped base of quality 7."""
    import _shim
    import oracle_bqsr as OQ
    line = 'clipped\t0\tref\t9\t255\t1M9H\t*\t0\t0\tA\t(\tOQ:Z:(\tRG:Z:0'
    read = _shim.AlignedSegment(line)
    ref = {'ref': SIMPLE_FASTA.split('\n')[1]}
    got = OQ.bam_to_bqsr_covariates([read], ['0'], ref, {'ref': [9]})
    z = lambda *s: np.zeros(s, dtype=np.int64)
    qt = z(1, 43); qt[0, 7] = 1
    pt = z(1, 43, 2); pt[0, 7, 0] = 1
    want = [np.array([6]), np.array([0]), np.array([1]), z(1, 43), qt, z(1, 43, 2), pt, z(1, 43, 16), z(1, 43, 16)]
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


def _load(name):
    with open(os.path.join(GOLD, name + '.json')) as fh:
        info = json.load(fh)
    return info, dict(np.load(os.path.join(GOLD, name + '.npz')))


def _inputs(name, tmp_path, oracle):
    import oracle_bqsr as OQ
    info, gold = _load(name)
    d = tmp_path / name
    d.mkdir()
    paths = OQ.synth_bqsr_set(str(d), **info['case'])
    for k, v in paths.items():
        assert oracle.sha256(open(v, 'rb').read()) == info['input_sha256'][k], 'synthetic input drifted'
    return info, gold, paths


@pytest.mark.parametrize('name', ['bqsr_a', 'bqsr_b'])
@pytest.mark.parametrize('reader', ['shim', 'product'])
def test_oracle_matches_reference_goldens(oracle, name, reader, tmp_path):
    import _shim
    import oracle_bqsr as OQ
    info, gold, paths = _inputs(name, tmp_path, oracle)
    bam = dict(readers())[reader](paths['sam'])
    reads = list(bam)
    assert np.array_equal(np.concatenate([OQ.bqsr_cycle(r) for r in reads]), gold['cycle'])
    assert np.array_equal(np.concatenate([OQ.bqsr_dinuc(r) for r in reads]), gold['dinuc'])
    assert np.array_equal(np.concatenate([OQ.trim(r) for r in reads]), gold['trim'].astype(bool))
    b = [OQ.adaptor_boundary(r) for r in reads]
    assert np.array_equal(np.array([-(2 ** 40) if x is None else x for x in b]), gold['boundary'])
    ref = {c: _shim.FastaFile(paths['fa']).fetch(c) for c in _shim.FastaFile(paths['fa']).references}
    var = {}
    for rec in _shim.VariantFile(paths['vcf']):
        var.setdefault(rec.chrom, []).extend(range(rec.start, rec.stop))
    got = OQ.bam_to_bqsr_covariates(reads, list(info['rg_to_pu']), ref, var)
    for k, g in zip(VEC, got):
        assert np.array_equal(g, gold[k]), k
    assert gold['trim'].sum() > 50 and (gold['boundary'] > -(2 ** 40)).sum() > 20     # the cases do exercise trimming


def test_product_read_attributes_match_stand_ins(oracle, tmp_path):
    """kbbq.aln.AlignedRead against the independent stand-in reader on every synthetic read."""
    import _shim
    from kbbq import aln
    _, _, paths = _inputs('bqsr_a', tmp_path, oracle)
    a, b = list(_shim.AlignmentFile(paths['sam'])), list(aln.AlignmentFile(paths['sam']))
    assert len(a) == len(b) == 300
    for x, y in zip(a, b):
        for attr in ('query_name', 'flag', 'reference_name', 'reference_start', 'reference_end', 'query_length',
                     'query_alignment_start', 'query_alignment_end', 'query_alignment_length', 'is_paired',
                     'is_unmapped', 'mate_is_unmapped', 'is_reverse', 'mate_is_reverse', 'is_read1', 'is_read2',
                     'next_reference_start', 'tlen', 'template_length', 'query_sequence'):
            assert getattr(x, attr) == getattr(y, attr), attr
        assert x.get_aligned_pairs() == y.get_aligned_pairs()
        assert x.get_tag('OQ') == y.get_tag('OQ') and x.get_tag('RG') == y.get_tag('RG')
    hdr = aln.AlignmentFile(paths['sam']).header.as_dict()
    assert [rg['ID'] for rg in hdr['RG']] == ['g0', 'g1', 'g2'] and hdr['RG'][1]['PU'] == 'unit1'


def test_product_host_functions_known_answers(simple_reads):
    """The product's adaptor boundary / trimming (host, per read) on the reference's answers."""
    from kbbq.gatk import bqsr
    r = simple_reads()
    assert bqsr.bamread_adaptor_boundary(r[0]) == 45 and bqsr.bamread_adaptor_boundary(r[1]) == 5
    assert not bqsr.trim_bamread(r[0]).any() and bqsr.trim_bamread(r[1]).shape == (9,)
    for boundary, first, sl in ((21, 0, slice(16, 17)), (13, 0, slice(7, 17)), (14, 0, slice(10, 17)),
                                (18, 0, slice(14, 17)), (36, 1, slice(0, 1))):
        want = np.zeros(r[first].query_length, dtype=bool); want[sl] = True
        assert np.array_equal(bqsr.trim_bamread(r[first], boundary), want), boundary
    r[0].tlen = 0
    assert bqsr.bamread_adaptor_boundary(r[0]) is None
    r[1].next_reference_start = 1000
    assert bqsr.bamread_adaptor_boundary(r[1]) is None
    r = simple_reads()
    r[0].cigartuples = [(0, 8), (1, 2), (0, 4), (2, 4)]
    assert not bqsr.trim_bamread(r[0], 18).any()
    r[1].cigartuples = [(2, 1), (0, 8)]
    assert not bqsr.trim_bamread(r[1], 36).any()


@pytest.mark.parametrize('name', ['bqsr_a', 'bqsr_b'])
def test_product_trim_matches_reference(oracle, name, tmp_path):
    from kbbq import aln
    from kbbq.gatk import bqsr
    _, gold, paths = _inputs(name, tmp_path, oracle)
    reads = list(aln.AlignmentFile(paths['sam']))
    assert np.array_equal(np.concatenate([bqsr.trim_bamread(r) for r in reads]), gold['trim'].astype(bool))
    b = [bqsr.bamread_adaptor_boundary(r) for r in reads]
    assert np.array_equal(np.array([-(2 ** 40) if x is None else x for x in b]), gold['boundary'])


@pytest.mark.parametrize('name', ['bqsr_a', 'bqsr_b'])
def test_product_per_read_covariates_match_reference(oracle, name, tmp_path):
    from kbbq import aln
    from kbbq.gatk import bqsr
    _, gold, paths = _inputs(name, tmp_path, oracle)
    reads = list(aln.AlignmentFile(paths['sam']))
    assert np.array_equal(np.concatenate([bqsr.bamread_bqsr_cycle(r) for r in reads]), gold['cycle'])
    assert np.array_equal(np.concatenate([bqsr.bamread_bqsr_dinuc(r) for r in reads]), gold['dinuc'])


def test_native_sam_arrays_match_stand_ins(oracle, tmp_path):
    """aln.AlignmentFile.batch(): the native SAM reader's arrays against the independent stand-in reader."""
    import _shim
    from kbbq import aln
    for name in ('bqsr_a', 'bqsr_b'):
        _, _, paths = _inputs(name, tmp_path, oracle)
        ref = list(_shim.AlignmentFile(paths['sam']))
        b = aln.AlignmentFile(paths['sam']).batch()
        assert b.n == len(ref)
        assert np.array_equal(b.flag, [r.flag for r in ref]) and np.array_equal(b.pos, [r.reference_start for r in ref])
        assert np.array_equal(b.pnext, [r.next_reference_start for r in ref]) and np.array_equal(b.tlen, [r.tlen for r in ref])
        assert np.array_equal(b.qlen, [r.query_length for r in ref])
        assert np.array_equal(b.ref_span, [r.reference_end - r.reference_start for r in ref])
        assert np.array_equal(b.clip & 0xFFFF, [r.query_alignment_start for r in ref])
        assert np.array_equal(b.clip >> 16, [r.query_alignment_end for r in ref])
        assert [b.contig_names[c] for c in b.contig] == [r.reference_name for r in ref]
        assert [b.rg_ids[g] for g in b.rg] == [r.get_tag('RG') for r in ref]
        ops = [[(int(x) & 15, int(x) >> 4) for x in b.cigar[o:o + m]] for o, m in zip(b.cig_off, b.cig_n)]
        assert ops == [list(r.cigartuples) for r in ref]
        S = int(b.qlen[0])
        pitch = (S + 15) // 16 * 16
        for which, get in ((0, lambda r: r.query_sequence), (2, lambda r: r.get_tag('OQ'))):
            plane = b.plane(which, pitch)
            assert all(bytes(plane[i, :S]).decode() == get(ref[i]) for i in range(b.n)) and not plane[:, S:].any()
        assert np.array_equal(b.qual_len, b.qlen) and np.array_equal(b.oq_len, b.qlen)
        assert b.names() == [r.query_name for r in ref]
    # malformed lines are a ValueError, as for the object reader; so is a damaged BAM (intact ones: test_aln_readers.py)
    bad = tmp_path / 'bad.sam'
    bad.write_text('@HD\tVN:1.6\nr1\t0\tc\t1\n')
    with pytest.raises(ValueError):
        aln.AlignmentFile(str(bad))
    bam = tmp_path / 'x.bam'
    bam.write_bytes(b'BAM\x01....')
    with pytest.raises(ValueError):
        aln.AlignmentFile(str(bam))


@pytest.mark.parametrize('name', ['bqsr_a', 'bqsr_b'])
def test_per_read_applybqsr_functions_match_the_reference(oracle, name, tmp_path):
    """kbbq.gatk.applybqsr.bamread_cycle_covariates / bamread_dinuc_covariates / recalibrate_bamread (reference
    applybqsr.py:46-78) against arrays the UNMODIFIED reference produced on the same alignments (oracle/gen_golden.py),
    through SAM text and through a BAM file of it."""
    import bamwriter
    from kbbq import aln, compare_reads
    from kbbq.gatk import applybqsr
    info, gold, paths = _inputs(name, tmp_path, oracle)
    vectors = [gold[k] for k in VEC]
    dqs = oracle.get_delta_qs(*vectors)                 # the CPU restatement of the solve (the product's runs on the GPU)
    for source in (paths['sam'], bamwriter.write_bam(tmp_path / 'a.bam', open(paths['sam']).read())):
        bam = aln.AlignmentFile(source)
        rg_to_int = {rg: i for i, rg in enumerate(compare_reads.get_rg_to_pu(bam))}
        reads = list(bam)
        assert np.array_equal(np.concatenate([applybqsr.bamread_cycle_covariates(r) for r in reads]), gold['ab_cycle'])
        assert np.array_equal(np.concatenate([applybqsr.bamread_dinuc_covariates(r) for r in reads]), gold['ab_dinuc'])
        got = np.concatenate([applybqsr.recalibrate_bamread(r, vectors[0], *dqs, rg_to_int) for r in reads])
        assert got.dtype == np.int_ and np.array_equal(got, gold['ab_recal'])


def test_native_adaptor_trim_is_the_per_read_walk(oracle, tmp_path):
    """SamBatch.adaptor_trim() (csrc/sam_host.cpp, every alignment at once -- what bam_to_bqsr_covariates uses) against
    the reference's own trim arrays on the goldens, and against the per-read bamread_adaptor_boundary + _trim_range on
    random pairs: every CIGAR operation, boundaries inside deletions / insertions / clips, both strands, odd flags."""
    from kbbq import aln
    from kbbq.gatk import bqsr

    def expand(trim, nq):
        out = np.zeros(int(nq.sum()), dtype=bool)
        at = np.concatenate([[0], np.cumsum(nq)])
        for i, t in enumerate(trim):
            out[at[i] + (int(t) & 0xFFFF):at[i] + (int(t) >> 16)] = True
        return out
    for name in ('bqsr_a', 'bqsr_b'):
        _, gold, paths = _inputs(name, tmp_path, oracle)
        b = aln.AlignmentFile(paths['sam']).batch()
        assert np.array_equal(expand(b.adaptor_trim(), b.qual_len), gold['trim'].astype(bool))
    rng = np.random.default_rng(11)
    lines = ['@HD\tVN:1.6', '@SQ\tSN:c\tLN:100000', '@RG\tID:g']
    for i in range(3000):
        ops, q = [], 0
        for k in range(int(rng.integers(1, 7))):
            op = 'MIDNS=X'[int(rng.integers(0, 7))] if k else 'MS=X'[int(rng.integers(0, 4))]
            l = int(rng.integers(1, 12))
            ops.append((l, op)); q += l if op in 'MIS=X' else 0
        if q == 0 or not any(op in 'M=X' for _, op in ops):
            ops.append((5, 'M')); q += 5
        span = sum(l for l, op in ops if op in 'MDN=X')
        pos = int(rng.integers(1, 500))
        flag = int(rng.choice([99, 147, 83, 163] * 3 + [67, 115, 73, 97, 0, 16, 1 | 16 | 8, 4 | 1 | 32]))
        pnext = max(1, pos + int(rng.integers(-span - 5, span + 6)))
        tlen = int(rng.integers(-span - 8, span + 9))
        qual = 'I' * q
        lines.append('\t'.join(['r%d' % i, str(flag), 'c', str(pos), '60', ''.join('%d%s' % o for o in ops), '=', str(pnext), str(tlen),
                                'A' * q, qual, 'RG:Z:g', 'OQ:Z:' + 'I' * q]))
    sam = tmp_path / 'rand.sam'
    sam.write_text('\n'.join(lines) + '\n')
    f = aln.AlignmentFile(str(sam))
    got = f.batch().adaptor_trim()
    reads = list(aln.AlignmentFile(str(sam)))
    want = [bqsr._trim_range(r) for r in reads]
    assert [(int(t) & 0xFFFF, int(t) >> 16) for t in got] == want
    assert sum(1 for lo, hi in want if hi > lo) > 300                      # the cases do trim
