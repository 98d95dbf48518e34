"""
The oracle against the known answers of the reference's OWN tests (restated as
data; reference paths relative to /root/reference/).  CPU only.
"""
import numpy as np
import pytest


def _write(tmp_path, name, recs):
    p = tmp_path / name
    p.write_text(''.join('@%s\n%s\n+\n%s\n' % r for r in recs))
    return str(p)


@pytest.fixture()
def atg_pair(tmp_path):
    # tests/test_recalibrate.py:19-35 -- ATG / ACG, quals '((#' = 7,7,2
    return (_write(tmp_path, 'u.fq', [('foo', 'ATG', '((#')]),
            _write(tmp_path, 'c.fq', [('foo', 'ACG', '((#')]))


@pytest.fixture()
def atg_pair_rg(tmp_path):
    # tests/test_recalibrate.py:37-51
    return (_write(tmp_path, 'ur.fq', [('foo/1_RG:Z:bar', 'ATG', '((#')]),
            _write(tmp_path, 'cr.fq', [('foo/1_RG:Z:bar', 'ACG', '((#')]))


def _expected_vectors():
    # tests/test_recalibrate.py:53-71
    pe = np.zeros((1, 43, 6), dtype=np.int64); pt = np.zeros((1, 43, 6), dtype=np.int64)
    pe[0, 7, 1] = 1; pt[0, 7, 0] = 1; pt[0, 7, 1] = 1
    de = np.zeros((1, 43, 16), dtype=np.int64); dt = np.zeros((1, 43, 16), dtype=np.int64)
    de[0, 7, 1] = 1; dt[0, 7, 1] = 1          # 'AT' -> 1
    return [np.array([6]), np.array([1]), np.array([2]),
            np.array([[0] * 7 + [1] + [0] * 35]), np.array([[0] * 7 + [2] + [0] * 35]),
            pe, pt, de, dt]


@pytest.mark.parametrize('use_rg', [False, True])
def test_tally_known_answer(oracle, atg_pair, atg_pair_rg, use_rg):
    files = atg_pair_rg if use_rg else atg_pair
    b = oracle.pack_records(oracle.read_fastq(files[0]), oracle.read_fastq(files[1]), use_rg)
    got = oracle.accumulate(b['seq'], b['cseq'], b['qual'], b['meta'], b['R'], b['S'])
    for a, g in zip(_expected_vectors(), got):
        assert np.array_equal(a, g)
    # the slow pure-Python twin agrees too
    got2 = oracle.py_accumulate(oracle.read_fastq(files[0]), oracle.read_fastq(files[1]), use_rg)
    for a, g in zip(_expected_vectors(), got2):
        assert np.array_equal(a, g)


def test_end_to_end_stdout_known_answer(oracle, atg_pair, atg_pair_rg):
    # tests/test_recalibrate.py:81-99 -- quals become ''# = 6,6,2
    text, _, _ = oracle.recalibrate_fastq_text(atg_pair)
    assert text == "@foo\nATG\n+\n''#\n"
    text, _, _ = oracle.recalibrate_fastq_text(atg_pair_rg, True)
    assert text == "@foo/1_RG:Z:bar\nATG\n+\n''#\n"


def test_prior_table(oracle):
    # tests/test_compare_reads.py:124-128
    assert oracle.PRIOR_DIST[0] == np.log(.9)
    assert np.all(oracle.PRIOR_DIST < 0)
    # SURVEY A7 [probed]: log(.9) - 2 d^2 up to d = 18, -inf beyond
    assert np.all(np.isfinite(oracle.PRIOR_DIST[:19])) and np.all(np.isinf(oracle.PRIOR_DIST[19:]))


def test_gatk_delta_q_signs(oracle):
    # tests/test_compare_reads.py:141-151
    prior_q = np.array([10, 20, 30])
    dq = oracle.gatk_delta_q(prior_q, np.array([10, 200, 0]), np.array([1000, 1000, 50000]))
    assert dq.shape == prior_q.shape
    assert dq[0] > 0 and dq[1] < 0 and dq[2] > 0
    assert np.all(dq + prior_q <= 42) and np.all(dq + prior_q > 0)


def test_p_to_q_q_to_p(oracle):
    # tests/test_compare_reads.py:153-166
    assert np.array_equal(oracle.p_to_q(np.array([.2, .3, .4, .1, .01, .001])),
                          np.array([6, 5, 3, 10, 20, 30]))
    allq = np.arange(43)
    diff = allq - oracle.p_to_q(oracle.q_to_p(allq))
    assert np.all((diff >= 0) & (diff <= 1))
    assert np.allclose(oracle.q_to_p(np.array([6, 10, 20, 30])).astype(float),
                       np.array([.251188643, .1, .01, .001]))


def test_dinuc_order_and_names(oracle):
    # tests/test_compare_reads.py:130-139, 210-217
    order = ['AA', 'AT', 'AG', 'AC', 'TA', 'TT', 'TG', 'TC', 'GA', 'GT', 'GG', 'GC',
             'CA', 'CT', 'CG', 'CC']
    assert [oracle._DINUC[d] for d in order] == list(range(16))
    assert oracle.infer_second('read1/2_RG:Z:FOO') and not oracle.infer_second('read1/1')
    assert oracle.infer_rg('read1/1_RG:Z:FOO') == 'FOO'


def test_dinuc_covariate_known_answers(oracle):
    # tests/test_compare_reads.py:172-189 through the apply table: give every dinuc
    # its own delta and read the context back.  ATGCATGC, q = 10.
    seq = np.frombuffer(b'ATGCATGC' + bytes(8), dtype=np.uint8).reshape(1, 16).copy()
    qual = np.full((1, 16), 43, dtype=np.uint8); qual[0, 8:] = 0
    meta = np.array([8], dtype=np.uint32)
    ddq = np.zeros((1, 43, 17), dtype=np.int64); ddq[0, :, :16] = np.arange(16) + 1
    z = lambda *s: np.zeros(s, dtype=np.int64)
    ctx = lambda: oracle.apply(seq, qual, meta, z(1), z(1), z(1, 43), z(1, 43, 16), ddq)[0, :8] - 1
    D = oracle._DINUC
    want = [-1] + [D[x] for x in ('AT', 'TG', 'GC', 'CA', 'AT', 'TG', 'GC')]
    assert list(ctx()) == want
    seq[0, 1] = ord('N'); want[1] = -1; want[2] = -1
    assert list(ctx()) == want
    qual[0, 6] = 2 + 33; want[6] = 2 - 1      # below minscore: passes through unchanged
    assert list(ctx()) == want


def test_apply_known_answer(oracle):
    # tests/test_compare_reads.py:219-233 -> [21, 21, 2]; tables with only 8 Q rows
    seq = np.frombuffer(b'ATG' + bytes(13), dtype=np.uint8).reshape(1, 16).copy()
    qual = np.frombuffer(b'((#' + bytes(13), dtype=np.uint8).reshape(1, 16).copy()
    meta = np.array([3], dtype=np.uint32)
    posdq = np.zeros((1, 8, 6), dtype=np.int64); posdq[0, 7, :] = 3
    ddq = np.zeros((1, 8, 16), dtype=np.int64); ddq[0, 7, :] = 5
    out = oracle.apply(seq, qual, meta, np.array([10]), np.array([1]),
                       np.array([[2] * 8]), posdq, ddq)
    assert list(out[0, :3]) == [21, 21, 2]


def test_get_delta_qs_known_answer(oracle):
    # tests/test_gatk_applybqsr.py:105-121
    a = oracle.get_delta_qs(np.array([10]), np.array([0]), np.array([1000]),
                            np.array([[0]]), np.array([[1000]]), np.array([[[0]]]),
                            np.array([[[1000]]]), np.array([[[0]]]), np.array([[[1000]]]))
    assert np.array_equal(a[0], np.array([3]))
    assert np.array_equal(a[1], np.array([[2]]))
    assert np.array_equal(a[2], np.array([[[1]]]))
    assert np.array_equal(a[3], np.array([[[1, 0]]]))


def test_reference_error_behaviour(oracle, tmp_path):
    # H2: a read shorter than the running maximum -> IndexError (recalibrate.py:89-101)
    u = _write(tmp_path, 'u.fq', [('a', 'ACGT', 'IIII'), ('b', 'ACG', 'III')])
    b = oracle.pack_records(oracle.read_fastq(u), oracle.read_fastq(u), False)
    with pytest.raises(IndexError):
        oracle.accumulate(b['seq'], b['cseq'], b['qual'], b['meta'], b['R'], b['S'])
    # H7: q > 42 -> IndexError ('L' = 43); lower-case base in a looked-up dinuc -> TypeError
    u = _write(tmp_path, 'u2.fq', [('a', 'ACGT', 'IILI')])
    b = oracle.pack_records(oracle.read_fastq(u), oracle.read_fastq(u), False)
    with pytest.raises(IndexError):
        oracle.accumulate(b['seq'], b['cseq'], b['qual'], b['meta'], b['R'], b['S'])
    u = _write(tmp_path, 'u3.fq', [('a', 'AcGT', 'IIII')])
    b = oracle.pack_records(oracle.read_fastq(u), oracle.read_fastq(u), False)
    with pytest.raises(TypeError):
        oracle.accumulate(b['seq'], b['cseq'], b['qual'], b['meta'], b['R'], b['S'])
    # name mismatch -> AssertionError (recalibrate.py:17)
    with pytest.raises(AssertionError):
        oracle.pack_records([('x1', 'A', 'I')], [('y1', 'A', 'I')], False)


def test_h1_negative_cycle_aliasing(oracle):
    # H1: short /2 read then a longer /1 read -- the /2 counts keep their absolute
    # column (2*len_t - 1 - i), i.e. they alias with forward cycles after growth.
    unc = [('a/2', 'AC', 'II'), ('b/1', 'ACGT', 'IIII')]
    got = oracle.py_accumulate(unc, unc)
    b = oracle.pack_records(unc, unc, False)
    got_c = oracle.accumulate(b['seq'], b['cseq'], b['qual'], b['meta'], b['R'], b['S'])
    for x, y in zip(got, got_c):
        assert np.array_equal(x, y)
    pt = got[6][0, 40]
    assert pt.shape == (8,) and list(pt) == [1, 1, 2, 2, 0, 0, 0, 0]
