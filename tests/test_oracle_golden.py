"""
The oracle against golden vectors produced by the UNMODIFIED reference
(oracle/gen_golden.py, run in the build container).  CPU only.
"""
import numpy as np
import pytest

from conftest import GOLDEN_CASES, load_golden

VEC = ['meanq', 'rg_errs', 'rg_total', 'q_errs', 'q_total', 'pos_errs', 'pos_total',
       'dinuc_errs', 'dinuc_total']
DQ = ['rgdq', 'qdq', 'posdq', 'dinucdq']


def regenerate(oracle, info, tmp_path):
    c = info['case']
    seq, cseq, qual, meta = oracle.synth(0, c['n'], c['n'], c['seed'], c['len_lo'], c['len_hi'],
                                         c['nrg'], c['qlo'], c['qhi'])
    names = oracle.synth_names(0, c['n'], c['nrg'], with_rg=c['infer_rg'])
    fa, fb = str(tmp_path / 'a.fq'), str(tmp_path / 'b.fq')
    oracle.write_fastq(fa, names, seq, qual, meta)
    oracle.write_fastq(fb, names, cseq, qual, meta)
    sha = [oracle.sha256(open(f, 'rb').read()) for f in (fa, fb)]
    assert sha == info['input_sha256'], 'synthetic input drifted from what the reference saw'
    return fa, fb, (seq, cseq, qual, meta)


@pytest.mark.parametrize('name', GOLDEN_CASES)
def test_oracle_matches_reference(oracle, name, tmp_path):
    info, gold = load_golden(name)
    fa, fb, planes = regenerate(oracle, info, tmp_path)
    text, vectors, dqs = oracle.recalibrate_fastq_text([fa, fb], info['case']['infer_rg'])
    for k, v in zip(VEC, vectors):
        assert v.dtype == np.int64 and np.array_equal(v, gold[k]), k
    for k, v in zip(DQ, dqs):
        assert np.array_equal(v, gold[k]), k
    assert len(text) == info['output_len']
    assert oracle.sha256(text) == info['output_sha256']
    assert text.startswith(info['first_records'])
    assert text.endswith(info['last_records'])
    # packed planes straight from synth() give the same tables as the FASTQ text route
    seq, cseq, qual, meta = planes
    c = info['case']
    v2 = oracle.accumulate(seq, cseq, qual, meta, c['nrg'], c['len_hi'])
    for k, v in zip(VEC, v2):
        assert np.array_equal(v, gold[k]), k


@pytest.mark.parametrize('name', ['short_64_1rg', 'q42_500_3rg'])
def test_python_twin_matches_reference(oracle, name, tmp_path):
    info, gold = load_golden(name)
    fa, fb, _ = regenerate(oracle, info, tmp_path)
    got = oracle.py_accumulate(oracle.read_fastq(fa), oracle.read_fastq(fb), info['case']['infer_rg'])
    for k, v in zip(VEC, got):
        assert np.array_equal(v, gold[k]), k


def test_numeric_tables(oracle):
    info, gold = load_golden('numeric')
    prior = [float(x).hex() if np.isfinite(x) else '-inf' for x in oracle.PRIOR_DIST]
    assert prior == info['prior_dist_hex']
    assert info['prior_dist_is_float64_exact']
    q = np.arange(43)
    assert [float(x).hex() for x in oracle.q_to_p(q)] == info['q_to_p_hex']
    assert [int(x) for x in oracle.p_to_q(oracle.q_to_p(q))] == info['p_to_q_of_q_to_p']
    s = info['p_to_q_samples']
    assert [int(x) for x in oracle.p_to_q(np.array(s['p']))] == s['q']
    errs, tot = gold['grid_errs'], gold['grid_total']
    prior_q = np.broadcast_to(np.arange(43)[:, None], (43, len(errs))).copy()
    dq = oracle.gatk_delta_q(prior_q, np.broadcast_to(errs, prior_q.shape).copy(),
                             np.broadcast_to(tot, prior_q.shape).copy())
    assert np.array_equal(dq, gold['grid_dq'])
    # SURVEY A7 [probed]: empty cell -> 0 except prior 0 -> +1
    empty = gold['grid_dq'][:, 0]
    assert empty[0] == 1 and np.all(empty[1:] == 0)


def test_synth_thresholds(oracle):
    from decimal import Decimal, getcontext
    getcontext().prec = 60
    want = [min(int((Decimal(2) ** 32) * (Decimal(10) ** (Decimal(-q) / Decimal(10)))), 2 ** 32 - 1)
            for q in range(43)]
    assert [int(x) for x in oracle.SYNTH_THR] == want
