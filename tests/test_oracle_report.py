"""
GATK-report model codec, CPU only.

  * the oracle (oracle/oracle_report.py) against reports written by the UNMODIFIED reference
    (tests/golden/report_*.json|txt) and against the reference's own known answers
    (tests/test_recaltable.py:75-83,97-141; tests/test_gatk_applybqsr.py:13-63), restated as data;
  * the product's host-only pieces -- kbbq.recaltable (parse / print) and
    kbbq.gatk.applybqsr.table_to_vectors -- against the same.
The product's vectors_to_report calls the device solve and is tested in test_gpu_report.py.
"""
import json
import os

import numpy as np
import pandas as pd
import pytest

from conftest import GOLD, GOLDEN_CASES, load_golden

VEC = ['meanq', 'rg_errs', 'rg_total', 'q_errs', 'q_total', 'pos_errs', 'pos_total',
       'dinuc_errs', 'dinuc_total']

# the reference's hand-written fixtures (data): tests/test_recaltable.py:75-83
EXTABLE = '''#:GATKTable:6:2:%s:%s:%.4f:%.4f:%d:%.2f:;
#:GATKTable:RecalTable0:
ReadGroup                   EventType  EmpiricalQuality  EstimatedQReported  Observations  Errors
HJCMTCCXX160113.5.AAGGATGT  M                   22.0000             24.3199        210398  1382.00
HK2WYCCXX160124.1.AAGGATGT  M                   22.0000             24.3994        196298  1391.00'''
EXTABLE = EXTABLE.replace('Errors\n', 'Errors \n')       # the header cell is padded to the 7-wide data

# tests/test_gatk_applybqsr.py:13-41 (also tests/conftest.py:171-202)
SMALL_REPORT = '''#:GATKReport.v1.1:5
#:GATKTable:2:17:%s:%s:;
#:GATKTable:Arguments:Recalibration argument collection values used in this run
Argument                    Value

#:GATKTable:3:94:%d:%d:%d:;
#:GATKTable:Quantized:Quality quantization map
QualityScore  Count    QuantizedScore

#:GATKTable:6:1:%s:%s:%.4f:%.4f:%d:%.2f:;
#:GATKTable:RecalTable0:
ReadGroup  EventType  EmpiricalQuality  EstimatedQReported  Observations  Errors
1          M                   23.0000              7.0000        200000  1000.00

#:GATKTable:6:1:%s:%d:%s:%.4f:%d:%.2f:;
#:GATKTable:RecalTable1:
ReadGroup  QualityScore  EventType  EmpiricalQuality  Observations  Errors
1                     7  M                   23.0000        200000  1000.00

#:GATKTable:8:50763:%s:%d:%s:%s:%s:%.4f:%d:%.2f:;
#:GATKTable:RecalTable2:
ReadGroup  QualityScore  CovariateValue  CovariateName  EventType  EmpiricalQuality  Observations  Errors
1                     7  1               Cycle          M                   23.0000        200000  1000.00
1                     7  AC              Context        M                   23.0000        200000  1000.00

'''


def small_report_answers():
    """tests/test_gatk_applybqsr.py:44-63"""
    z = lambda *s: np.zeros(s, dtype=np.int64)
    qe, qt = z(1, 43), z(1, 43); qe[0, 7], qt[0, 7] = 1000, 200000
    pe, pt = z(1, 43, 2), z(1, 43, 2); pe[0, 7, 0], pt[0, 7, 0] = 1000, 200000
    de, dt = z(1, 43, 16), z(1, 43, 16); de[0, 7, 3], dt[0, 7, 3] = 1000, 200000
    return [np.array([7.0]), np.array([1000]), np.array([200000]), qe, qt, pe, pt, de, dt]


def report_info(name):
    with open(os.path.join(GOLD, 'report_' + name + '.json')) as fh:
        return json.load(fh)


def oracle_text(name):
    import oracle_report as OR
    _, gold = load_golden(name)
    info = report_info(name)
    return OR.report_text(*[gold[k] for k in VEC], info['rg_order']), gold, info


# ---------------------------------------------------------------- the oracle
@pytest.mark.parametrize('name', GOLDEN_CASES)
def test_oracle_report_text_matches_reference(oracle, name):
    import oracle as O
    text, gold, info = oracle_text(name)
    lines = text.split('\n')
    assert [ln for ln in lines if ln.startswith('#:GATKTable:')] == info['table_heads']
    assert '\n'.join(lines[:40]) == info['head']
    for i, want in info['sampled'].items():
        assert lines[int(i)] == want, i
    assert len(text) == info['length'] and O.sha256(text) == info['sha256']
    assert info['reference_reread_is_identical']


def test_oracle_full_text_fixture(oracle):
    text, _, _ = oracle_text('short_64_1rg')
    assert text == open(os.path.join(GOLD, 'report_short_64_1rg.txt')).read()


@pytest.mark.parametrize('name', GOLDEN_CASES)
def test_oracle_round_trip(oracle, name):
    import oracle_report as OR
    text, gold, info = oracle_text(name)
    back = OR.table_to_vectors(text, info['rg_order'])
    S2 = back[5].shape[2]
    for k, b in zip(VEC[1:], back[1:]):
        g = gold[k]
        if g.ndim == 3 and g.shape[2] != 16 and g.shape[2] != S2:
            # the cycle axis shrinks to the largest cycle the report lists: 1..n | -n..-1
            n, full = S2 // 2, g.shape[2] // 2
            assert not g[:, :, n:2 * full - n].any()
            g = np.concatenate([g[:, :, :n], g[:, :, 2 * full - n:]], axis=2)
        assert np.array_equal(b, g), k
    # EstimatedQReported is the 4-decimal print of -10 * round(log10(mean error probability), 5)
    with np.errstate(all='ignore'):
        import oracle as O
        est = -10.0 * np.log10(np.sum(O.q_to_p(np.arange(43)) * gold['q_total'], axis=1) / gold['rg_total']).round(5).astype(float)
    assert np.allclose(back[0], est, atol=5.1e-5)


def test_oracle_known_answers_of_the_reference(oracle):
    import oracle_report as OR
    for got, want in zip(OR.table_to_vectors(SMALL_REPORT, ['1']), small_report_answers()):
        assert np.array_equal(got, want)
    title, desc, header, fmts, rows = OR.parse_report('#:GATKReport.v1.1:1\n' + EXTABLE + '\n\n')[0]
    assert (title, desc, len(rows), len(header)) == ('RecalTable0', '', 2, 6)
    typed = [(r[0], r[1], float(r[2]), float(r[3]), int(r[4]), float(r[5])) for r in rows]
    assert OR.render_table(title, desc, header, fmts, typed) == EXTABLE


# ---------------------------------------------------------------- the product's host side
def test_product_table_known_answers():
    from kbbq import recaltable
    t = recaltable.GATKTable.fromstring(EXTABLE)
    assert (t.title, t.description, t.data.shape) == ('RecalTable0', '', (2, 6))
    assert t.get_fmtstring() == '#:GATKTable:6:2:%s:%s:%.4f:%.4f:%d:%.2f:;'
    assert t.get_colfmts() == ['%s', '%s', '%.4f', '%.4f', '%d', '%.2f']
    assert t.get_titlestring() == '#:GATKTable:RecalTable0:'
    assert t.get_datastring() == '\n'.join(EXTABLE.splitlines()[2:])
    assert (t.get_nrows(), t.get_ncols()) == (2, 6)
    assert str(t) == EXTABLE
    empty_str = '\n'.join(EXTABLE.splitlines()[:2] + ['  '.join(EXTABLE.splitlines()[2].split())])
    empty = recaltable.GATKTable.fromstring(empty_str)
    assert empty.get_datastring() == empty_str.splitlines()[2]
    assert recaltable.GATKTable.parse_fmtstring(['foo', 'bar', 'baz', 'test'], '#:GATKTable:0:4:%d:%.4f:%s:%x:;') \
        == {'foo': np.int64, 'bar': np.float64, 'baz': str}
    assert recaltable.GATKTable('foo', 'bar', '').get_titlestring() == '#:GATKTable:foo:bar'
    full = recaltable.GATKTable('foo', 'bar', pd.DataFrame({'spam': ['eggs']}))
    assert repr(full) == '#:GATKTable:1:1:%s:;\n#:GATKTable:foo:bar\n   spam\n0  eggs'
    assert full == full and t == t and t != full and t != 5
    assert full != recaltable.GATKTable('', 'bar', pd.DataFrame({'spam': ['eggs']}))
    assert full != recaltable.GATKTable('foo', 'bar', pd.DataFrame())


def test_product_report_known_answers(tmp_path):
    from kbbq import recaltable
    r = recaltable.GATKReport([])
    assert r.tables == [] and r.get_headerstring() == '#:GATKReport.v1.1:0'
    assert repr(r) == '#:GATKReport.v1.1:0\n\n'
    r.tables = [0, 1, 2]
    assert r.get_headerstring() == '#:GATKReport.v1.1:3' and repr(r) == '#:GATKReport.v1.1:3\n0\n1\n2\n'
    assert r == recaltable.GATKReport([0, 1, 2]) and not r == recaltable.GATKReport([4, 5, 6])
    assert not r == recaltable.GATKReport([]) and not r == 3
    assert not recaltable.GATKReport([]) == recaltable.GATKReport([], version='0.0')
    with pytest.raises(ValueError):
        recaltable.RecalibrationReport([])


def test_product_reads_and_prints_reference_reports(tmp_path):
    from kbbq import recaltable
    path = os.path.join(GOLD, 'report_short_64_1rg.txt')
    text = open(path).read()
    rep = recaltable.RecalibrationReport.fromfile(path)
    assert [t.title for t in rep.tables] == ['Arguments', 'Quantized', 'RecalTable0', 'RecalTable1', 'RecalTable2']
    assert rep.tables[0].data.index.names == ['Argument']
    assert rep.tables[1].data.index.names == ['QualityScore']
    assert rep.tables[2].data.index.names == ['ReadGroup']
    assert rep.tables[3].data.index.names == ['ReadGroup', 'QualityScore']
    assert rep.tables[4].data.index.names == ['ReadGroup', 'QualityScore', 'CovariateName', 'CovariateValue']
    assert str(rep) == text
    assert rep.tables[4].data.index.names == ['ReadGroup', 'QualityScore', 'CovariateName', 'CovariateValue']
    out = tmp_path / 'again.txt'
    rep.write(str(out))
    assert out.read_text() == text
    assert recaltable.GATKReport.fromfile(path) == recaltable.GATKReport.fromfile(str(out))
    # a truncated file: the header still announces 5 tables (tests/test_recaltable.py:24-30)
    cut = tmp_path / 'cut.txt'
    cut.write_text(''.join(text.splitlines(keepends=True)[:130]))
    with pytest.raises(ValueError):
        recaltable.GATKReport.fromfile(str(cut))


@pytest.mark.parametrize('name', GOLDEN_CASES)
def test_product_parse_print_and_vectors_on_big_reports(oracle, name, tmp_path):
    """Reference-identical text (the oracle's, SHA-checked above) through the product's parser,
    printer and table_to_vectors."""
    import oracle_report as OR
    from kbbq import recaltable
    from kbbq.gatk import applybqsr
    text, gold, info = oracle_text(name)
    path = tmp_path / 'r.txt'
    path.write_text(text)
    rep = recaltable.RecalibrationReport.fromfile(str(path))
    assert str(rep) == text
    got = applybqsr.table_to_vectors(rep, info['rg_order'])
    want = OR.table_to_vectors(text, info['rg_order'])
    for k, g, w in zip(VEC, got, want):
        assert g.shape == w.shape and np.array_equal(g, w), k
        assert g.dtype == (np.float64 if k == 'meanq' else np.int64), k
    # a subset / another order of read groups
    if len(info['rg_order']) > 1:
        sub = info['rg_order'][::-1][:2]
        for g, w in zip(applybqsr.table_to_vectors(rep, sub), OR.table_to_vectors(text, sub)):
            assert np.array_equal(g, w)


def test_product_table_to_vectors_known_answers(tmp_path):
    from kbbq import recaltable
    from kbbq.gatk import applybqsr
    path = tmp_path / 'small.txt'
    path.write_text(SMALL_REPORT)
    rep = recaltable.RecalibrationReport.fromfile(str(path))
    got = applybqsr.table_to_vectors(rep, ['1'])
    for g, w in zip(got, small_report_answers()):
        assert np.array_equal(g, w)
    assert got[0].dtype == np.float64 and got[1].dtype == np.int64
    with pytest.raises(ValueError):
        applybqsr.table_to_vectors(rep, ['nope'])
