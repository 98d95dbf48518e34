"""
GPU tests (-m gpu) of the model file path: count tables -> GATK report
(EmpiricalQuality columns from the device solve, including the float64-prior call) and back,
and `recalibrate -g`.  Expected values: reports written by the UNMODIFIED reference
(tests/golden/report_*.json|txt) and the oracle.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLD, GOLDEN_CASES, load_golden
from test_gpu_parity import _files, dev, VEC            # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _info(name):
    with open(os.path.join(GOLD, 'report_' + name + '.json')) as fh:
        return json.load(fh)


def test_gatk_delta_q_float_prior_matches_oracle(dev, oracle):
    from kbbq import compare_reads as utils
    rng = np.random.default_rng(21)
    n = 20000
    tot = (10 ** rng.uniform(0, 10.5, n)).astype(np.int64)
    err = np.minimum((tot * 10 ** (-rng.uniform(0, 5, n))).astype(np.int64), tot)
    pq = rng.uniform(-0.999, 42.999, n)
    pq[:500] = np.round(pq[:500])
    pq[500:1000] = np.round(pq[500:1000]) + rng.choice([-1e-12, 1e-12, 2e-15, -2e-15], 500)
    pq = np.clip(pq, -0.999, 42.999)
    got = utils.gatk_delta_q(pq, err, tot)
    want = oracle.gatk_delta_q(pq, err, tot)
    assert got.dtype == np.float64 and np.array_equal(got, want)
    # 2-d shapes and integer priors keep working
    assert np.array_equal(utils.gatk_delta_q(pq.reshape(100, 200), err.reshape(100, 200), tot.reshape(100, 200)),
                          want.reshape(100, 200))
    for bad in (-1.0, 43.0, 43.5, -1.5, np.nan):
        with pytest.raises(IndexError):
            utils.gatk_delta_q(np.array([bad]), np.array([1]), np.array([10]))
    assert utils.gatk_delta_q(np.array([-0.5, 42.9]), np.array([1, 1]), np.array([10, 10])).shape == (2,)


@pytest.mark.parametrize('name', GOLDEN_CASES)
def test_vectors_to_report_matches_reference(dev, oracle, name, tmp_path):
    from kbbq import recaltable
    from kbbq.gatk import applybqsr, bqsr
    _, gold = load_golden(name)
    info = _info(name)
    rep = bqsr.vectors_to_report(*[gold[k] for k in VEC], info['rg_order'])
    text = str(rep)
    lines = text.split('\n')
    assert [ln for ln in lines if ln.startswith('#:GATKTable:')] == info['table_heads']
    assert '\n'.join(lines[:40]) == info['head']
    for i, want in info['sampled'].items():
        assert lines[int(i)] == want, i
    assert len(text) == info['length'] and oracle.sha256(text) == info['sha256']
    # write -> read -> print is the identity, and the counts come back
    path = str(tmp_path / 'r.txt')
    rep.write(path)
    again = recaltable.RecalibrationReport.fromfile(path)
    assert str(again) == text
    back = applybqsr.table_to_vectors(again, info['rg_order'])
    for k, b in zip(VEC[1:5], back[1:5]):
        assert np.array_equal(b, gold[k]), k
    assert np.array_equal(back[7], gold['dinuc_errs']) and np.array_equal(back[8], gold['dinuc_total'])
    assert back[6].sum() == gold['pos_total'].sum() and back[5].sum() == gold['pos_errs'].sum()


def test_quantize_and_cycle_labels():
    from kbbq.gatk import bqsr
    qt = np.zeros((2, 43), dtype=np.int64); qt[0, 7] = 3; qt[1, 30] = 1
    q = bqsr.quantize(qt, qt)
    assert q.shape == (94,) and q[7] == 7 and q[30] == 30 and (np.delete(q, [7, 30]) == 93).all()
    assert bqsr._cycle_labels(6).tolist() == [1, 2, 3, -3, -2, -1]


@pytest.mark.parametrize('name', ['c1_10k_1rg', 'c3cut_2k_8rg', 'short_64_1rg'])
def test_recalibrate_with_model_file(dev, oracle, name, tmp_path, capfd):
    """-g: first run saves the model (file = the reference's report for these vectors), second
    run loads it instead of pass 1 (the corrected FASTQ is not even opened) and prints the same
    FASTQ, which is the reference's output."""
    from kbbq import recalibrate
    info, _ = load_golden(name)
    rinfo = _info(name)
    fa, fb = _files(oracle, info, tmp_path)
    model = str(tmp_path / 'model.txt')
    capfd.readouterr()
    recalibrate.recalibrate(None, [fa, fb], infer_rg=info['case']['infer_rg'], gatkreport=model)
    first = capfd.readouterr().out
    assert oracle.sha256(first) == info['output_sha256']
    assert oracle.sha256(open(model).read()) == rinfo['sha256']
    recalibrate.recalibrate(None, [fa, str(tmp_path / 'does_not_exist.fq')],
                            infer_rg=info['case']['infer_rg'], gatkreport=model)
    second = capfd.readouterr().out
    assert second == first


def test_cli_gatkreport(dev, oracle, tmp_path, capfd, monkeypatch):
    from kbbq import main
    info, _ = load_golden('short_64_1rg')
    fa, fb = _files(oracle, info, tmp_path)
    model = str(tmp_path / 'm.txt')
    capfd.readouterr()
    monkeypatch.setattr('sys.argv', ['kbbq', 'recalibrate', '-f', fa, fb, '-g', model])
    main.main()
    out1 = capfd.readouterr().out
    assert os.path.exists(model) and oracle.sha256(out1) == info['output_sha256']
    monkeypatch.setattr('sys.argv', ['kbbq', 'recalibrate', '-f', fa, fb, '--gatkreport', model])
    main.main()
    assert capfd.readouterr().out == out1
