"""
CPU tests of the device solve's building blocks (no kernel launched):
  * kbbq._solve's decomposition of scipy.stats.binom.logpmf is bit-identical to logpmf;
  * csrc/solve_core.h compiled for the host (tests/native/solve_check.cpp) returns the
    reference's argmax on the golden gatk_delta_q grid and on adversarial cells.
"""
import os
import struct
import subprocess

import numpy as np
import pytest
import scipy.stats

from conftest import ROOT, load_golden
from kbbq import _solve


def _cells(rng, n):
    tot = (10 ** rng.uniform(0, 10.5, n)).astype(np.int64)
    err = (tot * 10 ** (-rng.uniform(0, 5, n))).astype(np.int64)
    err = np.minimum(err, tot)
    tot[:50] = rng.integers(0, 5, 50); err[:50] = rng.integers(0, 3, 50); err[:50] = np.minimum(err[:50], tot[:50])
    err[50:80] = tot[50:80]; err[80:110] = 0
    return err, tot


def test_native_gammaln_is_scipys_bit_for_bit():
    """csrc/solve_host.cpp restates SciPy 1.15's xsf::cephes::lgam; every positive argument must give the
    very float64 scipy.special.gammaln gives (same libm log, same operation order)."""
    from kbbq import _native as N
    lib = N.load()
    rng = np.random.default_rng(5)
    for x in (np.arange(1, 300001, dtype=np.float64), np.floor(10 ** rng.uniform(0, 10.5, 2_000_000)),
              rng.uniform(1e-3, 2e4, 500_000), 10 ** rng.uniform(-3, 300, 200_000),
              np.array([1, 2, 3, 12, 13, 14, 999, 1000, 1001, 1e8, 1e8 + 1, 2.0 ** 53, 1e300, 2.6e305, 1e308])):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty_like(x)
        N.check(lib.kbbq_gammaln_host(N.ptr(x), x.size, N.ptr(out)))
        with np.errstate(all='ignore'):
            want = scipy.special.gammaln(x)
        assert np.array_equal(out.view(np.int64), want.view(np.int64))
    err, tot = _cells(rng, 200_000)
    for threads in (1, 3, 8):
        _solve.COMBILN_THREADS = threads
        got = _solve.combiln(err, tot)
        assert np.array_equal(got.view(np.int64), _solve.combiln_scipy(err, tot).view(np.int64))
    _solve.COMBILN_THREADS = None
    # outside the support: NaN (never read by the solve), shapes are kept
    bad = _solve.combiln(np.array([[5, -3]]), np.array([[2, 7]]))
    assert bad.shape == (1, 2) and np.isnan(bad[0, 1])


def test_logpmf_decomposition_is_bit_identical():
    rng = np.random.default_rng(1)
    err, tot = _cells(rng, 4000)
    consts = _solve.model_consts()
    prior, logp, log1mp = consts[:43], consts[43:86], consts[86:]
    comb = _solve.combiln(err, tot)
    p = (10.0 ** (-(np.arange(43) / 10.0)))
    k = (err + 1).astype(np.float64); nk = (tot + 2).astype(np.float64) - k
    with np.errstate(all='ignore'):
        for cand in range(43):
            want = scipy.stats.binom.logpmf(err + 1, tot + 2, p[cand])
            got = (comb + k * logp[cand]) + nk * log1mp[cand]
            assert np.array_equal(want, got), cand
    assert np.isneginf(prior[19:]).all() and np.isfinite(prior[:19]).all()


@pytest.fixture(scope='module')
def solver(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp('native') / 'solve_check')
    subprocess.check_call(['g++', '-O1', '-std=c++17', '-ffp-contract=off', '-fsanitize=undefined', '-fno-sanitize-recover=undefined', '-o', exe,
                           os.path.join(ROOT, 'tests', 'native', 'solve_check.cpp')])

    def run(prior_q, errs, total):
        is_float = np.asarray(prior_q).dtype.kind == 'f'
        prior_q = np.ascontiguousarray(prior_q, dtype=np.float64 if is_float else np.int64).ravel()
        errs = np.ascontiguousarray(errs, dtype=np.int64).ravel()
        total = np.ascontiguousarray(total, dtype=np.int64).ravel()
        comb = _solve.combiln(errs, total)
        rec = np.empty(len(errs), dtype=[('p', '<i8'), ('e', '<i8'), ('t', '<i8'), ('c', '<f8')])
        rec['p'], rec['e'], rec['t'], rec['c'] = prior_q.view(np.int64), errs, total, comb
        blob = struct.pack('<q', len(errs)) + _solve.model_consts().tobytes() + rec.tobytes()
        out = subprocess.run([exe] + (['f'] if is_float else []), input=blob, capture_output=True, timeout=600)
        assert out.returncode == 0
        return np.frombuffer(out.stdout, dtype=np.int64) - prior_q
    return run


@pytest.mark.skipif(np.finfo(np.longdouble).nmant != 63, reason='np.longdouble is not x87 extended here')
def test_solve_core_matches_reference_grid(solver, oracle):
    _, gold = load_golden('numeric')
    errs, tot = gold['grid_errs'], gold['grid_total']
    prior_q = np.broadcast_to(np.arange(43)[:, None], (43, len(errs))).copy()
    be = np.broadcast_to(errs, prior_q.shape).copy(); bt = np.broadcast_to(tot, prior_q.shape).copy()
    assert np.array_equal(solver(prior_q, be, bt).reshape(prior_q.shape), gold['grid_dq'])
    # adversarial cells against the oracle (which makes the reference's SciPy/longdouble calls)
    rng = np.random.default_rng(7)
    err, t = _cells(rng, 6000)
    pq = rng.integers(0, 43, len(err))
    assert np.array_equal(solver(pq, err, t), oracle.gatk_delta_q(pq, err, t))
    # real tables
    for name in ('c1_10k_1rg', 'c5cut_2k_mixed'):
        _, g = load_golden(name)
        prior2 = (g['meanq'] + g['rgdq'])[:, None] + g['qdq']
        pp = np.broadcast_to(prior2[..., None], g['pos_total'].shape)
        assert np.array_equal(solver(pp, g['pos_errs'], g['pos_total']).reshape(pp.shape), g['posdq'])
        dd = np.broadcast_to(prior2[..., None], g['dinuc_total'].shape)
        assert np.array_equal(solver(dd, g['dinuc_errs'], g['dinuc_total']).reshape(dd.shape),
                              g['dinucdq'][..., :16])


@pytest.mark.skipif(np.finfo(np.longdouble).nmant != 63, reason='np.longdouble is not x87 extended here')
def test_solve_core_float_prior(solver, oracle):
    """The report builder calls gatk_delta_q with a float64 prior (reference gatk/bqsr.py:294):
    distance = float64 difference truncated toward zero."""
    rng = np.random.default_rng(11)
    err, t = _cells(rng, 4000)
    pq = rng.uniform(-0.999, 42.999, len(err))
    pq[:200] = np.round(pq[:200])                       # integral floats
    pq[200:400] = np.round(pq[200:400]) + rng.choice([-1e-12, 1e-12, 2e-15, -2e-15], 200)
    pq = np.clip(pq, -0.999, 42.999)
    got = solver(pq, err, t)
    want = oracle.gatk_delta_q(pq, err, t)
    assert got.dtype == np.float64 and np.array_equal(got, want)


def test_fused_solve_prep_matches_the_numpy_route():
    """kbbq_solve_prep_host (marginals + the gammaln term of every cell, one threaded pass over the flat table buffer)
    against vectors_from_tables + combiln on random tables, several read-group counts and thread counts."""
    rng = np.random.default_rng(9)
    for R, S2 in ((1, 300), (3, 64), (8, 302), (2, 2)):
        npos, ndn = R * 43 * S2, R * 43 * 16
        tot = rng.integers(0, 5_000_000, 2 * npos + 2 * ndn).astype(np.int64)
        flat = tot.copy()
        flat[:npos] = (tot[npos:2 * npos] * rng.random(npos) * 0.2).astype(np.int64)                  # errs <= total
        flat[2 * npos:2 * npos + ndn] = (tot[2 * npos + ndn:] * rng.random(ndn) * 0.2).astype(np.int64)
        flat[rng.random(flat.size) < 0.1] = 0
        p_e, p_t = flat[:npos].reshape(R, 43, S2), flat[npos:2 * npos].reshape(R, 43, S2)
        d_e, d_t = flat[2 * npos:2 * npos + ndn].reshape(R, 43, 16), flat[2 * npos + ndn:].reshape(R, 43, 16)
        p_e = np.minimum(p_e, p_t); d_e = np.minimum(d_e, d_t)
        flat = np.concatenate([p_e.ravel(), p_t.ravel(), d_e.ravel(), d_t.ravel()])
        want = _solve.vectors_from_tables(p_e, p_t, d_e, d_t)
        want_aux = np.concatenate([_solve.combiln(e, t).ravel() for e, t in ((want[1], want[2]), (want[3], want[4]), (p_e, p_t), (d_e, d_t))])
        for threads in (1, 5, None):
            _solve.COMBILN_THREADS = threads
            aux, q_e, q_t, rg_e, rg_t = _solve.solve_prep(flat, R, S2)
            assert np.array_equal(aux.view(np.int64), want_aux.view(np.int64)), (R, S2, threads)
            for got, w in ((rg_e, want[1]), (rg_t, want[2]), (q_e, want[3]), (q_t, want[4])):
                assert np.array_equal(got, w)
            assert np.array_equal(_solve.meanq_from_q_total(q_t), want[0])
    _solve.COMBILN_THREADS = None


def test_log_tables_without_scipy_equal_scipys():
    """_solve.model_consts() (xlogy / xlog1py by the C library's log / log1p, without importing SciPy) against
    the SciPy calls themselves -- the 129 model constants, and the two functions on a wider sample of probabilities."""
    from kbbq import _native as N, _solve
    import scipy.special
    got, want = _solve.model_consts(), _solve.model_consts_scipy()
    assert np.array_equal(got.view(np.uint64), want.view(np.uint64))
    rng = np.random.default_rng(11)
    p = np.concatenate([10.0 ** -rng.uniform(0, 12, 200000), rng.uniform(0, 1, 200000), [0.0, 1.0, 0.5, 1 - 2.0 ** -53, 0.2928932188134524, 0.29289321881345254]])
    logp, log1mp = np.empty_like(p), np.empty_like(p)
    N.check(N.load().kbbq_xlogy_tables_host(N.ptr(p), p.size, N.ptr(logp), N.ptr(log1mp)))
    with np.errstate(divide='ignore'):
        assert np.array_equal(logp.view(np.uint64), scipy.special.xlogy(1.0, p).view(np.uint64))
        assert np.array_equal(log1mp.view(np.uint64), scipy.special.xlog1py(1.0, -p).view(np.uint64))
