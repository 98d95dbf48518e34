"""
GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the
CPU oracle on the same seeded inputs, against the committed golden vectors
produced by the unmodified reference, and -- at large sizes -- through
size-independent properties.  Integer / byte results: bit-exact.
"""
import ctypes
import io
import contextlib

import numpy as np
import pytest

from conftest import GOLDEN_CASES, load_golden

pytestmark = pytest.mark.gpu

VEC = ['meanq', 'rg_errs', 'rg_total', 'q_errs', 'q_total', 'pos_errs', 'pos_total',
       'dinuc_errs', 'dinuc_total']
DQ = ['rgdq', 'qdq', 'posdq', 'dinucdq']


@pytest.fixture(scope='module')
def dev():
    import torch
    assert torch.cuda.is_available(), 'these tests need the MI355X'
    from kbbq import _device
    ctx = _device.context()
    assert 'gfx950' in ctx.name
    return _device


def _files(oracle, info, tmp_path):
    c = info['case']
    seq, cseq, qual, meta = oracle.synth(0, c['n'], c['n'], c['seed'], c['len_lo'], c['len_hi'],
                                         c['nrg'], c['qlo'], c['qhi'])
    names = oracle.synth_names(0, c['n'], c['nrg'], with_rg=c['infer_rg'])
    fa, fb = str(tmp_path / 'a.fq'), str(tmp_path / 'b.fq')
    oracle.write_fastq(fa, names, seq, qual, meta)
    oracle.write_fastq(fb, names, cseq, qual, meta)
    assert [oracle.sha256(open(f, 'rb').read()) for f in (fa, fb)] == info['input_sha256']
    return fa, fb


# ------------------------------------------------------------------ generator
@pytest.mark.parametrize('shape', [
    dict(first=0, n=1000, total=1000, seed=1, len_lo=150, len_hi=150, nrg=1),
    dict(first=123, n=777, total=5000, seed=99, len_lo=36, len_hi=300, nrg=8),
    dict(first=0, n=65, total=65, seed=5, len_lo=1, len_hi=16, nrg=3, qlo=2, qhi=42),
])
def test_device_generator_matches_oracle(dev, oracle, shape):
    b = dev.ReadBatch.synthetic(**shape)
    want = oracle.synth(**shape)
    n = shape['n']
    for got, w in zip((b.seq, b.cseq, b.qual), want[:3]):
        assert np.array_equal(got[:n].cpu().numpy(), w)
    assert np.array_equal(b.meta[:n].cpu().numpy().view(np.uint32), want[3])


# ------------------------------------------------------------------ K1 / K2 vs oracle
def _tables_via_device(dev, seq, cseq, qual, meta, R, S, minscore=6):
    batch = dev.ReadBatch.from_host(seq, qual, meta, cseq=cseq)
    t = dev.Tables(R, 2 * S)
    dev.accumulate(batch, t, minscore)
    return t.to_host()


@pytest.mark.parametrize('shape', [
    dict(n=1, len_lo=150, len_hi=150, nrg=1),
    dict(n=63, len_lo=150, len_hi=150, nrg=1),
    dict(n=64, len_lo=150, len_hi=150, nrg=2),
    dict(n=65, len_lo=150, len_hi=150, nrg=3),
    dict(n=5000, len_lo=150, len_hi=150, nrg=1),
    dict(n=5000, len_lo=150, len_hi=150, nrg=8),
    dict(n=3000, len_lo=36, len_hi=300, nrg=5),
    dict(n=700, len_lo=1, len_hi=16, nrg=1),           # pitch 16: one chunk per read
    dict(n=900, len_lo=100, len_hi=100, nrg=2, qlo=2, qhi=42),
    dict(n=40000, len_lo=151, len_hi=151, nrg=4),
])
@pytest.mark.parametrize('minscore', [6, 0, 20])
def test_accumulate_and_apply_match_oracle(dev, oracle, shape, minscore):
    import torch
    if minscore != 6 and shape['n'] > 5000:
        pytest.skip('one minscore is enough for the larger shapes')
    s = dict(first=0, total=shape['n'], seed=7 + shape['n'], qlo=0, qhi=41)
    s.update(shape)
    seq, cseq, qual, meta = oracle.synth(**s)
    R, S = s['nrg'], s['len_hi']
    got = _tables_via_device(dev, seq, cseq, qual, meta, R, S, minscore)
    want = oracle.accumulate(seq, cseq, qual, meta, R, S, minscore=minscore)
    for g, w, k in zip(got, want[5:], VEC[5:]):
        assert g.dtype == np.int64 and np.array_equal(g, w), k
    assert got[1].sum() == want[2].sum() > 0

    # apply with a dense pseudo-random model so that every LUT cell matters
    rng = np.random.default_rng(s['seed'])
    meanq = rng.integers(10, 30, R); rgdq = rng.integers(-2, 3, R); qdq = rng.integers(-3, 4, (R, 43))
    posdq = rng.integers(-6, 7, (R, 43, 2 * S)); ddq = rng.integers(-6, 7, (R, 43, 17)); ddq[..., 16] = 0
    lut, shp = dev.build_lut(meanq, rgdq, qdq, posdq, ddq, minscore=minscore)
    batch = dev.ReadBatch.from_host(seq, qual, meta)
    out = dev.apply(batch, dev.lut_to_device(lut), shp, minscore=minscore)[:s['n']].cpu().numpy()
    ref = oracle.apply(seq, qual, meta, meanq, rgdq, qdq, posdq, ddq, minscore=minscore)
    lens = (meta & 0xFFFF).astype(np.int64)
    inside = np.arange(seq.shape[1])[None, :] < lens[:, None]
    assert np.array_equal(out[inside].astype(np.int32) - 33, ref[inside])
    assert not out[~inside].any()
    # the checked kernel (int16 LUT, per-base range test) gives the same bytes
    chk = dev.apply(batch, dev.lut_to_device(lut), shp[:3] + (0,), minscore=minscore)[:s['n']].cpu().numpy()
    assert np.array_equal(chk, out)


def _random_planes(rng, n, L, pitch, qvals, qprob, pn=0.02, perr=0.05, R=1, lens=None):
    """Hand-made reads (not the synthetic generator): binned qualities, N runs, any alphabet mix."""
    seq = np.full((n, pitch), ord('N'), dtype=np.uint8); cseq = seq.copy()
    qual = np.zeros((n, pitch), dtype=np.uint8)
    lens = np.full(n, L) if lens is None else lens
    meta = np.zeros(n, dtype=np.uint32)
    acgt = np.frombuffer(b'ACGT', dtype=np.uint8)
    for i in range(n):
        l = int(lens[i])
        b = acgt[rng.integers(0, 4, l)]
        b[rng.random(l) < pn] = ord('N')
        if rng.random() < 0.05:                      # a run of N
            a = rng.integers(0, l); b[a:a + rng.integers(1, 20)] = ord('N')
        c = b.copy()
        e = rng.random(l) < perr
        c[e] = acgt[rng.integers(0, 4, int(e.sum()))]
        seq[i, :l] = b; cseq[i, :l] = c
        qual[i, :l] = 33 + rng.choice(qvals, size=l, p=qprob)
        meta[i] = l | (int(rng.integers(0, R)) << 16) | ((i & 1) << 31)
    return seq, cseq, qual, meta


@pytest.mark.parametrize('case', ['binned', 'all_low', 'one_base', 'many_rgs', 'homopolymer'])
def test_handmade_reads_match_oracle(dev, oracle, case):
    rng = np.random.default_rng(hash(case) % 2 ** 32)
    if case == 'binned':          # modern Illumina: a handful of quality bins, one below minscore
        seq, cseq, qual, meta = _random_planes(rng, 3000, 151, 160, [2, 11, 25, 37], [.05, .1, .25, .6]); R, S = 1, 151
    elif case == 'all_low':       # most reads entirely below minscore (never counted), a few not
        seq, cseq, qual, meta = _random_planes(rng, 800, 100, 112, [2, 3, 5, 30], [.4, .3, .29, .01]); R, S = 1, 100
    elif case == 'one_base':      # reads of length 1 and 2 (non-decreasing), 16-byte rows
        lens = np.sort(rng.integers(1, 3, 500))
        seq, cseq, qual, meta = _random_planes(rng, 500, 2, 16, [7, 40], [.5, .5], lens=lens); R, S = 1, 2
    elif case == 'many_rgs':      # 32 read groups in random order
        seq, cseq, qual, meta = _random_planes(rng, 4000, 75, 80, np.arange(2, 42), np.full(40, 1 / 40), R=32); R, S = 32, 75
    else:                         # homopolymers with constant quality: every count in very few bins
        seq, cseq, qual, meta = _random_planes(rng, 2000, 150, 160, [40], [1.0], pn=0.0, perr=0.0)
        seq[:, :150] = ord('A'); cseq[:, :150] = ord('A'); cseq[::3, 10] = ord('T'); R, S = 1, 150
    got = _tables_via_device(dev, seq, cseq, qual, meta, R, S)
    want = oracle.accumulate(seq, cseq, qual, meta, R, S)
    for g, w, k in zip(got, want[5:], VEC[5:]):
        assert np.array_equal(g, w), (case, k)
    if want[2].min() > 0:          # every read group has counted bases: the model is defined
        t = dev.Tables(R, 2 * S)
        import torch
        t.buf.copy_(torch.from_numpy(np.concatenate([x.ravel() for x in want[5:]])))
        lut, shape, vectors, dqs = dev.solve(t, want_dq=True)
        wdq = oracle.get_delta_qs(*want)
        for g, w, k in zip(dqs, wdq, DQ):
            assert np.array_equal(g, w), (case, k)
        batch = dev.ReadBatch.from_host(seq, qual, meta)
        out = dev.apply(batch, lut, shape)[:seq.shape[0]].cpu().numpy()
        ref = oracle.apply(seq, qual, meta, want[0], *wdq)
        inside = np.arange(seq.shape[1])[None, :] < (meta & 0xFFFF)[:, None]
        assert np.array_equal(out[inside].astype(np.int32) - 33, ref[inside]), case


def test_accumulate_adds_into_tables_and_is_linear(dev, oracle):
    seq, cseq, qual, meta = oracle.synth(0, 4000, 4000, 21, nrg=2)
    whole = _tables_via_device(dev, seq, cseq, qual, meta, 2, 150)
    t = dev.Tables(2, 300)
    for lo, hi in ((0, 1500), (1500, 1501), (1501, 4000)):
        b = dev.ReadBatch.from_host(seq[lo:hi], qual[lo:hi], meta[lo:hi], cseq=cseq[lo:hi])
        dev.accumulate(b, t)
    for a, b in zip(whole, t.to_host()):
        assert np.array_equal(a, b)


def test_host_buffer_entry_points(dev, oracle):
    """kbbq_accumulate / kbbq_apply with plain host pointers (what a ctypes binding of the
    reference would call with NumPy arrays)."""
    from kbbq import _native as N
    seq, cseq, qual, meta = oracle.synth(0, 2000, 2000, 31, nrg=3)
    R, S2 = 3, 300
    tabs = [np.zeros((R, 43, S2), np.int64), np.zeros((R, 43, S2), np.int64),
            np.zeros((R, 43, 16), np.int64), np.zeros((R, 43, 16), np.int64)]
    ctx = dev.context()
    for _ in range(2):      # called twice: the entry point ADDS
        N.check(N.load().kbbq_accumulate(ctx.handle, N.ptr(seq), N.ptr(cseq), N.ptr(qual), N.ptr(meta),
                                         2000, seq.shape[1], R, S2, 6, *[N.ptr(t) for t in tabs]))
    want = oracle.accumulate(seq, cseq, qual, meta, R, 150)
    for g, w in zip(tabs, want[5:]):
        assert np.array_equal(g, 2 * w)
    dqs = oracle.get_delta_qs(*want)
    a = [np.ascontiguousarray(x, dtype=np.int64) for x in (want[0],) + dqs]
    out = np.zeros_like(qual)
    N.check(N.load().kbbq_apply(ctx.handle, N.ptr(seq), N.ptr(qual), N.ptr(meta), 2000, seq.shape[1],
                                R, 43, S2, 17, 6, *[N.ptr(x) for x in a], N.ptr(out)))
    ref = oracle.apply(seq, qual, meta, want[0], *dqs)
    inside = np.arange(seq.shape[1])[None, :] < (meta & 0xFFFF)[:, None]
    assert np.array_equal(out[inside].astype(np.int32) - 33, ref[inside])


def test_host_buffer_entry_points_run_slab_by_slab(dev, oracle, monkeypatch):
    """kbbq_accumulate / kbbq_apply move a caller's rows through page-locked slabs (upload, kernel and download of successive
    slabs overlapping): with slabs of a few thousand rows -- KBBQ_STAGE_MB -- the tables and qualities are those of the
    oracle, a second call re-uses the staging, and a read the kernels flag is reported with its index in the WHOLE input
    (every slab numbers its rows from 0) without anything reaching the caller's tables."""
    import re
    from kbbq import _native as N
    monkeypatch.setenv('KBBQ_STAGE_MB', '1')                       # 1 MB of staging per slab: ~2 100 rows of 160 bytes x 3 planes
    n, R, S2 = 30_001, 2, 300
    seq, cseq, qual, meta = oracle.synth(0, n, n, 77, nrg=R)
    ctx, lib = dev.context(), N.load()
    tabs = [np.zeros((R, 43, S2), np.int64), np.zeros((R, 43, S2), np.int64), np.zeros((R, 43, 16), np.int64), np.zeros((R, 43, 16), np.int64)]
    for _ in range(2):
        N.check(lib.kbbq_accumulate(ctx.handle, N.ptr(seq), N.ptr(cseq), N.ptr(qual), N.ptr(meta), n, seq.shape[1], R, S2, 6, *[N.ptr(t) for t in tabs]))
    want = oracle.accumulate(seq, cseq, qual, meta, R, 150)
    for g, w in zip(tabs, want[5:]):
        assert np.array_equal(g, 2 * w)
    dqs = oracle.get_delta_qs(*want)
    a = [np.ascontiguousarray(x, dtype=np.int64) for x in (want[0],) + dqs]
    out = np.full_like(qual, 0xEE)
    N.check(lib.kbbq_apply(ctx.handle, N.ptr(seq), N.ptr(qual), N.ptr(meta), n, seq.shape[1], R, 43, S2, 17, 6, *[N.ptr(x) for x in a], N.ptr(out)))
    ref = oracle.apply(seq, qual, meta, want[0], *dqs)
    inside = np.arange(seq.shape[1])[None, :] < (meta & 0xFFFF)[:, None]
    assert np.array_equal(out[inside].astype(np.int32) - 33, ref[inside]) and not (out == 0xEE).all(1).any()
    # the device-plane entry point on the same rows writes the same bytes
    d_out = dev.apply(dev.ReadBatch.from_host(seq, qual, meta), *dev.solve_lut(dev.Tables.from_host(*want[5:9])))
    assert np.array_equal(d_out.cpu().numpy()[inside], out[inside])
    # a quality above 42 (IndexError of the reference) in a late slab, an earlier slab clean; then a second one before it
    bad = qual.copy()
    bad[25_000, 3] = 33 + 43
    before = [t.copy() for t in tabs]
    for first in (25_000, 7_777):
        bad[first, 3] = 33 + 43
        with pytest.raises(IndexError) as e:
            N.check(lib.kbbq_accumulate(ctx.handle, N.ptr(seq), N.ptr(cseq), N.ptr(bad), N.ptr(meta), n, seq.shape[1], R, S2, 6, *[N.ptr(t) for t in tabs]))
        assert int(re.search(r'read (\d+)', str(e.value)).group(1)) == first
        assert all(np.array_equal(t, b) for t, b in zip(tabs, before))
        with pytest.raises(IndexError) as e:
            N.check(lib.kbbq_apply(ctx.handle, N.ptr(seq), N.ptr(bad), N.ptr(meta), n, seq.shape[1], R, 43, S2, 17, 6, *[N.ptr(x) for x in a], N.ptr(out)))
        assert int(re.search(r'read (\d+)', str(e.value)).group(1)) == first
    ctx.status()                                                    # nothing left behind


# ------------------------------------------------------------------ K3: model solve
def test_k3_delta_q_matches_reference_grid_and_oracle(dev, oracle):
    from kbbq import compare_reads
    _, gold = load_golden('numeric')
    errs, tot = gold['grid_errs'], gold['grid_total']
    prior_q = np.broadcast_to(np.arange(43)[:, None], (43, len(errs))).copy()
    be = np.broadcast_to(errs, prior_q.shape).copy(); bt = np.broadcast_to(tot, prior_q.shape).copy()
    dq = compare_reads.gatk_delta_q(prior_q, be, bt)
    assert dq.shape == prior_q.shape and np.array_equal(dq, gold['grid_dq'])
    # reference tests/test_compare_reads.py:141-151
    p = np.array([10, 20, 30])
    d = compare_reads.gatk_delta_q(p, np.array([10, 200, 0]), np.array([1000, 1000, 50000]))
    assert d[0] > 0 and d[1] < 0 and d[2] > 0 and np.all(d + p <= 42) and np.all(d + p > 0)
    # adversarial cells: tiny and empty cells, errs == total, counts up to 3e10
    rng = np.random.default_rng(2025)
    n = 60000
    t = (10 ** rng.uniform(0, 10.5, n)).astype(np.int64)
    e = np.minimum((t * 10 ** (-rng.uniform(0, 5, n))).astype(np.int64), t)
    t[:500] = rng.integers(0, 6, 500); e[:500] = np.minimum(rng.integers(0, 4, 500), t[:500])
    e[500:800] = t[500:800]; e[800:1100] = 0
    pq = rng.integers(0, 43, n)
    assert np.array_equal(compare_reads.gatk_delta_q(pq, e, t), oracle.gatk_delta_q(pq, e, t))
    # outside the support / outside the prior table
    assert list(compare_reads.gatk_delta_q(np.array([5, 5]), np.array([10, -3]), np.array([3, 4]))) == \
        list(oracle.gatk_delta_q(np.array([5, 5]), np.array([10, -3]), np.array([3, 4])))
    with pytest.raises(IndexError):
        compare_reads.gatk_delta_q(np.array([43]), np.array([0]), np.array([0]))


@pytest.mark.parametrize('name', ['c1_10k_1rg', 'c3cut_2k_8rg', 'c5cut_2k_mixed', 'q42_500_3rg', 'short_64_1rg'])
def test_k3_get_delta_qs_and_fused_solve_match_reference(dev, name):
    import torch
    from kbbq.gatk import applybqsr
    _, g = load_golden(name)
    dqs = applybqsr.get_delta_qs(g['meanq'], g['rg_errs'], g['rg_total'], g['q_errs'], g['q_total'],
                                 g['pos_errs'], g['pos_total'], g['dinuc_errs'], g['dinuc_total'])
    for k, v in zip(DQ, dqs):
        assert np.array_equal(v, g[k]), k
    # fused: golden count tables on the device -> LUT + delta tables
    R, _, S2 = g['pos_total'].shape
    t = dev.Tables(R, S2)
    flat = np.concatenate([g[k].ravel() for k in ('pos_errs', 'pos_total', 'dinuc_errs', 'dinuc_total')])
    t.buf.copy_(torch.from_numpy(flat))
    lut, shape, vectors, fdq = dev.solve(t, want_dq=True)
    for k, v in zip(VEC, vectors):
        assert np.array_equal(v, g[k]), k
    for k, v in zip(DQ, fdq):
        assert np.array_equal(v, g[k]), k
    want_lut, want_shape = dev.build_lut(g['meanq'], g['rgdq'], g['qdq'], g['posdq'], g['dinucdq'])
    assert shape == want_shape
    assert np.array_equal(lut.cpu().numpy(), want_lut)     # canonical + table-driven parts + flags, byte for byte


# ------------------------------------------------------------------ K3 without the host (csrc/lgam_core.h)
def test_device_gammaln_is_the_hosts_bit_for_bit(dev):
    """The precondition of the all-device solve: gammaln evaluated by a kernel over the host libm's own log() constants
    equals the host routine (= scipy.special.gammaln, tests/test_solve_core_host.py) on every argument tried."""
    import torch
    from kbbq import _native as N
    tab = dev.device_logtab()
    assert tab is not None, 'the device restatement of gammaln was not accepted on this box: the host pass is in use'
    rng = np.random.default_rng(7)
    x = np.concatenate([np.floor(2.0 ** rng.uniform(0, 34, 2_000_000)), np.arange(1, 5001, dtype=np.float64)])
    want = np.empty_like(x)
    N.check(N.load().kbbq_gammaln_host(N.ptr(x), x.size, N.ptr(want)))
    d_x = torch.from_numpy(x).cuda(); d_out = torch.empty_like(d_x)
    N.check(N.load().kbbq_gammaln_dev(dev.context().handle, N.ptr(d_x), x.size, N.ptr(tab), N.ptr(d_out)))
    assert np.array_equal(d_out.cpu().numpy().view(np.uint64), want.view(np.uint64))


def _both_solves(dev, t):
    """(LUT of the host-fed solve, LUT of solve_lut) as byte arrays."""
    lut_host, shape_host, _, _ = dev.solve(t)
    lut_dev, shape_dev = dev.solve_lut(t)
    assert shape_dev == shape_host
    return lut_host.cpu().numpy(), lut_dev.cpu().numpy()


@pytest.mark.parametrize('name', ['c1_10k_1rg', 'c3cut_2k_8rg', 'c5cut_2k_mixed', 'q42_500_3rg', 'short_64_1rg'])
def test_all_device_solve_equals_the_host_fed_solve_on_golden_tables(dev, name):
    import torch
    _, g = load_golden(name)
    R, _, S2 = g['pos_total'].shape
    t = dev.Tables(R, S2)
    t.buf.copy_(torch.from_numpy(np.concatenate([g[k].ravel() for k in ('pos_errs', 'pos_total', 'dinuc_errs', 'dinuc_total')])))
    assert dev.device_logtab() is not None
    a, b = _both_solves(dev, t)
    assert np.array_equal(a, b)
    want_lut, _ = dev.build_lut(g['meanq'], g['rgdq'], g['qdq'], g['posdq'], g['dinucdq'])
    assert np.array_equal(b, want_lut)


@pytest.mark.parametrize('seed,R,S2,scale', [(1, 1, 300, 10.3), (2, 3, 300, 8.0), (3, 8, 100, 6.0), (4, 2, 600, 9.5), (5, 5, 32, 3.0)])
def test_all_device_solve_on_adversarial_tables(dev, seed, R, S2, scale):
    """Random count tables up to 2 * 10^10 per cell, empty cells, errs == total, tiny cells: the LUT of the all-device
    solve (kernel marginals, kernel gammaln, kernel meanq) equals the host-fed one byte for byte."""
    import torch
    rng = np.random.default_rng(seed)
    t = dev.Tables(R, S2)
    npos, ndn = R * 43 * S2, R * 43 * 16

    def cells(n):
        tot = (10 ** rng.uniform(0, scale, n)).astype(np.int64)
        err = np.minimum((tot * 10 ** (-rng.uniform(0, 5, n))).astype(np.int64), tot)
        k = rng.random(n)
        tot[k < 0.1] = 0; err[k < 0.1] = 0
        err[(k > 0.1) & (k < 0.15)] = tot[(k > 0.1) & (k < 0.15)]
        return err, tot
    pe, pt = cells(npos)
    de, dt = cells(ndn)
    t.buf.copy_(torch.from_numpy(np.concatenate([pe, pt, de, dt])))
    a, b = _both_solves(dev, t)
    assert np.array_equal(a, b)


def test_all_device_solve_hands_undecidable_meanq_to_the_host(dev):
    """One quality value only: the mean error is 10^(-q/10) itself and -10 log10 of it sits ON the truncation boundary (the
    reference's "float badness", tests/test_recalibrate.py:63).  The kernel must not guess: status -> host longdouble."""
    import torch
    from kbbq import _native as N
    t = dev.Tables(2, 300)
    pe, pt, de, dt = t.views()
    pt[0, 7, :] = 1000; pe[0, 7, :3] = 2
    pt[1, 20, :] = 500; pt[1, 30, :] = 500
    a, b = _both_solves(dev, t)
    assert np.array_equal(a, b)
    ctx = dev.context()
    dev.solve_lut(t, check=False)
    with pytest.raises(N.MeanqNeedsHost):
        ctx.status()
    empty = dev.Tables(1, 300)                               # no counted base at all: 0 / 0 in the reference
    dev.solve_lut(empty, check=False)
    with pytest.raises(N.MeanqNeedsHost):
        ctx.status()


def test_get_delta_qs_known_answer(dev):
    # reference tests/test_gatk_applybqsr.py:105-121
    from kbbq.gatk import applybqsr
    a = applybqsr.get_delta_qs(np.array([10]), np.array([0]), np.array([1000]), np.array([[0]]),
                               np.array([[1000]]), np.array([[[0]]]), np.array([[[1000]]]),
                               np.array([[[0]]]), np.array([[[1000]]]))
    assert [x.tolist() for x in a] == [[3], [[2]], [[[1]]], [[[1, 0]]]]


# ------------------------------------------------------------------ drop-in API vs reference goldens
@pytest.mark.parametrize('name', GOLDEN_CASES)
def test_dropin_api_matches_reference_goldens(dev, oracle, name, tmp_path, capfd):
    from kbbq import recalibrate
    from kbbq.gatk import applybqsr
    info, gold = load_golden(name)
    fa, fb = _files(oracle, info, tmp_path)
    vectors = recalibrate.fastq_to_covariate_arrays([fa, fb], infer_rg=info['case']['infer_rg'])
    assert len(vectors) == 9
    for k, v in zip(VEC, vectors):
        assert v.dtype == np.int64 and np.array_equal(v, gold[k]), k
    for k, v in zip(DQ, applybqsr.get_delta_qs(*vectors)):
        assert np.array_equal(v, gold[k]), k
    capfd.readouterr()
    recalibrate.recalibrate_fastq([fa, fb], infer_rg=info['case']['infer_rg'])
    text = capfd.readouterr().out
    assert len(text) == info['output_len']
    assert oracle.sha256(text) == info['output_sha256']
    assert text.startswith(info['first_records']) and text.endswith(info['last_records'])


@pytest.mark.parametrize('case', GOLDEN_CASES)
def test_the_command_line_runs_without_torch(dev, oracle, case, tmp_path):
    """`python -m kbbq.main recalibrate -f A B` on one GPU never imports PyTorch (kbbq/_hipmem.py: device memory, page-locked
    slabs, copies and events through the library's own C ABI) and prints the reference's bytes: the five golden cases,
    to stdout and with -o, with the model saved (-g) and loaded back; KBBQ_USE_TORCH=1 gives the same bytes through torch."""
    import os, subprocess, sys
    from conftest import ROOT
    info, _ = load_golden(case)
    fa, fb = _files(oracle, info, tmp_path)
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, 'kbbq-py_amd'))
    env.pop('KBBQ_USE_TORCH', None)
    base = [sys.executable, '-X', 'importtime', '-m', 'kbbq.main', 'recalibrate', '-f', fa, fb] + (['--infer-rg'] if info['case']['infer_rg'] else [])

    def imported(stderr):
        return {ln.split('|')[-1].strip() for ln in stderr.decode().split('\n') if ln.startswith('import time:')}
    r = subprocess.run(base, env=env, capture_output=True, timeout=300)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    mods = imported(r.stderr)
    assert 'kbbq._hipmem' in mods and not any(m == 'torch' or m.startswith('torch.') for m in mods)
    assert len(r.stdout) == info['output_len'] and oracle.sha256(r.stdout) == info['output_sha256']
    out, model = str(tmp_path / 'out.fq'), str(tmp_path / 'model.txt')
    for extra in (['-o', out], ['-o', out, '-g', model], ['-o', out, '-g', model]):          # save the model, then load it
        r = subprocess.run(base + extra, env=env, capture_output=True, timeout=300)
        assert r.returncode == 0 and not r.stdout, r.stderr.decode()[-2000:]
        assert not any(m == 'torch' for m in imported(r.stderr))
        assert oracle.sha256(open(out, 'rb').read()) == info['output_sha256'], extra
    assert os.path.getsize(model) > 500
    if case == GOLDEN_CASES[0]:
        r = subprocess.run(base, env=dict(env, KBBQ_USE_TORCH='1'), capture_output=True, timeout=300)
        assert r.returncode == 0 and 'torch' in imported(r.stderr) and oracle.sha256(r.stdout) == info['output_sha256']
        # bad input still ends the command with the reference's exception
        lines = open(fa).read().split('\n')
        lines[4 * 77 + 3] = lines[4 * 77 + 3][:10] + 'L' + lines[4 * 77 + 3][11:]                  # q = 43
        bad = str(tmp_path / 'bad.fq')
        open(bad, 'w').write('\n'.join(lines))
        r = subprocess.run([sys.executable, '-m', 'kbbq.main', 'recalibrate', '-f', bad, fb], env=env, capture_output=True, timeout=300)
        assert r.returncode != 0 and b'IndexError' in r.stderr


def _write(tmp_path, name, recs):
    p = tmp_path / name
    p.write_text(''.join('@%s\n%s\n+\n%s\n' % r for r in recs))
    return str(p)


def test_reference_known_answers_through_the_device(dev, tmp_path, capfd, monkeypatch):
    """reference tests/test_recalibrate.py:53-135 and tests/test_compare_reads.py:219-233."""
    import sys
    import kbbq.main
    from kbbq import recalibrate, compare_reads, fastx
    for nm, rg in (('foo', False), ('foo/1_RG:Z:bar', True)):
        fa = _write(tmp_path, 'u%d.fq' % rg, [(nm, 'ATG', '((#')])
        fb = _write(tmp_path, 'c%d.fq' % rg, [(nm, 'ACG', '((#')])
        v = recalibrate.fastq_to_covariate_arrays([fa, fb], infer_rg=rg)
        assert [x.tolist() for x in v[:3]] == [[6], [1], [2]]
        assert v[3][0, 7] == 1 and v[3].sum() == 1 and v[4][0, 7] == 2 and v[4].sum() == 2
        assert v[5].shape == (1, 43, 6) and v[5][0, 7, 1] == 1 and v[5].sum() == 1
        assert v[6][0, 7, 0] == 1 and v[6][0, 7, 1] == 1 and v[6].sum() == 2
        assert v[7].shape == (1, 43, 16) and v[7][0, 7, 1] == 1 and v[8][0, 7, 1] == 1 and v[8].sum() == 1
        capfd.readouterr()
        recalibrate.recalibrate_fastq([fa, fb], infer_rg=rg)
        assert capfd.readouterr().out == "@%s\nATG\n+\n''#\n" % nm
    recalibrate.recalibrate(bam=None, fastq=[fa, fb], infer_rg=True)
    assert capfd.readouterr().out == "@foo/1_RG:Z:bar\nATG\n+\n''#\n"
    fa = _write(tmp_path, 'u.fq', [('foo', 'ATG', '((#')]); fb = _write(tmp_path, 'c.fq', [('foo', 'ACG', '((#')])
    with monkeypatch.context() as m:
        m.setattr(sys, 'argv', [sys.argv[0], 'recalibrate', '-f', fa, fb])
        kbbq.main.main()
    assert capfd.readouterr().out == "@foo\nATG\n+\n''#\n"
    with pytest.raises(NotImplementedError), monkeypatch.context() as m:
        m.setattr(sys, 'argv', [sys.argv[0], 'recalibrate', '-b', 'foo'])
        kbbq.main.main()
    # per-read apply, tables with 8 Q rows and an unpadded 16-column dinuc table
    read = fastx.FastxRecord('foo', 'ATG', '((#')
    posdq = np.zeros((1, 8, 6)); posdq[0, 7, :] = 3
    ddq = np.zeros((1, 8, 16)); ddq[0, 7, :] = 5
    got = compare_reads.recalibrate_fastq(read, np.array([10]), np.array([1]), np.array([[2] * 8]),
                                          posdq, ddq, np.array([0]), compare_reads.Dinucleotide.dinuc_to_int)
    assert np.array_equal(got, [21, 21, 2])


def test_reference_error_behaviour(dev, tmp_path):
    from kbbq import recalibrate
    ok = ('r0', 'ACGTACGT', 'IIIIIIII')
    def run(recs_a, recs_b=None, **kw):
        fa = _write(tmp_path, 'ea.fq', recs_a); fb = _write(tmp_path, 'eb.fq', recs_b or recs_a)
        return recalibrate.fastq_to_covariate_arrays([fa, fb], **kw)
    with pytest.raises(IndexError):                       # H2 shorter than the running maximum
        run([ok, ('r1', 'ACG', 'III')])
    with pytest.raises(IndexError):                       # H7 q = 43 ('L')
        run([ok, ('r1', 'ACGTACGT', 'IIIILIII')])
    with pytest.raises(TypeError):                        # lower-case base in a looked-up dinucleotide
        run([ok, ('r1', 'ACGtACGT', 'IIIIIIII')])
    run([ok, ('r1', 'ACGtNCGT', "II##IIII")])              # ... but not when q < 6 / N hide it
    with pytest.raises(AssertionError):                   # name prefix
        run([ok, ('r1', 'ACGTACGT', 'IIIIIIII')], [ok, ('x1', 'ACGTACGT', 'IIIIIIII')])
    with pytest.raises(TypeError):                        # earlier device error beats a later host error
        run([ok, ('r1', 'ACGtACGT', 'IIIIIIII'), ('r2', 'ACG', 'III')])
    with pytest.raises(TypeError):                        # same read: dinuc lookup precedes the mask error
        run([ok, ('r1', 'AcG', 'III')])
    with pytest.raises(IndexError):                       # no RG field with --infer-rg
        run([ok], infer_rg=True)
    # pass 2 sees reads pass 1 never saw (zip truncation): a new RG there is an IndexError
    fa = _write(tmp_path, 'ta.fq', [('a/1_RG:Z:x', 'ACGT', 'IIII'), ('b/1_RG:Z:y', 'ACGT', 'IIII')])
    fb = _write(tmp_path, 'tb.fq', [('a/1_RG:Z:x', 'ACGT', 'IIII')])
    with pytest.raises(IndexError), contextlib.redirect_stdout(io.StringIO()):
        recalibrate.recalibrate_fastq([fa, fb], infer_rg=True)


def test_h1_negative_cycle_aliasing(dev, oracle, tmp_path):
    unc = [('a/2', 'AC', 'II'), ('b/1', 'ACGT', 'IIII'), ('c/2', 'ACGTA', 'IIIII')]
    from kbbq import recalibrate
    fa = _write(tmp_path, 'h1.fq', unc)
    got = recalibrate.fastq_to_covariate_arrays([fa, fa])
    want = oracle.py_accumulate(unc, unc)
    for g, w, k in zip(got, want, VEC):
        assert np.array_equal(g, w), k


def test_covariatedata_consume_read(dev, oracle):
    """reference tests/test_covariate.py:159-165 plus the intended semantics on real reads."""
    from kbbq import covariate, read
    try:
        r = read.ReadData(seq=np.array(['A', 'T', 'G']), qual=np.array([6, 10, 3]),
                          skips=np.array([False, False, True]), name='read01', rg=0, second=False,
                          errors=np.array([False, True, True]))
        cd = covariate.CovariateData()
        cd.consume_read(r)
        assert cd.qcov.rgcov[0] == (1, 2)
        assert cd.qcov[0, 10] == (1, 1)
        assert cd.cyclecov[0, 6, 0] == (0, 1)
        assert cd.dinuccov[0, 10, 1] == (1, 1)
        assert (cd.get_num_rgs(), cd.get_num_qs(), cd.get_num_cycles()) == (1, 11, 3)
        # a second, longer second-in-pair read with a low-quality unskipped base
        r2 = read.ReadData(seq=np.array(list('ACGNTA')), qual=np.array([30, 2, 30, 30, 30, 7]),
                           skips=np.zeros(6, bool), name='read02', rg=0, second=True,
                           errors=np.array([True, True, False, False, False, True]))
        cd.consume_read(r2)
        assert cd.cyclecov.shape() == (1, 31, 12)
        assert cd.cyclecov[0, 6, 0] == (0, 1)                 # first read's data kept at the front
        assert cd.cyclecov[0, 30, -1] == (1, 1) and cd.cyclecov[0, 2, -2] == (1, 1)   # q=2 counted (not skipped)
        assert cd.cyclecov[0, 7, -6] == (1, 1)
        assert cd.qcov.rgcov[0] == (4, 8)
        # dinuc: pos1 q<6 -> none; pos2 'CG' ok; pos3 N -> none; pos4 follows N -> none; pos5 'TA'
        d = compare_dinuc = {'CG': 14, 'TA': 4}
        assert cd.dinuccov[0, 30, d['CG']] == (0, 1) and cd.dinuccov[0, 7, d['TA']] == (1, 1)
        assert cd.dinuccov.total.sum() == 3
    finally:
        read.ReadData.rg_to_pu = dict(); read.ReadData.rg_to_int = dict(); read.ReadData.numrgs = 0


def test_adversarial_single_bin_does_not_overflow_lds_counters(dev):
    """Every base of every read falls into ONE (q, context) bin and ONE bin per cycle: the worst
    case for the 16-bit packed LDS counters of K1 (flush cadences in DESIGN.md)."""
    import torch
    n, L, pitch = 400_000, 150, 160
    b = dev.ReadBatch(n, pitch)
    row = torch.full((pitch,), ord('N'), dtype=torch.uint8, device='cuda'); row[:L] = ord('A')
    crow = row.clone(); crow[:L:7] = ord('C')                       # an error every 7th base
    qrow = torch.zeros(pitch, dtype=torch.uint8, device='cuda'); qrow[:L] = 33 + 40
    b.seq[:] = row; b.cseq[:] = crow; b.qual[:] = qrow
    meta = torch.full((n,), L, dtype=torch.int64, device='cuda')
    meta[1::2] += 1 << 31                                            # alternate mates
    b.meta[:] = torch.where(meta >= 2 ** 31, meta - 2 ** 32, meta).to(torch.int32)
    t = dev.Tables(1, 2 * L)
    dev.accumulate(b, t)
    pe, pt, de, dt = t.to_host()
    assert pt.sum() == n * L and pt[0, 40].sum() == n * L
    assert np.all(pt[0, 40, :L] == n // 2) and np.all(pt[0, 40, L:] == n // 2)
    nerr = len(range(0, L, 7))
    assert pe.sum() == n * nerr
    assert np.all(pe[0, 40, 0:L:7] == n // 2)
    assert dt.sum() == n * (L - 1) and dt[0, 40, 0] == n * (L - 1)   # context 'AA' = 0 for every base but the first
    assert de[0, 40, 0] == n * (nerr - 1)                            # position 0 has no context


# ------------------------------------------------------------------ large: properties
def test_large_batch_properties(dev):
    """4 M synthetic 2x150 reads (600 M bases) generated on the device: counts vs independent
    torch reductions, linearity over a split, apply invariants."""
    import torch
    n = 4_000_000
    b = dev.ReadBatch.synthetic(0, n, n, seed=2024, nrg=1)
    t = dev.Tables(1, 300)
    dev.accumulate(b, t)
    pe, pt, de, dt = [x.clone() for x in t.views()]
    q = b.qual[:, :150].to(torch.int16) - 33
    valid = q >= 6
    err = (b.seq[:, :150] != b.cseq[:, :150]) & valid
    assert int(pt.sum()) == int(valid.sum())
    assert int(pe.sum()) == int(err.sum())
    # per-score and per-cycle marginals
    qt = torch.bincount(q[valid].to(torch.int64), minlength=43)
    assert torch.equal(pt.sum(dim=(0, 2)), qt)
    second = (b.meta[:n] < 0)                                   # bit 31
    cyc_first = valid[~second].sum(dim=0)
    cyc_second = valid[second].sum(dim=0)
    assert torch.equal(pt[0].sum(dim=0)[:150], cyc_first)
    assert torch.equal(pt[0].sum(dim=0)[150:], cyc_second.flip(0))
    assert int(dt.sum()) <= int(pt.sum()) and int(de.sum()) <= int(pe.sum())
    # split into two launches == one launch
    t2 = dev.Tables(1, 300)
    for lo, hi in ((0, 1_234_567), (1_234_567, n)):
        part = dev.ReadBatch(hi - lo, b.pitch)
        part.seq, part.cseq, part.qual, part.meta = b.seq[lo:hi], b.cseq[lo:hi], b.qual[lo:hi], b.meta[lo:hi]
        dev.accumulate(part, t2)
    assert torch.equal(t.buf, t2.buf)
    # apply: identity model leaves every byte unchanged; q < 6 is always passed through
    z = lambda *s: np.zeros(s, dtype=np.int64)
    qdq = np.broadcast_to(np.arange(43), (1, 43)).copy()
    lut, shp = dev.build_lut(z(1), z(1), qdq, z(1, 43, 300), z(1, 43, 17))
    out = dev.apply(b, dev.lut_to_device(lut), shp)
    assert torch.equal(out, b.qual)
    lut, shp = dev.build_lut(z(1) + 5, z(1), qdq, z(1, 43, 300), z(1, 43, 17))
    out = dev.apply(b, dev.lut_to_device(lut), shp)
    low = (b.qual < 39) | (b.qual == 0)
    assert torch.equal(out[low], b.qual[low])
    assert torch.equal(out[~low], b.qual[~low] + 5)


def test_full_size_properties(dev):
    """BASELINE config 2 in full (50 M x 2x150 bp, 7.5 G bases): size-independent properties of the tally and the
    apply -- totals and per-quality / per-cycle marginals against independent torch reductions, the two device
    layouts against each other, linearity over a split, the apply's pass-through and padding invariants."""
    import torch
    n, S = 50_000_000, 150
    b = dev.ReadBatch.synthetic(0, n, n, seed=1, nrg=1)
    t = dev.Tables(1, 2 * S)
    dev.accumulate(b, t)
    pe, pt, de, dt = t.views()
    tot_valid = tot_err = 0
    qhist = torch.zeros(43, dtype=torch.int64, device='cuda')
    cyc1 = torch.zeros(S, dtype=torch.int64, device='cuda'); cyc2 = torch.zeros(S, dtype=torch.int64, device='cuda')
    step = 2_000_000
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        q = b.qual[lo:hi, :S]
        valid = q >= 33 + 6
        err = (b.seq[lo:hi, :S] != b.cseq[lo:hi, :S]) & valid
        tot_valid += int(valid.sum()); tot_err += int(err.sum())
        qhist += torch.bincount((q[valid].to(torch.int64) - 33), minlength=43)
        second = b.meta[lo:hi] < 0
        cyc1 += valid[~second].sum(dim=0); cyc2 += valid[second].sum(dim=0)
        del q, valid, err
    assert int(pt.sum()) == tot_valid and int(pe.sum()) == tot_err and tot_err > 0
    assert torch.equal(pt.sum(dim=(0, 2)), qhist)
    assert torch.equal(pt[0].sum(dim=0)[:S], cyc1) and torch.equal(pt[0].sum(dim=0)[S:], cyc2.flip(0))
    assert int(dt.sum()) <= tot_valid and int(de.sum()) <= tot_err
    # linearity: two uneven parts add up to the whole
    t2 = dev.Tables(1, 2 * S)
    for lo, hi in ((0, 17_000_002), (17_000_002, n)):
        part = dev.ReadBatch(hi - lo, b.pitch)
        part.seq, part.cseq, part.qual, part.meta = b.seq[lo:hi], b.cseq[lo:hi], b.qual[lo:hi], b.meta[lo:hi]
        dev.accumulate(part, t2)
    assert torch.equal(t.buf, t2.buf)
    # mate-pair rows: the same tables; the same new qualities
    pb = dev.PairBatch.from_reads(b)
    t3 = dev.Tables(1, 2 * S)
    dev.accumulate(pb, t3)
    assert torch.equal(t.buf, t3.buf)
    lut, shape, _, _ = dev.solve(t)
    out = dev.apply(b, lut, shape)
    # bases below the threshold pass through, padding stays zero, everything else is a valid quality character
    for lo in range(0, n, 5_000_000):
        hi = min(n, lo + 5_000_000)
        q, o = b.qual[lo:hi], out[lo:hi]
        low = (q[:, :S] < 33 + 6)
        assert torch.equal(o[:, :S][low], q[:, :S][low]) and not bool(o[:, S:].any())
        assert int(o[:, :S].min()) >= 33 and int(o[:, :S].max()) <= 33 + 93
        del q, o, low
    del b
    torch.cuda.empty_cache()
    out_pairs = pb.unpack(dev.apply(pb, lut, shape))
    assert torch.equal(out_pairs[:n], out[:n])


@pytest.mark.parametrize('R', [1, 8])
def test_the_headline_layout_at_full_size(dev, R):
    """BASELINE configs 2 and 3 in full (50 M x 2x150 bp, 1 and 8 read groups) in the layout bench.py's headline is measured
    on and the FASTQ packer writes -- mate-pair rows, 4-bit sequence planes, rows gathered by read-group segment: its
    count tables equal those of K1 on one character row per read, and the short-lived apply kernel's output on it (in
    grouped order, and stored through the permutation) equals the persistent kernel's on the character rows."""
    import torch
    n, S = 50_000_000, 150
    b = dev.ReadBatch.synthetic(0, n, n, seed=1, nrg=R)
    t = dev.Tables(R, 2 * S)
    dev.accumulate(b, t)
    laid = dev.lay_out(b, R, S, packed=True)
    assert laid.nib and isinstance(laid, dev.PairBatch) and (laid.seg is not None) == (R > 1) and laid.n == n // 2
    t2 = dev.Tables(R, 2 * S)
    dev.accumulate(laid, t2)
    assert torch.equal(t.buf, t2.buf) and int(t.buf.sum()) > 0
    lut, shape = dev.solve_lut(t)
    assert shape[3] == 1
    want = dev.apply(b, lut, shape)                       # one read per row, character planes: the persistent kernels
    del b
    torch.cuda.empty_cache()
    got = laid.unpack(dev.apply(laid, lut, shape))        # k2t_apply, rows in the layout's (grouped) order
    pitch = want.shape[1]
    if R == 1:
        assert torch.equal(got[:n, :S], want[:n, :S])
    else:
        pairs = want[:n].view(n // 2, 2 * pitch)
        step = 5_000_000
        for lo in range(0, n // 2, step):                 # row i of the layout is pair perm[i] of the input
            hi = min(n // 2, lo + step)
            assert torch.equal(got[2 * lo:2 * hi].view(hi - lo, 2 * pitch), pairs[laid.perm[lo:hi]]), lo
    del got
    torch.cuda.empty_cache()
    back = laid.unpack(dev.apply(laid, lut, shape, restore_order=True))      # stored through the permutation: input order
    assert torch.equal(back[:n, :S], want[:n, :S])
    assert not bool(back[:n, S:].any())


def test_bench_prints_one_json_line_with_the_contract_fields(dev):
    """bench.py's contract with the driver: exactly one JSON line on stdout, the metric / config of BASELINE.json, the
    roofline and cpu_baseline objects; a small run (the default sizes are the driver's business)."""
    import json, os, subprocess, sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
                        '--reads', '400000', '--cpu-sample', '20000'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.split('\n') if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, 'BASELINE.json')))
    assert d['metric'].split(' (')[0] in base['metric'] and d['unit'] == 'bases/s'
    assert (d['n_gpus'], d['steps'], d['warmup'], d['higher_is_better'], d['scaling'], d['vs_baseline']) == (1, 2, 1, True, 'weak', None)
    assert d['dtype'] == 'u8' and 'synthetic' in d['data'] and 'workload' in d['config'] and 'model' not in d['config']
    assert d['value'] > 1e9 and abs(d['value'] - 400000 * 150 / (d['ms_per_step'] / 1e3)) / d['value'] < 1e-6
    roof = d['roofline']
    assert roof['bound'] == 'hbm' and roof['unit'] == 'GB/s' and roof['peak'] == 8000.0
    assert abs(roof['frac'] - roof['achieved'] / roof['peak']) < 1e-9 and 0 < roof['frac'] < 1 and (roof['traffic'] is None or roof['traffic'] > 0)
    cpu = d['cpu_baseline']
    assert cpu['kind'] in ('port', 'reference') and cpu['cores'] >= 1 and cpu['value'] > 1e6 and cpu['unit'] == 'bases/s' and cpu['sample']
    # the figures next to it: the reference's own measured rate, the port on all host cores
    assert cpu['reference_bases_per_s'] > 1e5 and cpu['reference_cores'] == 1 and cpu['all_cores']['cores'] == cpu['host_cores']
    assert d['ranks_seen'] == 1 and d['kernels']['host_solve_and_sync_ms'] < 5
    assert d['verified'] is True and cpu['measured'].startswith('before the first GPU call')
    assert d['per_rank_ms_per_step']['min'] == d['per_rank_ms_per_step']['max'] == d['ms_per_step']
    # the other configurations and the rows next to the path, measured in the same process
    extra = d['extra']
    for key in ('config3_8rg', 'layout_pairs', 'layout_reads', 'aligned_read_kernels', 'file_path'):
        assert key in extra and 'error' not in extra[key], (key, extra.get(key))
    assert extra['config3_8rg']['layout_inclusive']['value'] < extra['config3_8rg']['value']
    assert all(extra['aligned_read_kernels'][k]['GB/s'] > 0 for k in ('k4_find_errors', 'k5_count_q', 'k6_canonical_reads'))
    assert extra['file_path']['value'] > 1e6 and extra['file_path']['stages_s']
    # the file path uploads the layout of the headline, written by the packer: 2 B/base, no layout pass, no unpack
    assert 'layout' not in extra['file_path']['stages_s'] and 2.0 <= extra['file_path']['h2d_bytes_per_base'] < 2.1
    assert all('mate-pair rows' in b['layout'] and '4-bit' in b['layout'] and 'kbbq_fastq_fill_rows' in b['written_by'] for b in extra['file_path']['bands'])
    # round 4: the same files within 256 MB of device memory and through two pipes -- the same bytes
    st, pp = extra['file_path']['streamed_within_256MB_of_device_memory'], extra['file_path']['through_two_pipes']
    assert st['same_bytes_as_resident'] is True and pp['same_bytes_as_resident'] is True, (st, pp)
    assert pp['run']['sequential'] and pp['run']['spooled'] and st['value'] > 1e6
    assert extra['config3_8rg']['verified'] is True and extra['single_end_150']['verified'] is True
    assert extra['from_input_order_rows']['ms_per_step'] == extra['layout_reads']['ms_per_step']


def test_bench_launches_its_own_two_rank_job(dev):
    """`python bench.py --gpus 2` from a plain command line: the parent starts torch.distributed.run as a child before it
    imports anything GPU-side; on this one-GPU box the two ranks share the device over gloo (a rehearsal, flagged)."""
    import json, os, subprocess, sys
    from conftest import ROOT
    env = dict(os.environ); env.pop('RANK', None); env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--reads', '200000'], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.split('\n') if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['ranks_seen'] == 2 and d['backend'] == 'gloo' and 'rehearsal' in d['data']
    assert abs(d['value'] - 2 * 200000 * 150 / (d['ms_per_step'] / 1e3)) / d['value'] < 1e-6
    assert 'extra' not in d and 'cpu_baseline' not in d
