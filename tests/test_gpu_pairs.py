"""
GPU tests (-m gpu) of the mate-pair row layout (include/kbbq_hip.h "mate-pair rows"): the same K1 / K2 on
rows that hold a read pair each.  Everything must equal the one-read-per-row path, which the other GPU tests
pin to the oracle and the reference's goldens; the oracle is consulted directly as well.
"""
import os

import numpy as np
import pytest

from test_gpu_parity import dev                      # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _pairs_numpy(plane, meta, S, ppitch, fill):
    n = plane.shape[0]
    out = np.full((n // 2, ppitch), fill, dtype=np.uint8)
    out[:, :S] = plane[0::2, :S]
    out[:, S + 1:2 * S + 1] = plane[1::2, :S]
    return out


@pytest.mark.parametrize('S,nrg', [(150, 1), (150, 4), (100, 2), (151, 1), (16, 3), (75, 1), (33, 2)])
def test_pack_accumulate_apply_equal_the_row_per_read_path(dev, oracle, S, nrg):
    import torch
    n = 6002 if S != 16 else 1280
    b = dev.ReadBatch.synthetic(0, n, n, seed=3 + S, len_lo=S, len_hi=S, nrg=nrg)
    pb = dev.PairBatch.from_reads(b)
    assert pb.n == n // 2 and pb.pitch == (2 * S + 1 + 15) // 16 * 16
    # the layout itself
    meta = b.meta[:n].cpu().numpy().view(np.uint32)
    for plane, pplane, fill in ((b.seq, pb.seq, ord('N')), (b.cseq, pb.cseq, ord('N')), (b.qual, pb.qual, 0)):
        assert np.array_equal(pplane[:pb.n].cpu().numpy(), _pairs_numpy(plane[:n].cpu().numpy(), meta, S, pb.pitch, fill))
    want_meta = (2 * S + 1) | (meta[0::2] & 0x7FFF0000)
    assert np.array_equal(pb.meta[:pb.n].cpu().numpy().view(np.uint32), want_meta)
    # K1: identical tables, and they are the oracle's
    t_reads, t_pairs = dev.Tables(nrg, 2 * S), dev.Tables(nrg, 2 * S)
    dev.accumulate(b, t_reads)
    dev.accumulate(pb, t_pairs)
    assert torch.equal(t_reads.buf, t_pairs.buf)
    host = [x.cpu().numpy() for x in (b.seq[:n], b.cseq[:n], b.qual[:n])]
    want = oracle.accumulate(host[0], host[1], host[2], meta, nrg, S)
    for got, w in zip(t_pairs.to_host(), want[5:9]):             # pos_errs, pos_total, dinuc_errs, dinuc_total
        assert np.array_equal(got, w)
    # K3 -> K2: identical new qualities after unpacking
    lut, shape, _, _ = dev.solve(t_pairs)
    out_reads = dev.apply(b, lut, shape)
    out_pairs = dev.apply(pb, lut, shape)
    assert torch.equal(pb.unpack(out_pairs)[:n], out_reads[:n])
    # padding and separator bytes of the pair plane stay zero
    op = out_pairs[:pb.n].cpu().numpy()
    assert not op[:, S].any() and not op[:, 2 * S + 1:].any()
    # adds into the tables like the other path
    dev.accumulate(pb, t_pairs)
    assert torch.equal(t_pairs.buf, 2 * t_reads.buf)


def test_minscore_and_split_thresholds(dev, oracle):
    import torch
    n, S = 4000, 150
    b = dev.ReadBatch.synthetic(0, n, n, seed=17, len_lo=S, len_hi=S, nrg=2, qlo=2, qhi=41)
    pb = dev.PairBatch.from_reads(b)
    for minscore, dmin in ((6, None), (20, None), (4, 6), (10, 6), (4, 12)):
        a, c = dev.Tables(2, 2 * S), dev.Tables(2, 2 * S)
        dev.accumulate(b, a, minscore, dinuc_minscore=dmin)
        dev.accumulate(pb, c, minscore, dinuc_minscore=dmin)
        assert torch.equal(a.buf, c.buf), (minscore, dmin)
    # more quality rows than the LDS tables hold: the pair path declines, the row-per-read path has a fallback
    from kbbq import _native as N
    with pytest.raises(N.LutNeedsCheckedApply):
        dev.accumulate(pb, dev.Tables(2, 2 * S), 0)
    dev.accumulate(b, dev.Tables(2, 2 * S), 0)


def test_not_pairable_and_error_reporting(dev):
    import torch
    n, S = 512, 60
    b = dev.ReadBatch.synthetic(0, n, n, seed=5, len_lo=40, len_hi=S, nrg=1)          # ragged lengths
    with pytest.raises(ValueError):
        dev.PairBatch.from_reads(b)
    b = dev.ReadBatch.synthetic(0, n - 1, n - 1, seed=5, len_lo=S, len_hi=S, nrg=1)  # odd count
    with pytest.raises(ValueError):
        dev.PairBatch.from_reads(b)
    b = dev.ReadBatch.synthetic(0, n, n, seed=5, len_lo=S, len_hi=S, nrg=1)
    b.meta[7] = b.meta[7] ^ (1 << 16)                                                  # mates in different read groups
    with pytest.raises(ValueError):
        dev.PairBatch.from_reads(b)
    b = dev.ReadBatch.synthetic(0, n, n, seed=5, len_lo=S, len_hi=S, nrg=1)
    assert not dev.PairBatch.worthwhile(75, 80) and dev.PairBatch.worthwhile(150, 160)
    pb = dev.PairBatch.from_reads(b)
    t = dev.Tables(1, 2 * S)
    pb.qual[9, S + 5] = 33 + 43                                                        # quality 43 in mate 2 of pair 9
    with pytest.raises(IndexError):
        dev.accumulate(pb, t)
    pb = dev.PairBatch.from_reads(b)
    pb.seq[3, 10] = ord('R'); pb.qual[3, 10] = 33 + 30
    with pytest.raises(TypeError):
        dev.accumulate(pb, dev.Tables(1, 2 * S))
    # a row the fast apply cannot serve: reported, never silently wrong
    pb = dev.PairBatch.from_reads(b)
    t = dev.Tables(1, 2 * S); dev.accumulate(pb, t)
    lut, shape, _, _ = dev.solve(t)
    pb.qual[5, 3] = 33 + 60
    from kbbq import _native as N
    with pytest.raises(N.LutNeedsCheckedApply):
        dev.apply(pb, lut, shape)


def test_large_batch_properties(dev):
    """20 M reads: pair rows give the very tables and qualities of the row-per-read path."""
    import torch
    n, S = 20_000_000, 150
    b = dev.ReadBatch.synthetic(0, n, n, seed=1, len_lo=S, len_hi=S, nrg=1)
    pb = dev.PairBatch.from_reads(b)
    a, c = dev.Tables(1, 2 * S), dev.Tables(1, 2 * S)
    dev.accumulate(b, a); dev.accumulate(pb, c)
    assert torch.equal(a.buf, c.buf)
    lut, shape, _, _ = dev.solve(c)
    out_reads = dev.apply(b, lut, shape)
    out_pairs = dev.apply(pb, lut, shape)
    del b
    un = pb.unpack(out_pairs)
    assert torch.equal(un[:n], out_reads[:n])


def test_product_fastq_path_takes_pair_rows_when_it_can(dev, oracle, tmp_path):
    """kbbq.recalibrate tallies uniform pairs on mate-pair rows (goldens c1 / c3cut: the printed FASTQ is the
    reference's, tests/test_gpu_parity.py) and ragged reads one read per row."""
    from conftest import load_golden
    from kbbq import recalibrate
    from test_gpu_parity import VEC, _files
    # c5cut: ragged lengths 36..300 in ascending order: several length bands, each at its own pitch; no pairs; every
    # band runs the table-driven K1 grouped by read group (the long ones on rows trimmed by the band's shortest read)
    for name, expect in (('c1_10k_1rg', ('pairs', False)), ('c3cut_2k_8rg', ('pairs', True)),
                         ('c5cut_2k_mixed', None), ('q42_500_3rg', ('pairs', True))):
        info, gold = load_golden(name)
        d = tmp_path / name; d.mkdir()
        fa, fb = _files(oracle, info, d)
        packed, tables = recalibrate._pack_and_tally([fa, fb], info['case']['infer_rg'], 6, 42)
        bands = packed['bands']
        if expect is None:
            assert len(bands) >= 6 and [b['pitch'] for b in bands] == sorted(b['pitch'] for b in bands)
            assert bands[0]['pitch'] <= 48 and bands[-1]['pitch'] == 304 and sum(b['n'] for b in bands) == packed['n']
            assert all(not isinstance(b['laid'], dev.PairBatch) for b in bands)
            assert all(b['laid'] is not None and 0 < b['Smin'] <= b['S'] for b in bands), [(b['S'], b['laid']) for b in bands]
        else:
            assert len(bands) == 1
            laid = bands[0]['laid']
            assert isinstance(laid, dev.PairBatch) and (getattr(laid, 'seg', None) is not None) == expect[1], name
        got = recalibrate._vectors_from_tables(*tables.to_host(), 42)
        for k, v in zip(VEC, got):
            assert np.array_equal(v, gold[k]), (name, k)


@pytest.mark.parametrize('short,long_', [(100, 150), (50, 100), (150, 250)])
def test_a_shorter_band_of_uniform_pairs_keeps_one_read_per_row(dev, oracle, short, long_, tmp_path, capfd):
    """2 x 100 bp pairs followed by 2 x 150 bp pairs (lengths non-decreasing: valid for the reference): the shorter band
    qualifies for mate-pair rows by itself, but the count tables have 2 x 150 columns -- it must be tallied one read per
    row (it used to raise ValueError out of pass 1).  Whole command against the oracle's text."""
    from kbbq import recalibrate
    n1, n2 = 600, 800
    s1 = oracle.synth(0, n1, n1 + n2, 5, short, short, 2)
    s2 = oracle.synth(n1, n2, n1 + n2, 5, long_, long_, 2)
    pitch = s2[0].shape[1]

    seq, cseq, qual = (np.concatenate([np.pad(a, ((0, 0), (0, pitch - a.shape[1])), constant_values=fill), b])
                       for a, b, fill in ((s1[0], s2[0], ord('N')), (s1[1], s2[1], ord('N')), (s1[2], s2[2], 0)))
    meta = np.concatenate([s1[3], s2[3]])
    names = oracle.synth_names(0, n1 + n2, 2, with_rg=True)
    fa, fb = str(tmp_path / 'a.fq'), str(tmp_path / 'b.fq')
    oracle.write_fastq(fa, names, seq, qual, meta)
    oracle.write_fastq(fb, names, cseq, qual, meta)
    want, wantv, _ = oracle.recalibrate_fastq_text([fa, fb], True)
    capfd.readouterr()
    recalibrate.recalibrate_fastq([fa, fb], infer_rg=True)
    assert capfd.readouterr().out == want
    vec = recalibrate.fastq_to_covariate_arrays([fa, fb], infer_rg=True)
    for g, w in zip(vec, wantv):
        assert np.array_equal(g, w)


@pytest.mark.parametrize('nrg,infer,n', [(1, False, 3000), (3, True, 3000), (1, False, 2999)])
def test_single_end_files_go_two_reads_to_a_row(dev, oracle, nrg, infer, n, tmp_path, capfd):
    """Single-end FASTQ files (no name ends in /2: every read is first in pair, compare_reads.py:304-306) of one length:
    the file path lays two neighbouring reads into one mate-pair row (KBBQ_ROWS_TWINS) -- the whole command against the
    oracle's text, with one read group and with read groups that change every second read."""
    from kbbq import recalibrate, _device as D
    seq, cseq, qual, meta = oracle.synth(0, n, n, 7, 150, 150, nrg)
    meta = meta & np.uint32(0x7FFFFFFF)
    names = ['s%d' % i + ('_RG:Z:g%d' % ((i >> 1) % nrg) if infer else '') for i in range(n)]
    fa, fb = str(tmp_path / 'a.fq'), str(tmp_path / 'b.fq')
    oracle.write_fastq(fa, names, seq, qual, meta)
    oracle.write_fastq(fb, names, cseq, qual, meta)
    want, wantv, _ = oracle.recalibrate_fastq_text([fa, fb], infer)
    seen = []
    real = D.laid_from_reader                     # the packer writes the layout itself: no device layout pass to look at
    D.laid_from_reader = lambda *a, **k: seen.append(real(*a, **k)) or seen[-1]
    try:
        capfd.readouterr()
        recalibrate.recalibrate_fastq([fa, fb], infer_rg=infer)
        assert capfd.readouterr().out == want
    finally:
        D.laid_from_reader = real
    assert seen and all(isinstance(b, D.PairBatch) and b.twins for b in seen)
    vec = recalibrate.fastq_to_covariate_arrays([fa, fb], infer_rg=infer)
    for g, w in zip(vec, wantv):
        assert np.array_equal(g, w)


def _run_ranks(world, argv, timeout=300, env=None):
    import os, socket, subprocess, sys
    from conftest import ROOT
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, KBBQ_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', **(env or {}))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world),
           '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(ROOT, 'tests', 'dist_cli_worker.py')] + argv
    return subprocess.run(cmd, env=env, capture_output=True, timeout=timeout)


@pytest.mark.parametrize('name', ['c1_10k_1rg', 'c3cut_2k_8rg', 'c5cut_2k_mixed', 'short_64_1rg'])
def test_two_ranks_print_the_reference_output(dev, oracle, name, tmp_path):
    """`kbbq recalibrate` under torch.distributed.run with 2 ranks (gloo, both on this GPU: a rehearsal of
    the one-process-per-GPU mode): shards, one allreduce of the tables, rank-ordered output = the reference's."""
    from conftest import load_golden
    from test_gpu_parity import _files
    info, _ = load_golden(name)
    fa, fb = _files(oracle, info, tmp_path)
    argv = ['recalibrate', '-f', fa, fb] + (['--infer-rg'] if info['case']['infer_rg'] else [])
    r = _run_ranks(2, argv)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    text = r.stdout.decode()
    assert len(text) == info['output_len'] and oracle.sha256(text) == info['output_sha256']


@pytest.mark.parametrize('n,world', [(4001, 2), (3000, 3)])
def test_ranks_recalibrate_single_end_files(dev, oracle, n, world, tmp_path):
    """Single-end FASTQ files under torch.distributed.run (gloo rehearsal on this GPU): every rank cuts its own byte
    range (no name ends in /2, so any record start is a cut point), lays its reads two to a row -- an odd shard
    included -- and the rank-ordered output is the oracle's text."""
    seq, cseq, qual, meta = oracle.synth(0, n, n, 11, 150, 150, 2)
    meta = meta & np.uint32(0x7FFFFFFF)
    names = ['s%d_RG:Z:g%d' % (i, (i >> 1) % 2) for i in range(n)]
    fa, fb = str(tmp_path / 'a.fq'), str(tmp_path / 'b.fq')
    oracle.write_fastq(fa, names, seq, qual, meta)
    oracle.write_fastq(fb, names, cseq, qual, meta)
    want = oracle.recalibrate_fastq_text([fa, fb], True)[0]
    r = _run_ranks(world, ['recalibrate', '-f', fa, fb, '--infer-rg'])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert r.stdout.decode() == want


@pytest.mark.parametrize('name,world', [('c1_10k_1rg', 2), ('c3cut_2k_8rg', 3), ('c5cut_2k_mixed', 2)])
def test_ranks_write_their_own_files(dev, oracle, name, world, tmp_path):
    """`kbbq recalibrate -o FILE` under torch.distributed.run: every rank cuts, indexes and scans its own byte range of
    the two files (no rank reads a whole file), tallies it, and writes FILE.rankNNNN at the same time; the files
    concatenated in rank order are the reference's output, and nothing goes to stdout."""
    import glob, os
    from conftest import load_golden
    from test_gpu_parity import _files
    info, _ = load_golden(name)
    fa, fb = _files(oracle, info, tmp_path)
    out = str(tmp_path / 'out.fq')
    argv = ['recalibrate', '-f', fa, fb, '-o', out] + (['--infer-rg'] if info['case']['infer_rg'] else [])
    env_timing = dict(KBBQ_TIMING='1')
    os.environ.update(env_timing)
    try:
        r = _run_ranks(world, argv)
    finally:
        os.environ.pop('KBBQ_TIMING')
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    assert r.stdout == b''
    parts = sorted(glob.glob(out + '.rank*'))
    assert len(parts) == world and all(os.path.getsize(p) > 0 for p in parts)
    text = b''.join(open(p, 'rb').read() for p in parts).decode()
    assert len(text) == info['output_len'] and oracle.sha256(text) == info['output_sha256']
    # the stage report of every rank names its own byte range, never a whole-file scan
    err = r.stderr.decode()
    assert err.count('open+index+scan (own byte range)') == world and 'open+index+scan (wait)' not in err


def test_two_ranks_agree_on_the_first_error(dev, oracle, tmp_path):
    """A quality above 42 in the second rank's shard: every rank raises the IndexError (no rank is left in the
    allreduce), nothing is printed."""
    from conftest import load_golden
    from test_gpu_parity import _files
    info, _ = load_golden('c1_10k_1rg')
    fa, fb = _files(oracle, info, tmp_path)
    lines = open(fa).read().split('\n')
    rec = 4 * 7777                                             # read 7777 lives on rank 1
    lines[rec + 3] = lines[rec + 3][:20] + chr(33 + 50) + lines[rec + 3][21:]
    open(fa, 'w').write('\n'.join(lines))
    r = _run_ranks(2, ['recalibrate', '-f', fa, fb], timeout=200)
    assert r.returncode != 0 and r.stdout == b''
    assert r.stderr.decode().count('IndexError') >= 2


@pytest.mark.parametrize('S,nrg,pairs', [(150, 8, False), (150, 8, True), (100, 3, True), (150, 1, True), (60, 16, False),
                                         (150, 40, True)])
def test_rows_grouped_by_read_group(dev, oracle, S, nrg, pairs):
    """Rows ordered by read group + segment table: same tables, same qualities (after ungroup) as the plain path."""
    import torch
    n = 20000
    b = dev.ReadBatch.synthetic(0, n, n, seed=11 + nrg, len_lo=S, len_hi=S, nrg=nrg)
    t0 = dev.Tables(nrg, 2 * S); dev.accumulate(b, t0)
    lut, shape, _, _ = dev.solve(t0)
    want = dev.apply(b, lut, shape)
    src = dev.PairBatch.from_reads(b) if pairs else b
    g = dev.group_by_rg(src, nrg)
    seg = g.seg.cpu().numpy()
    assert seg[0] == 0 and seg[-1] == src.n and (np.diff(seg) >= 0).all()
    rg = ((g.meta[:g.n].cpu().numpy().view(np.uint32) >> 16) & 0x7FFF)
    assert (np.diff(rg.astype(np.int64)) >= 0).all()
    t1 = dev.Tables(nrg, 2 * S); dev.accumulate(g, t1)
    assert torch.equal(t0.buf, t1.buf)
    out = dev.ungroup(g, dev.apply(g, lut, shape))
    if pairs:
        out = src.unpack(out)
    assert torch.equal(out[:n], want[:n])
    # a row outside its group's segment is reported, never silently mis-scored
    if nrg > 1 and not pairs:
        g.meta[0] = g.meta[0] ^ (1 << 16)
        from kbbq import _native as N
        with pytest.raises(N.LutNeedsCheckedApply):
            dev.apply(g, lut, shape)


def test_grouped_rows_with_empty_groups_and_tiny_batches(dev):
    """Read groups without rows (empty segments) and batches smaller than one wave."""
    import torch
    S = 50
    for n, R, keep in ((2, 1, None), (64, 6, (0, 3)), (130, 4, (1,)), (4000, 9, (2, 5, 8))):
        b = dev.ReadBatch.synthetic(0, n, n, seed=n, len_lo=S, len_hi=S, nrg=R)
        if keep is not None:
            # rewrite the read groups so that only `keep` occur (pairs still share theirs)
            meta = b.meta[:n].cpu().numpy().view(np.uint32)
            pair = np.arange(n) >> 1
            rg = np.array(keep, dtype=np.uint32)[pair % len(keep)]
            b.meta[:n] = torch.from_numpy(((meta & 0x8000FFFF) | (rg << 16)).view(np.int32)).cuda()
        t0 = dev.Tables(R, 2 * S); dev.accumulate(b, t0)
        lut, shape, _, _ = dev.solve(t0)
        want = dev.apply(b, lut, shape)
        for src in (b, dev.PairBatch.from_reads(b)):
            g = dev.group_by_rg(src, R)
            t1 = dev.Tables(R, 2 * S); dev.accumulate(g, t1)
            assert torch.equal(t0.buf, t1.buf), (n, R)
            out = dev.ungroup(g, dev.apply(g, lut, shape))
            if src is not b:
                out = src.unpack(out)
            assert torch.equal(out[:n], want[:n]), (n, R)


def test_randomised_layout_sweep_against_the_oracle(dev, oracle):
    """Seeded sweep over read lengths (around the 16-byte chunk and pitch boundaries), read-group counts and
    thresholds: every layout (rows, mate-pair rows, each grouped by read group) gives the oracle's tables and
    qualities."""
    import torch
    rng = np.random.default_rng(2024)
    lengths = [15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 100, 127, 128, 129, 150, 151, 199, 200]
    for S in lengths:
        nrg = int(rng.integers(1, 7))
        minscore = int(rng.choice([6, 6, 6, 10, 15]))
        n = int(rng.integers(300, 1500)) * 2
        b = dev.ReadBatch.synthetic(0, n, n, seed=S, len_lo=S, len_hi=S, nrg=nrg, qlo=int(rng.integers(0, 8)), qhi=int(rng.integers(30, 43)))
        meta = b.meta[:n].cpu().numpy().view(np.uint32)
        host = [x.cpu().numpy() for x in (b.seq[:n], b.cseq[:n], b.qual[:n])]
        want = oracle.accumulate(host[0], host[1], host[2], meta, nrg, S, minscore=minscore)
        dqs = oracle.get_delta_qs(*want)
        want_q = oracle.apply(host[0], host[2], meta, want[0], *dqs, minscore=minscore)
        layouts = [('rows', b)]
        if dev.PairBatch.worthwhile(S, b.pitch) or S in (16, 64, 128):
            layouts.append(('pairs', dev.PairBatch.from_reads(b)))
        for name, src in list(layouts):
            layouts.append((name + '+grouped', dev.group_by_rg(src, nrg)))
        for name, lay in layouts:
            t = dev.Tables(nrg, 2 * S)
            try:
                dev.accumulate(lay, t, minscore)
            except dev.N.LutNeedsCheckedApply:
                assert S >= 199 and name != 'rows', (S, name)      # longer than the table-driven K1 holds: only the plain rows have a fallback
                continue
            for got, w in zip(t.to_host(), want[5:9]):
                assert np.array_equal(got, w), (S, nrg, name)
            lut, shape, _, _ = dev.solve(t, minscore=minscore)
            out = dev.apply(lay, lut, shape, minscore=minscore)
            if getattr(lay, 'seg', None) is not None:
                out = dev.ungroup(lay, out)
            if isinstance(lay, dev.PairBatch):
                out = lay.unpack(out, b.pitch)
            got_q = out[:n].cpu().numpy()
            assert np.array_equal(got_q[:, :S].astype(np.int32) - 33, want_q[:, :S]), (S, nrg, name)   # the oracle returns raw qualities
            assert not got_q[:, S:].any()


def test_length_bands_of_a_mixed_length_input(dev, oracle):
    """Reads of 36..300 bases in ascending order (the only order the reference accepts), tallied and
    applied band by band at each band's own pitch -- the table-driven K1 laid out for the band's longest read --
    give the tables and qualities of the whole batch at the widest pitch."""
    import torch
    n, nrg = 6000, 2
    b = dev.ReadBatch.synthetic(0, n, n, seed=5, len_lo=36, len_hi=300, nrg=nrg)
    S = 300
    meta = b.meta[:n].cpu().numpy().view(np.uint32)
    lens = (meta & 0xFFFF).astype(np.int64)
    assert (np.diff(lens) >= 0).all() and lens.max() == S
    whole = dev.Tables(nrg, 2 * S); dev.accumulate(b, whole)
    lut, shape, _, _ = dev.solve(whole)
    want = dev.apply(b, lut, shape)[:n].cpu().numpy()
    host = [x[:n].cpu().numpy() for x in (b.seq, b.cseq, b.qual)]
    banded = dev.Tables(nrg, 2 * S)
    outs = []
    edges = [0] + [int(np.searchsorted(lens, c, side='right')) for c in (48, 64, 96, 128, 160, 208, 256)] + [n]
    used = 0
    for lo, hi in zip(edges[:-1], edges[1:]):
        if hi <= lo:
            continue
        smax, smin = int(lens[lo:hi].max()), int(lens[lo:hi].min())
        pitch = (smax + 15) // 16 * 16
        band = dev.ReadBatch.from_host(np.ascontiguousarray(host[0][lo:hi, :pitch]), np.ascontiguousarray(host[2][lo:hi, :pitch]),
                                       meta[lo:hi], cseq=np.ascontiguousarray(host[1][lo:hi, :pitch]))
        ref = dev.Tables(nrg, 2 * S)
        dev.accumulate(band, ref)                             # the same band without the hints: first-generation kernel
        for lay in (band, dev.group_by_rg(band, nrg)):
            for hint in (0, smin):                            # long bands fit the table-driven kernel only with s_min
                t = dev.Tables(nrg, 2 * S)
                try:
                    dev.accumulate(lay, t, s_band=smax, s_min=hint)
                except dev.N.LutNeedsCheckedApply:
                    # grouped rows have no first-generation fallback: without the promise the long bands do not fit
                    assert smax > 200 and lay is not band and hint == 0
                    continue
                assert torch.equal(t.buf, ref.buf), (lo, hi, hint)
        dev.accumulate(band, banded, s_band=smax, s_min=smin)
        out = dev.apply(band, lut, shape)[:hi - lo].cpu().numpy()
        assert np.array_equal(out, want[lo:hi, :pitch]) and not want[lo:hi, pitch:].any()
        used += 1
    assert used >= 6 and torch.equal(banded.buf, whole.buf)
    # a read longer than the band it was put in is reported, not mis-binned
    t = dev.Tables(nrg, 2 * S)
    short = dev.ReadBatch.from_host(np.ascontiguousarray(host[0][-8:, :304]), np.ascontiguousarray(host[2][-8:, :304]),
                                    meta[-8:], cseq=np.ascontiguousarray(host[1][-8:, :304]))
    with pytest.raises(IndexError):
        dev.accumulate(short, t, s_band=150)
    # ... and so is a read shorter than the band's promised minimum, when the promise was needed to fit the tables
    mixed = np.r_[np.arange(0, 8), np.arange(n - 64, n)]       # 8 of the shortest reads among 300-base ones
    odd = dev.ReadBatch.from_host(np.ascontiguousarray(host[0][mixed, :304]), np.ascontiguousarray(host[2][mixed, :304]),
                                  meta[mixed], cseq=np.ascontiguousarray(host[1][mixed, :304]))
    with pytest.raises(IndexError):
        dev.accumulate(dev.group_by_rg(odd, nrg), dev.Tables(nrg, 2 * S), s_band=300, s_min=int(lens[-64]))


@pytest.mark.parametrize('name', ['bench_a', 'bench_b'])
def test_two_ranks_print_the_reference_benchmark(dev, oracle, name, tmp_path):
    """`kbbq benchmark` under torch.distributed.run with 2 ranks: alignments (FASTQ reads) sharded, the per-quality
    counts summed with one allreduce, rank 0 prints the single-process (= reference) table."""
    import oracle_benchmark as OB
    from conftest import load_golden
    info, _ = load_golden(name)
    paths = OB.synth_truthset(str(tmp_path), **info['case'])
    for tag, extra in (('bam', []), ('fastq', ['-f', paths['fq']])):
        argv = ['benchmark', '-b', paths['sam'], '-r', paths['fa'], '-v', paths['vcf'], '-d', paths['bed'], '-l', 'lbl'] + extra
        r = _run_ranks(2, argv, env={'KBBQ_TIMING': '1'})
        assert r.returncode == 0, r.stderr.decode()[-2000:]
        assert r.stdout.decode() == info['printed'][tag], tag
        if tag == 'fastq':
            # every rank flags (K4) only the alignments its FASTQ shard maps to (+ its own share): about half each, not all
            import re
            seen = re.findall(r'rank (\d) of 2 counts FASTQ reads \[\d+, \d+\) and flagged alignments \[\d+, \d+\): (\d+) of (\d+)', r.stderr.decode())
            assert sorted(x[0] for x in seen) == ['0', '1'], r.stderr.decode()[-1500:]
            total = int(seen[0][2])
            assert all(int(k) <= 0.6 * total for _, k, _ in seen) and sum(int(k) for _, k, _ in seen) >= total


def test_two_ranks_tally_alignments_like_one(dev, oracle, tmp_path):
    """kbbq.gatk.bqsr.bam_to_bqsr_covariates on 2 ranks (K4 -> K6 -> K1 on shards of the alignments, one allreduce of
    the tables): the nine vectors of the reference golden; a read of another length stops every rank."""
    import json, os, socket, subprocess, sys
    from conftest import ROOT
    from test_oracle_bqsr import VEC, _inputs

    def run(paths, native, out):
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        env = dict(os.environ, KBBQ_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
               '--master-port', str(port), os.path.join(ROOT, 'tests', 'dist_api_worker.py'),
               paths['sam'], paths['fa'], paths['vcf'], native, out]
        return subprocess.run(cmd, env=env, capture_output=True, timeout=300)
    info, gold, paths = _inputs('bqsr_a', tmp_path, oracle)
    for native in ('1', '0'):
        out = str(tmp_path / ('vec%s.json' % native))
        r = run(paths, native, out)
        assert r.returncode == 0, r.stderr.decode()[-8000:]
        got = json.load(open(out))['vectors']
        for k, g in zip(VEC, got):
            assert np.array_equal(np.array(g, dtype=np.int64), gold[k]), (native, k)
    # the last alignment loses a base: the IndexError of the reference on every rank, after the reads before it
    lines = open(paths['sam']).read().rstrip('\n').split('\n')
    f = lines[-1].split('\t')
    f[5] = '%dM' % (len(f[9]) - 1); f[9] = f[9][:-1]; f[10] = f[10][:-1]
    lines[-1] = '\t'.join(x if not x.startswith('OQ:Z:') else x[:-1] for x in f)
    open(paths['sam'], 'w').write('\n'.join(lines) + '\n')
    out = str(tmp_path / 'err.json')
    r = run(paths, '1', out)
    assert r.returncode != 0 and json.load(open(out)) == {'error': 'IndexError'}
    assert r.stderr.decode().count('IndexError') >= 2


def test_planes_streamed_from_the_reader_equal_the_host_packed_ones(dev, oracle, tmp_path):
    """ReadBatch.from_reader (page-locked slabs, uploads overlapping the packer) against the host planes of
    NativeFastq.fill, for slab sizes that do and do not divide the batch, with and without the corrected file,
    a shard that starts in the middle, and the buffers handed on to the output pipeline afterwards."""
    import torch
    from kbbq import fastx
    n = 10000
    seq, cseq, qual, meta = oracle.synth(0, n, n, 4, 36, 151, 3)
    names = oracle.synth_names(0, n, 3, with_rg=True)
    fa, fb = str(tmp_path / 'a.fq'), str(tmp_path / 'b.fq')
    oracle.write_fastq(fa, names, seq, qual, meta)
    oracle.write_fastq(fb, names, cseq, qual, meta)
    A, B = fastx.NativeFastq(fa), fastx.NativeFastq(fb)
    assert A.scan(B, True)[0] == n
    for other in (B, None):
        for first, m, slab in ((0, n, 1 << 17), (0, n, 999), (3001, 5000, 1000), (n - 1, 1, 7), (5, 0, 64)):
            want = A.fill(other, True, m, 160, first=first)
            got = dev.ReadBatch.from_reader(A, other, True, first, m, 160, slab=slab)
            assert got.n == m and (got.cseq is None) == (other is None)
            pairs = [(got.seq, want[0]), (got.qual, want[2]), (got.meta, want[3].view(np.int32))]
            if other is not None:
                pairs.append((got.cseq, want[1]))
            for g, w in pairs:
                assert np.array_equal(g[:m].cpu().numpy(), w), (first, m, slab)
    assert not [k for k in dev._pinned if k[0] == 'ingest']          # released for the next user
    assert any(k[0] == '' for k in dev._pinned)


def test_two_ranks_when_pass_2_covers_more_than_pass_1(dev, oracle, tmp_path):
    """File B shorter than file A (the reference's zip() truncation: pass 1 stops, pass 2 prints all of A) and the
    -g model file (saved by one run, loaded by the next, file B not read): 2 ranks print what 1 rank prints.  Pass 1's
    shard readers hold only their byte ranges, so pass 2 has to open file A again."""
    from conftest import load_golden
    from test_gpu_parity import _files
    info, _ = load_golden('c1_10k_1rg')
    fa, fb = _files(oracle, info, tmp_path)
    lines = open(fb).read().split('\n')
    short = str(tmp_path / 'b_short.fq')
    open(short, 'w').write('\n'.join(lines[:4 * 3001]) + '\n')
    model = str(tmp_path / 'model.txt')
    for argv in (['recalibrate', '-f', fa, short], ['recalibrate', '-f', fa, fb, '-g', model], ['recalibrate', '-f', fa, fb, '-g', model]):
        one = _run_ranks(1, argv)
        two = _run_ranks(2, argv)
        assert one.returncode == 0 and two.returncode == 0, (one.stderr.decode()[-1500:], two.stderr.decode()[-1500:])
        assert len(one.stdout) > 1000 and two.stdout == one.stdout, argv
    assert os.path.getsize(model) > 1000


@pytest.mark.parametrize('nrg', [1, 3])
def test_a_lut_the_table_driven_kernels_cannot_serve_takes_checked_rows(dev, oracle, nrg, tmp_path, monkeypatch):
    """The file path with a model whose apply LUT is not range-safe for the table-driven kernels (a row whose smallest cycle
    entry + smallest context entry would leave 0..255: the blob's flags say so, shape mode APPLY_CHECKED): the layout the
    packer wrote cannot be applied, the band is filled again as one character row per read from the text and the checked
    kernel serves it -- same bytes as the oracle's apply with those tables.  (A solved model is always range-safe -- the
    prior forbids a posterior further than 18 from its prior at every level -- so the model is put in place of the solve.)"""
    from kbbq import recalibrate
    n, S = 3000, 150
    seq, cseq, qual, meta = oracle.synth(0, n, n, 23, S, S, nrg, 10, 41)
    names = oracle.synth_names(0, n, nrg, with_rg=nrg > 1)
    fa, fb, fo = (str(tmp_path / x) for x in ('a.fq', 'b.fq', 'out.fq'))
    oracle.write_fastq(fa, names, seq, qual, meta)
    oracle.write_fastq(fb, names, cseq, qual, meta)
    rng = np.random.default_rng(3)
    meanq = rng.integers(15, 25, nrg); rgdq = rng.integers(-2, 3, nrg); qdq = rng.integers(-3, 4, (nrg, 43))
    posdq = rng.integers(-6, 7, (nrg, 43, 2 * S)); ddq = rng.integers(-6, 7, (nrg, 43, 17)); ddq[..., 16] = 0
    posdq[:, 7, :] = -70                                       # a quality no read carries (10..41): only the row's range check sees it
    lut, shp = dev.build_lut(meanq, rgdq, qdq, posdq, ddq)
    assert shp[3] == dev.N.APPLY_CHECKED
    d_lut = dev.lut_to_device(lut)
    monkeypatch.setattr(dev, 'solve_lut', lambda tables, **k: (d_lut, shp))
    recalibrate.recalibrate_fastq([fa, fb], infer_rg=nrg > 1, output=fo)
    ref = oracle.apply(seq, qual, meta, meanq, rgdq, qdq, posdq, ddq)
    want = ''.join('@%s\n%s\n+\n%s\n' % (names[i], bytes(seq[i, :S]).decode(), bytes((ref[i, :S] + 33).astype(np.uint8)).decode()) for i in range(n))
    assert open(fo).read() == want
    bands = recalibrate.LAST_RUN['bands']
    assert all(b['output_rows'] == 'one read per row' for b in bands)
    # and with the model the path solves itself the same command writes mate-pair rows
    monkeypatch.undo()
    recalibrate.recalibrate_fastq([fa, fb], infer_rg=nrg > 1, output=fo)
    assert open(fo).read() == oracle.recalibrate_fastq_text([fa, fb], nrg > 1)[0]
    assert all('mate-pair rows' in b['output_rows'] and 'kbbq_fastq_fill_rows' in b['written_by'] for b in recalibrate.LAST_RUN['bands'])
