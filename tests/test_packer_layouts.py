"""The packer writes the device layout itself (kbbq_fastq_meta / kbbq_group_rows_host / kbbq_fastq_fill_rows /
kbbq_fastq_format_rows): checked on the CPU against a NumPy statement of the layout contract of include/kbbq_hip.h
(mate-pair rows, 4-bit sequence planes, rows gathered by read-group segment) applied to the character rows the
row-per-read packer (kbbq_fastq_fill_range) writes.  The GPU twin of this test (tests/test_gpu_layouts.py) compares the
same bytes with what kbbq_lay_out_dev makes of those rows."""
import numpy as np
import pytest

from kbbq import _native as N
from kbbq import fastx

CODE = {ord('A'): 0, ord('T'): 1, ord('G'): 2, ord('C'): 3, ord('N'): 4}


def np_nibbles(chars):
    """[rows, pitch] characters -> [rows, pitch / 2] code nibbles (include/kbbq_hip.h KBBQ_ROWS_NIBBLES)."""
    lut = np.full(256, 255, dtype=np.uint8)
    for ch, c in CODE.items():
        lut[ch] = c
    codes = lut[chars]
    assert codes.max(initial=0) <= 4
    rows, pitch = chars.shape
    c = codes.reshape(rows, pitch // 16, 2, 2, 4)             # chunk, word, low / high half, byte
    return (c[:, :, :, 0, :] | (c[:, :, :, 1, :] << 4)).reshape(rows, pitch // 2)


def np_layout(seq, cseq, qual, meta, flags, S, perm):
    """What kbbq_lay_out_dev makes of input-order character rows, in NumPy."""
    n, pitch = seq.shape
    pairs, nib = bool(flags & N.ROWS_PAIRS), bool(flags & N.ROWS_NIBBLES)
    if pairs:
        nrows = (n + 1) // 2
        dp = (2 * S + 1 + 15) // 16 * 16

        def two(plane, fill):
            out = np.full((nrows, dp), fill, dtype=np.uint8)
            out[:, :S] = plane[0::2, :S]
            out[:n // 2, S + 1:2 * S + 1] = plane[1::2, :S]
            return out
        dseq, dcseq, dqual = two(seq, ord('N')), None if cseq is None else two(cseq, ord('N')), two(qual, 0)
        dmeta = (np.uint32(2 * S + 1) | (meta[0::2] & np.uint32(0x7FFF0000))).astype(np.uint32)
    else:
        dseq, dcseq, dqual, dmeta = seq, cseq, qual, meta
    if perm is not None:
        dseq, dqual, dmeta = dseq[perm], dqual[perm], dmeta[perm]
        dcseq = None if dcseq is None else dcseq[perm]
    if nib:
        dseq, dcseq = np_nibbles(dseq), None if dcseq is None else np_nibbles(dcseq)
    return dseq, dcseq, dqual, dmeta


def _pair_files(tmp_path, rng, n, S, nrg, single_end=False, foreign=False, ragged=False):
    acgt = np.array(list('ACGTN'))
    ra, rb = [], []
    for i in range(n):
        L = S if not ragged else int(rng.integers(max(1, S - 20), S + 1))
        seq = ''.join(rng.choice(acgt, L, p=[.24, .24, .24, .24, .04]))
        if foreign and i == n // 3:
            seq = seq[:L // 2] + 'a' + seq[L // 2 + 1:]
        cseq = ''.join(c if rng.random() > 0.05 else str(rng.choice(acgt[:4])) for c in seq)
        qual = ''.join(chr(33 + int(q)) for q in rng.integers(0, 42, L))
        name = ('s%d' % i) if single_end else 'r%d/%d' % (i // 2, 1 + (i & 1))
        if nrg > 1:
            name += '_RG:Z:g%d' % (((i // 2) * 7) % nrg)
        ra.append((name, seq, qual)); rb.append((name, cseq, qual))
    if ragged:
        order = np.argsort([len(r[1]) for r in ra], kind='stable')
        ra, rb = [ra[i] for i in order], [rb[i] for i in order]
    fa, fb = tmp_path / 'a.fq', tmp_path / 'b.fq'
    for p, recs in ((fa, ra), (fb, rb)):
        p.write_text(''.join('@%s\n%s\n+\n%s\n' % r for r in recs))
    return str(fa), str(fb)


@pytest.mark.parametrize('n,S,nrg,single_end', [(40, 150, 1, False), (64, 37, 3, False), (33, 150, 1, True), (50, 16, 2, True),
                                                (2, 1, 1, False), (257, 75, 5, False)])
def test_fill_rows_writes_what_the_layout_pass_would(tmp_path, n, S, nrg, single_end):
    rng = np.random.default_rng(n * 1000 + S)
    fa, fb = _pair_files(tmp_path, rng, n, S, nrg, single_end)
    infer = nrg > 1
    A, B, info = fastx.PairScan(fa, fb, infer).result()
    assert info[0] == n and info[3] == 0
    R = info[2]
    pitch = fastx.pitch_for(S)
    seq, cseq, qual, meta = A.fill(B, infer, n, pitch)
    hmeta, st = A.meta(infer, 0, n)
    assert np.array_equal(hmeta, meta)
    lens = meta & 0xFFFF
    assert st['longest'] == int(lens.max()) and st['shortest'] == int(lens[lens > 0].min()) and st['empty'] == 0
    assert st['max_rg'] == int(((meta >> 16) & 0x7FFF).max())
    assert (st['pair_violations'] == 0) == (not single_end and n % 2 == 0)
    assert (st['twin_violations'] == 0) == single_end
    from kbbq import _device as dev
    for packed in (True, False):
        flags = dev.layout_flags(st, n, pitch, packed=packed)
        two = bool(flags & N.ROWS_PAIRS)
        assert two == ((2 * S + 1 + 15) // 16 * 16 < 2 * pitch)
        nrows = (n + 1) // 2 if two else n
        perm = seg = None
        if R > 1:
            perm, seg = np.empty(nrows, dtype=np.int64), np.empty(R + 1, dtype=np.int64)
            N.check(N.load().kbbq_group_rows_host(N.ptr(meta), nrows, 1 if two else 0, R, N.ptr(perm), N.ptr(seg)))
            key = ((meta[0::2] if two else meta) >> 16) & 0x7FFF
            assert np.array_equal(perm, np.argsort(key, kind='stable'))
            assert np.array_equal(seg, np.concatenate([[0], np.cumsum(np.bincount(key, minlength=R))]))
        want = np_layout(seq, cseq, qual, meta, flags, S, perm)
        dp = want[2].shape[1]
        sp = dp // 2 if flags & N.ROWS_NIBBLES else dp
        # in two slabs, to exercise row_lo
        got = [np.zeros((nrows, sp), np.uint8), np.zeros((nrows, sp), np.uint8), np.zeros((nrows, dp), np.uint8), np.zeros(nrows, np.uint32)]
        cut = nrows // 3
        for lo, m in ((0, cut), (cut, nrows - cut)):
            foreign = A.fill_rows(B, 0, n, meta, flags, 2 * S, dp, perm, lo, m, got[0][lo:lo + m], got[1][lo:lo + m], got[2][lo:lo + m], got[3][lo:lo + m])
            assert not foreign
        for g, w, name in zip(got, want, ('seq', 'cseq', 'qual', 'meta')):
            assert np.array_equal(g, w), (name, flags)
        # no corrected file: the same rows without a cseq plane
        only = [np.zeros((nrows, sp), np.uint8), np.zeros((nrows, dp), np.uint8), np.zeros(nrows, np.uint32)]
        assert not A.fill_rows(None, 0, n, meta, flags, 2 * S, dp, perm, 0, nrows, only[0], None, only[1], only[2])
        assert np.array_equal(only[0], want[0]) and np.array_equal(only[1], want[2]) and np.array_equal(only[2], want[3])
        # the writer reads new qualities out of the rows K2 writes them in: input-order rows of the same layout
        unperm = np_layout(seq, cseq, qual, meta, flags & ~N.ROWS_NIBBLES, S, None)[2]
        newq = unperm.copy()
        newq[newq != 0] = (newq[newq != 0] - 33 + 7) % 60 + 33
        rows = qual.copy()
        rows[rows != 0] = (rows[rows != 0] - 33 + 7) % 60 + 33
        text = A.format_rows_array(0, n, newq, flags & N.ROWS_PAIRS, 2 * S).tobytes()
        assert text == A.format(0, n, rows)
        if n >= 6:
            lo = 2
            sub = newq[lo // 2:] if two else newq[lo:]
            assert A.format_rows_array(lo, n - lo - 1, sub, flags & N.ROWS_PAIRS, 2 * S).tobytes() == A.format(lo, n - lo - 1, rows[lo:])


def test_fill_rows_reports_a_foreign_letter_and_mixed_lengths(tmp_path):
    rng = np.random.default_rng(5)
    fa, fb = _pair_files(tmp_path, rng, 20, 40, 1, foreign=True)
    A, B, info = fastx.PairScan(fa, fb, False).result()
    meta, st = A.meta(False, 0, 20)
    dp = (2 * 40 + 1 + 15) // 16 * 16
    bufs = lambda sp: (np.zeros((10, sp), np.uint8), np.zeros((10, sp), np.uint8), np.zeros((10, dp), np.uint8), np.zeros(10, np.uint32))
    s, c, q, m = bufs(dp // 2)
    assert A.fill_rows(B, 0, 20, meta, N.ROWS_PAIRS | N.ROWS_NIBBLES, 80, dp, None, 0, 10, s, c, q, m)      # 'a' is not a nucleotide code
    s, c, q, m = bufs(dp)
    assert not A.fill_rows(B, 0, 20, meta, N.ROWS_PAIRS, 80, dp, None, 0, 10, s, c, q, m)                   # character planes take it
    assert (s == ord('a')).sum() == 1
    # reads of several lengths: no pair rows (the statistics say so), nibble rows at the band's pitch
    d2 = tmp_path / 'r'; d2.mkdir()
    fa, fb = _pair_files(d2, rng, 30, 60, 1, ragged=True)
    A, B, info = fastx.PairScan(fa, fb, False).result()
    assert info[3] == 0
    meta, st = A.meta(False, 0, 30)
    assert st['pair_violations'] > 0 and st['twin_violations'] > 0
    with pytest.raises(ValueError):
        A.fill_rows(B, 0, 30, meta, N.ROWS_PAIRS, 120, 0, None, 0, 15, *bufs(dp))
    pitch = fastx.pitch_for(st['longest'])
    seq, cseq, qual, fmeta = A.fill(B, False, 30, pitch)
    got = (np.zeros((30, pitch // 2), np.uint8), np.zeros((30, pitch // 2), np.uint8), np.zeros((30, pitch), np.uint8), np.zeros(30, np.uint32))
    assert not A.fill_rows(B, 0, 30, meta, N.ROWS_NIBBLES, 2 * st['longest'], pitch, None, 0, 30, *got)
    want = np_layout(seq, cseq, qual, fmeta, N.ROWS_NIBBLES, st['longest'], None)
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    # a sub-range of the file as its own band (first > 0)
    sub_meta, sub_st = A.meta(False, 10, 12)
    assert np.array_equal(sub_meta, fmeta[10:22])
    got = (np.zeros((12, pitch // 2), np.uint8), np.zeros((12, pitch // 2), np.uint8), np.zeros((12, pitch), np.uint8), np.zeros(12, np.uint32))
    assert not A.fill_rows(B, 10, 12, sub_meta, N.ROWS_NIBBLES, 2 * st['longest'], pitch, None, 0, 12, *got)
    assert np.array_equal(got[0], want[0][10:22]) and np.array_equal(got[2], want[2][10:22])


def test_group_rows_host_refuses_a_group_beyond_r():
    meta = np.array([150 | (3 << 16), 150 | (1 << 16)], dtype=np.uint32)
    perm, seg = np.empty(2, dtype=np.int64), np.empty(3, dtype=np.int64)
    with pytest.raises(ValueError):
        N.check(N.load().kbbq_group_rows_host(N.ptr(meta), 2, 0, 2, N.ptr(perm), N.ptr(seg)))


def test_scalar_twins_of_the_simd_host_code():
    """The packer's 4-bit packing (SSSE3) has a scalar twin for other hosts: KBBQ_NO_SIMD=1 in a fresh process runs the layout and reader
    tests on it."""
    import os, subprocess, sys
    if os.environ.get('KBBQ_NO_SIMD'):
        pytest.skip('already the scalar run')
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, '-m', 'pytest', '-q', '-x', os.path.join(here, 'test_packer_layouts.py'), os.path.join(here, 'test_host_logic.py'),
                        '-k', 'fill_rows or fastq or pack_pair or reader or packer'], env=dict(os.environ, KBBQ_NO_SIMD='1'), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    assert ' passed' in r.stdout
