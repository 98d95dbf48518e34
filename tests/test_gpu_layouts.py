"""
GPU tests (-m gpu) of the native layout passes (csrc/kbbq_layout_kernels.h, include/kbbq_hip.h "layouts in one
interface"): sidecar statistics, the counting sort by read group, the one-pass lay_out (pair packing + gather by
read-group segment + 4-bit sequence planes) and K1 / K2 on what it writes, K2's store through the permutation.
Everything is compared with the CPU oracle on the same seeded reads (bit-exact: integer / byte work).
"""
import numpy as np
import pytest

from test_gpu_parity import dev                      # noqa: F401  (fixture)

pytestmark = pytest.mark.gpu


def _host(b, n):
    return [x[:n].cpu().numpy() for x in (b.seq, b.cseq, b.qual)] + [b.meta[:n].cpu().numpy().view(np.uint32)]


def _oracle_run(oracle, seq, cseq, qual, meta, R, S, minscore=6):
    want = oracle.accumulate(seq, cseq, qual, meta, R, S, minscore=minscore)
    dqs = oracle.get_delta_qs(*want)
    ref = oracle.apply(seq, qual, meta, want[0], *dqs, minscore=minscore)
    return want, ref


def test_meta_stats(dev):
    import torch
    b = dev.ReadBatch.synthetic(0, 5000, 5000, seed=5, len_lo=36, len_hi=150, nrg=7)
    meta = b.meta[:5000].cpu().numpy().view(np.uint32)
    lens = (meta & 0xFFFF).astype(int)
    st = dev.meta_stats(b)
    assert st['shortest'] == lens[lens > 0].min() and st['longest'] == lens.max()
    assert st['max_rg'] == int(((meta >> 16) & 0x7FFF).max()) and st['empty'] == int((lens == 0).sum())
    assert st['pair_violations'] > 0                       # lengths differ
    u = dev.ReadBatch.synthetic(0, 4000, 4000, seed=6, nrg=3)
    assert dev.meta_stats(u)['pair_violations'] == 0
    good = u.meta[:4000].cpu().numpy().view(np.uint32).copy()

    def with_meta(m):
        u.meta[:4000].copy_(torch.from_numpy(m.view(np.int32)))
        return dev.meta_stats(u)['pair_violations']
    m = good.copy(); m[1001] &= 0x7FFFFFFF                   # a "second" read that claims to be first
    assert with_meta(m) == 1
    m = good.copy(); m[2000] = (m[2000] & ~np.uint32(0x7FFF0000)) | np.uint32((((m[2000] >> 16) & 0x7FFF) + 1) % 3 << 16)
    assert with_meta(m) == 1                                 # mates in different read groups
    m = good.copy(); m[3000] = (m[3000] & ~np.uint32(0xFFFF)) | np.uint32(149)
    assert with_meta(m) == 1                                 # one read of another length
    assert with_meta(good) == 0
    e = dev.ReadBatch.synthetic(0, 7, 7, seed=1)
    assert dev.meta_stats(e)['pair_violations'] >= 1      # odd number of reads
    torch.cuda.synchronize()


@pytest.mark.parametrize('n,R,pairs', [(9000, 8, False), (9000, 8, True), (4096 * 3 + 17, 3, False), (70, 256, False),
                                       (200000, 5, True), (2, 1, True)])
def test_group_rows_is_the_stable_sort(dev, n, R, pairs):
    import torch
    n -= n % 2
    b = dev.ReadBatch.synthetic(0, n, n, seed=9 + R, nrg=R)
    nrows = n // 2 if pairs else n
    perm, seg = dev._group_perm(b.meta, nrows, pairs, R)
    meta = b.meta[:n].cpu().numpy().view(np.uint32)
    rg = ((meta[0::2] if pairs else meta) >> 16) & 0x7FFF
    want = np.argsort(rg, kind='stable')
    assert np.array_equal(perm.cpu().numpy(), want)
    assert np.array_equal(seg.cpu().numpy(), np.concatenate([[0], np.cumsum(np.bincount(rg, minlength=R))]))
    if int(rg.max()) > 0:
        with pytest.raises(ValueError):                      # a sidecar with a read group the caller did not announce
            dev._group_perm(b.meta, nrows, pairs, int(rg.max()))


@pytest.mark.parametrize('S,R,n', [(150, 1, 6002), (150, 8, 20000), (100, 3, 4000), (151, 2, 3000), (16, 2, 1280),
                                   (75, 1, 2000), (33, 5, 2000)])
def test_lay_out_tally_and_apply_match_the_oracle(dev, oracle, S, R, n):
    import torch
    b = dev.ReadBatch.synthetic(0, n, n, seed=21 + S + R, len_lo=S, len_hi=S, nrg=R)
    seq, cseq, qual, meta = _host(b, n)
    want, ref = _oracle_run(oracle, seq, cseq, qual, meta, R, S)
    laid = dev.lay_out(b, R, S, packed=True)
    pair_rows = dev.PairBatch.worthwhile(S, b.pitch)
    assert laid.nib and isinstance(laid, dev.PairBatch) == pair_rows and (laid.seg is not None) == (R > 1)
    # the planes: characters recovered from the nibbles equal the character layout of the same rows
    plain = dev.lay_out(b, R, S, packed=False)
    if plain is b:
        assert not pair_rows and R == 1
    for name in ('seq', 'cseq'):
        assert torch.equal(laid.chars(name)[:laid.n], getattr(plain, name)[:laid.n])
    assert torch.equal(laid.qual[:laid.n], plain.qual[:laid.n]) and torch.equal(laid.meta[:laid.n], plain.meta[:laid.n])
    # K1
    t = dev.Tables(R, 2 * S)
    dev.accumulate(laid, t)
    for got, w in zip(t.to_host(), want[5:9]):
        assert np.array_equal(got, w)
    # K3 -> K2, stored straight back into input order
    lut, shape, _, _ = dev.solve(t)
    out = dev.apply(laid, lut, shape, restore_order=True)
    if pair_rows:
        out = laid.unpack(out, b.pitch)
    assert np.array_equal(out[:n, :S].cpu().numpy().astype(np.int32) - 33, ref[:, :S])
    # ... and in grouped order + ungroup, the separate-pass form
    out2 = dev.apply(laid, lut, shape)
    if laid.seg is not None:
        out2 = dev.ungroup(laid, out2)
    if pair_rows:
        out2 = laid.unpack(out2, b.pitch)
    assert torch.equal(out2[:n], out[:n])


def test_lay_out_ragged_rows_keep_one_read_per_row(dev, oracle):
    n, R = 5000, 4
    b = dev.ReadBatch.synthetic(0, n, n, seed=77, len_lo=36, len_hi=150, nrg=R)
    seq, cseq, qual, meta = _host(b, n)
    want, ref = _oracle_run(oracle, seq, cseq, qual, meta, R, 150)
    laid = dev.lay_out(b, R, 150, packed=True)
    assert laid.nib and not isinstance(laid, dev.PairBatch) and laid.seg is not None
    t = dev.Tables(R, 300)
    dev.accumulate(laid, t)
    for got, w in zip(t.to_host(), want[5:9]):
        assert np.array_equal(got, w)
    lut, shape, _, _ = dev.solve(t)
    out = dev.apply(laid, lut, shape, restore_order=True)[:n].cpu().numpy().astype(np.int32)
    lens = (meta & 0xFFFF).astype(int)
    for i in (0, 1, 17, 999, n - 1):
        assert np.array_equal(out[i, :lens[i]] - 33, ref[i, :lens[i]]) and not out[i, lens[i]:].any()
    mask = np.arange(b.pitch)[None, :] < lens[:, None]
    assert np.array_equal((out - 33)[mask], ref[mask])


def test_a_base_outside_acgtn_keeps_character_planes(dev, oracle):
    """4-bit planes exist only for ACGTN batches: anything else stays in character planes, which carry the reference's
    TypeError rule (compare_reads.py:281-293) -- here a letter that is never looked up, so the run succeeds."""
    import torch
    n, S = 4000, 150
    b = dev.ReadBatch.synthetic(0, n, n, seed=4)
    b.seq[10, 5] = ord('X'); b.cseq[10, 5] = ord('X')
    b.qual[10, 5] = 33 + 2; b.qual[10, 6] = 33 + 3          # below minscore: neither (4,5) nor (5,6) is looked up
    seq, cseq, qual, meta = _host(b, n)
    want, ref = _oracle_run(oracle, seq, cseq, qual, meta, 1, S)
    laid = dev.lay_out(b, 1, S, packed=True)
    assert isinstance(laid, dev.PairBatch) and not laid.nib
    t = dev.Tables(1, 2 * S)
    dev.accumulate(laid, t)
    for got, w in zip(t.to_host(), want[5:9]):
        assert np.array_equal(got, w)
    # only the corrected read differs from ACGTN: still not packable (the comparison of codes would lose the letter)
    c = dev.ReadBatch.synthetic(0, n, n, seed=4)
    c.cseq[3, 7] = ord('a')
    assert not dev.lay_out(c, 1, S, packed=True).nib
    # a looked-up bad letter is the reference's TypeError, from the character planes
    d = dev.ReadBatch.synthetic(0, n, n, seed=4)
    d.seq[10, 5] = ord('X'); d.qual[10, 5] = 33 + 30; d.seq[10, 4] = ord('A')
    laid = dev.lay_out(d, 1, S, packed=True)
    assert not laid.nib
    with pytest.raises(TypeError):
        dev.accumulate(laid, dev.Tables(1, 2 * S))
    torch.cuda.synchronize()


def test_corrupt_nibble_planes_are_reported(dev):
    n, S = 2000, 150
    b = dev.ReadBatch.synthetic(0, n, n, seed=8)
    laid = dev.lay_out(b, 1, S, packed=True)
    assert laid.nib
    laid.seq[5, 3] = 0x7F                                    # nibbles 15 and 7: not codes
    with pytest.raises(dev.N.LutNeedsCheckedApply):
        dev.accumulate(laid, dev.Tables(1, 2 * S))


@pytest.mark.parametrize('S', [150, 100])                  # 19 and 13 chunks per mate-pair row: both chunk-position-major forms of K1
@pytest.mark.parametrize('minscore', [2, 6, 20])
def test_minscore_on_packed_rows(dev, oracle, minscore, S):
    n, R = 4000, 2
    b = dev.ReadBatch.synthetic(0, n, n, seed=31, nrg=R, qlo=2, len_lo=S, len_hi=S)
    seq, cseq, qual, meta = _host(b, n)
    want, ref = _oracle_run(oracle, seq, cseq, qual, meta, R, S, minscore=minscore)
    laid = dev.lay_out(b, R, S, packed=True)
    t = dev.Tables(R, 2 * S)
    dev.accumulate(laid, t, minscore)
    for got, w in zip(t.to_host(), want[5:9]):
        assert np.array_equal(got, w)
    lut, shape, _, _ = dev.solve(t, minscore=minscore)
    out = laid.unpack(dev.apply(laid, lut, shape, minscore=minscore, restore_order=True), b.pitch)
    assert np.array_equal(out[:n, :S].cpu().numpy().astype(np.int32) - 33, ref[:, :S])


def test_long_reads_measure_their_shortest_read_on_the_device(dev, oracle):
    """Plain accumulate() on 300-base reads: the shortest-read promise that lets K1's LDS tables fit is measured by
    k7_meta_stats instead of coming from the file path's length bands (no first-generation fallback)."""
    n, S = 3000, 300
    b = dev.ReadBatch.synthetic(0, n, n, seed=12, len_lo=280, len_hi=300)
    seq, cseq, qual, meta = _host(b, n)
    order = np.argsort(meta & 0xFFFF, kind='stable')         # the reference needs non-decreasing lengths
    want = oracle.accumulate(seq[order], cseq[order], qual[order], meta[order], 1, S)
    t = dev.Tables(1, 2 * S)
    ctx = dev.context()
    ctx.timing(True)
    dev.accumulate(b, t)
    ctx.timing(False)
    for got, w in zip(t.to_host(), want[5:9]):
        assert np.array_equal(got, w)


def test_large_packed_batch_properties(dev):
    """BASELINE-sized property checks that need no oracle: counts add up, output of the packed layout equals the
    character layout's, bytes past the reads stay zero."""
    import torch
    n, S = 4_000_000, 150
    b = dev.ReadBatch.synthetic(0, n, n, seed=1)
    laid = dev.lay_out(b, 1, S, packed=True)
    plain = dev.lay_out(b, 1, S, packed=False)
    t1, t2 = dev.Tables(1, 2 * S), dev.Tables(1, 2 * S)
    dev.accumulate(laid, t1); dev.accumulate(plain, t2)
    assert torch.equal(t1.buf, t2.buf)
    pe, pt, de, dt = t1.views()
    assert int(pt.sum()) == int(((b.qual[:n, :S] >= 33 + 6)).sum())
    lut, shape, _, _ = dev.solve(t1)
    assert torch.equal(dev.apply(laid, lut, shape), dev.apply(plain, lut, shape))


@pytest.mark.parametrize('n,R', [(30_000_000, 8), (26_000_000, 1), (6002, 3), (130, 2)])
def test_short_lived_k2_equals_the_persistent_k2(dev, n, R, monkeypatch):
    """K2 with short-lived workgroups (csrc/kbbq_k2_tile.h, the default on mate-pair rows with 4-bit planes) against the
    persistent kernel (KBBQ_K2_TILE=0) on the same batch: identical bytes, in grouped order and stored through the
    permutation.  The large sizes put chunk numbers beyond 2^28 (the chunk -> row division must be exact there) and
    every read group's run of workgroups to work."""
    import torch
    b = dev.ReadBatch.synthetic(0, n, n, seed=17, nrg=R)
    laid = dev.lay_out(b, R, 150, packed=True)
    assert laid.nib and isinstance(laid, dev.PairBatch)
    del b
    t = dev.Tables(R, 300)
    dev.accumulate(laid, t)
    lut, shape = dev.solve_lut(t)
    ctx = dev.context()
    for restore in (False, True):
        if restore and laid.seg is None:
            continue
        monkeypatch.setenv('KBBQ_K2_TILE', '0')
        want = dev.apply(laid, lut, shape, restore_order=restore)
        monkeypatch.delenv('KBBQ_K2_TILE')
        ctx.kernel_ms(1, reset=True); ctx.timing(True)
        got = dev.apply(laid, lut, shape, restore_order=restore)
        ctx.timing(False)
        assert torch.equal(got, want)
        del got, want


@pytest.mark.parametrize('n,S,R,lay', [(26_000_000, 150, 1, 'rows'), (6002, 150, 1, 'rows'), (130, 100, 1, 'rows'), (20_000, 75, 1, 'rows'),
                                       (8000, 150, 3, 'grouped'), (6000, 150, 1, 'pairs'), (6000, 100, 4, 'pairs+grouped'), (5000, 250, 1, 'rows')])
def test_short_lived_k2_on_character_planes(dev, oracle, n, S, R, lay, monkeypatch):
    """Round 4: the short-lived K2 on CHARACTER planes (k2t_apply<false>: rows as a caller holds them -- kbbq_apply_dev with one
    read group --, character rows grouped by read group, character mate-pair rows) against the persistent kernel
    (KBBQ_K2_TILE_CHARS=0) on the same batch: identical bytes, also through the permutation; at 26 M reads chunk numbers
    pass 2^28.  What it cannot serve it reports, and the checked kernel then decides as the reference does: a foreign letter
    is the TypeError, a quality above 42 the IndexError."""
    import torch
    b = dev.ReadBatch.synthetic(0, n, n, seed=23, len_lo=S, len_hi=S, nrg=R)
    t = dev.Tables(R, 2 * S)
    dev.accumulate(b, t)
    lut, shape = dev.solve_lut(t)
    src = b
    if 'pairs' in lay:
        src = dev.PairBatch.from_reads(b)
    if 'grouped' in lay:
        src = dev.group_by_rg(src, R)
    ctx = dev.context()
    for restore in ((False, True) if src.seg is not None else (False,)):
        monkeypatch.setenv('KBBQ_K2_TILE_CHARS', '0')
        want = dev.apply(src, lut, shape, restore_order=restore)
        monkeypatch.delenv('KBBQ_K2_TILE_CHARS')
        got = dev.apply(src, lut, shape, restore_order=restore)
        assert torch.equal(got, want), (lay, restore)
        del got, want
    if n > 1_000_000 or lay != 'rows':
        return
    # against the oracle, and the error cases through the path a caller takes (apply on plain rows)
    seq, cseq, qual, meta = _host(b, n)
    _, ref = _oracle_run(oracle, seq, cseq, qual, meta, R, S)
    got = dev.apply(b, lut, shape)[:n].cpu().numpy()
    assert np.array_equal(got[:, :S].astype(np.int32) - 33, ref[:, :S])
    k = n // 2
    bad = dev.ReadBatch.from_host(seq.copy(), qual.copy(), meta, cseq=cseq)
    at = int(np.flatnonzero(qual[k, 1:S] >= 33 + 6)[0]) + 1                 # a looked-up position (compare_reads.py:288-292: q[i] >= minscore)
    row = seq[k].copy(); row[at] = ord('x')
    bad.seq[k] = torch.from_numpy(row).cuda()
    with pytest.raises(TypeError):
        dev.apply(bad, lut, shape)
    bad = dev.ReadBatch.from_host(seq.copy(), qual.copy(), meta, cseq=cseq)
    row = qual[k].copy(); row[9] = 33 + 43
    bad.qual[k] = torch.from_numpy(row).cuda()
    with pytest.raises(IndexError):
        dev.apply(bad, lut, shape)


@pytest.mark.parametrize('lo,hi,R,packed', [(36, 48, 1, True), (20, 64, 3, True), (100, 150, 1, True), (161, 200, 2, False),
                                            (1, 16, 1, True), (250, 300, 1, False), (161, 208, 2, True), (257, 300, 1, True)])
def test_apply_with_the_lut_narrowed_to_the_rows_pitch(dev, oracle, lo, hi, R, packed, monkeypatch):
    """K2 on a length band whose rows are narrower than the tables (a band of a mixed-length input: tables of 2 x 300
    columns): the LUT holds only the columns such rows reach (k3_fill_row_lut) and 4-bit rows take the short-lived
    kernel.  Same bytes as the full LUT (KBBQ_K2_ROWLUT=0), as the persistent kernel (KBBQ_K2_TILE=0), and as the oracle
    -- for first- and second-in-pair reads (the mirrored half), grouped or not, with and without 4-bit planes."""
    import torch
    n, S = 6000, 300
    band = dev.ReadBatch.synthetic(0, n, 2 * n, seed=71 + lo, len_lo=lo, len_hi=hi, nrg=R)
    wide = dev.ReadBatch.synthetic(n, n, 2 * n, seed=71 + lo, len_lo=S - 40, len_hi=S, nrg=R)     # fills the far columns
    t = dev.Tables(R, 2 * S)
    dev.accumulate(band, t, s_band=hi)
    dev.accumulate(wide, t)
    lut, shape = dev.solve_lut(t)
    parts = [_host(b, n) for b in (band, wide)]
    pitch = wide.pitch
    pad = lambda a, fill: np.pad(a, ((0, 0), (0, pitch - a.shape[1])), constant_values=fill)
    seq = np.concatenate([pad(parts[0][0], ord('N')), parts[1][0]]); cseq = np.concatenate([pad(parts[0][1], ord('N')), parts[1][1]])
    qual = np.concatenate([pad(parts[0][2], 0), parts[1][2]]); meta = np.concatenate([parts[0][3], parts[1][3]])
    _, ref = _oracle_run(oracle, seq, cseq, qual, meta, R, S)
    inside = np.arange(band.pitch)[None, :] < (parts[0][3] & 0xFFFF)[:, None]      # the oracle's values are qualities, ours bytes
    want = np.where(inside, ref[:n, :band.pitch] + 33, 0).astype(np.uint8)
    rows = dev.lay_out(band, R, hi, packed=packed, pairs=False)
    assert rows.nib == packed and (R == 1 or rows.seg is not None)
    # K1 on the laid-out band (4-bit planes of long reads: the table-driven kernel with the shortest-read promise)
    t2 = dev.Tables(R, 2 * S)
    dev.accumulate(rows, t2, s_band=hi)
    dev.accumulate(wide, t2)
    assert torch.equal(t2.buf, t.buf)
    outs = {}
    for name, env in (('narrowed', {}), ('persistent', {'KBBQ_K2_TILE': '0'}), ('full', {'KBBQ_K2_ROWLUT': '0', 'KBBQ_K2_TILE': '0'})):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        outs[name] = dev.apply(rows, lut, shape, restore_order=True).cpu().numpy()
        plain = dev.apply(band, lut, shape).cpu().numpy()              # character planes, input order: kbbq_apply_dev
        for k in env:
            monkeypatch.delenv(k)
        assert np.array_equal(plain[:n], want), name
        assert np.array_equal(outs[name][:n], want), name
    torch.cuda.synchronize()


@pytest.mark.parametrize('S,R,n,packed', [(150, 1, 6000, True), (150, 4, 8000, True), (100, 2, 4000, True), (70, 1, 2002, False),
                                          (151, 3, 3000, True)])
def test_single_end_reads_take_mate_pair_rows_two_to_a_row(dev, oracle, S, R, n, packed):
    """Single-end input (no read is second in pair, compare_reads.py:304-306) of one length: lay_out packs two
    neighbouring reads into a mate-pair row (KBBQ_ROWS_TWINS) and K1 / K2 treat both halves as first-in-pair reads --
    the oracle's counts and qualities, which differ from what the same rows would give as first / second mates."""
    import torch
    b = dev.ReadBatch.synthetic(0, n, n, seed=91 + S + R, len_lo=S, len_hi=S, nrg=R)
    assert dev.meta_stats(b)['twin_violations'] > 0 and dev.meta_stats(b)['pair_violations'] == 0
    b.meta.bitwise_and_(0x7FFFFFFF)                                  # nobody is second in pair any more
    st = dev.meta_stats(b)
    assert st['twin_violations'] == 0 and st['pair_violations'] > 0
    seq, cseq, qual, meta = _host(b, n)
    want, ref = _oracle_run(oracle, seq, cseq, qual, meta, R, S)
    laid = dev.lay_out(b, R, S, packed=packed)
    assert isinstance(laid, dev.PairBatch) and laid.twins and laid.nib == packed and (laid.seg is not None) == (R > 1)
    t = dev.Tables(R, 2 * S)
    dev.accumulate(laid, t)
    for got, w in zip(t.to_host(), want[5:9]):
        assert np.array_equal(got, w)
    assert int(t.to_host()[1][..., S:].sum()) == 0                    # nothing in the second-in-pair columns
    lut, shape = dev.solve_lut(t)
    out = laid.unpack(dev.apply(laid, lut, shape, restore_order=True), b.pitch)
    assert np.array_equal(out[:n, :S].cpu().numpy().astype(np.int32) - 33, ref[:, :S])
    # an odd number of reads: the last row's second half stays padding -- same counts and bytes as one read per row
    odd = dev.ReadBatch.synthetic(0, n - 1, n - 1, seed=5, len_lo=S, len_hi=S, nrg=R)
    odd.meta.bitwise_and_(0x7FFFFFFF)
    lone = dev.lay_out(odd, R, S, packed=packed)
    assert isinstance(lone, dev.PairBatch) and lone.twins and lone.n == n // 2
    ta, tb = dev.Tables(R, 2 * S), dev.Tables(R, 2 * S)
    dev.accumulate(odd, ta)
    dev.accumulate(lone, tb)
    assert torch.equal(ta.buf, tb.buf)
    lut, shape = dev.solve_lut(ta)
    plain = dev.apply(odd, lut, shape)
    got = lone.unpack(dev.apply(lone, lut, shape, restore_order=True), odd.pitch)
    assert got.shape[0] == n and torch.equal(got[:n - 1], plain[:n - 1]) and int(got[n - 1].sum()) == 0
    # one read of another length and the batch keeps one read per row
    odd.meta[3] = (odd.meta[3] & ~0xFFFF) | (S - 1)
    assert not isinstance(dev.lay_out(odd, R, S, packed=packed), dev.PairBatch)
    torch.cuda.synchronize()


@pytest.mark.parametrize('n,S,nrg,single_end,foreign', [(6000, 150, 1, False, False), (6000, 150, 4, False, False), (5001, 150, 1, True, False),
                                                        (4000, 100, 3, True, False), (3000, 40, 1, False, True), (2000, 150, 2, False, True),
                                                        (2, 150, 1, False, False), (1, 20, 1, True, False)])
def test_the_packer_writes_the_rows_the_layout_pass_writes(dev, tmp_path, n, S, nrg, single_end, foreign):
    """_device.laid_from_reader (the C++ packer fills mate-pair / 4-bit / read-group-gathered rows straight into the
    page-locked slabs: kbbq_fastq_fill_rows) against dev.lay_out of the character rows the row-per-read packer uploads
    (k7_lay_out): every plane, the sidecars, perm and seg byte for byte; slab sizes that do and do not divide the band;
    a letter outside ACGTN sends both to character planes."""
    from kbbq import fastx
    from test_packer_layouts import _pair_files
    rng = np.random.default_rng(n + S)
    fa, fb = _pair_files(tmp_path, rng, n, S, nrg, single_end, foreign=foreign)
    infer = nrg > 1
    A, B, info = fastx.PairScan(fa, fb, infer).result()
    R = info[2]
    pitch = fastx.pitch_for(S)
    for other in (B, None):
        rows = dev.ReadBatch.from_reader(A, other, infer, 0, n, pitch)
        want = dev.lay_out(rows, R, S, packed=True)
        for slab in (1 << 17, 998):
            got = dev.laid_from_reader(A, other, infer, 0, n, pitch, R, packed=True, slab=slab)
            if want is rows:
                assert got is None
                continue
            assert type(got) is type(want) and got.n == want.n and got.pitch == want.pitch and got.nib == want.nib
            assert got.nib == (not foreign)
            if isinstance(want, dev.PairBatch):
                assert got.S == want.S and got.twins == want.twins == single_end
            m = want.n
            for name in ('seq', 'cseq', 'qual', 'meta'):
                g, w = getattr(got, name), getattr(want, name)
                assert (g is None) == (w is None), name
                if g is not None:
                    assert np.array_equal(g[:m].cpu().numpy(), w[:m].cpu().numpy()), (name, slab)
            assert (got.perm is None) == (want.perm is None) == (R == 1)
            if R > 1:
                assert np.array_equal(got.perm.cpu().numpy(), want.perm.cpu().numpy()) and np.array_equal(got.seg.cpu().numpy(), want.seg.cpu().numpy())
    assert not [k for k in dev._pinned if k[0] == 'ingest']


@pytest.mark.parametrize('R,n,S', [(1, 40000, 290), (3, 24000, 290), (1, 600, 290), (1, 30000, 150), (2, 9000, 150)])
def test_all_length_bands_in_one_launch(dev, R, n, S, monkeypatch):
    """kbbq_accumulate_bands_dev / kbbq_apply_bands_dev (one K1 and one K2 launch over the length bands of a mixed-length
    input, every band on its share of the workgroups) against a launch per band on the same bands: identical count
    tables, identical new qualities -- bands of 4-bit rows at every pitch the file path cuts (merged), a band of character
    planes and a band of mate-pair rows at the tables' full width (each launched alone by the same call), with one read
    group and with rows gathered by read-group segment (K1 merged over grid.y, K2 band by band)."""
    import torch
    spans = [(36, 48), (49, 64), (65, 96), (97, 128), (129, 160), (161, 208), (209, 256), (257, 290)]
    if S == 150:                                           # ... and a top band of uniform 2 x 150 pairs: mate-pair rows
        spans = [(20, 32), (36, 48), (49, 64), (65, 96), (97, 128), (129, 149)]
    items, first = [], 0
    for k, (lo, hi) in enumerate(spans):
        b = dev.ReadBatch.synthetic(first, n, 10 * n, seed=31 + k, len_lo=lo, len_hi=hi, nrg=R)
        first += n
        st = dev.meta_stats(b)
        packed = k != 4                                   # one band keeps character planes
        laid = dev.lay_out(b, R, st['longest'], packed=packed, pairs=False, stats=st)
        if laid is b:
            assert not packed and R == 1
        items.append((laid, st['longest'], st['shortest']))
    alone = 1                                              # the character band
    if S == 150:
        top = dev.ReadBatch.synthetic(first, n, 10 * n, seed=77, len_lo=S, len_hi=S, nrg=R)     # uniform pairs of the longest length
        laid = dev.lay_out(top, R, S, packed=True)
        assert isinstance(laid, dev.PairBatch) and laid.nib
        items.append((laid, S, S))
        alone += 1                                         # 19 chunks per row: the chunk-position-major K1, launched alone
    want_t = dev.Tables(R, 2 * S)
    for b, smax, smin in items:
        dev.accumulate(b, want_t, s_band=smax, s_min=smin)
    got_t = dev.Tables(R, 2 * S)
    ctx = dev.context()
    ctx.kernel_ms(0, reset=True); ctx.timing(True)
    dev.accumulate_bands(items, got_t)
    ctx.timing(False)
    assert 1 + alone <= ctx.kernel_ms(0)[1] <= 2 + alone   # one merged launch per workgroup shape (narrow rows: two workgroups per CU) + the bands it does not take
    assert torch.equal(got_t.buf, want_t.buf) and int(got_t.buf.sum()) > 0
    monkeypatch.setenv('KBBQ_K1_POSCOPIES', '1')          # without the copies of the cycle table / the extra trash rows: the same counts
    monkeypatch.setenv('KBBQ_K1_NTRASH', '1')
    single = dev.Tables(R, 2 * S)
    dev.accumulate_bands(items, single)
    assert torch.equal(single.buf, want_t.buf)
    monkeypatch.delenv('KBBQ_K1_POSCOPIES'); monkeypatch.delenv('KBBQ_K1_NTRASH')
    monkeypatch.setenv('KBBQ_K1_BANDS', '0')              # the A/B switch: a launch per band through the same entry point
    again = dev.Tables(R, 2 * S)
    dev.accumulate_bands(items, again)
    assert torch.equal(again.buf, want_t.buf)
    lut, shape = dev.solve_lut(want_t)
    for restore in (False, True):
        want_o = [dev.apply(b, lut, shape, restore_order=restore) for b, _, _ in items]
        ctx.kernel_ms(1, reset=True); ctx.timing(True)
        got_o = dev.apply_bands(items, lut, shape, restore_order=restore)
        ctx.timing(False)
        assert ctx.kernel_ms(1)[1] == (1 + alone if R == 1 else len(items))
        for g, w, (b, _, _) in zip(got_o, want_o, items):
            assert torch.equal(g[:b.n], w[:b.n]), b.describe()
    # bad input in one band is reported by the merged launch as by its own
    bad = items[2][0]
    bad.qual[5, 3] = 33 + 43
    with pytest.raises((IndexError, dev.N.LutNeedsCheckedApply)):
        dev.accumulate_bands(items, dev.Tables(R, 2 * S))
    with pytest.raises(dev.N.LutNeedsCheckedApply):
        dev.apply_bands(items, lut, shape)


def test_host_planes_of_a_caller_are_tallied_as_they_are(dev, oracle, tmp_path):
    """recalibrate._tally_local on bands that carry host planes (a caller's own packing: fastx.pack_pair(bands=True) without
    to_device): K1 and K2 run on those rows as they are -- no layout pass (a device pass into another layout costs more than it
    saves for one accumulate + apply)."""
    from conftest import load_golden
    from test_gpu_parity import _files
    from kbbq import fastx, recalibrate
    info, gold = load_golden('c1_10k_1rg')
    fa, fb = _files(oracle, info, tmp_path)
    packed = fastx.pack_pair(fa, fb, False, bands=True)
    assert all('source' not in band and 'seq' in band for band in packed['bands'])
    tables = recalibrate._tally_local(packed, 6, 42)
    for g, k in zip(tables.to_host(), ('pos_errs', 'pos_total', 'dinuc_errs', 'dinuc_total')):
        assert np.array_equal(g, gold[k]), k
    assert all(band['laid'] is None and band['batch'] is not None and not band['batch'].nib for band in packed['bands'])
