"""
GPU tests (-m gpu) of the BAM-sourced tally: K4 -> K6 -> K1 behind
kbbq.gatk.bqsr.bam_to_bqsr_covariates, against goldens from the UNMODIFIED reference
(tests/golden/bqsr_*), the reference's known answer (tests/test_gatk_bqsr.py:38-72) and the
oracle (oracle/oracle_bqsr.py) on further synthetic alignments and error cases.
"""
import numpy as np
import pytest

from test_gpu_parity import dev                      # noqa: F401  (fixture)
from test_oracle_bqsr import SIMPLE_FASTA, VEC, _inputs, _load

pytestmark = pytest.mark.gpu


def _var_pos(path):
    from kbbq import benchmark
    return benchmark.get_var_sites(path)


def _oracle_vectors(paths, minscore=6):
    import _shim
    import oracle_bqsr as OQ
    bam = _shim.AlignmentFile(paths['sam'])
    fa = _shim.FastaFile(paths['fa'])
    ref = {c: fa.fetch(c) for c in fa.references}
    rgs = [rg['ID'] for rg in bam.as_dict()['RG']]
    return OQ.bam_to_bqsr_covariates(list(bam), rgs, ref, _var_pos(paths['vcf']), minscore=minscore)


@pytest.mark.parametrize('name', ['bqsr_a', 'bqsr_b'])
def test_tally_matches_reference_goldens(dev, oracle, name, tmp_path):
    from kbbq import aln
    from kbbq.gatk import bqsr
    info, gold, paths = _inputs(name, tmp_path, oracle)
    got = bqsr.bam_to_bqsr_covariates(aln.AlignmentFile(paths['sam']), paths['fa'], _var_pos(paths['vcf']))
    assert len(got) == 9
    for k, g in zip(VEC, got):
        assert g.dtype == np.int64 and np.array_equal(g, gold[k]), k
    rep = str(bqsr.bam_to_report(aln.AlignmentFile(paths['sam']), paths['fa'], _var_pos(paths['vcf'])))
    assert len(rep) == info['report_len'] and oracle.sha256(rep) == info['report_sha256']


def test_known_answer_of_the_reference(dev, tmp_path):
    """One hard-clipped base of quality 7 (tests/test_gatk_bqsr.py:38-72); meanq 6 is the
    reference's own 'float badness'."""
    from kbbq import aln
    from kbbq.gatk import bqsr
    fa = tmp_path / 'simple.fa'; fa.write_text(SIMPLE_FASTA)
    sam = tmp_path / 't.sam'
    sam.write_text('@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:ref\tLN:45\n@RG\tID:0\tPU:0\n'
                   'clipped\t0\tref\t9\t255\t1M9H\t*\t0\t0\tA\t(\tOQ:Z:(\tRG:Z:0\n')
    got = bqsr.bam_to_bqsr_covariates(aln.AlignmentFile(str(sam)), str(fa), {'ref': [9]})
    z = lambda *s: np.zeros(s, dtype=np.int64)
    qt = z(1, 43); qt[0, 7] = 1
    pt = z(1, 43, 2); pt[0, 7, 0] = 1
    want = [np.array([6]), np.array([0]), np.array([1]), z(1, 43), qt, z(1, 43, 2), pt, z(1, 43, 16), z(1, 43, 16)]
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


@pytest.mark.parametrize('shape', [dict(seed=31, npairs=400, S=150), dict(seed=32, npairs=300, S=151),
                                   dict(seed=33, npairs=200, S=16, contigs=(('c', 2500),)),
                                   dict(seed=34, npairs=300, S=100, nrg=5), dict(seed=36, npairs=120, S=170),
                                   dict(seed=35, npairs=100, S=33, contigs=(('a', 1500), ('b', 1200), ('c', 1400)))])
@pytest.mark.parametrize('minscore', [6, 2, 15])
def test_tally_matches_oracle(dev, oracle, shape, minscore, tmp_path, monkeypatch):
    import oracle_bqsr as OQ
    from kbbq import aln
    from kbbq.gatk import bqsr
    paths = OQ.synth_bqsr_set(str(tmp_path), **shape)
    want = _oracle_vectors(paths, minscore)
    ctx = dev.context()
    for fused in ('2', '1', '0'):                          # one pass (the default: K4 and K6 folded into K1), K4 -> fused K6 + K1, K4 -> K6 -> K1
        monkeypatch.setenv('KBBQ_TALLY_FUSED', fused)
        ctx.kernel_ms(0, reset=True); ctx.timing(True)
        got = bqsr.bam_to_bqsr_covariates(aln.AlignmentFile(paths['sam']), paths['fa'], _var_pos(paths['vcf']),
                                          minscore=minscore)
        ctx.timing(False)
        for k, g, w in zip(VEC, got, want):
            assert np.array_equal(g, w), (k, fused)
        assert ctx.kernel_ms(0)[1] == 1                    # one tally launch either way (a refusal would show as two)
    assert want[2].sum() > 1000


def _edit_sam(paths, fn):
    lines = open(paths['sam']).read().split('\n')
    out = []
    idx = 0
    for ln in lines:
        if ln and not ln.startswith('@'):
            ln = fn(idx, ln.split('\t'))
            idx += 1
            ln = '\t'.join(ln)
        out.append(ln)
    open(paths['sam'], 'w').write('\n'.join(out))


def test_error_behaviour(dev, oracle, tmp_path):
    import oracle_bqsr as OQ
    from kbbq import aln
    from kbbq.gatk import bqsr
    run = lambda p: bqsr.bam_to_bqsr_covariates(aln.AlignmentFile(p['sam']), p['fa'], _var_pos(p['vcf']))

    # a letter outside ACGTN: harmless on a reverse-strand read (complemented to N), TypeError on a forward read
    d = tmp_path / 'a'; d.mkdir()
    paths = OQ.synth_bqsr_set(str(d), seed=41, npairs=60, S=50)

    def weird_reverse(i, f):
        if int(f[1]) & 16 and i % 3 == 0:
            f[9] = f[9][:20] + 'R' + f[9][21:]
            f[-1] = f[-1][:7 + 18] + 'IIIII' + f[-1][7 + 23:]            # OQ:Z: prefix is 5 chars... keep length
        return f
    _edit_sam(paths, weird_reverse)
    want = _oracle_vectors(paths)
    for k, g, w in zip(VEC, run(paths), want):
        assert np.array_equal(g, w), k

    # ... and harmless on a forward read too when no looked-up pair holds it (qualities below 6 around it): K6's 4-bit
    # planes cannot carry the letter, the pass is repeated with character planes and the tallies are the oracle's
    def weird_forward_unseen(i, f):
        if not int(f[1]) & 16 and i % 4 == 1:
            tag = f[-1]
            f[9] = f[9][:25] + 'R' + f[9][26:]
            f[-1] = tag[:5 + 25] + '$$' + tag[5 + 27:]
        return f
    _edit_sam(paths, weird_forward_unseen)
    want = _oracle_vectors(paths)
    for k, g, w in zip(VEC, run(paths), want):
        assert np.array_equal(g, w), k

    def weird_forward(i, f):
        if not int(f[1]) & 16 and i == 10:
            tag = f[-1]
            f[9] = f[9][:25] + 'R' + f[9][26:]
            f[-1] = tag[:5] + 'I' * (len(tag) - 5)                       # every base well above minscore
            f[5] = '%dM' % len(f[9])
        return f
    _edit_sam(paths, weird_forward)
    with pytest.raises(TypeError):
        _oracle_vectors(paths)
    with pytest.raises(TypeError):
        run(paths)

    # a read of another length than the first: the reference's boolean masks no longer fit
    d = tmp_path / 'b'; d.mkdir()
    paths = OQ.synth_bqsr_set(str(d), seed=42, npairs=20, S=40)

    def shorter(i, f):
        if i == 7:
            f[9], f[10], f[-1], f[5] = f[9][:30], f[10][:30], f[-1][:5 + 30], '30M'
        return f
    _edit_sam(paths, shorter)
    with pytest.raises(IndexError):
        _oracle_vectors(paths)
    with pytest.raises(IndexError):
        run(paths)

    # quality above 42 on a counted base: IndexError (np.add.at into a 43-wide axis)
    d = tmp_path / 'c'; d.mkdir()
    paths = OQ.synth_bqsr_set(str(d), seed=43, npairs=20, S=40)

    def high_q(i, f):
        if i == 3:
            f[-1] = f[-1][:5] + 'L' * 40                                  # 'L' = 43
            f[5] = '40M'
        return f
    _edit_sam(paths, high_q)
    with pytest.raises(IndexError):
        _oracle_vectors(paths)
    with pytest.raises(IndexError):
        run(paths)

    # a contig without variant sites: KeyError, as var_pos[chrom] in the reference
    with pytest.raises(KeyError):
        bqsr.bam_to_bqsr_covariates(aln.AlignmentFile(paths['sam']), paths['fa'], {'chr1': [5]})


@pytest.mark.parametrize('L', [150, 37, 16])
def test_canonical_reads_on_four_bit_planes_equal_the_character_planes(dev, L):
    """kbbq_canonical_reads_rows_dev with KBBQ_ROWS_NIBBLES: the same bases, qualities, sidecar and error positions as
    the character-plane form, and the same tallies from K1; a letter outside ACGTN on a forward read is refused
    (KBBQ_E_LUT), on a reverse-strand read it becomes N as in the character form."""
    import torch
    from kbbq import _native as N
    g = torch.Generator(device='cuda').manual_seed(600 + L)
    n, pitch = 3001, (L + 15) // 16 * 16
    rnd = lambda lo, hi, shape, dt=torch.uint8: torch.randint(lo, hi, shape, dtype=dt, device='cuda', generator=g)
    seq = torch.tensor([65, 67, 71, 84, 78], dtype=torch.uint8, device='cuda')[rnd(0, 5, (n, pitch)).long()]
    oq = rnd(33, 33 + 43, (n, pitch))
    flagsplane = rnd(0, 4, (n, pitch)) & rnd(0, 4, (n, pitch))         # bit 0 error, bit 1 skip: each 1 in 4
    lens = torch.full((n,), L, dtype=torch.int32, device='cuda')
    lo = rnd(0, L // 3 + 1, (n,), torch.int32)
    hi = L - rnd(0, L // 3 + 1, (n,), torch.int32)
    clip = lo | (hi << 16)
    tl = rnd(0, L, (n,), torch.int32)
    trim = torch.where(rnd(0, 4, (n,), torch.int32) == 0, tl | (torch.minimum(tl + 9, torch.tensor(L, device='cuda', dtype=torch.int32)) << 16), torch.zeros_like(tl))
    rflags = rnd(0, 4, (n,), torch.int32) | (rnd(0, 3, (n,), torch.int32) << 16)
    ctx, lib = dev.context(), N.load()

    def canonical(nib, s=seq):
        b = dev.ReadBatch(n, pitch, with_corrected=True, nib=nib)
        N.check(lib.kbbq_canonical_reads_rows_dev(ctx.handle, N.ptr(s), N.ptr(oq), N.ptr(flagsplane), None, N.ptr(lens), N.ptr(clip),
                                                  N.ptr(trim), N.ptr(rflags), n, pitch, L, 6, 6, N.ROWS_NIBBLES if nib else 0,
                                                  N.ptr(b.seq), N.ptr(b.cseq), N.ptr(b.qual), N.ptr(b.meta)))
        ctx.status()
        return b
    c, p = canonical(False), canonical(True)
    assert p.seq.shape == (n, pitch // 2)
    assert torch.equal(p.chars('seq'), c.seq) and torch.equal(p.qual, c.qual) and torch.equal(p.meta, c.meta)
    counted = c.seq != 78
    assert torch.equal((p.chars('cseq') != p.chars('seq')) & counted, (c.cseq != c.seq) & counted)
    assert int(((c.cseq != c.seq) & counted).sum()) > n
    tc, tp = dev.Tables(3, 2 * L), dev.Tables(3, 2 * L)
    dev.accumulate(c, tc, 6, dinuc_minscore=6)
    dev.accumulate(p, tp, 6, dinuc_minscore=6)
    assert torch.equal(tc.buf, tp.buf) and int(tc.buf.sum()) > 0
    # letters the planes cannot carry
    odd = seq.clone()
    rev = (rflags & 1).bool()
    odd[rev, L // 2] = ord('R')
    assert torch.equal(canonical(True, odd).chars('seq'), canonical(False, odd).seq)
    odd[(~rev).nonzero()[0], (lo[~rev][0] + 1).long()] = ord('R')
    with pytest.raises(N.LutNeedsCheckedApply):
        canonical(True, odd)


@pytest.mark.parametrize('L,G', [(150, 200_000), (37, 5_000), (100, 1_000), (32, 40)])
def test_the_one_pass_tally_counts_what_k4_then_the_fused_kernel_count(dev, L, G):
    """kbbq_tally_aligned_dev == kbbq_find_errors_dev (one plane of flags, no flip) + kbbq_accumulate_aligned_dev on the same
    arrays: reads of one M operation (compared with the reference by the tally kernel itself), insertions, deletions, soft
    clips, = / X operations, more than four operations, reads at both ends of the genome, both strands, several read groups."""
    import torch
    from kbbq import _native as N
    rng = np.random.default_rng(1000 + L)
    n, pitch = 5003, (L + 15) // 16 * 16
    genome = rng.choice(np.frombuffer(b'ACGTN', dtype=np.uint8), size=G, p=[.24, .24, .24, .24, .04])
    sites = (rng.random(G) < 0.05).astype(np.uint8)
    cig, cig_off, cig_n, ref_len = [], np.zeros(n, np.uint32), np.zeros(n, np.uint32), np.zeros(n, np.int32)
    start = np.zeros(n, np.int64)
    seq = np.zeros((n, pitch), dtype=np.uint8)
    kinds = rng.integers(0, 10, n)
    for i in range(n):
        k = kinds[i]
        if k <= 4:
            ops = [(0, L)]                                               # one M: the tally kernel's own comparison
        elif k == 5:
            a = int(rng.integers(1, L - 2)); ops = [(0, a), (1, int(rng.integers(1, 3))), (0, 0)]      # insertion
        elif k == 6:
            a = int(rng.integers(1, L - 1)); ops = [(0, a), (2, int(rng.integers(1, max(2, min(20, G - L + 1))))), (0, L - a)]  # deletion
        elif k == 7:
            a = int(rng.integers(1, min(10, L - 1))); ops = [(4, a), (0, L - a)]                         # soft clip
        elif k == 8:
            a = int(rng.integers(1, L - 1)); ops = [(7, a), (8, L - a)]                                  # = then X
        else:
            parts = np.diff(np.concatenate([[0], np.sort(rng.choice(np.arange(1, L), size=5, replace=False)), [L]]))
            ops = [(0, int(x)) for x in parts]                                                           # six operations
        ops = [(o, l) for o, l in ops]
        used = sum(l for o, l in ops if o in (0, 1, 4, 7, 8))
        ops = [(o, (l if (o, l) != (0, 0) else L - used)) for o, l in ops]
        rl = sum(l for o, l in ops if o in (0, 2, 3, 7, 8))
        edge = rng.integers(0, 40)
        start[i] = 0 if edge == 0 else (G - rl if edge == 1 else int(rng.integers(0, G - rl + 1)))
        ref_len[i] = rl
        cig_off[i] = len(cig); cig_n[i] = len(ops)
        cig += [(l << 4) | o for o, l in ops]
        # the read: the reference along its CIGAR, with substitutions
        r, g = [], int(start[i])
        for o, l in ops:
            if o in (0, 7, 8):
                r.append(genome[g:g + l]); g += l
            elif o in (1, 4):
                r.append(rng.choice(np.frombuffer(b'ACGT', dtype=np.uint8), size=l))
            else:
                g += l
        r = np.concatenate(r)
        assert len(r) == L
        sub = rng.random(L) < 0.05
        r = np.where(sub, rng.choice(np.frombuffer(b'ACGT', dtype=np.uint8), size=L), r)
        seq[i, :L] = r
    up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    d_seq, d_gen = up(seq), up(genome | (sites << 7))
    d_start, d_rl, d_co, d_cn = up(start), up(ref_len), up(cig_off.view(np.int32)), up(cig_n.view(np.int32))
    d_cig = up(np.array(cig, dtype=np.uint32).view(np.int32))
    d_len = up(np.full(n, L, np.int32))
    d_oq = up(rng.integers(33, 33 + 43, (n, pitch)).astype(np.uint8))
    lo = rng.integers(0, L // 3 + 1, n); hi = L - rng.integers(0, L // 3 + 1, n)
    d_clip = up((lo | (hi << 16)).astype(np.int32))
    tl = rng.integers(0, L, n)
    d_trim = up(np.where(rng.integers(0, 4, n) == 0, tl | (np.minimum(tl + 9, L) << 16), 0).astype(np.int32))
    d_flags = up((rng.integers(0, 4, n) | (rng.integers(0, 3, n) << 16)).astype(np.int32))
    ctx, lib = dev.context(), N.load()
    plane = torch.zeros((n, pitch), dtype=torch.uint8, device='cuda')
    N.check(lib.kbbq_find_errors_dev(ctx.handle, N.ptr(d_seq), N.ptr(d_len), n, pitch, N.ptr(d_start), N.ptr(d_rl), N.ptr(d_co), N.ptr(d_cn),
                                     N.ptr(d_cig), N.ptr(d_gen), None, G, N.ptr(torch.zeros(n, dtype=torch.uint8, device='cuda')), N.ptr(plane), None))
    ctx.status()
    for minscore in (6, 2):
        want, got = dev.Tables(3, 2 * L), dev.Tables(3, 2 * L)
        N.check(lib.kbbq_accumulate_aligned_dev(ctx.handle, N.ptr(d_seq), N.ptr(d_oq), N.ptr(plane), N.ptr(d_clip), N.ptr(d_trim), N.ptr(d_flags),
                                                n, pitch, L, 3, minscore, 6, N.ptr(want.buf)))
        ctx.status()
        scratch = torch.full((n, pitch), 0xA5, dtype=torch.uint8, device='cuda')       # whatever it held before
        N.check(lib.kbbq_tally_aligned_dev(ctx.handle, N.ptr(d_seq), N.ptr(d_oq), N.ptr(d_len), n, pitch, L, N.ptr(d_start), N.ptr(d_rl),
                                           N.ptr(d_co), N.ptr(d_cn), N.ptr(d_cig), N.ptr(d_gen), G, N.ptr(d_clip), N.ptr(d_trim), N.ptr(d_flags),
                                           N.ptr(scratch), 3, minscore, 6, N.ptr(got.buf)))
        ctx.status()
        assert torch.equal(got.buf, want.buf) and int(want.buf.sum()) > 0
        # only the rows of the reads K4 had to look at were written, and they hold K4's flags
        touched = (scratch != 0xA5).any(1).cpu().numpy()
        one_m = (kinds <= 4) & (start + pitch <= G)
        assert not touched[one_m].any()
        others = np.flatnonzero(~one_m)
        assert torch.equal(scratch[torch.from_numpy(others).cuda()], plane[torch.from_numpy(others).cuda()])
