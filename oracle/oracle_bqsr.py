"""
oracle_bqsr.py -- CPU restatement of the BAM-sourced covariate tally:
kbbq/gatk/bqsr.py:23-206 of the reference (strand-aware cycle and dinucleotide covariates,
adaptor trimming, bam_to_bqsr_covariates).  TEST INFRASTRUCTURE ONLY.

Scalar Python over duck-typed reads (anything with pysam.AlignedSegment's attributes: the
stand-ins of oracle/_shim.py or the product's kbbq.aln.AlignedRead).  Pinned by the
reference's known answers (tests/test_gatk_bqsr.py:9-122, restated in
tests/test_oracle_bqsr.py) and by golden vectors from the UNMODIFIED reference run on
synthetic alignments (oracle/gen_golden.py -> tests/golden/bqsr_*.npz).
"""
import numpy as np

import oracle as O
import oracle_benchmark as OB

_COMP = {'A': 'T', 'T': 'A', 'G': 'C', 'C': 'G'}
_CODE = {'A': 0, 'T': 1, 'G': 2, 'C': 3}


def read_oq(read):
    """compare_reads.py:332-336"""
    return np.array([ord(c) - 33 for c in read.get_tag('OQ')], dtype=np.int64)


def bqsr_cycle(read):
    """bqsr.py:23-31: cycles of the aligned part (soft clips stay 0), negative for read 2,
    reversed for reverse-strand reads."""
    out = np.zeros(read.query_length, dtype=np.int64)
    L = read.query_alignment_length
    for k in range(L):
        c = L - 1 - k if read.is_reverse else k
        out[read.query_alignment_start + k] = -(c + 1) if read.is_read2 else c
    return out


def _dinuc(seq, quals, minscore):
    """compare_reads.py:281-293 on a Python string."""
    out = np.full(len(seq), -1, dtype=np.int64)
    for i in range(1, len(seq)):
        if quals[i] < minscore or seq[i] == 'N' or seq[i - 1] == 'N':
            continue
        if seq[i] not in _CODE or seq[i - 1] not in _CODE:
            raise TypeError("int() argument must be a string, a bytes-like object or a real number, not 'NoneType'")
        out[i] = 4 * _CODE[seq[i - 1]] + _CODE[seq[i]]
    return out


def bqsr_dinuc(read, use_oq=True, minscore=6):
    """bqsr.py:33-50: context in sequencing orientation over the aligned part."""
    a, b = read.query_alignment_start, read.query_alignment_end
    seq = read.query_sequence[a:b]
    quals = (read_oq(read) if use_oq else np.array(read.query_qualities, dtype=np.int64))[a:b]
    if read.is_reverse:
        seq = ''.join(_COMP.get(x, 'N') for x in reversed(seq))
        quals = quals[::-1]
    d = _dinuc(seq, quals, minscore)
    if read.is_reverse:
        d = d[::-1]
    out = np.zeros(read.query_length, dtype=np.int64)
    out[a:b] = d
    return out


def adaptor_boundary(read):
    """bqsr.py:131-155"""
    if (read.tlen == 0 or not read.is_paired or read.is_unmapped or read.mate_is_unmapped
            or read.is_reverse == read.mate_is_reverse):
        return None
    if read.is_reverse:
        return read.next_reference_start - 1 if (read.reference_end - 1) > read.next_reference_start else None
    return read.reference_start + abs(read.tlen) if read.reference_start <= read.next_reference_start + read.tlen else None


def trim(read, boundary='compute'):
    """bqsr.py:158-206: bases past the adaptor boundary."""
    if boundary == 'compute':
        boundary = adaptor_boundary(read)
    skips = np.zeros(len(read.query_qualities), dtype=bool)
    if boundary is None:
        return skips
    pairs = read.get_aligned_pairs()
    if read.is_reverse:
        if boundary >= read.reference_start:
            reached, idx = False, 0
            for q, r in reversed(pairs):
                if r is not None and r <= boundary:
                    reached = True
                if reached and q is not None:
                    idx = q + 1
                    break
            skips[:idx] = True
        return skips
    if boundary <= read.reference_end - 1:
        reached, idx = False, len(skips)
        for q, r in pairs:
            if r is not None and r >= boundary:
                reached = True
            if reached and q is not None:
                idx = q
                break
        skips[idx:] = True
    return skips


def bam_to_bqsr_covariates(reads, rg_ids, ref, var_pos, minscore=6, maxscore=42):
    """bqsr.py:52-123.  reads: list; rg_ids: header read-group IDs in header order; ref: dict
    contig -> str; var_pos: dict contig -> list of 0-based variant positions."""
    rg_to_int = {rg: i for i, rg in enumerate(rg_ids)}
    R = len(rg_ids)
    refarr = {c: np.frombuffer(s.encode('ascii'), dtype=np.uint8) for c, s in ref.items()}    # as oracle_benchmark compares
    fullskips = {}
    for c in refarr:
        fullskips[c] = np.zeros(len(refarr[c]), dtype=bool)
        fullskips[c][np.array(var_pos[c], dtype=np.int64)] = True
    S = len(reads[0].query_qualities)
    expected = np.zeros(R, dtype=np.longdouble)
    rg_e = np.zeros(R, dtype=np.int64); rg_t = np.zeros(R, dtype=np.int64)
    q_e = np.zeros((R, maxscore + 1), dtype=np.int64); q_t = np.zeros_like(q_e)
    p_e = np.zeros((R, maxscore + 1, 2 * S), dtype=np.int64); p_t = np.zeros_like(p_e)
    d_e = np.zeros((R, maxscore + 1, 16), dtype=np.int64); d_t = np.zeros_like(d_e)
    for read in reads:
        rg = rg_to_int[read.get_tag('RG')]
        errors, skips = OB.find_read_errors(read, refarr, fullskips)
        q = read_oq(read)
        pos = bqsr_cycle(read)
        dn = bqsr_dinuc(read)
        trimmed = trim(read)
        if len(q) != S or len(errors) != S:
            raise IndexError('boolean index did not match indexed array along axis 0')
        for i in range(S):
            if skips[i] or q[i] < minscore or trimmed[i] or read.query_sequence[i] == 'N':
                continue
            expected[rg] += O.q_to_p(np.array([q[i]]))[0]
            e = bool(errors[i])
            rg_t[rg] += 1; q_t[rg, q[i]] += 1; p_t[rg, q[i], pos[i]] += 1
            if e:
                rg_e[rg] += 1; q_e[rg, q[i]] += 1; p_e[rg, q[i], pos[i]] += 1
            if dn[i] != -1:
                d_t[rg, q[i], dn[i]] += 1
                if e:
                    d_e[rg, q[i], dn[i]] += 1
    with np.errstate(all='ignore'):
        meanq = O.p_to_q(expected / rg_t)
    return meanq, rg_e, rg_t, q_e, q_t, p_e, p_t, d_e, d_t


# ---------------------------------------------------------------- synthetic alignments
def synth_bqsr_set(outdir, seed, npairs=150, S=60, contigs=(('chr1', 6000), ('chr2', 3000)), nrg=3):
    """Writes ref.fa, aln.sam, vars.vcf under outdir: paired reads of ONE query length S (the
    reference indexes every read with masks of the first read's length), FR pairs whose insert
    is sometimes shorter than the read (adaptor read-through -> trimming from either end),
    same-strand pairs, TLEN 0, unmapped mates, soft clips, indels, N bases, 3 read groups with
    PU names, OQ tags."""
    import os
    rng = np.random.default_rng(seed)
    bases = np.array(list('ACGT'))
    ref = {c: ''.join(rng.choice(bases, size=L)) for c, L in contigs}
    paths = {k: os.path.join(outdir, v) for k, v in dict(fa='ref.fa', sam='aln.sam', vcf='vars.vcf').items()}
    with open(paths['fa'], 'w') as fh:
        for c, _ in contigs:
            fh.write('>%s\n' % c)
            for i in range(0, len(ref[c]), 60):
                fh.write(ref[c][i:i + 60] + '\n')
    with open(paths['vcf'], 'w') as fh:
        fh.write('##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\n')
        for c, L in contigs:
            for p in np.sort(rng.choice(np.arange(1, L - 5), size=max(3, L // 50), replace=False)):
                fh.write('%s\t%d\t.\t%s\t%s\t30\t.\t.\n' % (c, p + 1, ref[c][p:p + int(rng.integers(1, 4))], 'A'))

    def make_read(c, anchor, anchored_end):
        """CIGAR + sequence of exactly S query bases; the read starts at `anchor` (forward) or
        ends there (anchored_end): returns (start, ops, seq)."""
        ops, seq, span = [], [], 0
        lead = int(rng.integers(1, 8)) if rng.random() < 0.25 else 0
        tail = int(rng.integers(1, 8)) if rng.random() < 0.25 else 0
        body = S - lead - tail
        blocks, left = [], body
        while left > 0:
            l = int(min(left, rng.integers(8, 40)))
            blocks.append(('M', l)); left -= l
            r = rng.random()
            if left > 3 and r < 0.15:
                il = int(rng.integers(1, 4)); blocks.append(('I', il)); left -= il
            elif left > 0 and r < 0.3:
                blocks.append(('D', int(rng.integers(1, 5))))
        span = sum(l for o, l in blocks if o in 'MD')
        start = anchor - span if anchored_end else anchor
        start = max(start, 0)
        rp = start
        if lead:
            ops.append((4, lead)); seq.append(''.join(rng.choice(bases, size=lead)))
        for o, l in blocks:
            if o == 'M':
                seg = list(ref[c][rp:rp + l])
                for k in np.flatnonzero(rng.random(l) < 0.04):
                    seg[k] = str(rng.choice(bases))
                for k in np.flatnonzero(rng.random(l) < 0.01):
                    seg[k] = 'N'
                ops.append((0, l)); seq.append(''.join(seg)); rp += l
            elif o == 'I':
                ops.append((1, l)); seq.append(''.join(rng.choice(bases, size=l)))
            else:
                ops.append((2, l)); rp += l
        if tail:
            ops.append((4, tail)); seq.append(''.join(rng.choice(bases, size=tail)))
        s = ''.join(seq)
        assert len(s) == S
        return start, ops, s

    sam = ['@HD\tVN:1.6\tSO:unsorted'] + ['@SQ\tSN:%s\tLN:%d' % c for c in contigs] \
        + ['@RG\tID:g%d\tPU:unit%d\tSM:s' % (i, i) for i in range(nrg)]
    for i in range(npairs):
        c, L = contigs[int(rng.integers(0, len(contigs)))]
        insert = int(rng.integers(S // 3, 3 * S)) if rng.random() < 0.5 else int(rng.integers(3 * S, 8 * S))
        fs = int(rng.integers(S + 10, L - 9 * S))
        kind = rng.random()
        same_strand = kind < 0.06
        zero_tlen = 0.06 <= kind < 0.11
        mate_unmapped = 0.11 <= kind < 0.14
        s1, ops1, seq1 = make_read(c, fs, False)
        s2, ops2, seq2 = make_read(c, fs + insert, not same_strand)
        rg = 'g%d' % (i % nrg)
        for mate, (st, ops, sq, other) in enumerate(((s1, ops1, seq1, s2), (s2, ops2, seq2, s1)), start=1):
            rev = (mate == 2) and not same_strand
            mate_rev = (mate == 1) and not same_strand
            flag = 1 | (64 if mate == 1 else 128) | (16 if rev else 0) | (32 if mate_rev else 0) \
                | (8 if mate_unmapped else 0)
            tlen = 0 if zero_tlen else (insert if mate == 1 else -insert)
            q = ''.join(chr(33 + int(x)) for x in rng.integers(2, 42, size=S))
            oq = ''.join(chr(33 + int(x)) for x in rng.integers(2, 42, size=S))
            cigar = ''.join('%d%s' % (l, 'MIDNSHP=X'[o]) for o, l in ops)
            sam.append('p%05d\t%d\t%s\t%d\t60\t%s\t=\t%d\t%d\t%s\t%s\tRG:Z:%s\tOQ:Z:%s'
                       % (i, flag, c, st + 1, cigar, other + 1, tlen, sq, q, rg, oq))
    with open(paths['sam'], 'w') as fh:
        fh.write('\n'.join(sam) + '\n')
    return paths
