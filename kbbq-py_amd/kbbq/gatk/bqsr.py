"""
kbbq.gatk.bqsr -- the reference's kbbq/gatk/bqsr.py on the device.

Report building (quantize :213-224, vectors_to_report :226-366): count tables (K1 output) ->
GATK recalibration report.  The EmpiricalQuality columns are the reference's gatk_delta_q
calls (:294, :307, :336, :352); they run on the device (K3), including the float64-prior call
of the read-group table.  Everything else is column assembly.

BAM-sourced tally (bam_to_bqsr_covariates :52-123 with the strand-aware covariates :23-50 and
the adaptor trimming :131-206): K4 flags errors / known sites by CIGAR walk, K6 rewrites every
read into sequencing orientation with the skipped bases uncounted, and K1 -- the very kernel of
the FASTQ path -- tallies.  Host work per read: the CIGAR-derived scalars (clip ends, adaptor
index); the per-read covariate functions exist for API parity and run on the host.
"""
import numpy as np
import pandas as pd

from .. import compare_reads as utils
from .. import recaltable


# ---------------------------------------------------------------- per-read covariates (host, API parity)
def bamread_bqsr_cycle(read):
    """Cycle of every base of the aligned part, 0 on soft clips (reference bqsr.py:23-31)."""
    full = np.zeros(read.query_length, dtype=np.int_)
    cyc = utils.generic_cycle_covariate(read.query_alignment_length, read.is_read2)
    full[read.query_alignment_start:read.query_alignment_end] = cyc[::-1] if read.is_reverse else cyc
    return full


def bamread_bqsr_dinuc(read, use_oq=True, minscore=6):
    """Dinucleotide context in sequencing orientation over the aligned part (reference
    bqsr.py:33-50)."""
    a, b = read.query_alignment_start, read.query_alignment_end
    quals = (utils.bamread_get_oq(read) if use_oq else np.array(read.query_qualities, dtype=np.int_))[a:b]
    letters = np.array(list(read.query_sequence[a:b]), dtype='U1')
    if read.is_reverse:
        comp = utils.Dinucleotide.complement
        letters = np.array([comp.get(x, 'N') for x in letters[::-1]], dtype='U1')
        quals = quals[::-1]
    ctx = utils.generic_dinuc_covariate(letters, quals, minscore) if b > a else np.zeros(0, dtype=np.int_)
    full = np.zeros(read.query_length, dtype=np.int_)
    full[a:b] = ctx[::-1] if read.is_reverse else ctx
    return full


# ---------------------------------------------------------------- adaptor trimming (host)
def bamread_adaptor_boundary(read):
    """Reference position where the adaptor starts, or None (reference bqsr.py:131-155, after
    GATK ReadUtils.getAdaptorBoundary)."""
    if (read.tlen == 0 or not read.is_paired or read.is_unmapped or read.mate_is_unmapped
            or read.is_reverse == read.mate_is_reverse):
        return None
    if read.is_reverse:
        if (read.reference_end - 1) > read.next_reference_start:
            return read.next_reference_start - 1
        return None
    if read.reference_start <= read.next_reference_start + read.tlen:
        return read.reference_start + abs(read.tlen)
    return None


_UNSET = object()


class _CigarView:
    """The few attributes _trim_range reads, for alignments that only exist as arrays."""

    def __init__(self, ops, start, end, reverse, nqual):
        self.cigartuples, self.reference_start, self.reference_end = ops, start, end
        self.is_reverse, self.query_qualities = reverse, range(nqual)


def _trim_range(read, boundary=_UNSET):
    """(lo, hi): query positions [lo, hi) lie past the adaptor boundary (reference
    bqsr.py:158-206), found by one pass over the CIGAR instead of get_aligned_pairs()."""
    if boundary is _UNSET:
        boundary = bamread_adaptor_boundary(read)
    n = len(read.query_qualities)
    if boundary is None:
        return 0, 0
    ops = read.cigartuples
    if read.is_reverse:
        if boundary < read.reference_start:
            return 0, 0
        # walking backwards: the last reference position <= boundary, then the nearest base at or before it
        q, r, reached = sum(l for op, l in ops if op in (0, 1, 4, 7, 8)), read.reference_end, False
        for op, l in reversed(ops):
            if op in (0, 7, 8):
                if reached:
                    return 0, q
                if r - l <= boundary:
                    return 0, q - ((r - 1) - min(boundary, r - 1))
                q -= l; r -= l
            elif op in (1, 4):
                if reached:
                    return 0, q
                q -= l
            elif op in (2, 3):
                if r - l <= boundary:
                    reached = True
                r -= l
        return 0, 0
    if boundary > read.reference_end - 1:
        return 0, 0
    q, r, reached = 0, read.reference_start, False
    for op, l in ops:
        if op in (0, 7, 8):
            if reached:
                return q, n
            if r + l > boundary:
                return q + max(boundary - r, 0), n
            q += l; r += l
        elif op in (1, 4):
            if reached:
                return q, n
            q += l
        elif op in (2, 3):
            if r + l > boundary:
                reached = True
            r += l
    return n, n


def trim_bamread(read, boundary=_UNSET):
    """Boolean array: bases to skip because they lie past the adaptor boundary (reference
    bqsr.py:158-206).  `boundary` defaults to bamread_adaptor_boundary(read), looked up at call
    time like the reference does."""
    lo, hi = _trim_range(read, bamread_adaptor_boundary(read) if boundary is _UNSET else boundary)
    out = np.zeros(len(read.query_qualities), dtype=bool)
    out[lo:hi] = True
    return out


# ---------------------------------------------------------------- the tally (device)
def bam_to_bqsr_covariates(bamfileobj, fastafilename, var_pos, minscore=6, maxscore=42):
    """The nine model vectors from aligned reads, a FASTA reference and known variant sites
    (reference bqsr.py:52-123): K4 -> K6 -> K1.  Every read must have the first read's length
    (the reference indexes with masks of that length: IndexError otherwise); quality = OQ tag.
    Under torch.distributed (parallel.init_from_env) every rank runs the three kernels on a contiguous shard of the
    alignments and the count tables are summed with one allreduce, as in kbbq.recalibrate."""
    from .. import _device as dev
    from .. import _native as N
    from .. import _solve, aln, benchmark, fastx, parallel
    if maxscore != 42:
        raise ValueError('the Q axis of the device tables is fixed at 43 (maxscore = 42)')
    torch = dev._torch()
    rg_to_pu = utils.get_rg_to_pu(bamfileobj)
    rg_to_int = {rg: i for i, rg in enumerate(rg_to_pu)}
    R = len(rg_to_int)
    fasta = aln.FastaFile(fastafilename)
    ref = {chrom: aln.chars(fasta.fetch(reference=chrom)) for chrom in fasta.references}
    fullskips = {}
    for chrom in ref:
        fullskips[chrom] = np.zeros(len(ref[chrom]), dtype=bool)
        fullskips[chrom][np.array(var_pos[chrom], dtype=np.int_)] = True      # KeyError: a contig without sites
    native = isinstance(bamfileobj, aln.AlignmentFile)
    reads = bamfileobj if native else list(bamfileobj)
    n = len(reads)
    if n == 0:
        raise StopIteration                                                    # next(bamfileobj) at :71
    genome = benchmark._Genome(ref, fullskips)
    world, rank = parallel.world_rank()
    bad_length = None
    if native:
        # arrays straight from the SAM reader; only reads whose adaptor boundary falls inside them need a CIGAR pass
        b = reads.batch()
        S = int(b.qual_len[0]) if b.qual_len[0] else int(b.qlen[0])            # len(read.query_qualities) of the first read
        if int(b.rg.min()) < 0:
            i = int(np.flatnonzero(b.rg < 0)[0])
            raise KeyError("tag 'RG' not present" if b.rg[i] == -1 else b._text(1, i).split('RG:Z:')[1].split('\t')[0])
        if int(b.oq_len.min()) < 0:
            raise KeyError("tag 'OQ' not present")
        lens = b.qlen.astype(np.uint32)
        wrong = np.flatnonzero((b.oq_len != S) | (lens != S))
        if wrong.size:
            bad_length = int(wrong[0])
        clip = b.clip.copy()
        rev = (b.flag & 16) != 0
        trim = b.adaptor_trim()                # boundary + CIGAR walk per alignment, in the reader (csrc/sam_host.cpp)
        flags = (rev.astype(np.uint32) | (((b.flag & 128) != 0).astype(np.uint32) << 1) | (b.rg.astype(np.uint32) << 16))
    else:
        S = len(reads[0].query_qualities)
        lens = np.array([len(r.query_sequence) for r in reads], dtype=np.uint32)
        pitch_all = fastx.pitch_for(int(lens.max()))
        oq_all = np.zeros((n, pitch_all), dtype=np.uint8)
        clip = np.zeros(n, dtype=np.uint32); trim = np.zeros(n, dtype=np.uint32); flags = np.zeros(n, dtype=np.uint32)
        for i, r in enumerate(reads):
            rg = rg_to_int[r.get_tag('RG')]
            q = aln.codes(r.get_tag('OQ'))
            if (len(q) != S or lens[i] != S) and bad_length is None:
                bad_length = i
            m = min(len(q), pitch_all)
            oq_all[i, :m] = q[:m]
            clip[i] = r.query_alignment_start | (r.query_alignment_end << 16)
            lo, hi = _trim_range(r)
            trim[i] = lo | (hi << 16)
            flags[i] = (1 if r.is_reverse else 0) | (2 if r.is_read2 else 0) | (rg << 16)
    upto = n if bad_length is None else bad_length       # reads before the offending one are still examined
    lo, hi = parallel.shard_range(upto, rank, world) if world > 1 else (0, upto)      # this rank's alignments
    m = hi - lo
    tables = dev.Tables(max(R, 1), 2 * S)

    def shard():
        import os
        u = benchmark._upload_reads(reads, genome, False, rows=(lo, hi))
        pitch = u['pitch']
        oq = reads.batch().plane(2, pitch, lo, m) if native else oq_all[lo:hi, :pitch]
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        d_seq, d_oq = u['seq'], up(oq)
        d_len, d_clip, d_trim, d_flags = (up(x[lo:hi].view(np.int32)) for x in (lens, clip, trim, flags))
        ctx = dev.context()
        mode = os.environ.get('KBBQ_TALLY_FUSED', '2')
        fits = pitch == (S + 15) // 16 * 16 and S <= 32767
        # K4 and K6 both folded into K1 (kbbq_tally_aligned_dev): reads that are one M / = / X operation are compared with the
        # reference by the tally kernel itself, K4 only looks at the others -- 3 B/base instead of 6.  Needs the site flags in
        # bit 7 of the reference bytes (benchmark._Genome: an ASCII reference); refusals as below, counted into tables of its own.
        if mode == '2' and fits and genome.mask is None:
            try:
                part = dev.Tables(max(R, 1), 2 * S)
                plane = torch.empty((max(m, 1), pitch), dtype=torch.uint8, device='cuda')     # rows of the reads K4 looks at
                N.check(N.load().kbbq_tally_aligned_dev(
                    ctx.handle, N.ptr(d_seq), N.ptr(d_oq), N.ptr(u['len']), m, pitch, S, N.ptr(u['ref_start']), N.ptr(u['ref_len']),
                    N.ptr(u['cig_off']), N.ptr(u['cig_n']), N.ptr(u['cigar']), N.ptr(genome.genome), genome.length,
                    N.ptr(d_clip), N.ptr(d_trim), N.ptr(d_flags), N.ptr(plane), max(R, 1), minscore, 6, N.ptr(part.buf)))
                ctx.status()
                tables.add(part)
                return
            except N.LutNeedsCheckedApply:
                pass
        err, skip = benchmark._find_errors(u, genome, fused=True)

        def canonical(nib):
            batch = dev.ReadBatch(m, pitch, with_corrected=True, nib=nib)
            N.check(N.load().kbbq_canonical_reads_rows_dev(
                ctx.handle, N.ptr(d_seq), N.ptr(d_oq), N.ptr(err), N.ptr(skip), N.ptr(d_len), N.ptr(d_clip),
                N.ptr(d_trim), N.ptr(d_flags), m, pitch, S, minscore, 6, N.ROWS_NIBBLES if nib else 0,
                N.ptr(batch.seq), N.ptr(batch.cseq), N.ptr(batch.qual), N.ptr(batch.meta)))
            ctx.status()
            return batch
        # K6 fused into K1 (kbbq_accumulate_aligned_dev): the tally straight from the reads as aligned, nothing written but the
        # tables -- unless a forward read carries a letter outside ACGTN or the tables do not fit the LDS beside it (both
        # reported before / without anything reaching `tables`: the attempt counts into tables of its own); then, as before:
        if mode != '0' and fits:
            try:
                part = dev.Tables(max(R, 1), 2 * S)
                N.check(N.load().kbbq_accumulate_aligned_dev(ctx.handle, N.ptr(d_seq), N.ptr(d_oq), N.ptr(err), N.ptr(d_clip), N.ptr(d_trim),
                                                             N.ptr(d_flags), m, pitch, S, max(R, 1), minscore, 6, N.ptr(part.buf)))
                ctx.status()
                tables.add(part)
                return
            except N.LutNeedsCheckedApply:
                pass
        # 4-bit sequence planes between K6 and K1 (one byte per base less to write and to read) unless a forward read
        # carries a letter outside ACGTN, the reads are longer than K1's packed form takes, or K1's tables for this
        # minscore do not fit beside it (both refusals come before anything is counted)
        # (tallied into tables of its own and added on success only: a refusal that arrives after the launch must not
        #  leave counts behind that the character-plane pass would add again)
        try:
            part = dev.Tables(max(R, 1), 2 * S)
            dev.accumulate(canonical(S <= dev.PACKED_READS), part, minscore, dinuc_minscore=6)
            tables.add(part)
        except N.LutNeedsCheckedApply:
            dev.accumulate(canonical(False), tables, minscore, dinuc_minscore=6)
    benchmark._on_all_ranks(shard if m else (lambda: None), lo)
    if world > 1:
        parallel.allreduce_tables(tables.buf)
    if bad_length is not None:
        raise IndexError('boolean index did not match indexed array along axis 0; size of axis is %d but size of '
                         'corresponding boolean axis is %d' % (S, int(lens[bad_length])))
    return _solve.vectors_from_tables(*tables.to_host(), maxscore)


# the fixed argument table GATK expects to find (reference bqsr.py:263-281)
_ARGUMENTS = (
    ('binary_tag_name', 'null'),
    ('covariate', 'ReadGroupCovariate,QualityScoreCovariate,ContextCovariate,CycleCovariate'),
    ('default_platform', 'null'),
    ('deletions_default_quality', '45'),
    ('force_platform', 'null'),
    ('indels_context_size', '3'),
    ('insertions_default_quality', '45'),
    ('low_quality_tail', '2'),
    ('maximum_cycle_value', '500'),
    ('mismatches_context_size', '2'),
    ('mismatches_default_quality', '-1'),
    ('no_standard_covs', 'false'),
    ('quantizing_levels', '16'),
    ('recalibration_report', 'null'),
    ('run_without_dbsnp', 'false'),
    ('solid_nocall_strategy', 'THROW_EXCEPTION'),
    ('solid_recal_mode', 'SET_Q_ZERO'),
)


def quantize(q_errs, q_total, maxscore=93):
    """Identity quantisation map; scores never observed map to maxscore (reference
    bqsr.py:213-224 -- not GATK's algorithm there either)."""
    seen = np.sum(q_total, axis=0)
    out = np.arange(maxscore + 1)
    out[:seen.shape[0]][seen == 0] = maxscore
    out[seen.shape[0]:] = maxscore
    return out


def _cycle_labels(width):
    """Cycle axis of the count tables -> GATK cycle numbers: 1..n, then -n..-1 (reference
    bqsr.py:345-346)."""
    n = width / 2
    fwd = np.arange(n) + 1
    return np.concatenate([fwd, -fwd[::-1]]).astype(np.int_)


def vectors_to_report(meanq, global_errs, global_total, q_errs, q_total,
                      pos_errs, pos_total, dinuc_errs, dinuc_total, rg_order, maxscore=42):
    """The nine model vectors + read-group names -> RecalibrationReport (reference
    bqsr.py:226-366).  Rows with no observations are dropped; the covariate table is ordered by
    (ReadGroup, QualityScore, CovariateValue as text, CovariateName)."""
    global_errs, global_total = np.asarray(global_errs), np.asarray(global_total)
    q_errs, q_total = np.asarray(q_errs), np.asarray(q_total)
    pos_errs, pos_total = np.asarray(pos_errs), np.asarray(pos_total)
    dinuc_errs, dinuc_total = np.asarray(dinuc_errs), np.asarray(dinuc_total)
    R, Q = q_total.shape
    rg_names = np.asarray(list(rg_order))

    arguments = pd.DataFrame({'Argument': [a for a, _ in _ARGUMENTS], 'Value': [v for _, v in _ARGUMENTS]})

    # read-group table: the reported quality is the mean error probability, 5 decimals of log10
    with np.errstate(divide='ignore', invalid='ignore'):
        expected = np.sum(utils.q_to_p(np.arange(Q)) * q_total, axis=1) / global_total
        est_q = -10.0 * np.log10(expected).round(decimals=5).astype(np.float64)
    est_q[np.isnan(est_q)] = 0
    emp = (utils.gatk_delta_q(est_q, global_errs.copy(), global_total.copy()) + est_q).astype(np.float64)
    rg_frame = pd.DataFrame({'ReadGroup': rg_order, 'EventType': 'M', 'EmpiricalQuality': emp,
                             'EstimatedQReported': est_q, 'Observations': global_total,
                             'Errors': global_errs.astype(np.float64)})
    rg_frame = rg_frame[rg_frame.Observations != 0]

    # quality table
    qs = np.tile(np.arange(Q), R)
    q_frame = pd.DataFrame({
        'ReadGroup': np.repeat(rg_names, Q), 'QualityScore': qs, 'EventType': np.full(R * Q, 'M'),
        'EmpiricalQuality': (utils.gatk_delta_q(qs, q_errs.ravel(), q_total.ravel()) + qs).astype(np.float64),
        'Observations': q_total.ravel(), 'Errors': q_errs.ravel().astype(np.float64)})
    q_frame = q_frame[q_frame.Observations != 0]

    # quantisation table (no quantisation: identity map over 0..93)
    counts = np.zeros(94)
    counts[np.arange(Q)] = np.sum(q_total, axis=0)
    quant_frame = pd.DataFrame({'QualityScore': np.arange(94), 'Count': counts,
                                'QuantizedScore': quantize(q_errs, q_total)})

    # covariate table: context rows and cycle rows together
    def covariate_rows(errs, total, labels, name):
        width = total.shape[2]
        q_of = np.repeat(np.tile(np.arange(total.shape[1]), total.shape[0]), width)
        e, t = errs.ravel(), total.ravel()
        emp_q = (utils.gatk_delta_q(q_of, e, t) + q_of).astype(np.float64)
        return dict(rg=np.repeat(rg_names, total.shape[1] * width), q=q_of,
                    value=np.tile(np.asarray(labels), total.shape[0] * total.shape[1]),
                    name=np.full(t.shape, name), emp=emp_q, obs=t, errs=e.astype(np.float64))

    parts = [covariate_rows(dinuc_errs, dinuc_total, np.array(utils.Dinucleotide.dinucs), 'Context'),
             covariate_rows(pos_errs, pos_total, _cycle_labels(pos_total.shape[2]).astype(np.str_), 'Cycle')]
    cat = {k: np.concatenate([p[k] for p in parts]) for k in parts[0]}
    keep = cat['obs'] != 0
    cat = {k: v[keep] for k, v in cat.items()}
    order = np.lexsort((cat['name'], cat['value'], cat['q'], cat['rg']))       # last key is primary
    cov_frame = pd.DataFrame({
        'ReadGroup': cat['rg'][order], 'QualityScore': cat['q'][order], 'CovariateValue': cat['value'][order],
        'CovariateName': cat['name'][order], 'EventType': np.full(order.shape, 'M'),
        'EmpiricalQuality': cat['emp'][order], 'Observations': cat['obs'][order], 'Errors': cat['errs'][order]})
    for col in ('ReadGroup', 'CovariateValue', 'CovariateName', 'EventType'):
        cov_frame[col] = cov_frame[col].astype(object)
    for frame in (q_frame,):
        for col in ('ReadGroup', 'EventType'):
            frame[col] = frame[col].astype(object)

    titles = recaltable.RecalibrationReport.TITLES
    descriptions = ('Recalibration argument collection values used in this run',
                    'Quality quantization map', '', '', '')
    frames = (arguments, quant_frame, rg_frame, q_frame, cov_frame)
    return recaltable.RecalibrationReport(
        [recaltable.GATKTable(t, d, f) for t, d, f in zip(titles, descriptions, frames)])


def bam_to_report(bamfileobj, fastafilename, var_pos):
    """Aligned reads -> recalibration report; read groups are named by their PU (reference
    bqsr.py:368-371)."""
    rgs = list(utils.get_rg_to_pu(bamfileobj).values())
    vectors = bam_to_bqsr_covariates(bamfileobj, fastafilename, var_pos)
    return vectors_to_report(*vectors, rgs)
