"""
kbbq.gatk.applybqsr -- get_delta_qs is on the hot path (reference
kbbq/gatk/applybqsr.py:80-103); table_to_vectors (:14-44) turns a stored GATK report back
into the nine model vectors.  The per-read ApplyBQSR emulation on
aligned reads (:46-78) is host NumPy, one read at a time, as in the reference (it has no batch
caller there; reads come from kbbq.aln or pysam).
"""
import numpy as np

from .. import compare_reads as utils


def table_to_vectors(table, rg_order, maxscore=42):
    """RecalibrationReport -> (meanq, global_errs, global_total, q_errs, q_total, pos_errs,
    pos_total, dinuc_errs, dinuc_total) for the read groups of rg_order, in that order
    (reference applybqsr.py:14-44).  Cells the report does not list are zero; the cycle axis is
    2 x the largest cycle number present, laid out 1..n then -n..-1; meanq is the report's
    EstimatedQReported (float64).  A read group missing from RecalTable0 cannot be cast to
    integer counts in the reference either (NaN): ValueError."""
    rg_order = list(rg_order)
    rg_index = {str(name): i for i, name in enumerate(rg_order)}
    R, Q = len(rg_order), maxscore + 1

    t0 = table.tables[2].data
    rows0 = {str(name): pos for pos, name in enumerate(t0.index)}
    missing = [name for name in rg_index if name not in rows0]
    if missing:
        raise ValueError('Cannot convert non-finite values (NA or inf) to integer: read group %r is not in the report'
                         % missing[0])
    pick = np.array([rows0[str(name)] for name in rg_order], dtype=np.int64)
    meanq = t0['EstimatedQReported'].to_numpy()[pick].astype(np.float64)
    global_errs = t0['Errors'].to_numpy()[pick].astype(np.int64)
    global_total = t0['Observations'].to_numpy()[pick]

    def scatter(shape, idx, values):
        out = np.zeros(shape, dtype=np.int64)
        out[idx] = np.asarray(values).astype(np.int64)
        return out

    t1 = table.tables[3].data
    rg1 = np.array([rg_index.get(str(x), -1) for x in t1.index.get_level_values('ReadGroup')], dtype=np.int64)
    q1 = t1.index.get_level_values('QualityScore').to_numpy().astype(np.int64)
    ok = (rg1 >= 0) & (q1 >= 0) & (q1 < Q)
    q_errs = scatter((R, Q), (rg1[ok], q1[ok]), t1['Errors'].to_numpy()[ok])
    q_total = scatter((R, Q), (rg1[ok], q1[ok]), t1['Observations'].to_numpy()[ok])

    t2 = table.tables[4].data
    rg2 = np.array([rg_index.get(str(x), -1) for x in t2.index.get_level_values('ReadGroup')], dtype=np.int64)
    q2 = t2.index.get_level_values('QualityScore').to_numpy().astype(np.int64)
    name2 = t2.index.get_level_values('CovariateName').to_numpy().astype(str)
    value2 = t2.index.get_level_values('CovariateValue').to_numpy().astype(str)
    errs2, obs2 = t2['Errors'].to_numpy(), t2['Observations'].to_numpy()
    inside = (rg2 >= 0) & (q2 >= 0) & (q2 < Q)

    cyc = inside & (name2 == 'Cycle')
    if not cyc.any():
        raise ValueError('cannot convert float NaN to integer: the report lists no Cycle rows for these read groups')
    cycles = value2[cyc].astype(np.int64)
    seqlen = int(cycles.max())
    # 1..n -> columns 0..n-1; -n..-1 -> columns n..2n-1; anything else is not on the reindexed grid
    col = np.where(cycles > 0, cycles - 1, 2 * seqlen + cycles)
    on_grid = (cycles != 0) & (cycles >= -seqlen)
    idx = (rg2[cyc][on_grid], q2[cyc][on_grid], col[on_grid])
    pos_errs = scatter((R, Q, 2 * seqlen), idx, errs2[cyc][on_grid])
    pos_total = scatter((R, Q, 2 * seqlen), idx, obs2[cyc][on_grid])

    ctx = inside & (name2 == 'Context')
    code = np.array([utils.Dinucleotide.dinuc_to_int.get(v, -1) for v in value2[ctx]], dtype=np.int64)
    known = code >= 0
    idx = (rg2[ctx][known], q2[ctx][known], code[known])
    dinuc_errs = scatter((R, Q, 16), idx, errs2[ctx][known])
    dinuc_total = scatter((R, Q, 16), idx, obs2[ctx][known])

    return meanq, global_errs, global_total, q_errs, q_total, pos_errs, pos_total, dinuc_errs, dinuc_total


def get_delta_qs(meanq, rg_errs, rg_total, q_errs, q_total, pos_errs, pos_total,
                 dinuc_errs, dinuc_total):
    """Hierarchical delta-Q solve: read group, then reported quality given the read
    group, then cycle and dinucleotide given both.  Returns
    (rgdeltaq[R], qscoredeltaq[R,Q], positiondeltaq[R,Q,2S], dinucdeltaq[R,Q,17]);
    the 17th dinucleotide column is zero so that context -1 adds nothing."""
    meanq = np.asarray(meanq)
    rg_dq = utils.gatk_delta_q(meanq, rg_errs, rg_total)
    level1 = np.broadcast_to((meanq + rg_dq)[:, np.newaxis], np.shape(q_total)).copy()
    q_dq = utils.gatk_delta_q(level1, q_errs, q_total)
    level2 = level1 + q_dq
    pos_dq = utils.gatk_delta_q(np.broadcast_to(level2[..., np.newaxis], np.shape(pos_total)).copy(),
                                pos_errs, pos_total)
    dn_dq = utils.gatk_delta_q(np.broadcast_to(level2[..., np.newaxis], np.shape(dinuc_total)).copy(),
                               dinuc_errs, dinuc_total)
    dn_dq = np.concatenate([dn_dq, np.zeros(dn_dq.shape[:-1] + (1,), dtype=dn_dq.dtype)], axis=-1)
    return rg_dq.copy(), q_dq.copy(), pos_dq.copy(), dn_dq.copy()


def _oriented_quals(read, use_oq):
    return utils.bamread_get_oq(read) if use_oq else np.array(read.query_qualities, dtype=np.int_)


def bamread_cycle_covariates(read):
    """Cycle of every base in ALIGNED order: 0..L-1 (first in pair) or -1..-L (second), reversed for a
    reverse-strand read (reference applybqsr.py:46-50)."""
    cycle = utils.generic_cycle_covariate(read.query_length, read.is_read2)
    return np.flip(cycle) if read.is_reverse else cycle


def bamread_dinuc_covariates(read, use_oq=True, minscore=6):
    """Dinucleotide context of every base in aligned order, computed in sequencing orientation (reverse-strand reads
    reverse-complemented, unknown letters -> N) and flipped back (reference applybqsr.py:52-63)."""
    seq, quals = read.query_sequence, _oriented_quals(read, use_oq)
    if read.is_reverse:
        seq = ''.join(utils.Dinucleotide.complement.get(x, 'N') for x in reversed(seq))
        quals = np.flip(quals)
    dinuc = utils.generic_dinuc_covariate(np.array(list(seq), dtype='U1'), quals, minscore)
    return np.flip(dinuc) if read.is_reverse else dinuc


def recalibrate_bamread(read, meanq, globaldeltaq, qscoredeltaq, positiondeltaq, dinucdeltaq, rg_to_int, use_oq=True,
                        minscore=6):
    """New qualities of one aligned read (reference applybqsr.py:65-78): bases at or above minscore get
    meanq + the four deltas of their read group / quality / context / cycle, the others keep their quality.
    As in the reference the context is always taken from the OQ tag's qualities with minscore 6."""
    original = _oriented_quals(read, use_oq)
    out = np.array(original, dtype=np.int_)
    rg = rg_to_int[read.get_tag('RG')]
    valid = original >= minscore
    q = original[valid]
    cycle = bamread_cycle_covariates(read)[valid]
    dinuc = bamread_dinuc_covariates(read)[valid]
    out[valid] = (meanq[rg] + globaldeltaq[rg] + qscoredeltaq[rg, q] + dinucdeltaq[rg, q, dinuc]
                  + positiondeltaq[rg, q, cycle]).astype(np.int_)
    return out

