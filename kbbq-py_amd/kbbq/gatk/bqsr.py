"""
kbbq.gatk.bqsr -- the report-building half of the reference's kbbq/gatk/bqsr.py
(quantize :213-224, vectors_to_report :226-366): count tables (K1 output) -> GATK
recalibration report.  The EmpiricalQuality columns are the reference's gatk_delta_q calls
(:294, :307, :336, :352); they run on the device (K3, compare_reads.gatk_delta_q), including
the float64-prior call of the read-group table.  Everything else is column assembly.
"""
import numpy as np
import pandas as pd

from .. import compare_reads as utils
from .. import recaltable

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
