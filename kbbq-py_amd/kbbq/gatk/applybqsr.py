"""
kbbq.gatk.applybqsr -- only get_delta_qs is on the hot path (reference
kbbq/gatk/applybqsr.py:80-103); the BAM / GATK-report emulation around it is
out of scope.
"""
import numpy as np

from .. import compare_reads as utils


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
