"""
Host half of the device model solve (K3).

The GPU cannot reproduce glibc/SciPy transcendentals bit for bit, so everything
transcendental in compare_reads.gatk_delta_q (reference compare_reads.py:235-260) is
evaluated HERE with the same SciPy functions scipy.stats.binom.logpmf itself calls
(scipy/stats/_discrete_distns.py, binom_gen._logpmf):

    logpmf(k; n, p) = combiln + xlogy(k, p) + xlog1py(n - k, -p)
    combiln         = gammaln(n + 1) - (gammaln(k + 1) + gammaln(n - k + 1))

`combiln` does not depend on the candidate quality, so it is one float64 per cell
(three gammaln calls); log(p) and log1p(-p) are 43-entry tables.  The device then only
multiplies and adds those doubles in the same order (IEEE, no contraction) and does the
longdouble prior add + argmax exactly (csrc/solve_core.h, csrc/x87add.h).
tests/test_solve_core_host.py pins this decomposition against logpmf itself.
"""
import numpy as np
import scipy.special

from . import compare_reads as utils

NQ = 43


def model_consts():
    """(prior[43], logp[43], log1mp[43]) as float64; prior is exactly representable
    (float64 values stored in a longdouble array by the reference)."""
    p = utils.q_to_p(np.arange(NQ, dtype=np.int_)).astype(np.float64)
    prior = utils.RescaledNormal.prior_dist.astype(np.float64)
    assert np.all((prior.astype(np.longdouble) == utils.RescaledNormal.prior_dist)
                  | ~np.isfinite(prior))
    with np.errstate(divide='ignore'):
        logp = scipy.special.xlogy(1.0, p)
        log1mp = scipy.special.xlog1py(1.0, -p)
    return np.ascontiguousarray(np.concatenate([prior, logp, log1mp]))


def combiln(numerrs, numtotal):
    """The candidate-independent term of logpmf(errs + 1; total + 2, p), float64 per cell."""
    x = np.asarray(numerrs) + 1
    n = np.asarray(numtotal) + 2
    k = np.floor(x)
    with np.errstate(all='ignore'):
        return scipy.special.gammaln(n + 1) - (scipy.special.gammaln(k + 1) + scipy.special.gammaln(n - k + 1))


def vectors_from_tables(pos_errs, pos_total, dinuc_errs, dinuc_total, maxscore=42):
    """The reference's 9-tuple from the four 3-d count arrays: q_* and rg_* are marginals of
    pos_* (every counted base has exactly one cycle); meanq = p_to_q(sum_q q_total * 10^(-q/10)
    / rg_total) in longdouble (recalibrate.py:111,120)."""
    q_errs, q_total = pos_errs.sum(axis=2), pos_total.sum(axis=2)
    rg_errs, rg_total = q_errs.sum(axis=1), q_total.sum(axis=1)
    expected = (q_total.astype(np.longdouble) * utils.q_to_p(np.arange(maxscore + 1))).sum(axis=1)
    with np.errstate(divide='ignore', invalid='ignore'):
        meanq = utils.p_to_q(expected / rg_total, maxscore)
    return meanq, rg_errs, rg_total, q_errs, q_total, pos_errs, pos_total, dinuc_errs, dinuc_total
