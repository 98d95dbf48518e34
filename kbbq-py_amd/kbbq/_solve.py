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

from . import compare_reads as utils

NQ = 43


def model_consts():
    """(prior[43], logp[43], log1mp[43]) as float64; prior is exactly representable
    (float64 values stored in a longdouble array by the reference)."""
    p = utils.q_to_p(np.arange(NQ, dtype=np.int_)).astype(np.float64)
    prior = utils.RescaledNormal.prior_dist.astype(np.float64)
    assert np.all((prior.astype(np.longdouble) == utils.RescaledNormal.prior_dist)
                  | ~np.isfinite(prior))
    # xlogy(1, p) and xlog1py(1, -p) as SciPy evaluates them, by the library's restatement (csrc/solve_host.cpp: the C
    # library's log and log1p) -- importing scipy.special here was 0.25 s of the command line's warm-up;
    # tests/test_solve_core_host.py compares the tables with SciPy's bit for bit
    from . import _native as N
    p = np.ascontiguousarray(p)
    logp, log1mp = np.empty(NQ), np.empty(NQ)
    N.check(N.load().kbbq_xlogy_tables_host(N.ptr(p), NQ, N.ptr(logp), N.ptr(log1mp)))
    return np.ascontiguousarray(np.concatenate([prior, logp, log1mp]))


def model_consts_scipy():
    """The same three tables with SciPy's own xlogy / xlog1py calls: the definition model_consts() is tested against."""
    import scipy.special
    p = utils.q_to_p(np.arange(NQ, dtype=np.int_)).astype(np.float64)
    prior = utils.RescaledNormal.prior_dist.astype(np.float64)
    with np.errstate(divide='ignore'):
        logp = scipy.special.xlogy(1.0, p)
        log1mp = scipy.special.xlog1py(1.0, -p)
    return np.ascontiguousarray(np.concatenate([prior, logp, log1mp]))


COMBILN_THREADS = None          # None: one thread per ~1700 cells, at most 16 (8 for a one-read-group model); an int fixes it


def combiln_scipy(numerrs, numtotal):
    """The candidate-independent term of logpmf(errs + 1; total + 2, p) with SciPy's own calls
    (scipy/stats/_discrete_distns.py binom_gen._logpmf): the definition the native routine is
    tested against, bit for bit."""
    import scipy.special
    x = np.asarray(numerrs) + 1
    n = np.asarray(numtotal) + 2
    k = np.floor(x)
    with np.errstate(all='ignore'):
        return scipy.special.gammaln(n + 1) - (scipy.special.gammaln(k + 1) + scipy.special.gammaln(n - k + 1))


def combiln(numerrs, numtotal):
    """The candidate-independent term of logpmf(errs + 1; total + 2, p), float64 per cell, by the
    library's restatement of SciPy's gammaln (csrc/solve_host.cpp: one fused pass on a few parked
    threads instead of three ufunc passes).  Cells outside the distribution's support (errs > total,
    negative counts) come back NaN; the solve never reads them."""
    from . import _native as N
    e = np.ascontiguousarray(np.asarray(numerrs), dtype=np.int64)
    t = np.ascontiguousarray(np.asarray(numtotal), dtype=np.int64)
    assert e.shape == t.shape
    out = np.empty(e.shape, dtype=np.float64)
    threads = COMBILN_THREADS if COMBILN_THREADS else max(1, min(16, e.size // 1700))
    N.check(N.load().kbbq_combiln_host(N.ptr(e), N.ptr(t), e.size, N.ptr(out), threads))
    return out


def _round64(t):
    """Round a positive Python int to 64 significant bits, ties to even -> (mantissa, shift)."""
    n = t.bit_length()
    if n <= 64:
        return t, 0
    sh = n - 64
    m, rem = t >> sh, t & ((1 << sh) - 1)
    half = 1 << (sh - 1)
    if rem > half or (rem == half and (m & 1)):
        m += 1
        if m == 1 << 64:
            m >>= 1; sh += 1
    return m, sh


def sequential_constant_sum(p, n):
    """Exactly the value an x87 longdouble accumulator holds after adding the float64 `p` to it
    `n` times, one add at a time (np.add.at on the longdouble expected_errs, recalibrate.py:111),
    in O(64) big-integer steps instead of n.  Needed when every counted base of a read group has
    the same quality: the mean error is then p itself up to the accumulated rounding, and p_to_q
    truncates right at that boundary (the reference's "float badness", tests/test_recalibrate.py:63)."""
    if n <= 0 or p == 0:
        return np.longdouble(0)
    mp, ep = np.frexp(np.float64(p))
    mp = int(mp * (1 << 53)); ep = int(ep) - 53               # p = mp * 2**ep exactly
    S, E = mp, ep                                             # accumulator after the first add: p itself
    n -= 1

    def add(S, E):                                            # one exact add, rounded to 64 bits
        e0 = min(E, ep)
        m, sh = _round64((S << (E - e0)) + (mp << (ep - e0)))
        return m, e0 + sh

    while n > 0:
        S1, E1 = add(S, E)
        if n < 8 or S1.bit_length() < 64 or E1 != E:
            S, E = S1, E1                                     # few adds left, still growing, or binade changed
            n -= 1
            continue
        # full-width mantissa in binade E: p = a*ulp + r with fixed a, r, so after at most one
        # irregular add (an exact tie landing on an odd mantissa) the increment is constant.  Look
        # two more adds ahead and jump while the mantissa stays below 2**64.
        S2, E2 = add(S1, E1)
        S3, E3 = add(S2, E2)
        if E2 != E or E3 != E or (S3 - S2) != (S2 - S1):
            S, E = S1, E1
            n -= 1
            continue
        inc = S2 - S1
        S, n = S1, n - 1
        k = min(n, ((1 << 64) - 1 - S) // inc) if inc > 0 else n
        # leave the last add before the binade boundary to the exact path
        k = max(0, k - 1)
        S += inc * k
        n -= k
    return np.ldexp(np.longdouble(S), E)


def meanq_from_q_total(q_total, maxscore=42):
    """meanq = p_to_q(sum_q q_total * 10^(-q/10) / rg_total) in longdouble (recalibrate.py:111,120)."""
    q_total = np.asarray(q_total)
    rg_total = q_total.sum(axis=1)
    p = utils.q_to_p(np.arange(maxscore + 1))
    expected = (q_total.astype(np.longdouble) * p).sum(axis=1)
    for r in range(q_total.shape[0]):
        nz = np.flatnonzero(q_total[r])
        if nz.size == 1:      # one quality value only: reproduce the reference's add-by-add rounding
            expected[r] = sequential_constant_sum(np.float64(p[nz[0]]), int(q_total[r, nz[0]]))
    with np.errstate(divide='ignore', invalid='ignore'):
        return utils.p_to_q(expected / rg_total, maxscore)


def solve_prep(flat_tables, R, S2):
    """The host half of one solve from the flat int64 table buffer as it comes off the device, in one threaded native
    pass (csrc/solve_host.cpp kbbq_solve_prep_host): (aux, q_errs, q_total, rg_errs, rg_total) -- aux = the gammaln
    term of every cell in kbbq_solve_dev's order, the others the reference's marginals (recalibrate.py:112-115)."""
    from . import _native as N
    flat = np.ascontiguousarray(flat_tables, dtype=np.int64)
    rows = R * NQ
    assert flat.size == 2 * rows * S2 + 2 * rows * 16
    aux = np.empty(R + rows + rows * S2 + rows * 16, dtype=np.float64)
    marg = np.empty(2 * rows + 2 * R, dtype=np.int64)
    threads = COMBILN_THREADS if COMBILN_THREADS else max(1, min(16, aux.size // 1700))
    N.check(N.load().kbbq_solve_prep_host(N.ptr(flat), R, S2, N.ptr(aux), N.ptr(marg), threads))
    return (aux, marg[:rows].reshape(R, NQ), marg[rows:2 * rows].reshape(R, NQ), marg[2 * rows:2 * rows + R], marg[2 * rows + R:])


def vectors_from_tables(pos_errs, pos_total, dinuc_errs, dinuc_total, maxscore=42):
    """The reference's 9-tuple from the four 3-d count arrays: q_* and rg_* are marginals of
    pos_* (every counted base has exactly one cycle); meanq = p_to_q(sum_q q_total * 10^(-q/10)
    / rg_total) in longdouble (recalibrate.py:111,120)."""
    q_errs, q_total = pos_errs.sum(axis=2), pos_total.sum(axis=2)
    rg_errs, rg_total = q_errs.sum(axis=1), q_total.sum(axis=1)
    meanq = meanq_from_q_total(q_total, maxscore)
    return meanq, rg_errs, rg_total, q_errs, q_total, pos_errs, pos_total, dinuc_errs, dinuc_total
