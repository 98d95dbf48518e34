// solve_core.h -- one cell of the delta-Q model solve, host/device.
//
// Restates compare_reads.gatk_delta_q (reference compare_reads.py:235-260) for ONE cell:
//     argmax over q' = 0..42 of   prior[|q' - prior_q|]  +  logpmf(errs+1; total+2, p[q'])
// with scipy.stats.binom.logpmf spelled out as SciPy 1.15 evaluates it
// (scipy/stats/_discrete_distns.py binom_gen._logpmf):
//     combiln + xlogy(k, p) + xlog1py(n - k, -p),  k = errs+1, n = total+2 (float64, left to right)
// The transcendental pieces are NOT recomputed here: `comb` (three gammaln calls) and the
// 43-entry tables logp[q'] = log(p), log1mp[q'] = log1p(-p) come from the host, produced by
// the very SciPy functions the reference calls, so every float64 below is the product or sum
// of the same doubles in the same order (compile with -ffp-contract=off).  The float64
// log-likelihood is then added to the longdouble prior with the x87 emulation of x87add.h
// and the FIRST maximum wins (np.argmax).
#pragma once
#include "x87add.h"

#define KSOLVE_NQ 43

struct SolveConsts {
    double prior[KSOLVE_NQ];    // RescaledNormal.prior_dist: log(.9) - 2 d^2, -inf from d = 19
    double logp[KSOLVE_NQ];     // scipy.special.xlogy(1.0, p[q'])
    double log1mp[KSOLVE_NQ];   // scipy.special.xlog1py(1.0, -p[q'])
};

// PRIOR selects how |q' - prior_q| is formed: the reference subtracts in the prior's own dtype and
// truncates toward zero (np.subtract.outer(...).astype(int), compare_reads.py:246): exact for an
// integer prior, and for a float64 prior (bqsr.vectors_to_report's EstimatedQReported) two
// neighbouring candidates can share distance 0.
template <typename PRIOR>
X87_HD int solve_cell(const SolveConsts& c, PRIOR prior_q, long long errs, long long total, double comb)
{
    const long long kk = errs + 1, nn = total + 2;
    // rv_discrete.logpmf masks: outside the support everything is -inf (or nan for n < 0):
    // every candidate ties and np.argmax returns 0
    if (nn < 0 || kk < 0 || kk > nn) return 0;
    const double k = (double)kk;
    const double nk = (double)nn - k;
    x87val best = x87_from_special(2);
    int arg = 0;
    for (int cand = 0; cand < KSOLVE_NQ; ++cand) {
        int diff = (int)((PRIOR)cand - prior_q); if (diff < 0) diff = -diff;
        const double pr = diff < KSOLVE_NQ ? c.prior[diff] : c.prior[KSOLVE_NQ - 1];
        const double t1 = k * c.logp[cand];
        const double t2 = nk * c.log1mp[cand];
        const double ll = (comb + t1) + t2;
        const x87val post = x87_add(pr, ll);
        if (cand == 0 || x87_gt(post, best)) { best = post; arg = cand; }
    }
    return arg;
}
