// lgam_core.h -- the candidate-independent term of the model solve, host/device, WITHOUT a call into libm.
//
//     combiln = gammaln(n + 1) - (gammaln(k + 1) + gammaln(n - k + 1)),   k = errs + 1, n = total + 2
// (scipy/stats/_discrete_distns.py binom_gen._logpmf, reached from compare_reads.py:254).  csrc/solve_host.cpp
// restates SciPy's gammaln (xsf::cephes::lgam) over the process's libm `log`; that `log` is the only piece a GPU
// cannot call.  This header restates it too, so that the whole solve can stay on the device (no D2H of the count
// tables, no host pass, no H2D of the terms inside a recalibration step):
//
//   * The algorithm is glibc 2.35's double-precision log (sysdeps/ieee754/dbl-64/e_log.c, the table-driven routine
//     glibc took from ARM's optimized-routines) in the form the x86-64 FMA variant of THIS image's libm executes:
//     the operation order below was read off that variant's machine code, including the multiply-adds the compiler
//     fused beyond the source's own fma().
//   * Its constants -- ln2 split in two, 5 polynomial coefficients, 128 (1/c, log c) pairs -- are not restated: they
//     are read at run time out of the libm the process has mapped (kbbq_libm_log_data, csrc/solve_host.cpp), so the
//     numbers are by construction the ones the host's `log` uses.
//   * Nothing is taken on trust: before the device path is switched on, a kernel evaluates gammaln on > 10^6
//     arguments and every bit is compared with the host routine (kbbq/_device.py device_gammaln_ok); on any mismatch
//     (another libm, a CPU without FMA) the host pass stays in use.
//
// Only what the solve needs: arguments are integer-valued doubles >= 1 (counts + 1, + 2, + 3), so log's near-1 branch
// is never taken except at exactly 1.0 (-> 0), and lgam's reflection / pole handling does not arise.
#pragma once
#include <stdint.h>
#include <math.h>

#ifndef LGAM_HD
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define LGAM_HD __host__ __device__ __forceinline__
#else
#define LGAM_HD static inline
#endif
#endif

// layout of the table kbbq_libm_log_data fills: [ln2hi, ln2lo, A0..A4 | 128 x (invc, logc)]
#define LGAM_LOGTAB_HEAD 7
#define LGAM_LOGTAB_DOUBLES (LGAM_LOGTAB_HEAD + 256)

LGAM_HD double lgam_fma(double a, double b, double c)
{
#ifdef __HIP_DEVICE_COMPILE__
    return __builtin_fma(a, b, c);
#else
    return fma(a, b, c);
#endif
}

// log(x) for finite x >= 1 that is either exactly 1 or outside glibc's near-1 interval [1 - 2^-4, 1 + 0x1.09p-4)
LGAM_HD double lgam_log(double x, const double* T)
{
    union { double d; uint64_t u; } v; v.d = x;
    const uint64_t ix = v.u;
    if (ix == 0x3FF0000000000000ull) return 0.0;
    const uint64_t tmp = ix - 0x3FE6000000000000ull;
    const int i = (int)((tmp >> 45) & 127u);
    const int k = (int)((int64_t)tmp >> 52);
    v.u = ix - (tmp & 0xFFF0000000000000ull);
    const double z = v.d;
    const double invc = T[LGAM_LOGTAB_HEAD + 2 * i], logc = T[LGAM_LOGTAB_HEAD + 2 * i + 1];
    const double ln2hi = T[0], ln2lo = T[1], A0 = T[2], A1 = T[3], A2 = T[4], A3 = T[5], A4 = T[6];
    const double kd = (double)k;
    const double r = lgam_fma(z, invc, -1.0);
    const double w = lgam_fma(kd, ln2hi, logc);
    const double p12 = lgam_fma(r, A2, A1);
    const double hi = r + w;
    const double r2 = r * r;
    double lo = (w - hi) + r;
    lo = lgam_fma(kd, ln2lo, lo);
    const double r3 = r * r2;
    const double p34 = lgam_fma(r, A4, A3);
    lo = lgam_fma(r2, A0, lo);
    const double pp = lgam_fma(p34, r2, p12);
    return lgam_fma(r3, pp, lo) + hi;
}

// log|Gamma(x)| as xsf::cephes::lgam evaluates it (scipy/special/xsf/cephes/gamma.h:278-360) for the arguments the
// solve produces; operation for operation the routine of csrc/solve_host.cpp with lgam_log in place of std::log
LGAM_HD double lgam_horner(double x, const double* c, int n)
{
    double v = c[0];
    for (int i = 1; i <= n; ++i) v = v * x + c[i];
    return v;
}

LGAM_HD double lgam_horner1(double x, const double* c, int n)
{
    double v = x + c[0];
    for (int i = 1; i < n; ++i) v = v * x + c[i];
    return v;
}

LGAM_HD double lgam_count(double x, const double* T)
{
    const double A[] = {8.11614167470508450300E-4, -5.95061904284301438324E-4, 7.93650340457716943945E-4,
                        -2.77777777730099687205E-3, 8.33333333333331927722E-2};
    const double B[] = {-1.37825152569120859100E3, -3.88016315134637840924E4, -3.31612992738871184744E5,
                        -1.16237097492762307383E6, -1.72173700820839662146E6, -8.53555664245765465627E5};
    const double C[] = {-3.51815701436523470549E2, -1.70642106651881159223E4, -2.20528590553854454839E5,
                        -1.13933444367982507207E6, -2.53252307177582951285E6, -2.01889141433532773231E6};
    const double LS2PI = 0.91893853320467274178;
    if (x < 13.0) {
        double z = 1.0, p = 0.0, u = x;
        while (u >= 3.0) { p -= 1.0; u = x + p; z *= u; }
        while (u < 2.0) { z /= u; p += 1.0; u = x + p; }
        if (z < 0.0) z = -z;
        if (u == 2.0) return lgam_log(z, T);
        p -= 2.0;
        x = x + p;
        p = x * lgam_horner(x, B, 5) / lgam_horner1(x, C, 6);
        return lgam_log(z, T) + p;
    }
    if (x >= 1000.0) {
        const double q = (x - 0.5) * lgam_log(x, T) - x + LS2PI;
        if (x > 1.0e8) return q;
        double p = 1.0 / (x * x);
        p = ((7.9365079365079365079365e-4 * p - 2.7777777777777777777778e-3) * p + 0.0833333333333333333333) / x;
        return q + p;
    }
    const double q = (x - 0.5) * lgam_log(x, T) - x + LS2PI;
    const double p = 1.0 / (x * x);
    return q + lgam_horner(p, A, 4) / x;
}

// the term itself; NaN outside the distribution's support (the solve ignores it there, csrc/solve_core.h)
LGAM_HD double lgam_combiln(long long errs, long long total, const double* T)
{
    const double k = (double)(errs + 1);
    const double n = (double)(total + 2);
    const double a = n + 1.0, b = k + 1.0, c = n - k + 1.0;
    if (!(a > 0.0 && b > 0.0 && c > 0.0)) {
        union { double d; uint64_t u; } v; v.u = 0x7FF8000000000000ull;
        return v.d;
    }
    return lgam_count(a, T) - (lgam_count(b, T) + lgam_count(c, T));
}
