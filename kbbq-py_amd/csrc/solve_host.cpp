// solve_host.cpp -- host half of the model solve (K3): the candidate-independent term of
// scipy.stats.binom.logpmf(errs + 1; total + 2, p),
//     combiln = gammaln(n + 1) - (gammaln(k + 1) + gammaln(n - k + 1)),   k = errs + 1, n = total + 2
// (scipy/stats/_discrete_distns.py binom_gen._logpmf, called by the reference at
// compare_reads.py:254), for every cell of the count tables.
//
// gammaln here is a restatement of the routine SciPy 1.15.3 -- the reference's pinned dependency in this
// image -- evaluates: xsf::cephes::lgam (scipy/special/xsf/cephes/gamma.h:278-360, Cephes Math Library
// 2.2, S. L. Moshier) for positive arguments: x < 13 by recurrence to [2, 3) and a rational function,
// x >= 13 by Stirling's series with Cephes' two coefficient sets.  Every operation is the same IEEE double
// operation in the same order (compiled with -ffp-contract=off; std::log is the process's libm, the one
// SciPy calls), so the value is bit-identical to scipy.special.gammaln: tests/test_solve_core_host.py
// checks that over millions of arguments.  Why it exists: three SciPy ufunc passes over 13.6 k cells cost
// 0.32 ms of a 10 ms bench step; one fused threaded pass costs a fraction of it.
#include "../../include/kbbq_hip.h"
#include "lgam_core.h"

#include <link.h>
#include <cstring>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <condition_variable>
#include <limits>
#include <mutex>
#include <functional>
#include <thread>
#include <vector>

int kbbq_set_error_(int code, const char* msg);      // defined in kbbq_hip.hip

namespace {

const double LGAM_A[] = {8.11614167470508450300E-4, -5.95061904284301438324E-4, 7.93650340457716943945E-4,
                         -2.77777777730099687205E-3, 8.33333333333331927722E-2};
const double LGAM_B[] = {-1.37825152569120859100E3, -3.88016315134637840924E4, -3.31612992738871184744E5,
                         -1.16237097492762307383E6, -1.72173700820839662146E6, -8.53555664245765465627E5};
const double LGAM_C[] = {-3.51815701436523470549E2, -1.70642106651881159223E4, -2.20528590553854454839E5,
                         -1.13933444367982507207E6, -2.53252307177582951285E6, -2.01889141433532773231E6};
const double LS2PI = 0.91893853320467274178;     // log(sqrt(2 pi))
const double MAXLGM = 2.556348e305;

inline double horner(double x, const double* c, int n)          // c[0] x^n + ... + c[n]
{
    double v = c[0];
    for (int i = 1; i <= n; ++i) v = v * x + c[i];
    return v;
}

inline double horner1(double x, const double* c, int n)         // x^n + c[0] x^(n-1) + ... + c[n-1]
{
    double v = x + c[0];
    for (int i = 1; i < n; ++i) v = v * x + c[i];
    return v;
}

// log|Gamma(x)| for finite x > 0 (the only arguments a valid cell produces: all >= 2)
double lgam_positive(double x)
{
    if (x < 13.0) {
        double z = 1.0, p = 0.0, u = x;
        while (u >= 3.0) { p -= 1.0; u = x + p; z *= u; }
        while (u < 2.0) { z /= u; p += 1.0; u = x + p; }
        if (z < 0.0) z = -z;
        if (u == 2.0) return std::log(z);
        p -= 2.0;
        x = x + p;
        p = x * horner(x, LGAM_B, 5) / horner1(x, LGAM_C, 6);
        return std::log(z) + p;
    }
    if (x > MAXLGM) return std::numeric_limits<double>::infinity();
    if (x >= 1000.0) {
        const double q = (x - 0.5) * std::log(x) - x + LS2PI;
        if (x > 1.0e8) return q;
        double p = 1.0 / (x * x);
        p = ((7.9365079365079365079365e-4 * p - 2.7777777777777777777778e-3) * p + 0.0833333333333333333333) / x;
        return q + p;
    }
    const double q = (x - 0.5) * std::log(x) - x + LS2PI;
    const double p = 1.0 / (x * x);
    return q + horner(p, LGAM_A, 4) / x;
}

void combiln_range(const int64_t* errs, const int64_t* total, int64_t lo, int64_t hi, double* out)
{
    for (int64_t i = lo; i < hi; ++i) {
        const double k = (double)(errs[i] + 1);
        const double n = (double)(total[i] + 2);
        const double a = n + 1.0, b = k + 1.0, c = n - k + 1.0;
        // outside the distribution's support the solve ignores this term (csrc/solve_core.h)
        if (!(a > 0.0 && b > 0.0 && c > 0.0)) { out[i] = std::numeric_limits<double>::quiet_NaN(); continue; }
        out[i] = lgam_positive(a) - (lgam_positive(b) + lgam_positive(c));
    }
}

// A few parked worker threads: spawning threads per call costs more than the work (0.1 ms vs 0.25 ms).
struct Pool {
    static const int MAX = 16;
    std::mutex m;
    std::condition_variable wake, done;
    std::vector<std::thread> workers;
    std::function<void(int64_t, int64_t)> job;      // [lo, hi) of the current job's items
    int64_t n = 0, per = 0;
    int parts = 0;            // slices of the current job (slice 0 is the caller's)
    uint64_t generation = 0;
    int pending = 0;
    bool stop = false;

    void worker(int id)
    {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(m);
            wake.wait(lk, [&] { return stop || generation != seen; });
            if (stop) return;
            seen = generation;
            const bool mine = id < parts;
            const int64_t lo = std::min<int64_t>(n, (int64_t)id * per), hi = std::min<int64_t>(n, lo + per);
            lk.unlock();
            if (mine) {
                job(lo, hi);                 // `job` is not reassigned before every slice has reported (pending == 0)
                lk.lock();
                if (--pending == 0) done.notify_one();
            }
        }
    }

    void run(std::function<void(int64_t, int64_t)> f, int64_t count, int threads)
    {
        std::unique_lock<std::mutex> lk(m);
        while ((int)workers.size() < threads - 1) {
            const int id = (int)workers.size() + 1;
            workers.emplace_back([this, id] { worker(id); });
            workers.back().detach();
        }
        job = std::move(f); n = count; parts = threads;
        per = (count + threads - 1) / threads;
        pending = threads - 1;
        ++generation;
        lk.unlock();
        wake.notify_all();
        job(0, std::min<int64_t>(count, per));
        lk.lock();
        done.wait(lk, [&] { return pending == 0; });
    }

};

// workers are started on first use and parked on a condition variable; the pool is never torn down
// (detached threads, no work at process exit: nothing to interleave with the HIP runtime's own shutdown)
Pool& pool() { static Pool* p = new Pool; return *p; }

}  // namespace

extern "C" {

int kbbq_gammaln_host(const double* x, int64_t n, double* out)
{
    if (n < 0 || (n > 0 && (!x || !out))) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_gammaln_host: bad argument");
    for (int64_t i = 0; i < n; ++i)
        out[i] = (x[i] > 0.0 && std::isfinite(x[i])) ? lgam_positive(x[i]) : std::numeric_limits<double>::quiet_NaN();
    return KBBQ_OK;
}

// The host half of one model solve in ONE pass over the count tables as they come off the device
// ([pos_errs R x 43 x S2 | pos_total | dinuc_errs R x 43 x 16 | dinuc_total], int64): the marginals the reference takes
// (recalibrate.py:112-115: q_* = sum over cycles, rg_* = sum over qualities) and the candidate-independent gammaln
// term of every cell, in the order kbbq_solve_dev reads them ([rg R | q R x 43 | pos R x 43 x S2 | dinuc R x 43 x 16]).
// marg = [q_errs R x 43 | q_total R x 43 | rg_errs R | rg_total R].  Rows (read group, quality) are spread over threads.
int kbbq_solve_prep_host(const int64_t* tables, int R, int S2, double* aux, int64_t* marg, int threads)
{
    if (!tables || !aux || !marg || R <= 0 || S2 <= 0) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_solve_prep_host: bad argument");
    const int Q = 43;
    const int64_t npos = (int64_t)R * Q * S2, ndn = (int64_t)R * Q * 16, rows = (int64_t)R * Q;
    const int64_t *pe = tables, *pt = tables + npos, *de = tables + 2 * npos, *dt = de + ndn;
    double* aux_rg = aux; double* aux_q = aux + R; double* aux_pos = aux_q + rows; double* aux_dn = aux_pos + npos;
    int64_t *q_errs = marg, *q_total = marg + rows, *rg_errs = marg + 2 * rows, *rg_total = rg_errs + R;
    auto do_rows = [&](int64_t lo, int64_t hi) {
        for (int64_t row = lo; row < hi; ++row) {
            const int64_t* e = pe + row * S2; const int64_t* t = pt + row * S2;
            int64_t se = 0, st = 0;
            for (int c = 0; c < S2; ++c) { se += e[c]; st += t[c]; }
            q_errs[row] = se; q_total[row] = st;
            combiln_range(e, t, 0, S2, aux_pos + row * S2);
            combiln_range(de + row * 16, dt + row * 16, 0, 16, aux_dn + row * 16);
        }
    };
    threads = std::max(1, std::min<int>(threads, (int)std::min<int64_t>(rows, Pool::MAX)));
    if (threads == 1) do_rows(0, rows);
    else pool().run(do_rows, rows, threads);
    for (int r = 0; r < R; ++r) {
        int64_t se = 0, st = 0;
        for (int q = 0; q < Q; ++q) { se += q_errs[(int64_t)r * Q + q]; st += q_total[(int64_t)r * Q + q]; }
        rg_errs[r] = se; rg_total[r] = st;
    }
    combiln_range(q_errs, q_total, 0, rows, aux_q);
    combiln_range(rg_errs, rg_total, 0, R, aux_rg);
    return KBBQ_OK;
}

// ---- the constants of the host's own `log`, for the device restatement (csrc/lgam_core.h) --------------------
// glibc keeps them in one read-only structure (__log_data: ln2hi, ln2lo, poly[5], poly1[11], 128 x {invc, logc}; not
// an exported symbol).  It is located in the libm image this process has mapped by the bit patterns of its first two
// members -- the split of ln 2 that the routine was published with -- and accepted only if the 128 pairs that follow
// look like what they must be: 1/c in (0.7, 1.5) descending, log c within 2^-20 of -log(invc).
namespace {
struct LogDataSearch { double* out; bool found; };

bool plausible_log_data(const double* d)
{
    if (!(d[2] < -0.49 && d[2] > -0.51)) return false;                 // A0 ~ -1/2
    const double* tab = d + 18;                                        // 2 + 5 + 11 doubles precede the table
    double prev = 2.0;
    for (int i = 0; i < 128; ++i) {
        const double invc = tab[2 * i], logc = tab[2 * i + 1];
        if (!(invc > 0.7 && invc < 1.5) || !(invc < prev)) return false;
        if (std::fabs(logc + std::log(invc)) > 1e-6) return false;
        prev = invc;
    }
    return true;
}

int scan_object(struct dl_phdr_info* info, size_t, void* arg)
{
    LogDataSearch* s = static_cast<LogDataSearch*>(arg);
    if (s->found || !info->dlpi_name || !strstr(info->dlpi_name, "libm")) return 0;
    const uint64_t ln2hi = 0x3FE62E42FEFA3800ull, ln2lo = 0x3D2EF35793C76730ull;
    for (int h = 0; h < info->dlpi_phnum; ++h) {
        const ElfW(Phdr)& ph = info->dlpi_phdr[h];
        if (ph.p_type != PT_LOAD || !(ph.p_flags & PF_R) || ph.p_filesz < (18 + 256) * 8) continue;
        const unsigned char* base = reinterpret_cast<const unsigned char*>(info->dlpi_addr + ph.p_vaddr);
        const size_t span = ph.p_filesz - (18 + 256) * 8;
        for (size_t off = ((8 - (reinterpret_cast<uintptr_t>(base) & 7)) & 7); off <= span; off += 8) {
            uint64_t a, b;
            memcpy(&a, base + off, 8);
            if (a != ln2hi) continue;
            memcpy(&b, base + off + 8, 8);
            if (b != ln2lo) continue;
            double d[18 + 256];
            memcpy(d, base + off, sizeof d);
            if (!plausible_log_data(d)) continue;
            memcpy(s->out, d, 7 * 8);                                  // ln2hi, ln2lo, A0..A4
            memcpy(s->out + LGAM_LOGTAB_HEAD, d + 18, 256 * 8);        // 128 x (invc, logc)
            s->found = true;
            return 1;
        }
    }
    return 0;
}
}  // namespace

extern "C" int kbbq_libm_log_data(double* out, int count)
{
    if (!out || count < LGAM_LOGTAB_DOUBLES) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_libm_log_data: need room for 263 doubles");
    (void)std::log(2.0);                                               // libm is mapped (and its ifunc resolved)
    LogDataSearch s{out, false};
    dl_iterate_phdr(scan_object, &s);
    if (!s.found) return kbbq_set_error_(KBBQ_E_HIP, "kbbq_libm_log_data: the constants of this libm's log() were not found; the host pass of the solve stays in use");
    return KBBQ_OK;
}

// gammaln of the solve's arguments through csrc/lgam_core.h on the HOST (no call into libm): what the device evaluates,
// for tests without a GPU.  x: integer-valued doubles >= 1.
extern "C" int kbbq_gammaln_restated_host(const double* x, int64_t n, const double* logtab, double* out)
{
    if (n < 0 || (n > 0 && (!x || !out)) || !logtab) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_gammaln_restated_host: bad argument");
    for (int64_t i = 0; i < n; ++i) out[i] = lgam_count(x[i], logtab);
    return KBBQ_OK;
}

int kbbq_combiln_host(const int64_t* errs, const int64_t* total, int64_t n, double* out, int threads)
{
    if (n < 0 || (n > 0 && (!errs || !total || !out))) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_combiln_host: bad argument");
    if (threads < 1) threads = 1;
    if (threads > Pool::MAX) threads = Pool::MAX;
    if (n < 4096 || threads == 1) { combiln_range(errs, total, 0, n, out); return KBBQ_OK; }
    pool().run([=](int64_t lo, int64_t hi) { combiln_range(errs, total, lo, hi, out); }, n, threads);
    return KBBQ_OK;
}

}  // extern "C"

// The two 43-entry log tables of the solve without importing SciPy (its import was 0.25 s of the command line's warm-up):
// scipy.stats.binom.logpmf forms xlogy(k, p) + xlog1py(n - k, -p) (compare_reads.py:254 -> scipy/stats/_discrete_distns.py
// binom_gen._logpmf); for real arguments SciPy 1.15 evaluates xlogy(x, y) = x * log(y) and xlog1py(x, y) = x * log1p(y) with
// the C library's log / log1p (measured: bit-equal to libm's on 400 000 arguments; NOT the cephes log1p its xsf headers also
// carry).  logp[i] = xlogy(1, p[i]), log1mp[i] = xlog1py(1, -p[i]); tests/test_solve_core_host.py compares both with
// SciPy's own calls, bit for bit, wherever the tests run.
extern "C" int kbbq_xlogy_tables_host(const double* p, int n, double* logp, double* log1mp)
{
    if (n < 0 || (n > 0 && (!p || !logp || !log1mp))) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_xlogy_tables_host: bad argument");
    for (int i = 0; i < n; ++i) {
        logp[i] = 1.0 * log(p[i]);
        log1mp[i] = 1.0 * log1p(-p[i]);
    }
    return KBBQ_OK;
}

