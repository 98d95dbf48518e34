// x87add.h -- exact emulation of an x87 80-bit extended-precision ADD of two doubles.
//
// Why: the reference's model solve adds a float64 log-likelihood to a np.longdouble
// prior (compare_reads.py:257) and takes np.argmax of the longdouble sums.  On x86-64
// np.longdouble is the x87 80-bit format (64-bit significand).  Two candidates whose
// exact sums differ can round to the same 80-bit value and then the FIRST wins
// (np.argmax), so the device solve (K3) must compare the ROUNDED values.  The GPU has
// no 80-bit type; this header rounds the exact sum of two doubles to 64 significant
// bits (round-to-nearest-even, as the x87 default control word does) with integer
// arithmetic and orders the results.
//
// Both inputs are doubles (the prior table holds float64 values widened to
// longdouble, see tests/golden/numeric.json "prior_dist_is_float64_exact"), finite or
// -inf.  The extended format's 15-bit exponent cannot overflow or underflow on sums of
// doubles, so (sign, exponent, 64-bit significand) describes every result exactly.
//
// Compiles as host code too (tests/native/x87_check.cpp checks it against the CPU's own
// long double arithmetic on a few million pairs).
#pragma once
#include <stdint.h>

#ifndef X87_HD
#ifdef __HIPCC__
#define X87_HD __host__ __device__ __forceinline__
#else
#define X87_HD static inline
#endif
#endif

struct x87val {
    uint64_t mant;   // significand, bit 63 set for finite non-zero values
    int32_t exp;     // exponent of bit 63 (value = mant * 2^(exp - 63))
    int32_t cls;     // 0: finite non-zero   1: zero   2: -inf   3: +inf   4: nan
    int32_t neg;     // sign (finite non-zero only)
};

X87_HD int x87_clz64(uint64_t x)
{
#ifdef __HIP_DEVICE_COMPILE__
    return __clzll((long long)x);
#else
    return __builtin_clzll(x);
#endif
}

X87_HD x87val x87_from_special(int cls)
{
    x87val r; r.mant = 0; r.exp = 0; r.cls = cls; r.neg = 0; return r;
}

// decompose a finite non-zero double into sign, integer significand m (<= 53 bits) and
// exponent e with |x| = m * 2^e
X87_HD void x87_split(double x, int& neg, uint64_t& m, int& e)
{
    union { double d; uint64_t u; } v; v.d = x;
    neg = (int)(v.u >> 63);
    const int be = (int)((v.u >> 52) & 0x7FF);
    const uint64_t frac = v.u & 0xFFFFFFFFFFFFFull;
    if (be == 0) { m = frac; e = -1074; }
    else { m = frac | (1ull << 52); e = be - 1075; }
}

X87_HD int x87_class(double x)
{
    union { double d; uint64_t u; } v; v.d = x;
    const int be = (int)((v.u >> 52) & 0x7FF);
    const uint64_t frac = v.u & 0xFFFFFFFFFFFFFull;
    if (be == 0x7FF) return frac ? 4 : ((v.u >> 63) ? 2 : 3);
    if (be == 0 && frac == 0) return 1;
    return 0;
}

// 128-bit helpers on (hi, lo) pairs -- kept explicit so host and device agree bit for bit
struct x87u128 { uint64_t hi, lo; };

X87_HD x87u128 x87_shr(x87u128 a, int s, bool& lost)      // 0 <= s; `lost` |= bits shifted out
{
    if (s == 0) return a;
    x87u128 r;
    if (s >= 128) { lost = lost || (a.hi | a.lo) != 0; r.hi = 0; r.lo = 0; return r; }
    if (s >= 64) {
        const int t = s - 64;
        lost = lost || a.lo != 0 || (t ? (a.hi & ((1ull << t) - 1)) != 0 : false);
        r.hi = 0; r.lo = t ? (a.hi >> t) : a.hi;
        return r;
    }
    lost = lost || (a.lo & ((1ull << s) - 1)) != 0;
    r.lo = (a.lo >> s) | (a.hi << (64 - s));
    r.hi = a.hi >> s;
    return r;
}

X87_HD x87u128 x87_add128(x87u128 a, x87u128 b)
{
    x87u128 r; r.lo = a.lo + b.lo; r.hi = a.hi + b.hi + (r.lo < a.lo ? 1 : 0); return r;
}

X87_HD x87u128 x87_sub128(x87u128 a, x87u128 b)             // a >= b
{
    x87u128 r; r.lo = a.lo - b.lo; r.hi = a.hi - b.hi - (a.lo < b.lo ? 1 : 0); return r;
}

X87_HD int x87_cmp128(x87u128 a, x87u128 b)
{
    if (a.hi != b.hi) return a.hi < b.hi ? -1 : 1;
    if (a.lo != b.lo) return a.lo < b.lo ? -1 : 1;
    return 0;
}

// round-to-nearest-even of (a + b) to a 64-bit significand
X87_HD x87val x87_add(double a, double b)
{
    const int ca = x87_class(a), cb = x87_class(b);
    if (ca == 4 || cb == 4) return x87_from_special(4);
    if (ca >= 2 || cb >= 2) {
        if (ca >= 2 && cb >= 2) return x87_from_special(ca == cb ? ca : 4);   // inf - inf = nan
        return x87_from_special(ca >= 2 ? ca : cb);
    }
    if (ca == 1 && cb == 1) return x87_from_special(1);
    int na = 0, nb = 0, ea = 0, eb = 0; uint64_t ma = 0, mb = 0;
    if (ca == 0) x87_split(a, na, ma, ea);
    if (cb == 0) x87_split(b, nb, mb, eb);
    if (ca == 1 || cb == 1) {                                  // x + 0: exact, just normalise
        const uint64_t m = ca == 1 ? mb : ma;
        const int e = ca == 1 ? eb : ea;
        const int lz = x87_clz64(m);
        x87val r; r.mant = m << lz; r.exp = e + 63 - lz; r.cls = 0; r.neg = ca == 1 ? nb : na;
        return r;
    }
    // operand with the larger exponent first
    if (eb > ea) { int t = ea; ea = eb; eb = t; t = na; na = nb; nb = t; uint64_t u = ma; ma = mb; mb = u; }
    const int d = ea - eb;
    x87u128 A; A.hi = ma; A.lo = 0;                            // ma * 2^64, unit 2^(ea - 64)
    x87u128 B; B.hi = mb; B.lo = 0;
    bool sticky = false;
    B = x87_shr(B, d, sticky);
    x87u128 R; int neg;
    if (na == nb) { R = x87_add128(A, B); neg = na; }
    else {
        const int c = x87_cmp128(A, B);
        if (c == 0 && !sticky) return x87_from_special(1);     // exact cancellation
        if (c > 0 || (c == 0 && sticky)) {
            // |a| > |b|: A - (B + fraction) = (A - B - 1) + (1 - fraction)
            R = x87_sub128(A, B);
            if (sticky) { x87u128 one; one.hi = 0; one.lo = 1; R = x87_sub128(R, one); }
            neg = na;
            if (c == 0) { /* cannot happen: B + fraction > A means |b| > |a| with d > 0 */ }
        } else {
            R = x87_sub128(B, A);                              // only when d == 0 (no sticky)
            neg = nb;
        }
    }
    // position of the leading one (R != 0 here, or R == 0 with sticky which cannot occur:
    // sticky needs d > 64, and then A - B - 1 >= 2^116 - 2^53)
    int top;
    if (R.hi) top = 127 - x87_clz64(R.hi); else top = 63 - x87_clz64(R.lo);
    x87val r; r.cls = 0; r.neg = neg;
    if (top <= 63) {                                           // fits: exact (sticky is false here)
        r.mant = R.lo << (63 - top);
        r.exp = ea - 64 + top;
        return r;
    }
    const int sh = top - 63;                                   // 1 .. 64 bits to drop
    bool rest = sticky;
    // guard bit = bit (sh - 1); bits below it go into `rest`
    x87u128 G = x87_shr(R, sh - 1, rest);
    const bool guard = (G.lo & 1ull) != 0;
    bool dummy = false;
    x87u128 M = x87_shr(G, 1, dummy);                          // now exactly 64 significant bits
    uint64_t m = M.lo;
    int e = ea - 64 + top;
    if (guard && (rest || (m & 1ull))) {
        m += 1;
        if (m == 0) { m = 1ull << 63; e += 1; }                // carried out of 64 bits
    }
    r.mant = m; r.exp = e;
    return r;
}

// strict "a > b" on rounded values; nan is never greater and nothing is greater than nan
// (the caller handles np.argmax's NaN rule separately)
X87_HD bool x87_gt(const x87val& a, const x87val& b)
{
    if (a.cls == 4 || b.cls == 4) return false;
    // map to an ordering: -inf < negative finite < zero < positive finite < +inf
    const int ra = a.cls == 2 ? -2 : a.cls == 3 ? 2 : a.cls == 1 ? 0 : (a.neg ? -1 : 1);
    const int rb = b.cls == 2 ? -2 : b.cls == 3 ? 2 : b.cls == 1 ? 0 : (b.neg ? -1 : 1);
    if (ra != rb) return ra > rb;
    if (ra == 1) return a.exp != b.exp ? a.exp > b.exp : a.mant > b.mant;
    if (ra == -1) return a.exp != b.exp ? a.exp < b.exp : a.mant < b.mant;
    return false;
}
