// Host check of kbbq-py_amd/csrc/x87add.h against the CPU's own x87 long double arithmetic.
// Built and run by tests/test_x87_emulation.py (g++, no GPU).  Prints "OK <n>" or the
// first mismatch.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../../kbbq-py_amd/csrc/x87add.h"

static uint64_t rng_state = 0x123456789abcdefull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

static double from_bits(uint64_t u) { double d; memcpy(&d, &u, 8); return d; }

static bool check(double a, double b, long long& n)
{
    volatile long double la = a, lb = b;
    volatile long double s = la + lb;           // x87 fadd, 64-bit significand, round to nearest even
    x87val r = x87_add(a, b);
    ++n;
    long double sv = s;
    bool ok;
    if (sv != sv) ok = r.cls == 4;
    else if (std::isinf(sv)) ok = r.cls == (sv < 0 ? 2 : 3);
    else if (sv == 0) ok = r.cls == 1;
    else {
        unsigned char raw[16]; memcpy(raw, (const void*)&sv, 10);
        uint64_t mant; memcpy(&mant, raw, 8);
        uint16_t se; memcpy(&se, raw + 8, 2);
        const int neg = se >> 15; const int e = (int)(se & 0x7FFF) - 16383;
        ok = r.cls == 0 && r.mant == mant && r.exp == e && r.neg == neg;
        if (!ok) fprintf(stderr, "MISMATCH a=%a b=%a native mant=%016llx exp=%d neg=%d  emu mant=%016llx exp=%d neg=%d cls=%d\n",
                         a, b, (unsigned long long)mant, e, neg, (unsigned long long)r.mant, r.exp, r.neg, r.cls);
    }
    if (!ok) fprintf(stderr, "MISMATCH (special) a=%a b=%a emu cls=%d\n", a, b, r.cls);
    return ok;
}

static bool check_order(double a1, double b1, double a2, double b2)
{
    volatile long double s1 = (long double)a1 + (long double)b1;
    volatile long double s2 = (long double)a2 + (long double)b2;
    const bool want = s1 > s2;
    const bool got = x87_gt(x87_add(a1, b1), x87_add(a2, b2));
    if (want != got) fprintf(stderr, "ORDER MISMATCH %a+%a vs %a+%a want %d got %d\n", a1, b1, a2, b2, want, got);
    return want == got;
}

int main(int argc, char** argv)
{
    const long long N = argc > 1 ? atoll(argv[1]) : 2000000;
    long long n = 0;
    const double specials[] = {0.0, -0.0, 1.0, -1.0, -INFINITY, 5e-324, -5e-324, 2.2250738585072014e-308,
                               1.7976931348623157e308, -1.7976931348623157e308, 0x1p-1022, 0x1.fffffffffffffp-1023,
                               -0.10536051565782628, -2.1053605156578263, -648.1053605156578};
    for (double a : specials) for (double b : specials) if (!check(a, b, n)) return 1;
    const int diffs[] = {0, 1, 2, 10, 11, 12, 13, 51, 52, 53, 54, 62, 63, 64, 65, 66, 75, 116, 117, 118, 119, 130, 300};
    for (long long i = 0; i < N; ++i) {
        // a: random sign/exponent/mantissa; b: exponent at a chosen distance below/above
        const uint64_t ma = rnd() & 0xFFFFFFFFFFFFFull, mb = rnd() & 0xFFFFFFFFFFFFFull;
        int ea = 1 + (int)(rnd() % 2045);
        const int d = diffs[rnd() % (sizeof diffs / sizeof diffs[0])];
        int eb = (rnd() & 1) ? ea - d : ea + d;
        if (eb < 0) eb = 0; if (eb > 2046) eb = 2046;
        uint64_t xa = ma, xb = mb;
        switch (rnd() % 6) {          // make low/high mantissa patterns that provoke ties and carries
        case 0: xa = 0; break;
        case 1: xb = 0xFFFFFFFFFFFFFull; break;
        case 2: xa &= ~0x7FFull; xb &= ~0x3FFull; break;
        case 3: xb = (xb & ~0xFFFull) | 0x400ull; break;
        default: break;
        }
        const double a = from_bits(((rnd() & 1) << 63) | ((uint64_t)ea << 52) | xa);
        const double b = from_bits(((rnd() & 1) << 63) | ((uint64_t)eb << 52) | xb);
        if (!check(a, b, n)) return 1;
        if ((i & 7) == 0) {
            // the solve's regime: large negative log-likelihoods plus a small negative prior
            const double ll = -std::ldexp((double)(rnd() >> 11), -20 - (int)(rnd() % 30));
            const double pr = -0.10536051565782628 - 2.0 * (double)((rnd() % 19) * (rnd() % 19));
            const double ll2 = std::nextafter(ll, (rnd() & 1) ? 0.0 : -INFINITY);
            if (!check(pr, ll, n) || !check_order(pr, ll, pr, ll2) || !check_order(pr, ll2, pr - 2.0, ll)) return 1;
        }
    }
    printf("OK %lld\n", n);
    return 0;
}
