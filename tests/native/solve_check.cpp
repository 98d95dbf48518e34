// Host harness for kbbq-py_amd/csrc/solve_core.h: reads cells from stdin (binary), writes
// the argmax per cell.  Driven by tests/test_solve_core_host.py.
//   header: int64 n; then SolveConsts (3*43 doubles); then n records {int64 prior_q, errs, total; double comb}
//   with argument "f" the first field of a record is a float64 prior (solve_cell<double>)
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#include "../../kbbq-py_amd/csrc/solve_core.h"

int main(int argc, char** argv)
{
    const bool float_prior = argc > 1 && argv[1][0] == 'f';
    int64_t n;
    if (fread(&n, 8, 1, stdin) != 1) return 2;
    SolveConsts c;
    if (fread(&c, sizeof c, 1, stdin) != 1) return 2;
    struct Rec { int64_t prior_q, errs, total; double comb; };
    std::vector<Rec> r((size_t)n);
    if (n && fread(r.data(), sizeof(Rec), (size_t)n, stdin) != (size_t)n) return 2;
    std::vector<int64_t> out((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const Rec& x = r[(size_t)i];
        if (float_prior) {
            double pf; memcpy(&pf, &x.prior_q, 8);
            out[(size_t)i] = solve_cell<double>(c, pf, x.errs, x.total, x.comb);
        } else {
            out[(size_t)i] = solve_cell(c, (int)x.prior_q, x.errs, x.total, x.comb);
        }
    }
    fwrite(out.data(), 8, (size_t)n, stdout);
    return 0;
}
