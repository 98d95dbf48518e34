// Host harness for kbbq-py_amd/csrc/solve_core.h: reads cells from stdin (binary), writes
// the argmax per cell.  Driven by tests/test_solve_core_host.py.
//   header: int64 n; then SolveConsts (3*43 doubles); then n records {int64 prior_q, errs, total; double comb}
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../../kbbq-py_amd/csrc/solve_core.h"

int main()
{
    int64_t n;
    if (fread(&n, 8, 1, stdin) != 1) return 2;
    SolveConsts c;
    if (fread(&c, sizeof c, 1, stdin) != 1) return 2;
    struct Rec { int64_t prior_q, errs, total; double comb; };
    std::vector<Rec> r((size_t)n);
    if (n && fread(r.data(), sizeof(Rec), (size_t)n, stdin) != (size_t)n) return 2;
    std::vector<int64_t> out((size_t)n);
    for (int64_t i = 0; i < n; ++i) out[(size_t)i] = solve_cell(c, (int)r[(size_t)i].prior_q, r[(size_t)i].errs, r[(size_t)i].total, r[(size_t)i].comb);
    fwrite(out.data(), 8, (size_t)n, stdout);
    return 0;
}
