// kbbq_solve_kernels.h -- K3: the delta-Q model solve on the device.
//
// k3_delta_q     generic: one thread per cell, mirrors compare_reads.gatk_delta_q (:235-260)
// k3_levels_ab   fused pipeline, levels "read group" and "reported quality" of
//                applybqsr.get_delta_qs (applybqsr.py:80-103): marginals of the pos tables,
//                then the two small solves; one 64-thread block per read group
// k3_level_c     fused pipeline, levels "cycle" and "dinucleotide": one thread per cell,
//                writes the K2 apply LUT directly (and the delta tables when asked)
// The arithmetic of a cell is csrc/solve_core.h (shared with the host test harness).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "solve_core.h"

struct K3CellParams {
    const long long* prior_q; const long long* errs; const long long* total; const double* comb;
    long long n; long long* dq; SolveConsts c;
};

struct K3FusedParams {
    const long long* tables;     // [pos_errs | pos_total | dinuc_errs | dinuc_total]
    int R; int S2; int rs;
    const int* meanq;            // [R]
    const double* aux;           // [comb_rg R | comb_q R*43 | comb_pos R*43*S2 | comb_dn R*43*16]
    int* post_q;                 // [R*43] scratch: posterior quality after the first two levels
    int16_t* lut;                // K2 layout, rows of rs
    int* dq;                     // optional [rgdq R | qdq R*43 | posdq R*43*S2 | dinucdq R*43*17]
    SolveConsts c;
};

__global__ __launch_bounds__(256) void k3_delta_q(K3CellParams p)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n;
         i += (long long)gridDim.x * blockDim.x) {
        const long long pq = p.prior_q[i];
        // the reference indexes prior_dist[|q' - prior_q|]: a prior outside 0..42 is an IndexError
        // there; the host wrapper rejects it before the launch
        p.dq[i] = (long long)solve_cell(p.c, (int)pq, p.errs[i], p.total[i], p.comb[i]) - pq;
    }
}

struct K3PostParams {
    const double* prior_q; const long long* errs; const long long* total; const double* comb;
    long long n; long long* post; SolveConsts c;
};

// the same cell solve for a float64 prior; returns the posterior quality itself (the caller
// subtracts the prior in float64 as the reference's `posterior_q - prior_q` does)
__global__ __launch_bounds__(256) void k3_posterior_q(K3PostParams p)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n;
         i += (long long)gridDim.x * blockDim.x)
        p.post[i] = (long long)solve_cell<double>(p.c, p.prior_q[i], p.errs[i], p.total[i], p.comb[i]);
}

__global__ __launch_bounds__(64) void k3_levels_ab(K3FusedParams p)
{
    __shared__ long long qe[KSOLVE_NQ], qt[KSOLVE_NQ];
    __shared__ int post_rg;
    const int r = blockIdx.x, t = threadIdx.x;
    const size_t npos = (size_t)p.R * KSOLVE_NQ * p.S2;
    if (t < KSOLVE_NQ) {
        const long long* pe = p.tables + ((size_t)r * KSOLVE_NQ + t) * p.S2;
        const long long* pt = pe + npos;
        long long e = 0, tt = 0;
        for (int c = 0; c < p.S2; ++c) { e += pe[c]; tt += pt[c]; }
        qe[t] = e; qt[t] = tt;
    }
    __syncthreads();
    if (t == 0) {
        long long e = 0, tt = 0;
        for (int q = 0; q < KSOLVE_NQ; ++q) { e += qe[q]; tt += qt[q]; }
        const int prior = p.meanq[r];
        const int post = solve_cell(p.c, prior, e, tt, p.aux[r]);
        post_rg = post;
        if (p.dq) p.dq[r] = post - prior;
    }
    __syncthreads();
    if (t < KSOLVE_NQ) {
        const int prior = post_rg;
        const int post = solve_cell(p.c, prior, qe[t], qt[t], p.aux[p.R + (size_t)r * KSOLVE_NQ + t]);
        p.post_q[r * KSOLVE_NQ + t] = post;
        if (p.dq) p.dq[p.R + r * KSOLVE_NQ + t] = post - prior;
    }
}

__global__ __launch_bounds__(256) void k3_level_c(K3FusedParams p)
{
    const long long npos = (long long)p.R * KSOLVE_NQ * p.S2;
    const long long ndn = (long long)p.R * KSOLVE_NQ * 16;
    const long long* pos_errs = p.tables;
    const long long* pos_total = p.tables + npos;
    const long long* dn_errs = p.tables + 2 * npos;
    const long long* dn_total = dn_errs + ndn;
    const double* comb_pos = p.aux + p.R + (size_t)p.R * KSOLVE_NQ;
    const double* comb_dn = comb_pos + npos;
    int* dq_pos = p.dq ? p.dq + p.R + p.R * KSOLVE_NQ : nullptr;
    int* dq_dn = p.dq ? dq_pos + npos : nullptr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < npos + ndn;
         i += (long long)gridDim.x * blockDim.x) {
        if (i < npos) {
            const long long cell = i / p.S2;               // r * 43 + q
            const int col = (int)(i - cell * p.S2);
            const int prior = p.post_q[cell];
            const int post = solve_cell(p.c, prior, pos_errs[i], pos_total[i], comb_pos[i]);
            p.lut[cell * p.rs + col] = (int16_t)post;      // meanq + rgdq + qdq + posdq
            if (dq_pos) dq_pos[i] = post - prior;
        } else {
            const long long j = i - npos;
            const long long cell = j >> 4;
            const int d = (int)(j & 15);
            const int prior = p.post_q[cell];
            const int v = solve_cell(p.c, prior, dn_errs[j], dn_total[j], comb_dn[j]) - prior;
            int16_t* row = p.lut + cell * p.rs + p.S2;
            row[5 * (d >> 2) + (d & 3)] = (int16_t)v;
            if (dq_dn) dq_dn[cell * 17 + d] = v;
            if (d == 0) {
                // contexts with an N / without a previous base: the zero pad column
                // (applybqsr.py:98-101); then the row padding
                for (int a = 0; a < 5; ++a) { row[5 * a + 4] = 0; row[20 + a] = 0; }
                for (int s = p.S2 + 25; s < p.rs; ++s) p.lut[cell * p.rs + s] = 0;
                if (dq_dn) dq_dn[cell * 17 + 16] = 0;
            }
        }
    }
}
