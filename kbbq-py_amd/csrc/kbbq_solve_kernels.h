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
#include "lgam_core.h"

#ifndef ST_MEANQ
#define ST_MEANQ 4   // status word: a read group whose meanq the all-device solve cannot decide (see k3_levels_ab)
#endif

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
    // all-device form (aux == NULL): the gammaln terms come from csrc/lgam_core.h over `logtab` (the host libm's own
    // constants, 263 doubles on the device), meanq from the marginals and perr[q] = q_to_p(q)
    const double* logtab;
    double perr[KSOLVE_NQ];
    int* meanq_out;              // optional [R]
    unsigned long long* status;
};

struct K3GammalnParams { const double* x; long long n; const double* logtab; double* out; };

// gammaln of integer-valued arguments >= 1 through csrc/lgam_core.h (self-check and tests of the all-device solve)
__global__ __launch_bounds__(256) void k3_gammaln(K3GammalnParams p)
{
    __shared__ double T[LGAM_LOGTAB_DOUBLES];
    for (int i = threadIdx.x; i < LGAM_LOGTAB_DOUBLES; i += blockDim.x) T[i] = p.logtab[i];
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (long long)gridDim.x * blockDim.x)
        p.out[i] = lgam_count(p.x[i], T);
}

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

// ---- wave-cooperative cell solve: lane = candidate quality -------------------------------------
// All 64 lanes pass the SAME cell.  Lane q' < 43 forms the posterior of candidate q' exactly as
// solve_cell does (same doubles, same order, same 80-bit add); the argmax is a butterfly over an
// order-preserving key of the rounded 80-bit value, ties to the lower index (np.argmax: first
// maximum), NaN neither greater nor smaller than anything (it can only arise at candidate 0, and
// then wins every tie as the sequential scan does).  `sc` = the 129 model constants in LDS.
__device__ __forceinline__ int k3_lane() { return (int)(threadIdx.x & 63u); }

__device__ __forceinline__ int solve_cell_wave(const double* sc, int prior_q, long long errs, long long total, double comb)
{
    const long long kk = errs + 1, nn = total + 2;
    if (nn < 0 || kk < 0 || kk > nn) return 0;                       // uniform: outside the support argmax is 0
    const int lane = k3_lane();
    const int cand = lane < KSOLVE_NQ ? lane : KSOLVE_NQ - 1;         // spare lanes repeat the last candidate
    const double k = (double)kk;
    const double nk = (double)nn - k;
    int diff = cand - prior_q; if (diff < 0) diff = -diff;
    const double pr = sc[diff < KSOLVE_NQ ? diff : KSOLVE_NQ - 1];
    const double t1 = k * sc[KSOLVE_NQ + cand];
    const double t2 = nk * sc[2 * KSOLVE_NQ + cand];
    const double ll = (comb + t1) + t2;
    const x87val v = x87_add(pr, ll);
    // order-preserving key: (class rank, exponent, significand), mirrored for negative values
    const int rank = v.cls == 2 ? 0 : v.cls == 3 ? 4 : v.cls == 1 ? 2 : (v.neg ? 1 : 3);
    int hi = rank << 20;
    unsigned long long lo = 0ull;
    if (rank == 3) { hi += v.exp + 40000; lo = v.mant; }
    if (rank == 1) { hi += 40000 - v.exp; lo = ~v.mant; }
    int nan = v.cls == 4 ? 1 : 0;
    int idx = lane < KSOLVE_NQ ? lane : 64 + lane;                    // a repeat never beats the original
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int ohi = __shfl_xor(hi, off), onan = __shfl_xor(nan, off), oidx = __shfl_xor(idx, off);
        const unsigned long long olo = (unsigned long long)__shfl_xor((long long)lo, off);
        const bool comparable = !(nan | onan);
        const bool other_gt = comparable && (ohi != hi ? ohi > hi : olo > lo);
        const bool mine_gt = comparable && (ohi != hi ? hi > ohi : lo > olo);
        if (other_gt || (!mine_gt && oidx < idx)) { hi = ohi; lo = olo; nan = onan; idx = oidx; }
    }
    return idx;
}

__device__ __forceinline__ long long wave_sum_ll(long long v)
{
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) v += __shfl_xor(v, off);
    return v;
}

// one 1024-thread workgroup per read group: marginals of the cycle tables by row (coalesced, wave
// sums), the read-group cell, then the 43 reported-quality cells, three per wave
__global__ __launch_bounds__(1024) void k3_levels_ab(K3FusedParams p)
{
    __shared__ long long qe[KSOLVE_NQ], qt[KSOLVE_NQ];
    __shared__ double sc[3 * KSOLVE_NQ];
    __shared__ double T[LGAM_LOGTAB_DOUBLES];
    __shared__ double perr[KSOLVE_NQ];
    __shared__ int post_rg;
    const int r = blockIdx.x, lane = k3_lane(), wave = (int)(threadIdx.x >> 6), nwaves = (int)(blockDim.x >> 6);
    for (int i = threadIdx.x; i < 3 * KSOLVE_NQ; i += blockDim.x)
        sc[i] = i < KSOLVE_NQ ? p.c.prior[i] : i < 2 * KSOLVE_NQ ? p.c.logp[i - KSOLVE_NQ] : p.c.log1mp[i - 2 * KSOLVE_NQ];
    if (!p.aux) {
        for (int i = threadIdx.x; i < LGAM_LOGTAB_DOUBLES; i += blockDim.x) T[i] = p.logtab[i];
        for (int i = threadIdx.x; i < KSOLVE_NQ; i += blockDim.x) perr[i] = p.perr[i];
    }
    const size_t npos = (size_t)p.R * KSOLVE_NQ * p.S2;
    for (int q = wave; q < KSOLVE_NQ; q += nwaves) {
        const long long* pe = p.tables + ((size_t)r * KSOLVE_NQ + q) * p.S2;
        const long long* pt = pe + npos;
        long long e = 0, t = 0;
        for (int c = lane; c < p.S2; c += 64) { e += pe[c]; t += pt[c]; }
        e = wave_sum_ll(e); t = wave_sum_ll(t);
        if (lane == 0) { qe[q] = e; qt[q] = t; }
    }
    __syncthreads();
    if (wave == 0) {
        const long long e = wave_sum_ll(lane < KSOLVE_NQ ? qe[lane] : 0ll);
        const long long t = wave_sum_ll(lane < KSOLVE_NQ ? qt[lane] : 0ll);
        int prior;
        if (p.aux) prior = p.meanq[r];
        else {
            // meanq = p_to_q(expected_errs / rg_total) (recalibrate.py:111,120; compare_reads.py:262-267): the reference
            // TRUNCATES -10 log10 of a longdouble quotient.  Formed here in double precision; whenever the value lies
            // within 1e-7 of an integer -- far wider than the double arithmetic's and the reference's own accumulated
            // rounding (its sum takes one 80-bit add per base: <= 2e-9 in this quantity at 10^10 bases) -- the read group
            // is reported and the caller solves with the host's longdouble meanq instead.  One quality value only puts
            // the quotient exactly ON such a boundary (the reference's "float badness", tests/test_recalibrate.py:63).
            const long long mine = lane < KSOLVE_NQ ? qt[lane] : 0ll;
            double ex = lane < KSOLVE_NQ ? (double)mine * perr[lane] : 0.0;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) ex += __shfl_xor(ex, off);
            const int kinds = __popcll(__ballot(mine != 0ll));
            const double v = -10.0 * log10(ex / (double)t);
            const bool decided = t > 0 && kinds > 1 && v == v && fabs(v) < 1e6 && fabs(v - rint(v)) >= 1e-7;
            int mq = decided ? (int)trunc(v) : 0;
            mq = mq < 0 ? 0 : mq > KSOLVE_NQ - 1 ? KSOLVE_NQ - 1 : mq;
            if (!decided && lane == 0) atomicMin(&p.status[ST_MEANQ], (unsigned long long)r);
            if (lane == 0 && p.meanq_out) p.meanq_out[r] = mq;
            prior = mq;
        }
        const int post = solve_cell_wave(sc, prior, e, t, p.aux ? p.aux[r] : lgam_combiln(e, t, T));
        if (lane == 0) { post_rg = post; if (p.dq) p.dq[r] = post - prior; }
    }
    __syncthreads();
    const int prior = post_rg;
    for (int q = wave; q < KSOLVE_NQ; q += nwaves) {
        const int post = solve_cell_wave(sc, prior, qe[q], qt[q],
                                         p.aux ? p.aux[p.R + (size_t)r * KSOLVE_NQ + q] : lgam_combiln(qe[q], qt[q], T));
        if (lane == 0) {
            p.post_q[r * KSOLVE_NQ + q] = post;
            if (p.dq) p.dq[p.R + r * KSOLVE_NQ + q] = post - prior;
        }
    }
}

// one WAVE per (read group, quality, cycle | context) cell
__global__ __launch_bounds__(256) void k3_level_c(K3FusedParams p)
{
    __shared__ double sc[3 * KSOLVE_NQ];
    __shared__ double T[LGAM_LOGTAB_DOUBLES];
    for (int i = threadIdx.x; i < 3 * KSOLVE_NQ; i += blockDim.x)
        sc[i] = i < KSOLVE_NQ ? p.c.prior[i] : i < 2 * KSOLVE_NQ ? p.c.logp[i - KSOLVE_NQ] : p.c.log1mp[i - 2 * KSOLVE_NQ];
    if (!p.aux) for (int i = threadIdx.x; i < LGAM_LOGTAB_DOUBLES; i += blockDim.x) T[i] = p.logtab[i];
    __syncthreads();
    const int lane = k3_lane(), wave = (int)(threadIdx.x >> 6), nwaves = (int)(blockDim.x >> 6);
    const long long npos = (long long)p.R * KSOLVE_NQ * p.S2;
    const long long ndn = (long long)p.R * KSOLVE_NQ * 16;
    const long long* pos_errs = p.tables;
    const long long* pos_total = p.tables + npos;
    const long long* dn_errs = p.tables + 2 * npos;
    const long long* dn_total = dn_errs + ndn;
    const double* comb_pos = p.aux ? p.aux + p.R + (size_t)p.R * KSOLVE_NQ : nullptr;
    const double* comb_dn = p.aux ? comb_pos + npos : nullptr;
    int* dq_pos = p.dq ? p.dq + p.R + p.R * KSOLVE_NQ : nullptr;
    int* dq_dn = p.dq ? dq_pos + npos : nullptr;
    for (long long i = (long long)blockIdx.x * nwaves + wave; i < npos + ndn; i += (long long)gridDim.x * nwaves) {
        if (i < npos) {
            const long long cell = i / p.S2;               // r * 43 + q
            const int col = (int)(i - cell * p.S2);
            const int prior = p.post_q[cell];
            const long long ce = pos_errs[i], ct = pos_total[i];
            const int post = solve_cell_wave(sc, prior, ce, ct, comb_pos ? comb_pos[i] : lgam_combiln(ce, ct, T));
            if (lane == 0) {
                p.lut[cell * p.rs + col] = (int16_t)post;  // meanq + rgdq + qdq + posdq
                if (dq_pos) dq_pos[i] = post - prior;
            }
        } else {
            const long long j = i - npos;
            const long long cell = j >> 4;
            const int d = (int)(j & 15);
            const int prior = p.post_q[cell];
            const long long ce = dn_errs[j], ct = dn_total[j];
            const int v = solve_cell_wave(sc, prior, ce, ct, comb_dn ? comb_dn[j] : lgam_combiln(ce, ct, T)) - prior;
            int16_t* row = p.lut + cell * p.rs + p.S2;
            if (lane == 0) {
                row[5 * (d >> 2) + (d & 3)] = (int16_t)v;
                if (dq_dn) dq_dn[cell * 17 + d] = v;
            }
            if (d == 0) {
                // contexts with an N / without a previous base: the zero pad column
                // (applybqsr.py:98-101); then the row padding
                if (lane < 5) { row[5 * lane + 4] = 0; row[20 + lane] = 0; }
                for (int s2 = p.S2 + 25 + lane; s2 < p.rs; s2 += 64) p.lut[cell * p.rs + s2] = 0;
                if (lane == 0 && dq_dn) dq_dn[cell * 17 + 16] = 0;
            }
        }
    }
}
