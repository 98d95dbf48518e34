/*
 * kbbq_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded restatement of the reference's recalibrate hot
 * path (adamjorr/kbbq-py @ v1).  It is the CHECKER for the HIP path: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * Nothing under kbbq-py_amd/ imports, links or calls anything in oracle/.
 *
 * Parity pinning: see oracle/oracle.py header and tests/test_oracle_*.py --
 * this file is checked against (1) the known answers of the reference's own
 * tests and (2) golden vectors produced by running the unmodified reference
 * in the build container (oracle/gen_golden.py -> tests/golden/).
 *
 * Every function cites the reference lines (paths relative to
 * /root/reference/) it follows.  The loops deliberately walk read by read and
 * base by base in the reference's order instead of being clever: this is the
 * referee, not the product.
 *
 * Input layout (shared with the product, see include/kbbq_hip.h): three byte
 * planes seq / cseq / qual with one row of `pitch` bytes per read (ASCII, qual
 * is phred+33 as in the FASTQ text) and a uint32 sidecar per read:
 *     bits  0..15  read length
 *     bits 16..30  read-group id (first-appearance order)
 *     bit  31      second-in-pair flag
 */
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#define KO_OK            0
#define KO_INDEX_ERROR  -2   /* reference raises IndexError  */
#define KO_TYPE_ERROR   -3   /* reference raises TypeError   */

static inline int meta_len(uint32_t m)    { return (int)(m & 0xFFFFu); }
static inline int meta_rg(uint32_t m)     { return (int)((m >> 16) & 0x7FFFu); }
static inline int meta_second(uint32_t m) { return (int)(m >> 31); }

/* compare_reads.py:199 -- nucleotides = ['A','T','G','C']; dinuc index is
 * 4*first + second (compare_reads.py:213-219).  -1: not a nucleotide.      */
static inline int nuc_code(uint8_t c)
{
    switch (c) {
    case 'A': return 0;
    case 'T': return 1;
    case 'G': return 2;
    case 'C': return 3;
    default:  return -1;
    }
}

/* compare_reads.py:281-293 generic_dinuc_covariate for ONE read.
 * dinuc[0] = -1; for i >= 1: -1 when q[i] < minscore or either base is 'N',
 * else the dictionary lookup.  A character outside ACGTN at a position that
 * is actually looked up makes dict.get return None -> TypeError
 * (compare_reads.py:224,292).                                              */
static int dinuc_row(const uint8_t* s, const int* q, int len, int minscore, int* out)
{
    if (len > 0) out[0] = -1;
    for (int i = 1; i < len; ++i) {
        int invalid = (q[i] < minscore) || (s[i] == 'N') || (s[i - 1] == 'N');
        if (invalid) { out[i] = -1; continue; }
        int a = nuc_code(s[i - 1]), b = nuc_code(s[i]);
        if (a < 0 || b < 0) return KO_TYPE_ERROR;
        out[i] = 4 * a + b;
    }
    return KO_OK;
}

/*
 * Pass 1: recalibrate.py:22-121 fastq_to_covariate_arrays, loop body :57-119.
 *
 * Tables are caller-allocated, zero-initialised, C-contiguous int64 with the
 * FINAL shapes  q_*[R][Q], pos_*[R][Q][S2], dinuc_*[R][Q][16]  (Q = maxscore+1,
 * S2 = 2 * longest read).  The reference grows its arrays as it goes
 * (recalibrate.py:61-87); growth appends zeros at the END of the cycle axis, so
 * a negative (second-in-pair) cycle -(i+1) lands on ABSOLUTE column
 * 2*seqlen_t-(i+1) where seqlen_t is the running maximum read length when the
 * read is tallied (SURVEY hazard H1) -- reproduced here with `seqlen`.
 *
 * A read shorter than the running maximum makes the boolean masks mismatch
 * (recalibrate.py:89-101) -> IndexError (hazard H2).  q > maxscore indexes
 * past the Q axis (recalibrate.py:114-115) -> IndexError (hazard H7).
 *
 * expected_errs accumulates q_to_p(q) in long double, read order, exactly as
 * np.add.at on the longdouble vector does (recalibrate.py:45,111); q2p[] is
 * the float64 table np.power(10.0, -(q/10.0)) supplied by the caller so the
 * very same doubles are summed (compare_reads.py:269-271).
 */
int kbbq_oracle_accumulate(const uint8_t* seq, const uint8_t* cseq, const uint8_t* qual,
                           const uint32_t* meta, int64_t nreads, int64_t pitch,
                           int R, int S2, int minscore, int maxscore,
                           const double* q2p, long double* expected_errs,
                           int64_t* rg_errs, int64_t* rg_total,
                           int64_t* q_errs, int64_t* q_total,
                           int64_t* pos_errs, int64_t* pos_total,
                           int64_t* dinuc_errs, int64_t* dinuc_total,
                           int64_t* bad_read)
{
    const int Q = maxscore + 1;
    int seqlen = 0;                       /* recalibrate.py:34 */
    static __thread int q[65536];
    static __thread int dn[65536];

    for (int64_t r = 0; r < nreads; ++r) {
        const uint8_t* s  = seq  + r * pitch;
        const uint8_t* c  = cseq + r * pitch;
        const uint8_t* ql = qual + r * pitch;
        const int len = meta_len(meta[r]);
        const int rg  = meta_rg(meta[r]);
        const int second = meta_second(meta[r]);
        if (bad_read) *bad_read = r;
        if (rg >= R) return KO_INDEX_ERROR;

        if (len > seqlen) seqlen = len;   /* recalibrate.py:81-87 */
        if (2 * seqlen > S2) return KO_INDEX_ERROR;
        if (len < seqlen) return KO_INDEX_ERROR;   /* hazard H2, recalibrate.py:89-101 */

        for (int i = 0; i < len; ++i) q[i] = (int)ql[i] - 33;   /* recalibrate.py:92 */
        int rc = dinuc_row(s, q, len, minscore, dn);            /* recalibrate.py:94 */
        if (rc != KO_OK) return rc;

        for (int i = 0; i < len; ++i) {
            /* recalibrate.py:93 + compare_reads.py:275-279: cycle i, or -(i+1)
             * used as a negative index into an axis of length 2*seqlen.      */
            const int cyc = second ? (2 * seqlen - (i + 1)) : i;
            const int err   = (s[i] != c[i]);                   /* recalibrate.py:13-20,91 */
            const int valid = !(q[i] < minscore);               /* recalibrate.py:96-98 */
            const int dvalid = (dn[i] != -1) && valid;          /* recalibrate.py:99 */
            if (valid) {
                if (q[i] > maxscore || q[i] < 0) return KO_INDEX_ERROR;
                expected_errs[rg] += (long double)q2p[q[i]];    /* recalibrate.py:111 */
                rg_total[rg] += 1;                              /* :113 */
                q_total[(int64_t)rg * Q + q[i]] += 1;           /* :115 */
                pos_total[((int64_t)rg * Q + q[i]) * S2 + cyc] += 1;   /* :117 */
                if (err) {
                    rg_errs[rg] += 1;                           /* :112 */
                    q_errs[(int64_t)rg * Q + q[i]] += 1;        /* :114 */
                    pos_errs[((int64_t)rg * Q + q[i]) * S2 + cyc] += 1; /* :116 */
                }
            }
            if (dvalid) {
                dinuc_total[((int64_t)rg * Q + q[i]) * 16 + dn[i]] += 1;      /* :119 */
                if (err)
                    dinuc_errs[((int64_t)rg * Q + q[i]) * 16 + dn[i]] += 1;   /* :118 */
            }
        }
    }
    if (bad_read) *bad_read = -1;
    return KO_OK;
}

/*
 * Pass 2: compare_reads.py:320-328 recalibrate_fastq (per read) driven by
 * recalibrate.py:141-156.  For q >= minscore:
 *   new = meanq[rg] + rgdq[rg] + qdq[rg,q] + dinucdq[rg,q,dinuc] + posdq[rg,q,cycle]
 * with cycle = i or -(i+1) and dinuc = -1 for "no context", both resolved by
 * Python negative indexing on the FINAL table sizes (S2 columns / 17 columns,
 * applybqsr.py:98-101).  q < minscore passes through.  No clipping (hazard H3):
 * the raw integer is returned in out_q; turning it into a character is the
 * caller's business (recalibrate.py:152).
 *
 * Index errors mirrored: rg beyond the tables; q beyond the Q axis of the
 * tables actually passed (Qt may be < 43, see tests/test_compare_reads.py:219-233
 * of the reference where Qt = 8); cycle beyond S2.
 */
int kbbq_oracle_apply(const uint8_t* seq, const uint8_t* qual, const uint32_t* meta,
                      int64_t nreads, int64_t pitch, int R, int Qt, int S2, int D,
                      int minscore,
                      const int64_t* meanq, const int64_t* rgdq, const int64_t* qdq,
                      const int64_t* posdq, const int64_t* dinucdq,
                      int32_t* out_q, int64_t* bad_read)
{
    static __thread int q[65536];
    static __thread int dn[65536];
    for (int64_t r = 0; r < nreads; ++r) {
        const uint8_t* s  = seq  + r * pitch;
        const uint8_t* ql = qual + r * pitch;
        int32_t* o = out_q + r * pitch;
        const int len = meta_len(meta[r]);
        const int rg  = meta_rg(meta[r]);
        const int second = meta_second(meta[r]);
        if (bad_read) *bad_read = r;
        for (int i = 0; i < len; ++i) q[i] = (int)ql[i] - 33;       /* compare_reads.py:321 */
        int rc = dinuc_row(s, q, len, minscore, dn);                /* :326 */
        if (rc != KO_OK) return rc;
        for (int i = 0; i < len; ++i) {
            if (q[i] < minscore) { o[i] = q[i]; continue; }         /* :322-323 */
            if (rg >= R || q[i] >= Qt) return KO_INDEX_ERROR;
            int cyc = second ? -(i + 1) : i;                        /* :325 */
            if (cyc < 0) cyc += S2;
            if (cyc < 0 || cyc >= S2) return KO_INDEX_ERROR;
            int d = dn[i];
            if (d < 0) d += D;                                      /* index -1 -> last column */
            if (d < 0 || d >= D) return KO_INDEX_ERROR;
            const int64_t cell = (int64_t)rg * Qt + q[i];
            o[i] = (int32_t)(meanq[rg] + rgdq[rg] + qdq[cell]       /* :327 */
                             + dinucdq[cell * D + d] + posdq[cell * S2 + cyc]);
        }
        for (int64_t i = len; i < pitch; ++i) o[i] = 0;
    }
    if (bad_read) *bad_read = -1;
    return KO_OK;
}

/* ------------------------------------------------------------------------
 * Synthetic reads: a counter-based generator that
 * the HIP library re-implements independently on the device
 * (kbbq-py_amd/csrc/synth.hip); tests check the two produce identical bytes.
 *
 *   mix(x)   = splitmix64 finaliser of x + 0x9E3779B97F4A7C15
 *   r        = mix( mix(seed + read) ^ pos )          64 random bits per base
 *   base     = "ACGT"[r & 3];  'N' when ((r >> 2) & 1023) == 0   (p = 1/1024)
 *   q        = (((r >> 12) & 0xFFFF) * nq) >> 16  + qlo          (uniform)
 *   error    = (uint32)(r >> 32) < thr[q], thr[q] = floor(2^32 * 10^(-q/10))
 *              (thr[0] = 2^32 - 1)
 *   cseq     = error ? "ACGT"[(b + 1 + ((r >> 28) & 15) % 3) & 3] : base
 *              ('N' with error -> "ACGT"[b])
 *   length   = len_lo + (read * (len_hi - len_lo + 1)) / total_reads
 *              (non-decreasing ramp: the only order the reference survives, H2)
 *   rg       = (read >> 1) % nrg  ; second = read & 1
 * Row bytes at and beyond the read length: 'N' in seq / cseq, 0 in qual.
 * ------------------------------------------------------------------------ */
static inline uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

void kbbq_oracle_synth(uint8_t* seq, uint8_t* cseq, uint8_t* qual, uint32_t* meta,
                       int64_t first_read, int64_t nreads, int64_t total_reads,
                       int64_t pitch, uint64_t seed, int len_lo, int len_hi,
                       int nrg, int qlo, int qhi, const uint32_t* thr)
{
    static const char ACGT[4] = { 'A', 'C', 'G', 'T' };
    const uint64_t nq = (uint64_t)(qhi - qlo + 1);
    for (int64_t k = 0; k < nreads; ++k) {
        const uint64_t read = (uint64_t)(first_read + k);
        const int len = len_lo + (int)((read * (uint64_t)(len_hi - len_lo + 1)) / (uint64_t)total_reads);
        const uint32_t rg = (uint32_t)((read >> 1) % (uint64_t)nrg);
        meta[k] = (uint32_t)len | (rg << 16) | ((uint32_t)(read & 1u) << 31);
        uint8_t* s = seq + k * pitch; uint8_t* c = cseq + k * pitch; uint8_t* ql = qual + k * pitch;
        const uint64_t hr = mix64(seed + read);
        for (int i = 0; i < len; ++i) {
            const uint64_t r = mix64(hr ^ (uint64_t)i);
            const int b = (int)(r & 3u);
            const int isn = (((r >> 2) & 1023u) == 0);
            const int q = (int)((((r >> 12) & 0xFFFFu) * nq) >> 16) + qlo;
            const int err = ((uint32_t)(r >> 32) < thr[q]);
            const int sub = (b + 1 + (int)(((r >> 28) & 15u) % 3u)) & 3;
            s[i]  = isn ? 'N' : ACGT[b];
            c[i]  = err ? (isn ? ACGT[b] : ACGT[sub]) : s[i];
            ql[i] = (uint8_t)(q + 33);
        }
        for (int64_t i = len; i < pitch; ++i) { s[i] = 'N'; c[i] = 'N'; ql[i] = 0; }
    }
}
