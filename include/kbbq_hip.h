/*
 * kbbq_hip.h -- C ABI of libkbbq_hip.so, the MI355X (gfx950) implementation of
 * kbbq's recalibrate hot path.
 *
 * The reference (adamjorr/kbbq-py @ v1) is pure Python and has no FFI layer;
 * these entry points are what a ctypes binding of its hot-path functions binds
 * (INTEGRATION.md shows the stub).  Each entry point cites the reference code it
 * replaces (paths relative to the reference checkout).
 *
 * Conventions
 *   - plain C, no torch types; every function returns 0 (KBBQ_OK) or a negative
 *     KBBQ_E_* code; kbbq_last_error() returns text for the calling thread.
 *   - "_dev" functions take DEVICE pointers and enqueue on the context's stream
 *     without synchronising; kernels report data errors (the reference's
 *     IndexError / TypeError cases) through the context's status word, read
 *     with kbbq_ctx_status() (which synchronises).
 *   - host-pointer functions (no suffix) stage through device memory owned by
 *     the context and synchronise before returning.
 *   - one context per device; a context is not thread-safe -- with one exception the file path relies on (kbbq/_egress.py, kbbq/_stream.py:
 *     the output pipeline copies slab k off the device while the producer thread launches K2 on slab k + 1): kbbq_dev_alloc / _free,
 *     kbbq_dev_copy_async and kbbq_event_create / _record / _sync touch nothing of the context but its stream and may be called
 *     from a second thread while the first one launches kernels and reads the status.
 *
 * Read layout ("padded SoA"): three byte planes seq / cseq / qual, one row of
 * `pitch` bytes per read, pitch a multiple of 16, rows 16-byte aligned.  Bytes
 * are the FASTQ characters (qual is phred+33).  Bytes at and beyond the read
 * length MUST be zero in the qual plane and SHOULD be 'N' in seq / cseq (any other
 * value is still handled correctly, through a slower exact check of that chunk).
 * One uint32 of metadata per read:
 *     bits  0..15  length          (<= pitch)
 *     bits 16..30  read-group id   (first-appearance order, recalibrate.py:59-64)
 *     bit  31      second in pair  (compare_reads.py:304-306)
 *
 * Count tables: ONE int64 buffer of kbbq_tables_count(R, S2) elements,
 *     [ pos_errs[R][43][S2] | pos_total[R][43][S2] | dinuc_errs[R][43][16] | dinuc_total[R][43][16] ]
 * i.e. the four 3-D arrays of recalibrate.py:51-54 at their final size
 * (S2 = 2 * longest read).  q_* and rg_* are marginals of pos_* (SURVEY 7).
 * kbbq_accumulate* ADD into the buffer, so batches, shards and ranks compose.
 */
#ifndef KBBQ_HIP_H
#define KBBQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KBBQ_OK          0
#define KBBQ_E_HIP      -1   /* HIP runtime error (no device, out of memory, ...)        */
#define KBBQ_E_INDEX    -2   /* the reference raises IndexError on this input            */
#define KBBQ_E_TYPE     -3   /* the reference raises TypeError (base outside ACGTN)      */
#define KBBQ_E_ARG      -4   /* bad argument (alignment, sizes, read too long for LDS)   */
#define KBBQ_E_RANGE    -5   /* recalibrated quality + 33 outside 0..255     */
#define KBBQ_E_NAME     -6   /* corrected read name does not start with the read name    */
#define KBBQ_E_LUT      -7   /* a device-built LUT needs kbbq_apply_dev(KBBQ_APPLY_CHECKED)  */
#define KBBQ_E_MEANQ    -8   /* kbbq_solve_device_dev: meanq sits on a truncation boundary, solve with the host's longdouble meanq */

#define KBBQ_APPLY_CHECKED 0 /* per-base range test, int16 LUT                            */
#define KBBQ_APPLY_FAST    1 /* table-driven int8 LUT, no per-base tests (LUT flags == 0) */

#define KBBQ_NQ         43   /* maxscore + 1, recalibrate.py:36                          */
#define KBBQ_NDINUC     16
#define KBBQ_ABI_VERSION 1

typedef struct kbbq_ctx kbbq_ctx;

/* ---- library / context ------------------------------------------------ */
int         kbbq_abi_version(void);
const char* kbbq_last_error(void);
int         kbbq_device_count(int* count);
/* One process per GPU (torch.distributed.run): the host threads a rank starts, and where they run.
 * kbbq_host_threads: threads the library's host stages (scan, fill, format) start for `work_bytes` of work -- at most this
 * process's SHARE of the CPUs it may use: usable CPUs / LOCAL_WORLD_SIZE (the launcher's variable; KBBQ_LOCAL_RANKS for
 * other launchers; KBBQ_HOST_THREADS overrides the ceiling), so eight ranks of one node do not start eight times the
 * host's threads.  kbbq_bind_host_to_device / _to_pci: bind the calling thread, and every thread started from it
 * afterwards, to the CPUs of the NUMA node the GPU (PCI address as hipDeviceGetPCIBusId prints it) hangs on; *numa_node
 * = -1 and nothing changed when the system names none.  The reference has no counterpart: it is single-threaded
 * (recalibrate.py:56-57,141-156 walk the reads in one Python loop).                                                    */
int         kbbq_host_threads(size_t work_bytes);
/* Ask for huge pages behind a large host buffer the caller is about to fill for the first time (madvise; a no-op where
 * transparent huge pages are off or KBBQ_HUGE_PAGES=0): the egress pipeline's render buffers.  Always KBBQ_OK. */
int         kbbq_host_advise_huge(void* p, size_t bytes);
int         kbbq_bind_host_to_pci(const char* pci_bus_id, int* numa_node, int* ncpus);
int         kbbq_bind_host_to_device(int device, int* numa_node, int* ncpus);
int         kbbq_ctx_create(int device, kbbq_ctx** out);
int         kbbq_ctx_destroy(kbbq_ctx* ctx);
/* Run on a caller-owned hipStream_t (e.g. torch's current stream).  NULL selects the
 * device's default (null) stream -- which IS torch's current stream unless the caller
 * changed it.  A fresh context runs on a private non-blocking stream until this is called. */
int         kbbq_ctx_set_stream(kbbq_ctx* ctx, void* hip_stream);
int         kbbq_ctx_sync(kbbq_ctx* ctx);
/* Synchronise, fetch and clear the kernels' status word.  Returns KBBQ_OK or the
 * KBBQ_E_INDEX / KBBQ_E_TYPE / KBBQ_E_RANGE the reference would have raised
 * first (lowest read index; TypeError wins a tie, recalibrate.py:94 precedes
 * :114).  *read_index receives that read's index within the launch (or -1).  */
int         kbbq_ctx_status(kbbq_ctx* ctx, int64_t* read_index);
/* Device properties the host code sizes launches with.                     */
int         kbbq_ctx_info(kbbq_ctx* ctx, int* compute_units, int* lds_bytes, char* name, int name_len);

/* ---- device memory plumbing (for callers without torch) --------------- */
int kbbq_dev_alloc(kbbq_ctx* ctx, size_t bytes, void** dptr);
int kbbq_dev_free(kbbq_ctx* ctx, void* dptr);
int kbbq_dev_zero(kbbq_ctx* ctx, void* dptr, size_t bytes);                       /* async */
int kbbq_dev_upload(kbbq_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);   /* sync */
int kbbq_dev_download(kbbq_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes); /* sync */
int kbbq_dev_mem_info(kbbq_ctx* ctx, size_t* free_bytes, size_t* total_bytes);          /* hipMemGetInfo: what the file path sizes its device budget with */
/* the rest of what the file path needs to run WITHOUT torch (kbbq/_hipmem.py: the single-GPU command line never imports
 * it): page-locked host buffers for the ingest / egress slabs, copies enqueued on the context's stream (kind 1 host to
 * device, 2 device to host, 3 device to device; no synchronisation), events on that stream (when has a slab's upload
 * left its buffer). */
int kbbq_host_alloc(size_t bytes, void** hptr);
int kbbq_host_free(void* hptr);
int kbbq_dev_copy_async(kbbq_ctx* ctx, void* dst, const void* src, size_t bytes, int kind);
int kbbq_event_create(kbbq_ctx* ctx, void** event);
int kbbq_event_record(kbbq_ctx* ctx, void* event);
int kbbq_event_sync(void* event);
int kbbq_event_destroy(void* event);

/* ---- geometry ---------------------------------------------------------- */
size_t kbbq_tables_count(int R, int S2);          /* int64 elements in a count-table buffer */
size_t kbbq_lut_count(int R, int Qt, int S2);     /* int16 elements in an apply LUT         */

/* ---- K1: error flagging + covariate binning ---------------------------
 * Replaces the loop body of recalibrate.fastq_to_covariate_arrays
 * (recalibrate.py:57-119) with find_corrected_sites (:13-20),
 * fastq_cycle_covariates / fastq_dinuc_covariates (compare_reads.py:275-302).
 * Second-in-pair cycles land on column 2*len-(i+1) (valid inputs
 * have non-decreasing lengths, so the running maximum IS the read's length).
 * q > 42 -> KBBQ_E_INDEX; a base outside ACGTN in a looked-up dinucleotide ->
 * KBBQ_E_TYPE (through kbbq_ctx_status).                                    */
int kbbq_accumulate_dev(kbbq_ctx* ctx, const uint8_t* d_seq, const uint8_t* d_cseq,
                        const uint8_t* d_qual, const uint32_t* d_meta,
                        int64_t nreads, int pitch, int R, int S2, int minscore,
                        int64_t* d_tables);
/* Same, with a separate quality threshold for the dinucleotide context: bases with
 * q >= minscore are counted, a context exists only where q >= dinuc_minscore.  This is
 * the rule of ReadData / CovariateData.consume_read (read.py:336-369, covariate.py:406-425:
 * `skips` decide what is counted -- the caller zeroes the quality byte of skipped bases
 * and passes minscore = 0 -- while minscore only gates the context).               */
int kbbq_accumulate_ex_dev(kbbq_ctx* ctx, const uint8_t* d_seq, const uint8_t* d_cseq,
                           const uint8_t* d_qual, const uint32_t* d_meta,
                           int64_t nreads, int pitch, int R, int S2, int minscore,
                           int dinuc_minscore, int64_t* d_tables);
/* The same for ONE LENGTH BAND of a mixed-length input: S2 sizes the (global) count tables as before, S_band >= the
 * longest read of THIS batch (0 = S2 / 2) sizes the kernel's LDS tables, so that short reads packed at a narrow pitch
 * use the table-driven kernel even when the input's longest read would not fit it.  Counts land in the same cells
 * (a second-in-pair column 2*len - 1 - i does not depend on the table width).  S_min: a promise that no (non-empty)
 * read of the batch is shorter (0 = no promise); long bands (~200-300 bases) need it to fit the LDS -- a cycle row
 * then has 3*S_band - S_min words -- and a read that breaks it is reported as KBBQ_E_INDEX, not counted.        */
int kbbq_accumulate_band_dev(kbbq_ctx* ctx, const uint8_t* d_seq, const uint8_t* d_cseq, const uint8_t* d_qual,
                             const uint32_t* d_meta, int64_t nreads, int pitch, int R, int S2, int S_band, int S_min,
                             int minscore, int dinuc_minscore, int64_t* d_tables);
/* Host-buffer form (what a ctypes binding of the reference calls with NumPy arrays; ADDS into the four count arrays):
 * the rows travel slab by slab through page-locked staging of the context's own -- host threads copy slab k + 1 while the
 * copy engine uploads slab k and the kernel runs on slab k - 1 -- so device memory is two slabs whatever the input's size
 * (KBBQ_STAGE_MB: staging bytes per slab, default 96 MB) and the call runs at the PCIe rate of its 3 planes.  A read the
 * kernel flags is reported with its index in the WHOLE input; nothing is added to the caller's arrays then.            */
int kbbq_accumulate(kbbq_ctx* ctx, const uint8_t* seq, const uint8_t* cseq,
                    const uint8_t* qual, const uint32_t* meta,
                    int64_t nreads, int pitch, int R, int S2, int minscore,
                    int64_t* pos_errs, int64_t* pos_total,
                    int64_t* dinuc_errs, int64_t* dinuc_total);

/* ---- K2: delta-Q table lookup / apply ----------------------------------
 * Replaces compare_reads.recalibrate_fastq (compare_reads.py:320-328) as driven
 * by recalibrate.py:141-152.  The five model arrays are folded into one LUT blob
 * of kbbq_lut_bytes(R, Qt, S2) bytes (kbbq_build_lut on the host, kbbq_solve_dev on
 * the device):
 *   canonical part, int16, one row of kbbq_lut_row_stride(S2) entries per
 *   (read group, quality):
 *     row[0 .. S2-1]    = meanq[rg] + rgdq[rg] + qdq[rg][q] + posdq[rg][q][cycle]
 *     row[S2 .. S2+24]  = dinucdq[rg][q][d] indexed by 5*code(prev)+code(cur), codes
 *                         A0 T1 G2 C3 (compare_reads.py:199) and 4 = N / no previous
 *                         base; entries involving code 4 hold dinucdq[rg][q][-1]
 *   table-driven part, int8, indexed by the RAW quality byte (padding row, identity
 *   rows below minscore, a mirrored copy of the cycle entries for second-in-pair
 *   reads) -- see csrc/kbbq_kernels_v3.h; and an int of flags: bit 0 = a value
 *   does not fit int8, bit 1 = some (cycle, context) pair can leave 0..255.
 * new_q = cycle entry + context entry for q >= minscore, q otherwise; the output byte
 * is new_q + 33 (no clipping, compare_reads.py:327; outside 0..255 -> KBBQ_E_RANGE).
 * Cycle -(i+1) wraps on the final S2 (Python negative index).  q >= Qt, rg >= R or a
 * cycle beyond S2 -> KBBQ_E_INDEX.
 * mode KBBQ_APPLY_FAST may only be used when the blob's flags are 0 (known for a
 * host-built blob; for a device-built one kbbq_ctx_status returns KBBQ_E_LUT after
 * the fact and the caller re-runs with KBBQ_APPLY_CHECKED).                       */
int    kbbq_lut_row_stride(int S2);
size_t kbbq_full_lut_bytes(int R, int Qt, int S2);
size_t kbbq_lut_bytes(int R, int Qt, int S2);
int kbbq_build_lut(int R, int Qt, int S2, int D, int minscore,
                   const int64_t* meanq, const int64_t* rgdq, const int64_t* qdq,
                   const int64_t* posdq, const int64_t* dinucdq, void* lut_blob_out,
                   int* flags_out);
int kbbq_apply_dev(kbbq_ctx* ctx, const uint8_t* d_seq, const uint8_t* d_qual,
                   const uint32_t* d_meta, int64_t nreads, int pitch,
                   int R, int Qt, int S2, int minscore,
                   const void* d_lut_blob, int mode, uint8_t* d_qual_out);
/* Host-buffer form: LUT built on the host (kbbq_build_lut), rows through the same page-locked slabs as kbbq_accumulate --
 * upload of slab k + 1, kernel on slab k and download of slab k - 1 overlap (PCIe is full duplex).  On an error status the
 * read index is that of the WHOLE input and the contents of qual_out are unspecified.                                  */
int kbbq_apply(kbbq_ctx* ctx, const uint8_t* seq, const uint8_t* qual, const uint32_t* meta,
               int64_t nreads, int pitch, int R, int Qt, int S2, int D, int minscore,
               const int64_t* meanq, const int64_t* rgdq, const int64_t* qdq,
               const int64_t* posdq, const int64_t* dinucdq, uint8_t* qual_out);

/* ---- K3: delta-Q model solve ------------------------------------------
 * Replaces compare_reads.gatk_delta_q (compare_reads.py:235-260) and, fused,
 * applybqsr.get_delta_qs (gatk/applybqsr.py:80-103).  Split of labour (DESIGN.md
 * "K3"): the host evaluates everything transcendental with the SciPy calls the
 * reference makes -- per cell comb = gammaln(n+1) - (gammaln(k+1) + gammaln(n-k+1)),
 * k = errs+1, n = total+2, and the 3 x 43 table h_consts129 = [prior_dist |
 * log p | log1p(-p)] -- and the device multiplies/adds those float64 values in
 * SciPy's order, adds the longdouble prior with an exact 80-bit emulation and
 * takes the first maximum over the 43 candidates.  Results are integers
 * identical to the reference's.
 *
 * kbbq_delta_q_dev: dq[i] = argmax_i - prior_q[i] for ncells independent cells
 * (prior_q must lie in 0..42).
 * kbbq_posterior_q_dev: the same cell solve for a float64 prior (the reference calls
 * gatk_delta_q with EstimatedQReported when it builds a report, gatk/bqsr.py:294): the
 * distance |q' - prior| is the float64 difference truncated toward zero, -1 < prior < 43;
 * returns the posterior quality argmax_i itself.
 * kbbq_solve_dev: the whole hierarchy from the device count tables: marginals,
 * read-group and quality levels (d_post_q[R*43] scratch), then cycle and
 * dinucleotide levels, writing the K2 LUT blob (kbbq_lut_bytes bytes) and, when
 * d_dq != NULL, int32 [rgdq R | qdq R*43 | posdq R*43*S2 | dinucdq R*43*17].
 * d_aux (kbbq_solve_aux_count doubles) = [comb_rg | comb_q | comb_pos | comb_dn].  */
int    kbbq_delta_q_dev(kbbq_ctx* ctx, const int64_t* d_prior_q, const int64_t* d_errs,
                        const int64_t* d_total, const double* d_comb, int64_t ncells,
                        const double* h_consts129, int64_t* d_dq);
int    kbbq_posterior_q_dev(kbbq_ctx* ctx, const double* d_prior_q, const int64_t* d_errs,
                            const int64_t* d_total, const double* d_comb, int64_t ncells,
                            const double* h_consts129, int64_t* d_post_q);
/* host helpers of the solve (no GPU): kbbq_combiln_host fills comb[i] for cells (errs[i], total[i]) with
 * the restated SciPy gammaln (csrc/solve_host.cpp; bit-identical to scipy.special.gammaln for positive
 * arguments, NaN for cells outside the distribution's support, which the solve ignores), on `threads`
 * threads; kbbq_gammaln_host exposes the gammaln itself for the tests.                              */
int    kbbq_combiln_host(const int64_t* errs, const int64_t* total, int64_t ncells, double* comb, int threads);
/* The host half of one solve in one threaded pass over the count tables as they come off the device (layout of
 * kbbq_accumulate_dev's d_tables): marg = [q_errs R*43 | q_total R*43 | rg_errs R | rg_total R] (recalibrate.py:112-115)
 * and aux = the gammaln term of every cell in kbbq_solve_dev's order [rg R | q R*43 | pos R*43*S2 | dinuc R*43*16]. */
int kbbq_solve_prep_host(const int64_t* tables, int R, int S2, double* aux, int64_t* marg, int threads);
int    kbbq_gammaln_host(const double* x, int64_t n, double* out);
/* the solve's two log tables without importing SciPy: logp[i] = xlogy(1, p[i]) = log(p[i]) and log1mp[i] = xlog1py(1, -p[i])
 * = log1p(-p[i]), both with the C library's routines, which is what SciPy 1.15 evaluates for real arguments (compared bit
 * for bit in tests/test_solve_core_host.py); the reference reaches both through scipy.stats.binom.logpmf
 * (compare_reads.py:254). */
int    kbbq_xlogy_tables_host(const double* p, int n, double* logp, double* log1mp);
/* The solve without the host (csrc/lgam_core.h): kbbq_libm_log_data reads the constants of the host libm's own log()
 * (ln 2 split in two, 5 coefficients, 128 x {1/c, log c}: 263 doubles) out of the mapped library, so that a kernel can
 * evaluate the same gammaln bit for bit; KBBQ_E_HIP when they are not found (then kbbq_solve_dev's host-fed form stays
 * in use).  kbbq_gammaln_restated_host evaluates that restatement on the host (no call into libm; x integer-valued
 * and >= 1), kbbq_gammaln_dev on the device: the caller checks either against kbbq_gammaln_host before relying on
 * kbbq_solve_device_dev.  */
int    kbbq_libm_log_data(double* out263, int count);
int    kbbq_gammaln_restated_host(const double* x, int64_t n, const double* logtab263, double* out);
int    kbbq_gammaln_dev(kbbq_ctx* ctx, const double* d_x, int64_t n, const double* d_logtab263, double* d_out);
/* kbbq_solve_dev with NOTHING from the host inside the step: marginals, the gammaln term of every cell (d_logtab263:
 * kbbq_libm_log_data's table on the device) and meanq = p_to_q(sum_q q_total[q] * 10^(-q/10) / rg_total)
 * (recalibrate.py:111,120; compare_reads.py:262-271) are formed by the kernels.  h_consts172 = the 129 model constants
 * of kbbq_solve_dev + the 43 float64 values q_to_p(q).  meanq is a TRUNCATION of a longdouble quotient in the
 * reference; the kernel forms it in double precision and, when -10 log10(mean) lies within 1e-7 of an integer (where
 * the reference's own 80-bit rounding decides; always so when a read group has one quality value only) or a read
 * group has no counted base, reports KBBQ_E_MEANQ through kbbq_ctx_status: the caller then solves through
 * kbbq_solve_dev with the host's longdouble meanq.  d_meanq_out [R] (may be NULL) receives the kernel's meanq.   */
int    kbbq_solve_device_dev(kbbq_ctx* ctx, const int64_t* d_tables, int R, int S2, int minscore,
                             const double* d_logtab263, const double* h_consts172, int32_t* d_post_q,
                             void* d_lut, int32_t* d_dq, int32_t* d_meanq_out);
size_t kbbq_solve_aux_count(int R, int S2);
size_t kbbq_solve_dq_count(int R, int S2);
int    kbbq_solve_dev(kbbq_ctx* ctx, const int64_t* d_tables, int R, int S2, int minscore,
                      const int32_t* d_meanq, const double* d_aux, const double* h_consts129,
                      int32_t* d_post_q, void* d_lut_blob, int32_t* d_dq);

/* ---- K4 / K5: benchmark-path error flagging -----------------
 * kbbq_find_errors_dev replaces compare_reads.find_read_errors (compare_reads.py:84-139)
 * for a batch of aligned reads: per read a CIGAR walk against its reference window
 * [ref_start, ref_start + ref_len) of the concatenated genome / skip-mask byte arrays
 * (skip mask = benchmark.get_full_skips, benchmark.py:22-39); cigar ops are
 * (length << 4 | op) with BAM op codes.  Outputs one byte per base (0 / 1) in the err and
 * skip planes; flip[r] != 0 reverses both for reverse-strand reads (benchmark.py:70-72).
 * The planes (seq, err, skip) are 16-byte aligned; genome / skipmask hold genome_len bytes each and
 * need no alignment or padding (reference windows are read with unaligned 16-byte loads, byte by
 * byte at the very end of the arrays).
 * Python's negative-index wraps of the reference (skips[-1], subset[-1]) are reproduced;
 * its IndexError / ValueError cases arrive through kbbq_ctx_status as KBBQ_E_INDEX /
 * KBBQ_E_RANGE.
 * kbbq_count_q_dev replaces the two np.bincount calls of benchmark.calculate_q
 * (benchmark.py:76-91) over the unskipped bases: counts[0..255] = observations per
 * quality (byte - qoffset), counts[256..511] = errors; ADDS into d_counts512.
 * ONE PLANE OF FLAGS: with d_skip == NULL kbbq_find_errors_dev writes both answers into d_err -- bit 0 error,
 * bit 1 skip -- and kbbq_count_q_dev / kbbq_canonical_reads_dev called with d_skip == NULL read that plane: one
 * byte per base less to write and to read when no caller wants the two boolean arrays themselves.
 * Implementation (csrc/kbbq_aligned_kernels.h): the first four CIGAR operations of every read are copied into one
 * 16-byte record per read (context-owned scratch), a lane walks them in registers and fetches the one reference
 * window its chunk needs; chunks on an operation boundary, reads with more operations and malformed input take the
 * sequential walk.  KBBQ_K4=v1 in the environment selects the first form (A/B timing).          */
int kbbq_find_errors_dev(kbbq_ctx* ctx, const uint8_t* d_seq, const uint32_t* d_len, int64_t nreads, int pitch,
                         const int64_t* d_ref_start, const int32_t* d_ref_len,
                         const uint32_t* d_cig_off, const uint32_t* d_cig_n, const uint32_t* d_cigar,
                         const uint8_t* d_genome, const uint8_t* d_skipmask, int64_t genome_len,
                         const uint8_t* d_flip, uint8_t* d_err, uint8_t* d_skip);
/* d_skipmask may be NULL: the reference is then FUSED -- bit 7 of every d_genome byte is that site's skip flag, the
 * low 7 bits its letter (ASCII) -- and a chunk fetches one scattered 16-byte window instead of two. */
/* kbbq_canonical_reads_dev: first half of gatk.bqsr.bam_to_bqsr_covariates (gatk/bqsr.py:52-123;
 * strand-aware covariates :23-50, skips :86-88).  Rewrites aligned reads into sequencing
 * orientation so that kbbq_accumulate_dev tallies them: per read the aligned part
 * [clip & 0xFFFF, clip >> 16) only, reverse-strand reads (flags bit 0) reverse-complemented
 * (letters outside ACGT -> N), padded to the common length S with uncounted bases; a base is
 * given quality byte 0 (never counted) when its K4 skip flag is set, its original quality is
 * below minscore, it lies in the trimmed range [trim & 0xFFFF, trim >> 16) (adaptor, host) or
 * it is N; out_cseq differs from out_seq exactly where the K4 error flag is set; out_meta =
 * S | read group (flags >> 16) << 16 | read 2 (flags bit 1) << 31.  The reference's TypeError
 * (a looked-up dinucleotide with a letter outside ACGT on a forward read, decided on the
 * ORIGINAL qualities with dinuc_minscore) arrives as KBBQ_E_TYPE.  All planes 16-byte
 * aligned, nreads * pitch bytes each, no padding rows needed.                           */
int kbbq_canonical_reads_dev(kbbq_ctx* ctx, const uint8_t* d_seq, const uint8_t* d_oq, const uint8_t* d_err,
                             const uint8_t* d_skip, const uint32_t* d_len, const uint32_t* d_clip,
                             const uint32_t* d_trim, const uint32_t* d_flags, int64_t nreads, int pitch, int S,
                             int minscore, int dinuc_minscore, uint8_t* d_out_seq, uint8_t* d_out_cseq,
                             uint8_t* d_out_qual, uint32_t* d_out_meta);
/* The same with the layout of the output rows chosen: 0 (character planes, as above) or KBBQ_ROWS_NIBBLES -- d_out_seq /
 * d_out_cseq are then 4-bit planes of nreads * pitch / 2 bytes (the layout kbbq_accumulate_rows_dev takes with that
 * flag; one byte per base less to write here and to read there).  A corrected base differs from its base in the low
 * bit of its code (an N: code 5 in d_out_cseq only).  Rows that cannot be packed -- a letter outside ACGTN on a forward-strand read -- make
 * kbbq_ctx_status return KBBQ_E_LUT: repeat the call with layout 0 (which is also where KBBQ_E_TYPE is decided). */
int kbbq_canonical_reads_rows_dev(kbbq_ctx* ctx, const uint8_t* d_seq, const uint8_t* d_oq, const uint8_t* d_err,
                                  const uint8_t* d_skip, const uint32_t* d_len, const uint32_t* d_clip,
                                  const uint32_t* d_trim, const uint32_t* d_flags, int64_t nreads, int pitch, int S,
                                  int minscore, int dinuc_minscore, int layout, uint8_t* d_out_seq,
                                  uint8_t* d_out_cseq, uint8_t* d_out_qual, uint32_t* d_out_meta);
/* kbbq_accumulate_aligned_dev: kbbq_canonical_reads_rows_dev + kbbq_accumulate_rows_dev in ONE kernel (K6 fused into K1): the
 * tally of gatk.bqsr.bam_to_bqsr_covariates (gatk/bqsr.py:52-123) straight from the reads as aligned -- d_seq, d_oq (OQ
 * characters) and d_flagplane (kbbq_find_errors_dev's one plane of flags: bit 0 error, bit 1 skip) are [nreads, pitch]
 * planes of reads of the common length S (pitch = S rounded up to 16); d_clip / d_trim / d_flags per read as for
 * kbbq_canonical_reads_dev.  Counts ADD into d_tables (R, S2 = 2 S).  3 B/base read instead of 3 + 2 written + 2 read
 * again.  A forward-strand read with a letter outside ACGTN makes kbbq_ctx_status return KBBQ_E_LUT: tally through
 * kbbq_canonical_reads_rows_dev(layout 0), where the reference's TypeError is decided; a shape whose tables do not fit the
 * LDS, and reads shorter than 32 bases, return KBBQ_E_LUT at once (nothing launched). */
int kbbq_accumulate_aligned_dev(kbbq_ctx* ctx, const uint8_t* d_seq, const uint8_t* d_oq, const uint8_t* d_flagplane,
                                const uint32_t* d_clip, const uint32_t* d_trim, const uint32_t* d_flags, int64_t nreads,
                                int pitch, int S, int R, int minscore, int dinuc_minscore, int64_t* d_tables);
/* kbbq_tally_aligned_dev: kbbq_find_errors_dev (no flip, one plane of flags, fused reference) + kbbq_accumulate_aligned_dev --
 * the whole device side of gatk.bqsr.bam_to_bqsr_covariates (gatk/bqsr.py:52-123: compare_reads.find_read_errors per read,
 * compare_reads.py:84-139, then the covariate counts) -- with ONE pass over the reads for every read that is a single
 * M / = / X operation over all of its bases: the tally kernel compares such a read with its reference window itself (read
 * bytes, OQ, reference: 3 B/base instead of 6 for the two calls).  Every other read (insertions, deletions, clips in the
 * CIGAR, more or odd operations, a window at the very end of the genome) is listed on the device, kbbq_find_errors_dev's kernel
 * runs over the listed reads only and writes THEIR rows of d_flagplane, which the tally kernel reads for them.  d_flagplane:
 * [nreads, pitch] scratch of the caller's (contents on entry irrelevant; rows of listed reads hold their flags afterwards).
 * d_genome: FUSED reference (bit 7 of a byte = the site's skip flag), genome_len bytes.  d_len[i] == S for every read (the
 * caller's promise, as for kbbq_accumulate_aligned_dev).  Counts ADD into d_tables and equal those of the two calls; the
 * same statuses arrive through kbbq_ctx_status (KBBQ_E_INDEX / KBBQ_E_RANGE from the CIGAR walk, KBBQ_E_LUT for a forward
 * read with a letter outside ACGTN); a shape the tally kernel does not serve returns KBBQ_E_LUT before anything is launched. */
int kbbq_tally_aligned_dev(kbbq_ctx* ctx, const uint8_t* d_seq, const uint8_t* d_oq, const uint32_t* d_len, int64_t nreads,
                           int pitch, int S, const int64_t* d_ref_start, const int32_t* d_ref_len,
                           const uint32_t* d_cig_off, const uint32_t* d_cig_n, const uint32_t* d_cigar,
                           const uint8_t* d_genome, int64_t genome_len,
                           const uint32_t* d_clip, const uint32_t* d_trim, const uint32_t* d_flags,
                           uint8_t* d_flagplane, int R, int minscore, int dinuc_minscore, int64_t* d_tables);
int kbbq_count_q_dev(kbbq_ctx* ctx, const uint8_t* d_qual, const uint8_t* d_err, const uint8_t* d_skip,
                     const uint32_t* d_len, int64_t nreads, int pitch, int qoffset, int64_t* d_counts512);

/* ---- mate-pair rows: an optional device layout for paired reads of one length S --------------
 * One row per pair: [mate 1: S bytes][separator][mate 2: S bytes][padding to a multiple of 16];
 * separator and padding are 'N' (seq, cseq) / 0 (qual): uncounted bases.  For 2 x 150 bp the pitch
 * is 304 instead of 2 x 160: 5 % fewer bytes through HBM for the same bases.  K1 / K2 run on these
 * rows as on single reads (byte offset = stored cycle index; the separator gives mate 2's first
 * base "no previous base"); results are identical to the one-read-per-row path.  Preconditions
 * (the caller checks): reads alternate first / second in pair, both mates have length S = S2 / 2
 * and the same read group.  Data errors (quality > 42, alphabet) and rows the fast apply cannot
 * serve are reported through kbbq_ctx_status with ROW indices / as KBBQ_E_LUT: re-run on
 * one-read-per-row planes for the reference's exact error.
 * sidecar of a pair row: (2S + 1) | read group << 16.
 * kbbq_pack_pairs_dev: planes [2 * npairs, pitch] + sidecars -> pair planes [npairs, kbbq_pair_pitch(S2)]
 * (d_cseq / d_pcseq may be NULL); kbbq_unpack_pairs_dev: one pair plane (the K2 output) -> [2 * npairs, pitch].
 * kbbq_pair_lut_dev derives the pair-row apply LUT (kbbq_pair_lut_bytes) from the blob kbbq_solve_dev /
 * kbbq_build_lut produced (its flags must be 0: values fit int8 and stay in 0..255).          */
int    kbbq_pair_pitch(int S2);
size_t kbbq_pair_lut_bytes(int R, int Qt, int S2);
int    kbbq_pack_pairs_dev(kbbq_ctx* ctx, const uint8_t* d_seq, const uint8_t* d_cseq, const uint8_t* d_qual,
                           const uint32_t* d_meta, int64_t npairs, int pitch, int S2,
                           uint8_t* d_pseq, uint8_t* d_pcseq, uint8_t* d_pqual, uint32_t* d_pmeta);
int    kbbq_unpack_pairs_dev(kbbq_ctx* ctx, const uint8_t* d_pplane, int64_t npairs, int S2, int pitch, uint8_t* d_plane);
int    kbbq_accumulate_pairs_dev(kbbq_ctx* ctx, const uint8_t* d_pseq, const uint8_t* d_pcseq, const uint8_t* d_pqual,
                                 const uint32_t* d_pmeta, int64_t npairs, int R, int S2, int minscore,
                                 int dinuc_minscore, int64_t* d_tables);
int    kbbq_pair_lut_dev(kbbq_ctx* ctx, const void* d_lut_blob, int R, int S2, int minscore, void* d_pair_lut);
/* the same for rows described by layout flags (below): KBBQ_ROWS_PAIRS [| KBBQ_ROWS_NIBBLES] [| KBBQ_ROWS_TWINS] */
int    kbbq_pair_lut_rows_dev(kbbq_ctx* ctx, const void* d_lut_blob, int R, int S2, int minscore, int flags, void* d_pair_lut);
int    kbbq_apply_pairs_dev(kbbq_ctx* ctx, const uint8_t* d_pseq, const uint8_t* d_pqual, const uint32_t* d_pmeta,
                            int64_t npairs, int R, int S2, int minscore, const void* d_lut_blob,
                            const void* d_pair_lut, uint8_t* d_pout);

/* ---- rows grouped by read group -----------------------------------------------------------
 * With many read groups each K1 workgroup (its LDS tables hold ONE group) used to scan every row and keep
 * its own, and K2's table-driven LUT outgrew the LDS.  If the caller orders the rows by read group
 * (stable: any order within a group gives the same counts) and passes d_seg[R + 1] -- group g owns rows
 * [d_seg[g], d_seg[g + 1]) -- every slice walks only its rows with all lanes busy and stages only its
 * group's LUT rows: any number of read groups runs at the single-group rate.  `pairs` selects mate-pair
 * rows (then pitch must be kbbq_pair_pitch(S2) and d_pair_lut comes from kbbq_pair_lut_dev; NULL otherwise).
 * The fast apply LUT must be range-safe (blob flags 0); rows it cannot serve are reported (KBBQ_E_LUT).  */
int    kbbq_accumulate_grouped_dev(kbbq_ctx* ctx, const uint8_t* d_seq, const uint8_t* d_cseq, const uint8_t* d_qual,
                                   const uint32_t* d_meta, int64_t nrows, int pitch, int pairs, int R, int S2,
                                   int S_band, int S_min, int minscore, int dinuc_minscore, const int64_t* d_seg,
                                   int64_t* d_tables);      /* S_band, S_min as in kbbq_accumulate_band_dev (0 with pairs) */
int    kbbq_apply_grouped_dev(kbbq_ctx* ctx, const uint8_t* d_seq, const uint8_t* d_qual, const uint32_t* d_meta,
                              int64_t nrows, int pitch, int pairs, int R, int S2, int minscore,
                              const void* d_lut_blob, const void* d_pair_lut, const int64_t* d_seg, uint8_t* d_out);

/* ---- layouts in one interface: flags, 4-bit sequence planes, the layout pass -----------------
 * KBBQ_ROWS_PAIRS   the rows are mate-pair rows (above; pitch = kbbq_pair_pitch(S2)).
 * KBBQ_ROWS_NIBBLES the seq / cseq planes hold one NIBBLE per base instead of one character: the nucleotide
 *                   code of compare_reads.py:199 (A0 T1 G2 C3; 4 = N, separator, padding), row stride pitch / 2,
 *                   8 bytes per 16-base chunk; word w (0, 1) of a chunk holds bases 8w..8w+3 in the low nibbles of
 *                   its bytes and bases 8w+4..8w+7 in the high nibbles.  Only batches whose seq AND cseq are
 *                   entirely ACGTN have such planes (kbbq_lay_out_dev reports KBBQ_E_LUT otherwise: keep byte
 *                   planes, which carry the reference's TypeError semantics); find_corrected_sites
 *                   (recalibrate.py:13-20) then compares codes.  K1 reads 2 B/base instead of 3, K2 2.5 instead of 3.
 * kbbq_accumulate_rows_dev / kbbq_apply_rows_dev: K1 / K2 on any combination (d_seg NULL: rows not grouped).
 * d_perm (apply, optional, with d_seg): row i of the batch is STORED as row d_perm[i] of d_out -- the rows go
 * straight back into the order they had before kbbq_group_rows_dev, no separate pass.
 * kbbq_meta_stats_dev: one pass over the sidecars (synchronises): h_stats8[0] shortest non-empty read, [1] longest
 * read, [2] largest read-group id, [3] violations of the mate-pair preconditions (0 = uniform first / second pairs
 * of one length and read group), [4] empty reads, [5] violations of the KBBQ_ROWS_TWINS preconditions (0 = single-end reads of one
 * length whose neighbours 2p / 2p+1 share a read group).
 * kbbq_group_rows_dev: stable counting sort of the rows (pairs != 0: of the PAIRS, by the first mate's sidecar)
 * by read group, R <= 256: d_perm[nrows] (row i of the grouped order is row d_perm[i]) and d_seg[R + 1];
 * d_work: kbbq_group_rows_work_bytes(nrows, R) bytes.  A sidecar with read group >= R -> KBBQ_E_RANGE (status).
 * kbbq_lay_out_dev: input-order rows [nreads, pitch] -> the destination layout in ONE pass (pair packing, gather by
 * d_perm (may be NULL), nibble packing fused): destination planes [nrows, dpitch] (seq / cseq: dpitch / 2 with
 * KBBQ_ROWS_NIBBLES), nrows = nreads / 2 and dpitch = kbbq_pair_pitch(S2) with KBBQ_ROWS_PAIRS, else nreads and pitch.  */
#define KBBQ_ROWS_PAIRS    1
#define KBBQ_ROWS_NIBBLES  2
/* KBBQ_ROWS_TWINS (with KBBQ_ROWS_PAIRS): the two reads of every row are BOTH first in pair -- single-end input of one
 * length packed two reads to a mate-pair row (same planes, same sidecar, same 5 % fewer bytes); the second half of a row
 * then counts into / looks up the forward cycle columns [0, S) like the first half instead of the mirrored ones
 * (compare_reads.py:304-306: no "/2" in the name).  Precondition (kbbq_meta_stats_dev h_stats8[5] == 0): no read is second
 * in pair, reads 2p and 2p+1 share their read group, all reads have one length (an odd count is fine: the last row's second
 * half is padding, kbbq_lay_out_dev writes (nreads + 1) / 2 rows).  kbbq_accumulate_rows_dev and
 * kbbq_pair_lut_rows_dev take the flag (the apply kernel reads it out of the LUT); kbbq_lay_out_dev packs such rows as any
 * other pair rows.  */
#define KBBQ_ROWS_TWINS    4
int    kbbq_accumulate_rows_dev(kbbq_ctx* ctx, const uint8_t* d_seq, const uint8_t* d_cseq, const uint8_t* d_qual,
                                const uint32_t* d_meta, int64_t nrows, int pitch, int flags, int R, int S2, int S_band,
                                int S_min, int minscore, int dinuc_minscore, const int64_t* d_seg, int64_t* d_tables);
int    kbbq_apply_rows_dev(kbbq_ctx* ctx, const uint8_t* d_seq, const uint8_t* d_qual, const uint32_t* d_meta, int64_t nrows,
                           int pitch, int flags, int R, int S2, int minscore, const void* d_lut_blob, const void* d_pair_lut,
                           const int64_t* d_seg, const int64_t* d_perm, uint8_t* d_out);
int    kbbq_meta_stats_dev(kbbq_ctx* ctx, const uint32_t* d_meta, int64_t nreads, int32_t* h_stats8);
/* ---- all length bands of a mixed-length input in one launch per kernel (BASELINE config 5) --------------------
 * The reference grows its count arrays as the reads get longer (recalibrate.py:81-101) and tallies / applies read by read;
 * the file path cuts a (length-sorted: SURVEY H2) input into bands of one row width each and used to launch K1 and K2 once
 * per band -- pipeline fill, table zeroing and flush paid per band.  kbbq_accumulate_bands_dev / kbbq_apply_bands_dev take
 * the whole list: bands the merged kernels serve (K1: every shape the table-driven kernel fits except the
 * chunk-position-major form of 2 x 150 bp mate-pair rows; K2: one-read-per-row 4-bit planes, one read group, the shapes the
 * short-lived kernel takes) run as ONE launch, every band on its share of the workgroups with its own pitch, LDS table
 * geometry (S_band, S_min as in kbbq_accumulate_band_dev) and pitch-narrowed LUT; the others are launched as
 * kbbq_accumulate_rows_dev / kbbq_apply_rows_dev (or kbbq_accumulate_band_dev / kbbq_apply_dev for flags == 0 without d_seg)
 * would launch them.  Results are those of the per-band calls (counts ADD into d_tables; every band's d_out gets its new
 * qualities); d_cseq is not read by apply, d_perm / d_out / d_pair_lut not by accumulate. */
typedef struct kbbq_band {
    const uint8_t* d_seq; const uint8_t* d_cseq; const uint8_t* d_qual; const uint32_t* d_meta;
    int64_t nrows; int32_t pitch; int32_t flags;           /* KBBQ_ROWS_* */
    int32_t S_band; int32_t S_min;                         /* longest / shortest read of the band (0: as kbbq_accumulate_band_dev) */
    const int64_t* d_seg; const int64_t* d_perm;           /* rows grouped by read group (may be NULL) */
    uint8_t* d_out; const void* d_pair_lut;                /* apply: output plane; mate-pair rows: kbbq_pair_lut_rows_dev's LUT */
} kbbq_band;
int    kbbq_accumulate_bands_dev(kbbq_ctx* ctx, const kbbq_band* bands, int nbands, int R, int S2, int minscore,
                                 int dinuc_minscore, int64_t* d_tables);
int    kbbq_apply_bands_dev(kbbq_ctx* ctx, const kbbq_band* bands, int nbands, int R, int S2, int minscore, const void* d_lut_blob);
size_t kbbq_group_rows_work_bytes(int64_t nrows, int R);
int    kbbq_group_rows_dev(kbbq_ctx* ctx, const uint32_t* d_meta, int64_t nrows, int pairs, int R, void* d_work,
                           int64_t* d_perm, int64_t* d_seg);
int    kbbq_lay_out_dev(kbbq_ctx* ctx, const uint8_t* d_seq, const uint8_t* d_cseq, const uint8_t* d_qual,
                        const uint32_t* d_meta, int64_t nreads, int pitch, int flags, int S2, const int64_t* d_perm,
                        uint8_t* d_lseq, uint8_t* d_lcseq, uint8_t* d_lqual, uint32_t* d_lmeta);
int    kbbq_unpack_nibbles_dev(kbbq_ctx* ctx, const uint8_t* d_nib, int64_t nbases, uint8_t* d_chars);
/* d_dst[i] += d_src[i], i < n: count tables tallied apart (a length band, a layout tried first) into the file's */
int    kbbq_tables_add_dev(kbbq_ctx* ctx, int64_t* d_dst, const int64_t* d_src, size_t n);

/* ---- the exchange step of the sharded path without torch: RCCL allreduce of the count tables -------------
 * Reads shard across GPUs by contiguous record ranges, every rank tallies its shard (K1) and the ONLY exchange is the
 * sum of the int64 count-table buffers; marginals, meanq and the solve are then replicated and K2 is
 * local.  One process (or thread) per GPU, each with its own kbbq_ctx.  Rank 0 obtains a 128-byte id
 * (kbbq_comm_unique_id) and hands it to the other ranks by whatever channel the caller has (file, socket, MPI);
 * every rank then calls kbbq_comm_create with the same id -- collective, as ncclCommInitRank is.
 * kbbq_allreduce_tables: in place, enqueued on the context's stream (ordered between K1 and the solve, no host wait).
 * librccl is bound at the first call with dlopen (a copy already loaded by the process, else KBBQ_RCCL_LIB /
 * kbbq_comm_library(path), else the loader's librccl.so.1, else /opt/rocm/lib): libkbbq_hip.so itself does not link it.
 * The Python layer of this repository uses torch.distributed for the same step (kbbq/parallel.py); these entry points
 * are for C / C++ / other-language callers.  */
#define KBBQ_COMM_ID_BYTES 128
typedef struct kbbq_comm kbbq_comm;
int kbbq_comm_library(const char* path);
int kbbq_comm_unique_id(void* id128);
int kbbq_comm_create(kbbq_ctx* ctx, const void* id128, int nranks, int rank, kbbq_comm** out);
int kbbq_allreduce_tables(kbbq_comm* comm, int64_t* d_buf, size_t n);
int kbbq_comm_destroy(kbbq_comm* comm);

/* ---- host SAM / BAM reader (no GPU) ----------------------------------------------------------
 * Replaces, for the truth-set benchmark and the BAM-sourced tally, what the reference gets from
 * pysam.AlignmentFile / AlignedSegment (benchmark.py:57-74,102-143; gatk/bqsr.py:23-123): per alignment
 * the flag, contig, position, CIGAR, mate position, template length, sequence, qualities and the RG / OQ
 * tags, as arrays (the kernels take whole batches).  SAM text, gzip-compressed SAM or BAM (the reference's own input);
 * mapped, inflated where needed, indexed and parsed in parallel.
 * kbbq_sam_info: { alignments, CIGAR operations, longest SEQ, contigs seen, @RG lines, header lines }.
 * kbbq_sam_fields (any pointer may be NULL): flag; contig = index into the first-appearance list of RNAME;
 * pos / pnext 0-based; tlen; qlen = len(SEQ) (0 for '*'); ref_span = reference_end - reference_start;
 * clip = query_alignment_start | query_alignment_end << 16; cig_off / cig_n into kbbq_sam_cigar's array
 * (length << 4 | BAM op code, unknown letter = 15); rg = index of the RG:Z tag among the header's @RG IDs
 * (-1 no tag, -2 not in the header); has_qual_oq = len(QUAL) (0 for '*') | len(OQ:Z) << 16 (-1 << 16 when absent).
 * kbbq_sam_fill: plane rows [0, n) <- SEQ (0) / QUAL (1) / OQ (2) of alignments [first, first + n), zero padded.
 * kbbq_sam_text: 0 QNAME, 1 whole alignment line, 2 header line, 3 contig name, 4 @RG ID (not NUL-terminated).  */
typedef struct kbbq_sam kbbq_sam;
/* path: SAM text, gzip / bgzip-compressed SAM, or BAM (inflated with zlib, records rendered as SAM lines) */
int kbbq_sam_open(const char* path, kbbq_sam** out);
int kbbq_sam_close(kbbq_sam* f);
int kbbq_sam_info(const kbbq_sam* f, int64_t* info6);
int kbbq_sam_fields(const kbbq_sam* f, int32_t* flag, int32_t* contig, int64_t* pos, int64_t* pnext, int64_t* tlen,
                    int32_t* qlen, int32_t* ref_span, uint32_t* clip, uint32_t* cig_off, uint32_t* cig_n, int32_t* rg,
                    int32_t* has_qual_oq);
int kbbq_sam_cigar(const kbbq_sam* f, uint32_t* ops);
/* trim[i] = lo | hi << 16: query positions [lo, hi) of alignment i lie past its adaptor boundary (0: none) -- the reference's
 * bamread_adaptor_boundary + trim_bamread (gatk/bqsr.py:131-206) for every alignment of the file, by CIGAR walk. */
int kbbq_sam_adaptor_trim(const kbbq_sam* f, uint32_t* trim);
/* A whole regular file as bytes, inflated when it is gzip / bgzip (bgzip blocks side by side, gzip members chunk-wise on all host threads):
 * where kbbq/aln.py's FASTA / VCF / BED readers (the reference reads those through pysam: benchmark.py:9-55) take their text from.
 * *data stays valid until kbbq_text_close. */
typedef struct kbbq_text kbbq_text;
int kbbq_text_open(const char* path, kbbq_text** out, const uint8_t** data, size_t* n);
int kbbq_text_close(kbbq_text* t);
int kbbq_sam_fill(const kbbq_sam* f, int64_t first, int64_t n, int pitch, int which, uint8_t* plane);
int kbbq_sam_text(const kbbq_sam* f, int what, int64_t i, const char** p, int64_t* len);


/* ---- host FASTQ ingest / egress (no GPU) ----------------------------------
 * Replaces, for this path, pysam.FastxFile iteration (recalibrate.py:56-57,141-142),
 * the name parsing of compare_reads.py:304-318 / recalibrate.py:59-64 and the print()
 * calls of recalibrate.py:153-156.  4-line records; name = header up to whitespace.
 * kbbq_fastq_scan(a, b|NULL, infer_rg, info[5]) validates in the reference's order and
 * returns info = { usable reads, longest read S, read groups R, error kind, error index }
 * with kinds 1 RG inference IndexError, 2 RG AssertionError, 3 name prefix
 * (AssertionError), 4 length mismatch (ValueError), 5 shorter than the running maximum
 * (IndexError; that read is still included so the device can check it first).
 * kbbq_fastq_fill packs reads [0, n) into the padded planes (multi-threaded);
 * kbbq_fastq_format renders "@name\nseq\n+\nqual\n" records with the new qualities.  */
typedef struct kbbq_fastq kbbq_fastq;
int         kbbq_fastq_open(const char* path, kbbq_fastq** out);
int         kbbq_fastq_close(kbbq_fastq* f);
/* multi-GPU ingest: rank 0 opens and scans the whole file, the other ranks index only their shard's byte range
 * [byte_lo, byte_hi) (record starts from kbbq_fastq_record_offset; byte_hi < 0: to the end; uncompressed files only,
 * kbbq_fastq_is_plain) and take the file-wide read-group names from rank 0 (count NUL-terminated strings). */
int         kbbq_fastq_open_range(const char* path, int64_t byte_lo, int64_t byte_hi, kbbq_fastq** out);
/* The first record start at or after `offset` (the file size when there is none; -1 on error): a rank of a multi-GPU
 * run cuts ITS byte range out of an uncompressed file with it, nobody indexes the whole file (kbbq/fastx.py). */
int64_t     kbbq_fastq_sync_offset(const char* path, int64_t offset);
/* skip_second != 0: never cut in front of a second-in-pair record (name field ends with "/2"): mates stay together */
int64_t     kbbq_fastq_sync_offset_ex(const char* path, int64_t offset, int skip_second);
int64_t     kbbq_fastq_record_offset(const kbbq_fastq* f, int64_t i);
int         kbbq_fastq_is_plain(const kbbq_fastq* f);
int         kbbq_fastq_set_rg_names(kbbq_fastq* f, const char* names, int count);
int64_t     kbbq_fastq_count(const kbbq_fastq* f);
int         kbbq_fastq_name(const kbbq_fastq* f, int64_t i, const char** name, int* len);
int         kbbq_fastq_rg_count(const kbbq_fastq* f);
const char* kbbq_fastq_rg_name(const kbbq_fastq* f, int i);
int         kbbq_fastq_scan(kbbq_fastq* a, const kbbq_fastq* b, int infer_rg, int64_t* info5);
/* ---- the same files read SEQUENTIALLY (csrc/fastq_stream.cpp): inputs that cannot be mapped or sought, inputs of any size ----
 * The reference walks its inputs read by read in constant memory (recalibrate.py:56-57 zip(FastxFile(A), FastxFile(B)),
 * :141-156 a second walk over A), so `-f <(zcat a.fq.gz) <(zcat b.fq.gz)` and files larger than memory work there.  A
 * kbbq_fastq_stream hands out SEGMENTS -- whole records, each an ordinary kbbq_fastq over memory of its own (scan / meta /
 * fill_rows / format work on it; close it with kbbq_fastq_close): the leading file about `max_bytes` per segment
 * (records = 0), the following file exactly the leader's number of records (records > 0; fewer only when it ends first: zip()
 * stops there).  *segment = NULL when the input has ended; *at_end = 1 when it ends behind this segment.  path: a regular
 * file, a named pipe / process substitution, "-" = standard input; gzip / bgzip bytes (a .fq.gz file of any size, a pipe that
 * carries them) are inflated as they are read, member after member, in constant memory -- as pysam.FastxFile reads them.
 * kbbq_fastq_stream_tee: every byte handed out is also appended to `fd` -- the spool a pipe's file A is kept in for pass 2.
 * kbbq_fastq_scan_next: kbbq_fastq_scan of one segment, with what the reference's walk carries from read to read handed
 * over by the caller: the read groups met so far (kbbq_fastq_set_rg_names first; new ones are appended, first-appearance
 * order over the whole input, recalibrate.py:59-64) and the longest read so far (recalibrate.py:89-101).                  */
typedef struct kbbq_fastq_stream kbbq_fastq_stream;
int         kbbq_fastq_stream_open(const char* path, kbbq_fastq_stream** out);
int         kbbq_fastq_stream_is_regular(const kbbq_fastq_stream* s);
int         kbbq_fastq_stream_tee(kbbq_fastq_stream* s, int fd);
int         kbbq_fastq_stream_next(kbbq_fastq_stream* s, size_t max_bytes, int64_t records, kbbq_fastq** segment, int* at_end);
int         kbbq_fastq_stream_prefetch(kbbq_fastq_stream* s, size_t bytes);   /* read ahead until `bytes` wait in the stream (another thread, while the leading file is read) */
int         kbbq_fastq_stream_close(kbbq_fastq_stream* s);
int         kbbq_fastq_scan_next(kbbq_fastq* a, const kbbq_fastq* b, int infer_rg, int64_t prior_longest, int64_t* info5);
/* kbbq_fastq_open of both files (b may be NULL) + kbbq_fastq_scan on the library's own threads; _begin returns at
 * once, _wait joins, hands over the readers and info5 and frees the job (call it exactly once).  File A's error is
 * reported before file B's.  Lets a Python caller set its device up while the files are being read. */
typedef struct kbbq_fastq_job kbbq_fastq_job;
int         kbbq_fastq_pair_begin(const char* path_a, const char* path_b, int infer_rg, kbbq_fastq_job** job);
int         kbbq_fastq_pair_wait(kbbq_fastq_job* job, kbbq_fastq** a, kbbq_fastq** b, int64_t* info5);
int         kbbq_fastq_lengths(const kbbq_fastq* f, int64_t first, int64_t n, uint32_t* out);   /* sequence lengths */
/* length bands of reads [first, first + n): maximal runs of reads of one length class (class = first entry of the
 * ascending classes[] >= the length); out[run] = {lo, hi, longest, shortest non-empty}; returns the number of runs,
 * or 1 run covering everything when there are more than max_runs */
int         kbbq_fastq_length_runs(const kbbq_fastq* f, int64_t first, int64_t n, const uint32_t* classes, int nclasses,
                                   int max_runs, int64_t* out);
int         kbbq_fastq_fill(const kbbq_fastq* a, const kbbq_fastq* b, int infer_rg, int64_t n, int pitch,
                            uint8_t* seq, uint8_t* cseq, uint8_t* qual, uint32_t* meta);
/* the same for reads [first, first + n) -> rows [0, n): one rank's shard; read-group ids are those of the scan */
int kbbq_fastq_fill_range(const kbbq_fastq* a, const kbbq_fastq* b, int infer_rg, int64_t first, int64_t n, int pitch,
                          uint8_t* seq, uint8_t* cseq, uint8_t* qual, uint32_t* meta);
/* ---- the packer writes the device layout itself (no kbbq_lay_out_dev pass, 2 B/base over PCIe instead of 3) ----
 * Replaces, like kbbq_fastq_fill_range, the pysam iteration + per-read arrays of recalibrate.py:56-57,89-101,141-148, but
 * hands the kernels the layout they are measured on: mate-pair rows / 4-bit sequence planes / rows gathered by
 * read-group segment (KBBQ_ROWS_* above), byte for byte what kbbq_lay_out_dev would make of kbbq_fastq_fill_range's rows.
 * kbbq_fastq_meta: the sidecar words of reads [first, first + n) (read-group ids of the scan) and the statistics of
 * kbbq_meta_stats_dev over them (stats8, same slots) -- what decides which layout the reads qualify for.
 * kbbq_group_rows_host: the stable counting sort of kbbq_group_rows_dev on the host (perm, seg).
 * kbbq_fastq_fill_rows: destination rows [row_lo, row_lo + nrows) of that band into plane rows 0 .. nrows - 1 (seq / cseq:
 * row stride pitch / 2 with KBBQ_ROWS_NIBBLES); KBBQ_ROWS_PAIRS: pitch = kbbq_pair_pitch(S2) (the argument is ignored) and
 * every read has length S2 / 2; perm (may be NULL): destination row r holds source row perm[r].  *foreign is set to 1 when
 * a seq / cseq letter outside ACGTN met the nibble packing: repeat the band without KBBQ_ROWS_NIBBLES (character planes
 * carry the reference's TypeError rule, compare_reads.py:224,292).
 * kbbq_fastq_format_rows: kbbq_fastq_format reading the new qualities out of the layout K2 wrote them in (mate-pair rows:
 * read first + i at row i / 2, byte offset (i & 1) * (S2 / 2 + 1)) -- recalibrate.py:153-156 without an unpack pass.  */
int         kbbq_fastq_meta(const kbbq_fastq* a, int infer_rg, int64_t first, int64_t n, uint32_t* meta, int32_t* stats8);
int         kbbq_group_rows_host(const uint32_t* meta, int64_t nrows, int pairs, int R, int64_t* perm, int64_t* seg);
int         kbbq_fastq_fill_rows(const kbbq_fastq* a, const kbbq_fastq* b, int64_t first, int64_t n, const uint32_t* meta,
                                 int flags, int S2, int pitch, const int64_t* perm, int64_t row_lo, int64_t nrows,
                                 uint8_t* seq, uint8_t* cseq, uint8_t* qual, uint32_t* dmeta, int* foreign);
int64_t     kbbq_fastq_format_rows(const kbbq_fastq* a, int64_t first, int64_t n, int flags, int S2, int pitch,
                                   const uint8_t* newqual, char* out, int64_t cap);
/* benchmark.py:102-124: idx[i] = the alignment whose QNAME + "/1"|"/2" equals the name of FASTQ read i up to its
 * first '_' (the last such alignment), -1 if none.                                                   */
int         kbbq_sam_match_fastq(const kbbq_sam* f, const kbbq_fastq* fq, int64_t* idx);
int64_t     kbbq_fastq_format(const kbbq_fastq* a, int64_t first, int64_t n, int pitch,
                              const uint8_t* newqual, char* out, int64_t cap);

/* ---- synthetic reads (bench / tests) -----------------------
 * Device twin of the generator documented in oracle/kbbq_oracle.c.          */
int kbbq_synth_dev(kbbq_ctx* ctx, uint8_t* d_seq, uint8_t* d_cseq, uint8_t* d_qual,
                   uint32_t* d_meta, int64_t first_read, int64_t nreads, int64_t total_reads,
                   int pitch, uint64_t seed, int len_lo, int len_hi, int nrg,
                   int qlo, int qhi, const uint32_t* thr43);

/* ---- timing of the most recent kernels (bench.py roofline) -------------
 * When enabled, every K1 / K2 launch is bracketed by HIP events on the launch
 * stream; kbbq_ctx_kernel_ms returns the accumulated milliseconds and launch
 * counts since the last reset (synchronises).  which: 0 = K1, 1 = K2.       */
int kbbq_ctx_timing(kbbq_ctx* ctx, int enable);
int kbbq_ctx_kernel_ms(kbbq_ctx* ctx, int which, double* total_ms, int64_t* launches, int reset);

#ifdef __cplusplus
}
#endif
#endif /* KBBQ_HIP_H */
