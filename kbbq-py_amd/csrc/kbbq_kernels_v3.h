// kbbq_kernels_v3.h -- table-driven, branch-free forms of K1 and K2 (the default path).
//
// Measured on MI355X (profiles/r01_pmc_v1.md): the first kernels (kbbq_kernels.h) were bound
// by VALU issue -- ~24 (K1) / ~22 (K2) vector instructions per base at 4 cycles per wave64
// integer instruction -- not by LDS (removing every LDS op bought 7 %) and not by HBM.
// These versions move per-base decisions into table layout so that a base costs ~7 (K1) /
// ~6 (K2) vector instructions:
//   K1: no "counted?" / "context defined?" tests: the LDS histogram has a TRASH row (all
//       qualities below minscore clamp onto it with one v_max) and TRASH context slots (the
//       context index is 5*code(prev)+code(cur) with code 4 = N / no previous base, so it is
//       always in range); trash is never flushed.  Second-in-pair columns are stored
//       mirrored so that both mates walk upwards and the per-base column step is an
//       instruction immediate.
//   K2: the LUT is indexed by the RAW quality byte: row 0 (padding) yields 0, rows of
//       uncounted qualities are identity rows, so there is no clamp, no "q >= minscore"
//       test and no select; a mirrored copy of the cycle entries serves second-in-pair
//       reads with ascending addresses.
// Same read-block / 16-byte-chunk decomposition, same error semantics (rare exact paths)
// and the same C ABI as the first kernels, which remain as the fallback for shapes these
// layouts do not fit in LDS (very long reads, many read groups).
#pragma once
#include "kbbq_kernels.h"
#include <type_traits>

#define K1V3_THREADS 1024        // one workgroup per CU: the replicated context table needs ~150 KB of LDS
#define K1V3_DNREP 16            // copies of the context-count table (copy = lane & (DN - 1)); long reads use 8 to fit the LDS
#ifndef K2V3_THREADS
#define K2V3_THREADS 512
#endif
#ifndef K2V3_NBUF
#define K2V3_NBUF 2             // chunk buffers in the prefetch ring
#endif
#ifndef K2V3_WAVES
#define K2V3_WAVES 4            // waves per SIMD the register allocation aims at
#endif
#define K1V3_FLUSH_ITERS 48          // 48 * 16 waves * 64 reads = 49,152 reads per workgroup between flushes (< 65,535)

struct K1v3Params {
    const uint8_t* seq; const uint8_t* cseq; const uint8_t* qual; const u32* meta;
    long long nreads; int pitch; int cpr; u32 cpr_magic; int R; int S;
    int minscore;               // counting threshold: row r of the LDS tables is quality minscore + r - 1
    int type_minscore;          // threshold of the dinucleotide lookup (TypeError rule); differs in SPLIT
    u32 qlo_m1;                 // 32 + minscore: quality bytes <= this hit the trash row
    u32 dlo;                    // SPLIT only: 33 + dinucleotide minscore
    int nrows;                  // 44 - minscore: one row per counted quality (row = 42 - q) + the trash row LAST
    u32 row_bytes;              // pos row stride in bytes ((3S | 1) words)
    u32 slack_bytes;            // after the last (trash) row: padding bytes of short reads index past its end
    int minlen;                 // rows are trimmed to 3S - minlen words: a shorter read is reported (ST_INDEX), not counted
    int dn_flush_iters;         // workgroup iterations between flushes of the (16-bit packed) context table
    int maxlen;                 // longest row the tables take: S, or 2S + 1 for mate-pair rows
    int gap;                    // mate-pair rows: 1 (the separator byte between the mates), else 0
    int twins;                  // mate-pair rows holding two FIRST-in-pair reads (single-end input packed two to a row): the second
                                // half counts into the forward columns [0, S) like the first, not into the mirrored ones
    int gS2;                    // columns of the GLOBAL cycle tables (2 x the longest read of the whole input); S may be
                                // smaller -- the longest read of THIS batch -- because a second-in-pair column
                                // 2*len-1-i does not depend on the S the LDS table is laid out for
    const long long* seg;       // rows grouped by read group: slice g owns rows [seg[g], seg[g + 1]); NULL: every slice scans all rows
    u64* tables; u64* status;
    // ALN form (aligned reads tallied in place, K6 fused into K1: see k1v3_body): seq = the reads as aligned, cseq = K4's plane of
    // flags (bit 0 error, bit 1 skip), qual = the OQ characters; per read, instead of `meta`:
    const u32* aflags;          // bit 0 reverse strand, bit 1 read 2, bits 16.. read group
    const u32* aclip;           // query_alignment_start | query_alignment_end << 16
    const u32* atrim;           // skipped range lo | hi << 16 (adaptor; lo == hi: none)
    // REF form (K4 folded in as well, for reads that are ONE M / = / X operation over all their bases): bit 2 of a read's aflags
    // word says "no plane of flags for this read: compare it with the reference here" -- cseq is then only read for the other
    // reads (k4v2_find_errors wrote their rows), and a chunk of such a read loads 16 bytes of `genome` (bit 7 of a byte: the
    // site's skip flag) at ag0[read] + its offset in the read instead
    const uint8_t* genome; const long long* ag0;
    // copies of the cycle table (1, 2 or 4; KJ == 0 forms): a chunk counts into copy (row slot & (copies - 1)), the flush sums them.
    // Narrow rows put many lanes of a wave on the same few columns (pitch 48: 21 lanes per column), i.e. onto the same (quality,
    // column) words: same-address LDS atomics serialise.  Their tables are small, so copies fit the LDS beside the context table.
    int pos_copies; u32 pos_copy_bytes;      // bytes from one copy to the next (rows + slack)
    // trash rows (1, 2, 4 or 8; KJ == 0 forms): every uncounted byte -- below minscore, and ALL the padding behind a read's last
    // base -- is binned on the trash row at its column, so the lanes of a wave that sit on the last chunk of their rows hit the same
    // few trash words over and over (a 16- to 21-way same-address atomic for narrow rows).  With several trash rows a chunk uses row
    // (row slot & (ntrash - 1)) of them: the clamp's upper bound becomes a per-chunk value, nothing is added per base.
    int ntrash;
};

// nucleotide decode of 4 bases: code (A0 T1 G2 C3, N/other 4), 5*code, and the character each
// code stands for (alphabet check), all by v_perm_b32 with the data as selector
__device__ __forceinline__ void decode4x(u32 w, u32& code, u32& code5, u32& expect)
{
    const u32 h = (w >> 1) & 0x07070707u;                                // A->0 C->1 T->2 G->3 N->7
    code   = __builtin_amdgcn_perm(0x04040404u, 0x02010300u, h);
    code5  = __builtin_amdgcn_perm(0x14141414u, 0x0A050F00u, h);         // 5 * code: A0 C15 T5 G10, N/other 20
    expect = __builtin_amdgcn_perm(0x4E000000u, 0x47544341u, h);
}

// increment of both count tables for base B of a word: errs << 16 | total = (seq byte != cseq byte) ? 0x10001 : 1.
// One sub-dword compare on byte B of xw = seq ^ cseq and one select (the compiler's own sequence masks the
// byte out first: three instructions).
template <int B>
__device__ __forceinline__ u32 k1_increment(u32 xw, u32 zero, u32 both)
{
    u32 inc;
    if (B == 0)      asm("v_cmp_ne_u32_sdwa vcc, %1, %2 src0_sel:BYTE_0 src1_sel:DWORD\n\tv_cndmask_b32_e32 %0, 1, %3, vcc" : "=v"(inc) : "v"(xw), "v"(zero), "v"(both) : "vcc");
    else if (B == 1) asm("v_cmp_ne_u32_sdwa vcc, %1, %2 src0_sel:BYTE_1 src1_sel:DWORD\n\tv_cndmask_b32_e32 %0, 1, %3, vcc" : "=v"(inc) : "v"(xw), "v"(zero), "v"(both) : "vcc");
    else if (B == 2) asm("v_cmp_ne_u32_sdwa vcc, %1, %2 src0_sel:BYTE_2 src1_sel:DWORD\n\tv_cndmask_b32_e32 %0, 1, %3, vcc" : "=v"(inc) : "v"(xw), "v"(zero), "v"(both) : "vcc");
    else             asm("v_cmp_ne_u32_sdwa vcc, %1, %2 src0_sel:BYTE_3 src1_sel:DWORD\n\tv_cndmask_b32_e32 %0, 1, %3, vcc" : "=v"(inc) : "v"(xw), "v"(zero), "v"(both) : "vcc");
    return inc;
}


// ---- 4-bit sequence planes ("packed" layouts) ------------------------------------------------
// A seq / cseq plane may hold one NIBBLE per base instead of one character: the nucleotide code itself
// (A0 T1 G2 C3, compare_reads.py:199; 4 = N, the separator and the padding), 8 bytes per 16-base chunk, row
// stride pitch / 2.  Word w of a chunk (w = 0, 1) carries bases 8w .. 8w+3 in the LOW nibbles of its four bytes and
// bases 8w+4 .. 8w+7 in the HIGH nibbles, so that `w & 0x0F0F0F0F` and `(w >> 4) & 0x0F0F0F0F` are the codes of four
// consecutive bases as bytes -- the form the per-base code below (and the quality words) use.  Only reads whose
// seq AND cseq are entirely ACGTN can be packed (k7_lay_out reports anything else and the caller keeps the
// byte planes, which carry the reference's exact TypeError semantics); the comparison of A1
// (recalibrate.py:13-20) is then a comparison of codes.  HBM traffic: K1 2 B/base instead of 3, K2 2.5 instead of 3.
#define NIBM 0x0F0F0F0Fu
__device__ __forceinline__ u32 nib_lo(u32 w) { return w & NIBM; }
__device__ __forceinline__ u32 nib_hi(u32 w) { return (w >> 4) & NIBM; }
// some nibble of w is not a code (>= 5)
__device__ __forceinline__ u32 nib_invalid(u32 w) { return (((w & 0x77777777u) + 0x33333333u) | w) & 0x88888888u; }
// 4 characters -> 4 codes as bytes (0..4), and whether all four are in ACGTN
__device__ __forceinline__ u32 chars_to_codes(u32 w, u32& bad)
{
    const u32 h = (w >> 1) & 0x07070707u;
    bad |= __builtin_amdgcn_perm(0x4E000000u, 0x47544341u, h) ^ w;
    return __builtin_amdgcn_perm(0x04040404u, 0x02010300u, h);
}
// 4 codes as bytes -> their characters
__device__ __forceinline__ u32 codes_to_chars(u32 c) { return __builtin_amdgcn_perm(0x4E4E4E4Eu, 0x43475441u, c & 0x07070707u); }

// 0x01 in every byte of x that is non-zero
__device__ __forceinline__ u32 nonzero_bytes(u32 x)
{
    return ((x | ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu)) >> 7) & 0x01010101u;
}

// 16 bytes starting at ANY byte offset: one global_load_dwordx4 with an unaligned address (the
// amdhsa targets run with unaligned access mode on; the compiler itself emits this for a
// 1-byte-aligned 16-byte copy).  The caller guarantees the 16 bytes are readable.
__device__ __forceinline__ void load16_any(const uint8_t* base, long long off, u32 out[4])
{
    uint4 v;
    __builtin_memcpy(&v, base + off, 16);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
}

__device__ __forceinline__ void reverse16(u32 v[4])                      // byte i <- byte 15 - i
{
    const u32 a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
    v[0] = __builtin_amdgcn_perm(0u, a3, 0x00010203u); v[1] = __builtin_amdgcn_perm(0u, a2, 0x00010203u);
    v[2] = __builtin_amdgcn_perm(0u, a1, 0x00010203u); v[3] = __builtin_amdgcn_perm(0u, a0, 0x00010203u);
}

typedef __attribute__((address_space(3))) u32 lds_u32;

struct K1Chunk { u32 s[4], c[4], q[4]; u32 mk; u32 off; int k; int j; int nb; u32 fl, clip, trim; int jj; u32 g0lo, g0hi; };

// LDS: dn [nrows][32][16] u32 context counts (errs << 16 | total), 16 copies (copy = lane & 15:
//          the table is tiny and hot -- measured 15 LDS cycles per wave-atomic unreplicated --
//          copies cut the bank and same-address collisions), slot = 5*code(prev)+code(cur).
//          16-bit halves: a copy can receive 64 lanes x 16*cpr bases per workgroup iteration in
//          the worst case (every base the same bin), so this table alone is flushed every
//          dn_flush_iters = 65535 / (1024 * cpr) iterations (6 for 2x150): ~1 % of the time.
//      pos[nrows][row_bytes/4] u32 (errs << 16 | total): [0,S) first-in-pair cycle = position;
//          [S, 3S) second-in-pair, index x = position + 2*(S - len)  <->  column 2S-1-x
//      rows are ordered by the INVERTED quality byte (row = 42 - q, one v_min clamps every
//      uncounted byte onto the trash row, which is last); bytes past the end of a short read
//      (quality 0 -> trash) can index beyond the trash row's end: `slack_bytes` absorb that.
// KJ > 0 (mate-pair rows of exactly KJ chunks): the cycle table is laid out CHUNK-POSITION MAJOR -- position 16 j + k of
// a row lives in word k * KJ + j of a table row whose stride is a multiple of 32 words.  The bank of a lane's atomic is
// then (k * KJ + j) mod 32 whatever its quality: for the 16-base unroll's fixed k, lanes of one row hit DIFFERENT banks
// (only the lanes of the next row, 19 chunks on, meet them again: 2 LDS cycles per half-wave), where the
// position-major layout lets the random quality row pick the bank (a 32-lane half-wave on 32 banks: ~3.5 cycles).
// Measured (profiles/r02_k1_lds.md): the cycle-table atomics were the larger half of K1's LDS time once the 4-bit
// sequence planes had made K1 LDS-bound.  The per-base offset 4 * k * KJ is an instruction immediate.
// The kernel's body: workgroup `bx` of the `gx` that share this batch (k1v3_accumulate: the whole grid; k1v3_bands: the
// workgroups one length band was given), read-group slice `g`.
// ALN (K6 fused into K1, gatk/bqsr.py:52-123 in one pass over the aligned reads instead of K6 writing canonical reads for K1 to
// read back): the lane's chunk is 16 bytes of the read AS ALIGNED -- for a reverse-strand read the chunks are taken last to
// first and byte-reversed in registers, bases complemented on their codes (A0 T1 G2 C3: code ^ 1), so that lane order and
// byte order are sequencing order for either strand and everything below the prologue is the FASTQ kernel's: canonical
// position c = i - qs (forward) or qe - 1 - i (reverse) is the cycle, read 2 takes the mirrored columns, the previous base
// in sequencing order gives the context.  A base is uncounted (quality byte 0 -> trash row) outside the aligned part
// [qs, qe), inside the trimmed range, where K4 set the skip flag or where it is N (bqsr.py:86-88); bases outside the
// aligned part are no context either (code 4).  The error flag is K4's bit 0.  A forward read with a letter outside ACGTN
// is reported (ST_LUT): the caller repeats the tally through K6's character planes, where the reference's TypeError is decided.
template <bool SPLIT, int DN, bool NIB, int KJ, bool ALN = false, bool REF = false>
__device__ __forceinline__ void k1v3_body(const K1v3Params& p, u32* lds, const int bx, const int gx, const int g)
{
    static_assert(!ALN || (!NIB && KJ == 0), "the aligned-read form reads character planes");
    static_assert(!REF || ALN, "the reference comparison belongs to the aligned-read form");
    const int ntrash = p.ntrash;                       // (the host gives the 19-chunk form one trash row: several were measured, nothing)
    const int dn_words = (p.nrows - 1 + ntrash) * 32 * DN;
    const int ncopies = KJ > 0 ? 1 : p.pos_copies;
    const int pos_words = KJ > 0 ? (p.nrows - 1 + ntrash) * (int)(p.row_bytes >> 2) + (int)(p.slack_bytes >> 2) : ncopies * (int)(p.pos_copy_bytes >> 2);
    u32* dnt = lds;
    u32* pos = lds + dn_words;
    for (int i = threadIdx.x; i < dn_words + pos_words; i += blockDim.x) lds[i] = 0u;
    __syncthreads();

    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));   // wave-uniform: block bases live in SGPRs
    const int nwaves = blockDim.x >> 6;
    // rows grouped by read group: this slice's rows are contiguous (every lane useful, no compaction)
    const long long seg_lo = p.seg ? p.seg[g] : 0ll, seg_hi = p.seg ? p.seg[g + 1] : p.nreads;
    const long long nblocks = (seg_hi - seg_lo + 63) >> 6;
    const long long iters = (nblocks + nwaves - 1) / nwaves;
    const int S = p.S, S2 = 2 * p.S;
    const u32 row_bytes = p.row_bytes;
    const u32 tclamp = 255u - p.qlo_m1;                                // inverted bytes >= this are uncounted
    const u32 pos_base = (u32)dn_words * 4u - 180u * row_bytes;        // 180 = 255 - 'K': inverted byte of q = 42 is row 0
    const u32 dnt_row = 128u * DN;                             // bytes per row of the replicated totals
    // LDS byte address (the low half of the flat address is the LDS offset) of this lane's copy, less the row bias
    const u32 dnt_base = (u32)reinterpret_cast<size_t>(lds) - 180u * dnt_row + 4u * (u32)(lane_id() & (DN - 1));
    int since_flush = 0, since_dn_flush = 0;
    // work item of this lane at step 0 and the per-step advances (64 / 128 items)
    const int lane_k0 = p.cpr == 1 ? lane : (int)__umulhi((u32)lane, p.cpr_magic);
    const int lane_j0 = lane - lane_k0 * p.cpr;
    const int dk1 = 64 / p.cpr, dj1 = 64 - dk1 * p.cpr;
    const int dk2 = 128 / p.cpr, dj2 = 128 - dk2 * p.cpr;

    u64* pos_errs = p.tables;
    u64* pos_total = p.tables + (size_t)p.R * KQ * p.gS2;
    u64* dn_errs = p.tables + 2 * (size_t)p.R * KQ * p.gS2;
    u64* dn_total = dn_errs + (size_t)p.R * KQ * KND;

    auto flush_dn = [&]() {
        for (int r = wave; r < p.nrows - 1; r += nwaves) {           // the last row is trash
            const int q = KQ - 1 - r;
            if (lane < 25) {
                const int a = lane / 5, b = lane - 5 * a;
                if (a < 4 && b < 4) {
                    u32 vt = 0u, ve = 0u;
                    for (int cpy = 0; cpy < DN; ++cpy) {
                        const u32 v = dnt[(r * 32 + lane) * DN + cpy];
                        dnt[(r * 32 + lane) * DN + cpy] = 0u;
                        vt += v & 0xFFFFu; ve += v >> 16;
                    }
                    if (vt) {
                        const size_t e = ((size_t)g * KQ + q) * KND + 4 * a + b;
                        atomicAdd(&dn_total[e], (u64)vt);
                        if (ve) atomicAdd(&dn_errs[e], (u64)ve);
                    }
                }
            }
        }
    };
    auto flush_pos = [&]() {
        for (int r = wave; r < p.nrows - 1; r += nwaves) {
            const int q = KQ - 1 - r;
            const size_t grow = ((size_t)g * KQ + q) * p.gS2;
            u32* prow = pos + (size_t)r * (row_bytes >> 2);
            if (KJ > 0) {
                for (int x = lane; x < 16 * KJ; x += 64) {
                    const u32 v = prow[x];
                    if (v) {
                        prow[x] = 0u;
                        const int k = x / KJ, j = x - k * KJ, at = 16 * j + k;       // byte offset within the mate-pair row
                        if (at == S || at > S2) continue;                           // separator / padding: never counted
                        const int col = at < S ? at : (p.twins ? at - S - 1 : S2 - 1 - (at - S - 1));
                        atomicAdd(&pos_total[grow + col], (u64)(v & 0xFFFFu));
                        if (v >> 16) atomicAdd(&pos_errs[grow + col], (u64)(v >> 16));
                    }
                }
                continue;
            }
            for (int x = lane; x < 3 * S - p.minlen; x += 64) {
                u32 v = prow[x];
                if (ncopies > 1) {                                            // sum the copies (16-bit halves apart: they are counts)
                    u32 vt = v & 0xFFFFu, ve = v >> 16;
                    for (int cpy = 1; cpy < ncopies; ++cpy) {
                        u32* q = prow + (size_t)cpy * (p.pos_copy_bytes >> 2);
                        const u32 w = q[x];
                        if (w) { q[x] = 0u; vt += w & 0xFFFFu; ve += w >> 16; }
                    }
                    if (vt | ve) {
                        prow[x] = 0u;
                        if (p.gap && x == S) continue;
                        const int col = x < S ? x : (p.twins ? x - S - p.gap : S2 - 1 - (x - S - p.gap));
                        atomicAdd(&pos_total[grow + col], (u64)vt);
                        if (ve) atomicAdd(&pos_errs[grow + col], (u64)ve);
                    }
                    continue;
                }
                if (v) {
                    prow[x] = 0u;
                    if (p.gap && x == S) continue;                            // the separator byte of a mate-pair row is never counted
                    const int col = x < S ? x : (p.twins ? x - S - p.gap : S2 - 1 - (x - S - p.gap));
                    atomicAdd(&pos_total[grow + col], (u64)(v & 0xFFFFu));
                    if (v >> 16) atomicAdd(&pos_errs[grow + col], (u64)(v >> 16));
                }
            }
        }
    };

    // (every XCD's workgroups walking a contiguous eighth of the rows instead of every eighth iteration: measured, four alternating
    //  rounds on the headline layout, 3.40-3.44 against 3.36-3.43 ms -- nothing; K2's short-lived tiles do gain from it, kbbq_k2_tile.h)
    for (long long it = bx; it < iters; it += gx) {
        const long long blk = it * nwaves + wave;
        if (blk < nblocks) {
            const long long read0 = seg_lo + (blk << 6);
            const long long myread = read0 + lane;
            u32 m, afl = 0u, aclip = 0u, atrim = 0u, ag0lo = 0u, ag0hi = 0u;
            if constexpr (ALN) {
                afl = myread < seg_hi ? p.aflags[myread] : 0u;
                aclip = myread < seg_hi ? p.aclip[myread] : 0u;
                atrim = myread < seg_hi ? p.atrim[myread] : 0u;
                if constexpr (REF) {
                    const long long g0 = myread < seg_hi ? p.ag0[myread] : 0ll;
                    ag0lo = (u32)g0; ag0hi = (u32)((u64)g0 >> 32);
                }
                m = myread < seg_hi ? ((u32)p.S | ((afl >> 16) << 16) | ((afl & 2u) << 30)) : 0u;     // every read has the common length S
            } else {
                m = myread < seg_hi ? p.meta[myread] : 0u;
            }
            const bool match = myread < seg_hi && (int)((m >> 16) & 0x7FFFu) == g && (m & 0xFFFFu) != 0u;
            const u64 mask = __ballot(match);
            const int n = __popcll(mask);
            u32 cm = m, coff = (u32)lane;
            if (n != 64 && n > 0) {            // compact the matching reads to lanes 0..n-1
                const int rank = __popcll(mask & ((1ull << lane) - 1ull));
                const int dst = match ? rank : 63;
                cm = (u32)__builtin_amdgcn_ds_permute(dst << 2, (int)(match ? m : 0u));
                coff = (u32)__builtin_amdgcn_ds_permute(dst << 2, (int)(match ? (u32)lane : 0u));
                if constexpr (ALN) {
                    afl = (u32)__builtin_amdgcn_ds_permute(dst << 2, (int)(match ? afl : 0u));
                    aclip = (u32)__builtin_amdgcn_ds_permute(dst << 2, (int)(match ? aclip : 0u));
                    atrim = (u32)__builtin_amdgcn_ds_permute(dst << 2, (int)(match ? atrim : 0u));
                    if constexpr (REF) {
                        ag0lo = (u32)__builtin_amdgcn_ds_permute(dst << 2, (int)(match ? ag0lo : 0u));
                        ag0hi = (u32)__builtin_amdgcn_ds_permute(dst << 2, (int)(match ? ag0hi : 0u));
                    }
                }
            }
            const uint8_t* bseq = p.seq + (size_t)read0 * (NIB ? p.pitch >> 1 : p.pitch);
            const uint8_t* bcseq = p.cseq + (size_t)read0 * (NIB ? p.pitch >> 1 : p.pitch);
            const uint8_t* bqual = p.qual + (size_t)read0 * p.pitch;
            const int total = n * p.cpr;
            u32 carry_code = 20u, carry_char = 0u;      // 5 * code of the previous chunk's last base

            // fetch: issue the three 16-byte loads of one step (no wait); process: bin them.
            // The loop below keeps one step in flight while the previous one is binned.
            // (k, j) = (read slot, chunk) of the lane's work item; advanced incrementally.
            auto advance = [&](K1Chunk& ch, int dk, int dj) {
                const int jj = ch.j + dj;
                const bool wrap = jj >= p.cpr;
                ch.j = wrap ? jj - p.cpr : jj;
                ch.k = ch.k + dk + (wrap ? 1 : 0);
            };
            auto fetch = [&](K1Chunk& ch) {
                const bool act0 = ch.k < n;
                const int k = act0 ? ch.k : 0;
                ch.mk = bperm(cm, k);
                ch.off = n == 64 ? (u32)k : bperm(coff, k);
                ch.nb = act0 ? ((int)(ch.mk & 0xFFFFu) - 16 * ch.j) : 0;
                ch.jj = ch.j;
                if constexpr (ALN) {
                    ch.fl = bperm(afl, k); ch.clip = bperm(aclip, k); ch.trim = bperm(atrim, k);
                    if (ch.fl & 1u) ch.jj = p.cpr - 1 - ch.j;          // a reverse-strand read: its chunks last to first
                    if constexpr (REF) { ch.g0lo = bperm(ag0lo, k); ch.g0hi = bperm(ag0hi, k); }
                }
                // lanes without work re-read the block's first chunk (valid memory, result unused)
#ifndef KBBQ_ABL_NOLOAD
                const u32 rowoff = ch.nb > 0 ? __umul24(ch.off, (u32)p.pitch) + (u32)(16 * ch.jj) : 0u;
#else
                const u32 rowoff = (u32)lane * 16u;      // timing only: every step re-reads the same cached KiB
#endif
                if constexpr (NIB) {
                    const uint2 sv = *reinterpret_cast<const uint2*>(bseq + (rowoff >> 1));
                    const uint2 cv = *reinterpret_cast<const uint2*>(bcseq + (rowoff >> 1));
                    ch.s[0] = sv.x; ch.s[1] = sv.y; ch.c[0] = cv.x; ch.c[1] = cv.y;
                } else if constexpr (REF) {
                    // a one-operation read: the chunk's reference window (any byte offset) instead of its row of K4's flags --
                    // ONE unconditional 16-byte load from whichever address applies (a load under a branch would be waited for
                    // at the branch's end)
                    const uint4 sv = *reinterpret_cast<const uint4*>(bseq + rowoff);
                    const bool inref = ch.nb > 0 && (ch.fl & 4u) != 0u;
                    const long long g0 = (long long)(((u64)ch.g0hi << 32) | (u64)ch.g0lo);
                    const uint8_t* src = inref ? p.genome + g0 + 16 * ch.jj : bcseq + rowoff;
                    load16_any(src, 0, ch.c);
                    ch.s[0] = sv.x; ch.s[1] = sv.y; ch.s[2] = sv.z; ch.s[3] = sv.w;
                } else {
                    const uint4 sv = *reinterpret_cast<const uint4*>(bseq + rowoff);
                    const uint4 cv = *reinterpret_cast<const uint4*>(bcseq + rowoff);
                    ch.s[0] = sv.x; ch.s[1] = sv.y; ch.s[2] = sv.z; ch.s[3] = sv.w;
                    ch.c[0] = cv.x; ch.c[1] = cv.y; ch.c[2] = cv.z; ch.c[3] = cv.w;
                }
#ifdef KBBQ_Q6_PROBE
                uint4 qv;                                            // TIMING ONLY (wrong results): see kbbq_k2_tile.h
                {
                    const uint3 q3 = *reinterpret_cast<const uint3*>(bqual + ((rowoff >> 2) * 3u));
                    qv.x = (q3.x & 0x1F1F1F1Fu) + 0x27272727u; qv.y = (q3.y & 0x1F1F1F1Fu) + 0x27272727u; qv.z = (q3.z & 0x1F1F1F1Fu) + 0x27272727u;
                    qv.w = (((q3.x >> 6) & 0x03030303u) | ((q3.y >> 4) & 0x0C0C0C0Cu) | ((q3.z >> 2) & 0x10101010u)) + 0x27272727u;
                }
#else
                const uint4 qv = *reinterpret_cast<const uint4*>(bqual + rowoff);
#endif
                ch.q[0] = qv.x; ch.q[1] = qv.y; ch.q[2] = qv.z; ch.q[3] = qv.w;
            };
            auto process = [&](const K1Chunk& ch) {
                const int j = ch.j, nb = ch.nb;
                const bool act = nb > 0;
                const int len = (int)(ch.mk & 0xFFFFu);
                const bool second = (ch.mk >> 31) != 0u;
                const int pos0 = 16 * j;
                // byte-parallel decode; alphabet and q-range screening (no byte masks: bytes past
                // the read are 'N' in seq/cseq and 0 in qual by the layout contract; anything else
                // only costs a visit to the exact checker)
                u32 code[4], code5[4], xw[4], qv[4], badbits = 0u, hiq = 0u;
                int c0 = pos0;                       // canonical position (= cycle) of the chunk's byte 0
                if constexpr (ALN) {
                    // the prologue of the aligned-read form (see above): from 16 bytes of the read as aligned to codes, error
                    // flags and qualities in sequencing order
                    const bool rev = (ch.fl & 1u) != 0u;
                    const int qs = (int)(ch.clip & 0xFFFFu), qe = (int)(ch.clip >> 16);
                    const int tlo = (int)(ch.trim & 0xFFFFu), thi = (int)(ch.trim >> 16);
                    const int i0 = 16 * ch.jj;                                       // position in the read of the loaded byte 0
                    auto bits = [](int lo, int hi) -> u32 {                          // bit b set for lo <= b < hi, 0 <= b < 16
                        lo = lo < 0 ? 0 : (lo > 16 ? 16 : lo); hi = hi < 0 ? 0 : (hi > 16 ? 16 : hi);
                        return hi > lo ? ((1u << hi) - 1u) & ~((1u << lo) - 1u) : 0u;
                    };
                    u32 aligned = bits(qs - i0, qe - i0);                            // inside the aligned part
                    u32 counted = aligned & ~bits(tlo - i0, thi - i0);               // ... and not trimmed away (adaptor)
                    u32 sv[4] = {ch.s[0], ch.s[1], ch.s[2], ch.s[3]}, fv[4] = {ch.c[0], ch.c[1], ch.c[2], ch.c[3]};
                    u32 ov[4] = {ch.q[0], ch.q[1], ch.q[2], ch.q[3]};
                    if constexpr (REF) {
                        // compare_reads.py:109-114 for a chunk inside the read's one M / = / X operation, as k4v2_find_errors' plain
                        // path makes them: error = read byte != reference byte, skip = the site's flag (bit 7 of the reference byte)
                        if (ch.fl & 4u) {
#pragma unroll
                            for (int wd = 0; wd < 4; ++wd)
                                fv[wd] = nonzero_bytes(sv[wd] ^ (fv[wd] & 0x7F7F7F7Fu)) | ((fv[wd] >> 6) & 0x02020202u);
                        }
                    }
                    if (rev) {
                        reverse16(sv); reverse16(fv); reverse16(ov);
                        aligned = __brev(aligned) >> 16; counted = __brev(counted) >> 16;
                    }
                    c0 = rev ? qe - i0 - 16 : i0 - qs;
                    const u32 rev1 = rev ? 0x01010101u : 0u;
#pragma unroll
                    for (int wd = 0; wd < 4; ++wd) {
                        const u32 am = (((aligned >> (4 * wd)) & 0xFu) * 0x00204081u & 0x01010101u) * 0xFFu;    // 4 bits -> 4 byte masks
                        const u32 cm_ = (((counted >> (4 * wd)) & 0xFu) * 0x00204081u & 0x01010101u) * 0xFFu;
                        u32 cd, cd5, expect;
                        decode4x(sv[wd], cd, cd5, expect);
                        const u32 odd = nonzero_bytes(expect ^ sv[wd]) * 0xFFu;      // letters that are none of ACGTN
                        badbits |= rev ? 0u : (odd & am);                            // a forward read keeps its letters: all must be ACGTN
                        cd = (cd & ~odd) | (0x04040404u & odd);                      // reverse strand: complement.get(x, 'N') (bqsr.py:43-45)
                        const u32 not_n = nonzero_bytes(sv[wd] ^ 0x4E4E4E4Eu);       // bqsr.py:86-88: an N is skipped
                        const u32 skipped = ((fv[wd] >> 1) & 0x01010101u) | (not_n ^ 0x01010101u);
                        cd ^= (~cd >> 2) & rev1;                                     // complement on codes (A0 T1 G2 C3; 4 stays 4)
                        cd = (cd & am) | (0x04040404u & ~am);                        // outside the aligned part: no base, no context
                        code[wd] = cd; code5[wd] = (cd << 2) + cd;
                        xw[wd] = fv[wd] & 0x01010101u;                               // K4's error flag
                        qv[wd] = ov[wd] & cm_ & ~(skipped * 0xFFu);
                    }
                } else {
                    qv[0] = ch.q[0]; qv[1] = ch.q[1]; qv[2] = ch.q[2]; qv[3] = ch.q[3];
                }
                if constexpr (NIB) {
                    code[0] = nib_lo(ch.s[0]); code[1] = nib_hi(ch.s[0]); code[2] = nib_lo(ch.s[1]); code[3] = nib_hi(ch.s[1]);
                    const u32 x0 = ch.s[0] ^ ch.c[0], x1 = ch.s[1] ^ ch.c[1];        // recalibrate.py:13-20 on codes
                    xw[0] = nib_lo(x0); xw[1] = nib_hi(x0); xw[2] = nib_lo(x1); xw[3] = nib_hi(x1);
                    badbits = nib_invalid(ch.s[0]) | nib_invalid(ch.s[1]);          // not a plane k7_lay_out wrote
                }
#pragma unroll
                for (int wd = 0; wd < 4; ++wd) {
                    if constexpr (NIB) {
                        code5[wd] = (code[wd] << 2) + code[wd];
                    } else if constexpr (!ALN) {
                        u32 expect;
                        decode4x(ch.s[wd], code[wd], code5[wd], expect);
                        badbits |= expect ^ ch.s[wd];
                        xw[wd] = ch.s[wd] ^ ch.c[wd];
                    }
                    hiq |= (qv[wd] + 0x34343434u) | qv[wd];                          // bit 7 of a byte: q > 42
                }
                hiq &= 0x80808080u;
                const u32 last_code5 = code5[3] >> 24;
                u32 prev_code5 = wave_shr1(last_code5, carry_code);
                carry_code = (u32)__builtin_amdgcn_readlane((int)last_code5, 63);
                u32 prev_char = 0u;
                if constexpr (!NIB && !ALN) {
                    const u32 last_char = ch.s[3] >> 24;
                    prev_char = wave_shr1(last_char, carry_char);
                    carry_char = (u32)__builtin_amdgcn_readlane((int)last_char, 63);
                }
                if (j == 0) { prev_code5 = 20u; prev_char = 0u; }                    // dinuc[0] = -1
                if (act) {
                    const long long read = read0 + ch.off;
                    const bool fits_tables = (u32)(len - p.minlen) <= (u32)(p.maxlen - p.minlen);
                    if (hiq || !fits_tables) flag(p.status, ST_INDEX, read);         // recalibrate.py:114-115; read longer (shorter) than the tables
                    if constexpr (NIB || ALN) {
                        if (badbits) flag(p.status, ST_LUT, 0);                      // the caller must use byte planes (ALN: K6's character planes)
                    } else {
                        if (badbits && chunk_type_error(ch.s[0], ch.s[1], ch.s[2], ch.s[3], ch.q[0], ch.q[1], ch.q[2], ch.q[3],
                                                        prev_char, nb, pos0, p.type_minscore))
                            flag(p.status, ST_TYPE, read);                           // compare_reads.py:224,292
                    }
                    if (!hiq && fits_tables && !((NIB || ALN) && badbits)) {
                        // A: pos address less the row term; both mates ascend with the base index
                        const u32 half = second ? (u32)(S + 2 * (S - len)) : 0u;     // column 2*len-1-pos
                        const u32 copy_off = KJ > 0 ? 0u : (u32)(ch.k & (ncopies - 1)) * p.pos_copy_bytes;     // this chunk's copy of the cycle table
                        const u32 tcl = tclamp + (u32)(ch.k & (ntrash - 1));                                   // ... and its trash row
                        const u32 A = KJ > 0 ? pos_base + 4u * (u32)j : pos_base + copy_off + (half + (u32)pos0) * 4u;
                        // ALN: bases before the aligned part have NEGATIVE canonical positions.  They are uncounted, i.e. they land on
                        // the trash row (the last one) -- up to 15 words before its start when the chunk holds the first aligned
                        // base: the tail of the row before it, which this form never uses and never flushes (every read has length S:
                        // columns < 2S + 16 of 3S words; the host sets minlen = S, the flush stops at 2S) as long as S >= 32 (the host
                        // sends shorter reads through K6); a chunk wholly outside the aligned part is sent to column 0.
                        int colA = ((int)half + c0) * 4;
                        if constexpr (ALN) { if (colA < -60) colA = 0; }
                        u32 pc5 = prev_code5 << 24;
#pragma unroll
                        for (int wd = 0; wd < 4; ++wd) {
                            const u32 pw5 = __builtin_amdgcn_alignbyte(code5[wd], pc5, 3);
                            const u32 d5 = pw5 + code[wd];                            // 5*prev + cur per byte, <= 24
                            pc5 = code5[wd];
                            const u32 xwd = xw[wd];
                            const u32 qn = ~qv[wd];
                            const u32 zero = 0u, both = 0x10001u;
                            auto one_base = [&](auto bsel) {
                                constexpr int b = decltype(bsel)::value;
                                const u32 qi = (qn >> (8 * b)) & 0xFFu;                 // 255 - quality byte
                                const u32 tq = qi < tcl ? qi : tcl;                    // below minscore (and padding): a trash row
#ifndef KBBQ_K1_PLAIN_INC
                                const u32 inc = k1_increment<b>(xwd, zero, both);      // recalibrate.py:13-20: errs << 16 | total
#else
                                const u32 inc = ((xwd >> (8 * b)) & 0xFFu) != 0u ? 0x10001u : 1u;
#endif
                                u32 a;
                                if constexpr (ALN) {
                                    const int co = colA + 4 * (4 * wd + b);
                                    a = __umul24(tq, row_bytes) + pos_base + copy_off + (u32)co;
                                } else {
                                    a = __umul24(tq, row_bytes) + A + (u32)(KJ > 0 ? 4 * KJ * (4 * wd + b) : 4 * (4 * wd + b));
                                }
#ifndef KBBQ_ABL_NOPOS
                                atomicAdd(reinterpret_cast<u32*>(reinterpret_cast<char*>(lds) + a), inc);   // recalibrate.py:116-117
#else
                                asm volatile("" :: "v"(a), "v"(inc));
#endif
                                u32 slot = (d5 >> (8 * b)) & 0xFFu;
                                if (SPLIT) slot = qi <= 255u - p.dlo ? slot : 24u;             // context needs q >= its own threshold
                                const u32 trow = tq * dnt_row + dnt_base;                      // constant powers of two: two shift-adds,
                                const u32 ad = slot * (4u * DN) + trow;                // the LDS base is inside dnt_base
#ifndef KBBQ_ABL_NODN
                                __hip_atomic_fetch_add(reinterpret_cast<lds_u32*>(ad), inc, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_WORKGROUP);          // recalibrate.py:118-119
#else
                                asm volatile("" :: "v"(ad), "v"(inc));
#endif
                            };
                            one_base(std::integral_constant<int, 0>{}); one_base(std::integral_constant<int, 1>{});
                            one_base(std::integral_constant<int, 2>{}); one_base(std::integral_constant<int, 3>{});
                        }
                    }
                }
            };

            if (total > 0) {
                // one step in flight while the previous one is binned; fetches are unconditional
                // (past the end they re-read the block's first chunk) so that the outstanding
                // loads can be counted (s_waitcnt vmcnt(3)) instead of drained
                K1Chunk ca, cb;
                ca.k = lane_k0; ca.j = lane_j0;
                cb.k = lane_k0; cb.j = lane_j0; advance(cb, dk1, dj1);
                fetch(ca);
                for (int w0 = 0; w0 < total; w0 += 128) {
                    fetch(cb);
                    process(ca);
                    if (w0 + 64 < total) {
                        advance(ca, dk2, dj2);
                        fetch(ca);
                        process(cb);
                        advance(cb, dk2, dj2);
                    }
                }
            }
        }
        ++since_flush; ++since_dn_flush;
        if (since_flush == K1V3_FLUSH_ITERS || since_dn_flush == p.dn_flush_iters) {
            __syncthreads();
            flush_dn(); since_dn_flush = 0;
            if (since_flush == K1V3_FLUSH_ITERS) { flush_pos(); since_flush = 0; }
            __syncthreads();
        }
    }
    __syncthreads();
    flush_dn(); flush_pos();
}

template <bool SPLIT, int DN = K1V3_DNREP, bool NIB = false, int KJ = 0>
__global__ __launch_bounds__(K1V3_THREADS) void k1v3_accumulate(K1v3Params p)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    k1v3_body<SPLIT, DN, NIB, KJ>(p, lds, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y);
}

// the aligned-read form (K6 fused into K1): the BAM-sourced tally of gatk/bqsr.py:52-123 straight from the reads as aligned
template <bool SPLIT, int DN>
__global__ __launch_bounds__(K1V3_THREADS) void k1v3_aligned(K1v3Params p)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    k1v3_body<SPLIT, DN, false, 0, true>(p, lds, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y);
}

// ... and K4 folded in for reads of one M / = / X operation (kbbq_tally_aligned_dev): read bytes, OQ and the reference window
template <bool SPLIT, int DN>
__global__ __launch_bounds__(K1V3_THREADS) void k1v3_aligned_ref(K1v3Params p)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    k1v3_body<SPLIT, DN, false, 0, true, true>(p, lds, (int)blockIdx.x, (int)gridDim.x, (int)blockIdx.y);
}

// ONE launch over all length bands of a mixed-length input (BASELINE config 5; recalibrate.py:81-101 grows its arrays as the
// reads get longer, the file path cuts the reads into bands of one row width each: kbbq/fastx.py BAND_CLASSES).  A launch
// per band is 9-10 iterations per workgroup for the narrow bands -- pipeline fill, table zeroing and the final flush of a
// 100 KB table paid eight times per step; here every band gets its share of the CUs' workgroups (wg_start, by the host:
// proportional to the band's chunks plus a per-row term) and they all run at once, each workgroup with ITS band's
// parameters -- row pitch, LDS table geometry for the band's longest / shortest read, copies of the context table.
#define K1V3_MAX_BANDS 16
struct K1BandsParams {
    int nbands;
    int wg_start[K1V3_MAX_BANDS + 1];     // band b owns workgroups [wg_start[b], wg_start[b + 1]) of blockIdx.x
    int dn[K1V3_MAX_BANDS];               // copies of the context table: K1V3_DNREP or 8
    unsigned long long* dbg;              // diagnostic (KBBQ_K1_BANDS_DBG): [2 * workgroups] start / end of every workgroup (s_memrealtime, 100 MHz)
    K1v3Params band[K1V3_MAX_BANDS];
};

template <bool SPLIT, bool NIB>
__global__ __launch_bounds__(K1V3_THREADS) void k1v3_bands(K1BandsParams t)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    int b = 0;
    while (b + 1 < t.nbands && (int)blockIdx.x >= t.wg_start[b + 1]) ++b;
    const K1v3Params& p = t.band[b];
    const int bx = (int)blockIdx.x - t.wg_start[b], gx = t.wg_start[b + 1] - t.wg_start[b];
    if (t.dbg && threadIdx.x == 0 && blockIdx.y == 0) t.dbg[blockIdx.x] = wall_clock64();
    if (t.dn[b] == K1V3_DNREP) k1v3_body<SPLIT, K1V3_DNREP, NIB, 0>(p, lds, bx, gx, (int)blockIdx.y);
    else k1v3_body<SPLIT, 8, NIB, 0>(p, lds, bx, gx, (int)blockIdx.y);
    if (t.dbg && threadIdx.x == 0 && blockIdx.y == 0) t.dbg[gridDim.x + blockIdx.x] = wall_clock64();
}

// ---------------------------------------------------------------- K2 (table-driven)
// "Full" LUT, int8, one row per (read group, RAW quality byte qb in [0, 33+Qt)):
//     row[0 .. W)         cycle entry for first-in-pair reads, index = position   (W = S2 + 16:
//     row[W .. 2W)        the same entries mirrored: W + position <-> column S2-1-position
//     row[2W .. +32)      context entry, index 5*code(prev)+code(cur) (25 used)
// the 16 extra entries per region keep the padding positions of a 16-byte chunk in-region)
// qb = 0 (padding): cycle entries -33, contexts 0  -> output byte 0
// qb below 33 + minscore: identity rows (cycle entry = qb - 33, contexts 0) -> byte unchanged
// otherwise the model rows.  new byte = cycle entry + context entry + 33.
__host__ __device__ __forceinline__ int full_lut_width(int S2) { return S2 + 16; }
__host__ __device__ __forceinline__ int full_lut_row_bytes(int S2)
{
    int rb = (2 * full_lut_width(S2) + 32 + 3) & ~3;
    if (((rb >> 2) & 1) == 0) rb += 4;        // odd number of dwords per row: rows start on different banks
    return rb;
}

struct K2Chunk { u32 s[4], q[4]; u32 mk; u32 dlo, dhi; int kk; int k; int j; int nb; bool act0; };

struct K2v3Params {
    const uint8_t* seq; const uint8_t* qual; const u32* meta;
    long long nreads; int pitch; int cpr; u32 cpr_magic; int R; int Qt; int S2; int minscore; u32 qlo;
    const int16_t* lut16;      // canonical LUT (exact path)
    const int8_t* full;        // full LUT (global copy, staged into LDS)
    int full_bytes; int rs16;
    u32 rb;                    // bytes per row of the staged LUT
    u32 W;                     // offset of the second-in-pair (mirrored) cycle entries within a row
    u32 ctx_off;               // offset of the context entries within a row
    int maxlen;                // longest row the LUT serves: S2 (H1: any single read), or S2 + 1 for mate-pair rows
    int pairs;                 // mate-pair rows: rows the fast path cannot serve are reported (KBBQ_E_LUT), not emulated
    const long long* seg;      // rows grouped by read group: slice blockIdx.y stages only its group's LUT rows; NULL: all groups
    int rpb;                   // rows per wave block (<= 64): a wave's contiguous footprint is rpb * pitch bytes per plane
    int parts;                 // > 1: the workgroups walk the rows as `parts` sequential fronts (workgroup b: part b % parts) instead of one: the 2 read +
                               // 1 written traversal of 50 M character rows 4.68-4.78 -> 4.44-4.48 ms with 4 ... 32 fronts (KBBQ_K2_PARTS, default 8; kbbq_k2_tile.h
                               // has the same for the short-lived kernel; K1's read-only traversal does not care)
    const long long* perm;     // rows grouped by read group: row i is stored as row perm[i] of `out` (straight back into input order); NULL: row i
    uint8_t* out; u64* status;
};

template <bool NIB = false>
__global__ __launch_bounds__(K2V3_THREADS) __attribute__((amdgpu_waves_per_eu(K2V3_WAVES, 8))) void k2v3_apply(K2v3Params p)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const u32 rb = p.rb;
    const u32 rg_bytes = (u32)(33 + p.Qt) * rb;
    const int g = blockIdx.y;                                   // grouped rows: the read group of this slice
    {
        // grouped: only the rows of this slice's read group are staged (any number of groups at the LDS cost of one)
        const u32* src = reinterpret_cast<const u32*>(p.full + (p.seg ? (size_t)g * rg_bytes : 0));
        const int nw = (int)((p.seg ? rg_bytes : (u32)p.full_bytes) >> 2);
        for (int i = threadIdx.x; i < nw; i += blockDim.x) lds[i] = src[i];
        __syncthreads();
    }
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = blockDim.x >> 6;
    const long long seg_lo = p.seg ? p.seg[g] : 0ll, seg_hi = p.seg ? p.seg[g + 1] : p.nreads;
    const long long nblocks = (seg_hi - seg_lo + p.rpb - 1) / p.rpb;
    const u32 hi_add = (u32)(0x80 - (p.Qt + 33)) * 0x01010101u;   // byte >= Qt+33 <=> bit 7 after the add
    const int lane_k0 = p.cpr == 1 ? lane : (int)__umulhi((u32)lane, p.cpr_magic);
    const int lane_j0 = lane - lane_k0 * p.cpr;
    const int dk1 = 64 / p.cpr, dj1 = 64 - dk1 * p.cpr;
    const int dkn = (64 * K2V3_NBUF) / p.cpr, djn = 64 * K2V3_NBUF - dkn * p.cpr;

    long long blk0 = (long long)blockIdx.x * nwaves + wave, blk_step = (long long)gridDim.x * nwaves, blk_end = nblocks;
    if (p.parts > 1 && (int)gridDim.x % p.parts == 0) {
        const long long per = (nblocks + p.parts - 1) / p.parts;
        const int part = (int)blockIdx.x % p.parts;
        blk0 = part * per + (long long)((int)blockIdx.x / p.parts) * nwaves + wave;
        blk_step = (long long)((int)gridDim.x / p.parts) * nwaves;
        blk_end = (part + 1) * per < nblocks ? (part + 1) * per : nblocks;
    }
    for (long long blk = blk0; blk < blk_end; blk += blk_step) {
        const long long read0 = seg_lo + blk * p.rpb;
        const long long myread = read0 + lane;
        const int n = (int)((seg_hi - read0) < p.rpb ? (seg_hi - read0) : p.rpb);
        const u32 m = lane < n ? p.meta[myread] : 0u;
        const long long pm = (p.perm && lane < n) ? p.perm[myread] : 0ll;
        const int total = n * p.cpr;
        const uint8_t* bseq = p.seq + (size_t)read0 * (NIB ? p.pitch >> 1 : p.pitch);
        const uint8_t* bqual = p.qual + (size_t)read0 * p.pitch;
        uint8_t* bout = p.out + (size_t)read0 * p.pitch;
        u32 carry_code = 20u, carry_char = 0u;

        auto advance = [&](K2Chunk& ch, int dk, int dj) {
            const int jj = ch.j + dj;
            const bool wrap = jj >= p.cpr;
            ch.j = wrap ? jj - p.cpr : jj;
            ch.kk = ch.kk + dk + (wrap ? 1 : 0);
        };
        auto fetch = [&](K2Chunk& ch) {
            ch.act0 = ch.kk < n;
            ch.k = ch.act0 ? ch.kk : 0;
            ch.mk = bperm(m, ch.k);
            if (p.perm) { ch.dlo = bperm((u32)pm, ch.k); ch.dhi = bperm((u32)(pm >> 32), ch.k); }
            ch.nb = ch.act0 ? ((int)(ch.mk & 0xFFFFu) - 16 * ch.j) : 0;
#ifndef KBBQ_ABL_NOLOAD
            const u32 rowoff = ch.nb > 0 ? __umul24((u32)ch.k, (u32)p.pitch) + (u32)(16 * ch.j) : 0u;
#else
            const u32 rowoff = (u32)lane * 16u;
#endif
            if constexpr (NIB) {
                const uint2 sv = *reinterpret_cast<const uint2*>(bseq + (rowoff >> 1));
                ch.s[0] = sv.x; ch.s[1] = sv.y;
            } else {
                const uint4 sv = *reinterpret_cast<const uint4*>(bseq + rowoff);
                ch.s[0] = sv.x; ch.s[1] = sv.y; ch.s[2] = sv.z; ch.s[3] = sv.w;
            }
            const uint4 qv = *reinterpret_cast<const uint4*>(bqual + rowoff);
            ch.q[0] = qv.x; ch.q[1] = qv.y; ch.q[2] = qv.z; ch.q[3] = qv.w;
        };
        auto process = [&](const K2Chunk& ch) {
            const int j = ch.j, nb = ch.nb, k = ch.k;
            const bool act = nb > 0;
            const int len = (int)(ch.mk & 0xFFFFu);
            const int rg = (int)((ch.mk >> 16) & 0x7FFFu);
            const bool second = (ch.mk >> 31) != 0u;
            const int pos0 = 16 * j;
            u32 code[4], code5[4], badbits = 0u, hiq = 0u;
            if constexpr (NIB) {
                code[0] = nib_lo(ch.s[0]); code[1] = nib_hi(ch.s[0]); code[2] = nib_lo(ch.s[1]); code[3] = nib_hi(ch.s[1]);
                badbits = nib_invalid(ch.s[0]) | nib_invalid(ch.s[1]);
            }
#pragma unroll
            for (int wd = 0; wd < 4; ++wd) {
                if constexpr (NIB) {
                    code5[wd] = (code[wd] << 2) + code[wd];
                } else {
                    u32 expect;
                    decode4x(ch.s[wd], code[wd], code5[wd], expect);
                    badbits |= expect ^ ch.s[wd];
                }
                hiq |= ((ch.q[wd] & 0x7F7F7F7Fu) + hi_add) | ch.q[wd];
            }
            hiq &= 0x80808080u;
            const u32 last_code5 = code5[3] >> 24;
            u32 prev_code5 = wave_shr1(last_code5, carry_code);
            carry_code = (u32)__builtin_amdgcn_readlane((int)last_code5, 63);
            u32 prev_char = 0u;
            if constexpr (!NIB) {
                const u32 last_char = ch.s[3] >> 24;
                prev_char = wave_shr1(last_char, carry_char);
                carry_char = (u32)__builtin_amdgcn_readlane((int)last_char, 63);
            }
            if (j == 0) { prev_code5 = 20u; prev_char = 0u; }
#ifdef KBBQ_ABL_COPY
            if (ch.act0) {      // timing-only build: the kernel's memory traffic with (almost) no work
                *reinterpret_cast<uint4*>(bout + (__umul24((u32)k, (u32)p.pitch) + (u32)pos0)) =
                    make_uint4(ch.q[0] ^ ch.s[0], ch.q[1] ^ ch.s[1], ch.q[2] ^ ch.s[NIB ? 0 : 2], ch.q[3] ^ ch.s[NIB ? 1 : 3]);
                return;
            }
#endif
            if (ch.act0) {
                u32 o[4] = {0u, 0u, 0u, 0u};
                if (act) {
                    const long long read = read0 + k;
                    if constexpr (NIB) {
                        if (badbits) flag(p.status, ST_LUT, 0);                      // not a plane k7_lay_out wrote
                    } else {
                        if (badbits && chunk_type_error(ch.s[0], ch.s[1], ch.s[2], ch.s[3], ch.q[0], ch.q[1], ch.q[2], ch.q[3],
                                                        prev_char, nb, pos0, p.minscore))
                            flag(p.status, ST_TYPE, read);
                    }
                    u32 d5[4];
                    u32 pc5 = prev_code5 << 24;
#pragma unroll
                    for (int wd = 0; wd < 4; ++wd) {
                        d5[wd] = __builtin_amdgcn_alignbyte(code5[wd], pc5, 3) + code[wd];   // 5*prev + cur per byte
                        pc5 = code5[wd];
                    }
                    const bool trouble = hiq != 0u || rg >= p.R || len > p.maxlen || (p.seg && rg != g) || (NIB && badbits);
                    if (trouble && (p.pairs || (p.seg && rg != g) || (NIB && badbits))) {
                        flag(p.status, ST_LUT, 0);                       // the caller re-runs on one-read-per-row planes
                    } else if (trouble) {
                        const uint4 e = chunk_apply_exact(p.lut16, p.rs16, p.R, p.Qt, p.S2, p.qlo, rg, second, pos0, nb,
                                                          ch.q[0], ch.q[1], ch.q[2], ch.q[3], d5[0], d5[1], d5[2], d5[3],
                                                          p.status, read);
                        o[0] = e.x; o[1] = e.y; o[2] = e.z; o[3] = e.w;
                    } else {
                        const u32 rgb = p.seg ? 0u : (u32)rg * rg_bytes;
                        const u32 A = rgb + (second ? p.W : 0u) + (u32)pos0;
                        const u32 C = rgb + p.ctx_off;
#pragma unroll
                        for (int wd = 0; wd < 4; ++wd) {
                            // issue the word's eight LDS reads before combining any of them
                            int v1[4], v2[4];
#pragma unroll
                            for (int b = 0; b < 4; ++b) {
                                const u32 qb = (ch.q[wd] >> (8 * b)) & 0xFFu;
                                const u32 rowq = __umul24(qb, rb);
                                const u32 dd = (d5[wd] >> (8 * b)) & 0xFFu;
#ifndef KBBQ_ABL_NOLUT
                                v1[b] = *reinterpret_cast<const int8_t*>(reinterpret_cast<const char*>(lds) + (rowq + A + (u32)(4 * wd + b)));
                                v2[b] = *reinterpret_cast<const int8_t*>(reinterpret_cast<const char*>(lds) + (rowq + C + dd));
#else
                                v1[b] = (int)(rowq + A); v2[b] = (int)(C + dd);
#endif
                            }
                            // FAST mode is only legal for range-safe LUTs (flags == 0): every sum is 0..255
#pragma unroll
                            for (int b = 0; b < 4; ++b) o[wd] |= (u32)(v1[b] + v2[b] + 33) << (8 * b);
                        }
                    }
                }
                uint8_t* dst = p.perm ? p.out + (size_t)(((u64)ch.dhi << 32) | ch.dlo) * (size_t)p.pitch + (size_t)pos0
                                      : bout + (__umul24((u32)k, (u32)p.pitch) + (u32)pos0);
                *reinterpret_cast<uint4*>(dst) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        };

        if (total > 0) {
            // ring of K2V3_NBUF chunk buffers: while one step is processed the next NBUF-1 are in flight
            K2Chunk c[K2V3_NBUF];
#pragma unroll
            for (int b = 0; b < K2V3_NBUF; ++b) {
                c[b].kk = lane_k0; c[b].j = lane_j0;
#pragma unroll
                for (int t = 0; t < b; ++t) advance(c[b], dk1, dj1);
            }
#pragma unroll
            for (int b = 0; b < K2V3_NBUF - 1; ++b) fetch(c[b]);
            for (int w0 = 0; w0 < total; w0 += 64 * K2V3_NBUF) {
#pragma unroll
                for (int b = 0; b < K2V3_NBUF; ++b) {
                    if (b == 0 || w0 + 64 * b < total) {
                        fetch(c[(b + K2V3_NBUF - 1) % K2V3_NBUF]);      // unconditional: inactive chunks load row 0
                        process(c[b]);
                        advance(c[b], dkn, djn);
                    }
                }
            }
        }
    }
}

// Builds the full int8 LUT from the canonical int16 LUT and reports whether it is usable:
// flags[0] |= 1 when some value does not fit int8, flags[0] |= 2 when some (cycle, context)
// combination of a row could leave 0..255 (then the checked kernel must be used).
struct LutFillParams {
    const int16_t* lut16; int rs16; int R; int Qt; int S2; int minscore;
    int8_t* full; int8_t* compact8; int* flags; u64* status;
};

// one workgroup per row (read group, raw quality byte): the row of the full LUT, and for a model row also
// its int8 copy of the canonical row and its range check (block-wide min / max)
__global__ __launch_bounds__(256) void k3_fill_full_lut(LutFillParams p)
{
    __shared__ int red[4][4];
    const int rb = full_lut_row_bytes(p.S2);
    const int NR = 33 + p.Qt;
    const int row = blockIdx.x;                       // r * NR + qb
    const int r = row / NR, qb = row - r * NR;
    const int W = full_lut_width(p.S2);
    const bool model = qb >= 33 + p.minscore;
    const int16_t* src = p.lut16 + ((size_t)r * p.Qt + (qb >= 33 ? qb - 33 : 0)) * p.rs16;
    int bad = 0;
    int8_t* dst = p.full + (size_t)row * rb;
    for (int x = threadIdx.x; x < rb; x += blockDim.x) {
        int v = 0;
        if (!model) {
            if (x < 2 * W) v = qb == 0 ? -33 : qb - 33;                   // padding -> 0 ; uncounted -> unchanged
        } else {
            if (x < p.S2) v = src[x];
            else if (x >= W && x < W + p.S2) v = src[p.S2 - 1 - (x - W)];   // mirrored copy for second-in-pair
            else if (x >= 2 * W && x < 2 * W + 25) v = src[p.S2 + (x - 2 * W)];
        }
        if (v < -128 || v > 127) bad |= 1;
        dst[x] = (int8_t)v;
    }
    if (qb >= 33) {
        // int8 copy of the canonical row (same row stride, one byte per entry)
        int8_t* c8 = p.compact8 + ((size_t)r * p.Qt + (qb - 33)) * p.rs16;
        for (int x = threadIdx.x; x < p.rs16; x += blockDim.x) {
            const int v = src[x];
            if (v < -128 || v > 127) bad |= 1;
            c8[x] = (int8_t)v;
        }
    }
    if (model) {
        // range safety of the row: min/max over cycles + min/max over contexts must stay in 0..255 after +33
        int lo1 = 32767, hi1 = -32768, lo2 = 32767, hi2 = -32768;
        for (int x = threadIdx.x; x < p.S2; x += blockDim.x) { const int v = src[x]; lo1 = v < lo1 ? v : lo1; hi1 = v > hi1 ? v : hi1; }
        for (int x = threadIdx.x; x < 25; x += blockDim.x) { const int v = src[p.S2 + x]; lo2 = v < lo2 ? v : lo2; hi2 = v > hi2 ? v : hi2; }
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            int o;
            o = __shfl_xor(lo1, off); lo1 = o < lo1 ? o : lo1;
            o = __shfl_xor(hi1, off); hi1 = o > hi1 ? o : hi1;
            o = __shfl_xor(lo2, off); lo2 = o < lo2 ? o : lo2;
            o = __shfl_xor(hi2, off); hi2 = o > hi2 ? o : hi2;
        }
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { red[w][0] = lo1; red[w][1] = hi1; red[w][2] = lo2; red[w][3] = hi2; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int k = 1; k < (int)(blockDim.x >> 6); ++k) {
                lo1 = red[k][0] < lo1 ? red[k][0] : lo1; hi1 = red[k][1] > hi1 ? red[k][1] : hi1;
                lo2 = red[k][2] < lo2 ? red[k][2] : lo2; hi2 = red[k][3] > hi2 ? red[k][3] : hi2;
            }
            if (lo1 + lo2 + 33 < 0 || hi1 + hi2 + 33 > 255) bad |= 2;
        }
    }
    if (bad) { atomicOr(p.flags, bad); atomicMin(&p.status[ST_LUT], 0ull); }
}

// The full LUT narrowed to the cycle columns rows of one pitch can reach (K2 on one-read-per-row planes): a read in a
// row of `pitch` bytes has at most Sb = min(pitch, S2) bases, so of a model row's S2 cycle entries only [0, Sb) (first
// in pair) and [S2 - Sb, S2) (second in pair: column S2 - 1 - pos) are ever indexed.  Same geometry as the full LUT with
// Sb in place of S2 (full_lut_width / full_lut_row_bytes): W = Sb + 16, [0, W) forward, [W, 2W) mirrored, 25 contexts at
// 2W.  A length band of a mixed-length input (tables of 600 columns, rows of 48 bytes) stages 12 KB instead of 96 KB:
// the LDS then holds several workgroups per CU again, and the short-lived K2 (kbbq_k2_tile.h) can afford the staging.
struct RowLutParams { const int16_t* lut16; int rs16; int R; int Qt; int S2; int Sb; int minscore; int8_t* out; };

__global__ __launch_bounds__(256) void k3_fill_row_lut(RowLutParams p)
{
    const int rb = full_lut_row_bytes(p.Sb), W = full_lut_width(p.Sb);
    const int NR = 33 + p.Qt;
    const int row = blockIdx.x;                       // r * NR + qb
    const int r = row / NR, qb = row - r * NR;
    const bool model = qb >= 33 + p.minscore;
    const int16_t* src = p.lut16 + ((size_t)r * p.Qt + (qb >= 33 ? qb - 33 : 0)) * p.rs16;
    int8_t* dst = p.out + (size_t)row * rb;
    for (int x = threadIdx.x; x < rb; x += blockDim.x) {
        int v = 0;
        if (!model) {
            if (x < 2 * W) v = qb == 0 ? -33 : qb - 33;                   // padding -> 0 ; uncounted -> unchanged
        } else {
            if (x < p.Sb) v = src[x];
            else if (x >= W && x < W + p.Sb) v = src[p.S2 - 1 - (x - W)];   // second in pair: column S2 - 1 - pos
            else if (x >= 2 * W && x < 2 * W + 25) v = src[p.S2 + (x - 2 * W)];
        }
        dst[x] = (int8_t)v;                            // the blob's flags said every value fits (FAST mode only)
    }
}

// ---------------------------------------------------------------- K4 / K5: benchmark path
// (SURVEY.md 8(f) #1).  K4 restates compare_reads.find_read_errors (compare_reads.py:84-139):
// a CIGAR walk that compares the read with the reference and marks sites to skip; K5 is the
// two np.bincount calls of benchmark.calculate_q (benchmark.py:76-91) over the unskipped bases.
struct K4Params {
    const uint8_t* seq; const u32* len; long long nreads; int pitch;
    const long long* ref_start;      // offset of reference_start in `genome` / `skipmask`
    const int* ref_len;              // reference_end - reference_start (the read's reference window)
    const u32* cig_off; const u32* cig_n; const u32* cigar;   // per read: first op, op count; ops = len << 4 | op
    const uint8_t* genome; const uint8_t* skipmask; const uint8_t* flip;     // skipmask == NULL: bit 7 of a genome byte is its skip flag
    long long genome_len;            // bytes in genome / skipmask
    uint8_t* err; uint8_t* skip; u64* status;
};

// the same, for windows that may run past the end of the array (`limit` = its size in bytes): bytes past
// the end read as zero.  Only the last chunk of the last rows / of the genome ever takes the byte path.
__device__ __forceinline__ void load16_upto(const uint8_t* base, long long off, long long limit, u32 out[4])
{
    if (off + 16 <= limit) { load16_any(base, off, out); return; }
    out[0] = out[1] = out[2] = out[3] = 0u;
    for (int b = 0; b < 16 && off + b < limit; ++b) out[b >> 2] |= (u32)base[off + b] << (8 * (b & 3));
}


__device__ __forceinline__ void set_byte(u32 v[4], int i, u32 val)      // i in 0..15, slow paths only
{
    const u32 m = 0xFFu << (8 * (i & 3)), x = (val & 0xFFu) << (8 * (i & 3));
    v[0] = (i >> 2) == 0 ? (v[0] & ~m) | x : v[0];
    v[1] = (i >> 2) == 1 ? (v[1] & ~m) | x : v[1];
    v[2] = (i >> 2) == 2 ? (v[2] & ~m) | x : v[2];
    v[3] = (i >> 2) == 3 ? (v[3] & ~m) | x : v[3];
}

__device__ __forceinline__ u32 get_byte(const u32 v[4], int i)
{
    const u32 w = (i >> 2) == 0 ? v[0] : (i >> 2) == 1 ? v[1] : (i >> 2) == 2 ? v[2] : v[3];
    return (w >> (8 * (i & 3))) & 0xFFu;
}

// 0xFF in the bytes of word w (of a 16-byte vector) whose position p = 4w + k lies in [lo, hi)
__device__ __forceinline__ u32 range_mask(int lo, int hi, int w) { return byte_mask(hi, w) & ~byte_mask(lo, w); }


__device__ __forceinline__ void shr_bytes16(u32 v[4], int nb)            // byte i <- byte i + nb (0 <= nb <= 15), zero fill
{
    const int ws = nb >> 2; const u32 bs = (u32)(nb & 3) * 8u;
    const u32 a0 = ws == 0 ? v[0] : ws == 1 ? v[1] : ws == 2 ? v[2] : v[3];
    const u32 a1 = ws == 0 ? v[1] : ws == 1 ? v[2] : ws == 2 ? v[3] : 0u;
    const u32 a2 = ws == 0 ? v[2] : ws == 1 ? v[3] : 0u;
    const u32 a3 = ws == 0 ? v[3] : 0u;
    v[0] = __builtin_amdgcn_alignbit(a1, a0, bs); v[1] = __builtin_amdgcn_alignbit(a2, a1, bs);
    v[2] = __builtin_amdgcn_alignbit(a3, a2, bs); v[3] = a3 >> bs;
}

// lane <-> one 16-byte OUTPUT chunk of one read.  The lane walks the read's (short) CIGAR in
// order and applies every operation's effect to its own 16 positions in that order, which is
// exactly the reference's sequential semantics (an M assigns, a later D/N ORs into the base
// before it, Python's index -1 wraps: skips[-1], subset[-1]).  A chunk that lies wholly inside
// one M/=/X operation -- the common case -- is 16 bytes of read against 16 bytes of reference
// and of the site mask (unaligned 16-byte loads), compared
// byte-parallel.  For reverse-strand reads the OUTPUT is reversed (benchmark.py:70-72): the
// lane's input positions are then [n-16j-16, n-16j) and its 16 result bytes are byte-reversed.
// Work items (read block, chunk) of a thread are software-pipelined two deep, as in K6: the per-read fields of item
// i + 2, then -- for item i + 1 -- the chunk's read bytes, the first CIGAR operation and, SPECULATIVELY, the reference
// and mask windows the chunk needs if no insertion / deletion precedes it (offset ref_start + in_lo: every chunk of a
// one-block read, and the chunks before the first indel of any other) are in flight while item i is walked.  The walk
// uses the speculative windows when an M block asks for exactly that offset and loads its own otherwise.
struct K4Item { long long rb; int j; };
struct K4Meta { int n, rl; u32 f, nc; long long g0; const u32* ops; bool valid; };
struct K4Win { u32 sw[4], gw[4], mw[4]; u32 op0; long long goff; int in_lo, cnt; bool has, spec; };

template <bool FUSED>
__global__ __launch_bounds__(256) void k4_find_errors(K4Params p)
{
    // a workgroup takes 256 / cpr whole reads per iteration: (slot, chunk) of a thread are fixed
    const int cpr = p.pitch >> 4;
    const int rpb = cpr <= 256 ? 256 / cpr : 0;
    const int slot = cpr <= 256 ? (int)threadIdx.x / cpr : 0;
    const int j0 = (int)threadIdx.x - slot * cpr;
    const long long step = rpb ? rpb : 1;
    const long long gstep = (long long)gridDim.x * step;
    const bool idle = (rpb && slot >= rpb) || j0 >= cpr;
    // fused reference: no separate mask array, bit 7 of every reference byte is the site's skip flag (one scattered
    // window per chunk instead of two: the windows of a read straddle 2-3 cache lines each, so they are fetched ~1.8x)
    constexpr bool fused = FUSED;
    auto mask_at = [&](long long i) -> u32 { return fused ? (u32)(p.genome[i] >> 7) : (u32)(p.skipmask[i] != 0); };
    auto next = [&](K4Item it) { it.j += 256; if (it.j >= cpr) { it.j = j0; it.rb += gstep; } return it; };
    auto live = [&](const K4Item& it) { return !idle && it.rb + slot < p.nreads; };
    auto fetch_meta = [&](const K4Item& it, K4Meta& m) {
        m.valid = live(it);
        const long long r = m.valid ? it.rb + slot : 0;
        m.n = (int)p.len[r]; m.rl = p.ref_len[r]; m.f = p.flip[r]; m.g0 = p.ref_start[r];
        m.nc = m.valid ? p.cig_n[r] : 0u;
        m.ops = p.cigar + p.cig_off[r];
    };
    auto fetch_win = [&](const K4Item& it, const K4Meta& m, K4Win& w) {
        w.has = false; w.spec = false; w.cnt = 0; w.in_lo = 0; w.op0 = 0u; w.goff = 0;
        if (!m.valid) return;
        const int n = m.n, out_lo = 16 * it.j, out_hi = out_lo + 16 < n ? out_lo + 16 : n;
        if (out_lo >= n) return;
        w.has = true;
        w.cnt = out_hi - out_lo;
        w.in_lo = m.f ? n - out_hi : out_lo;                       // input (unflipped) positions [in_lo, in_lo + cnt)
        load16_any(p.seq + (size_t)(it.rb + slot) * p.pitch, w.in_lo, w.sw);   // the chunk's read bytes (may run into the next row)
        if (m.nc) w.op0 = m.ops[0];
        w.goff = m.g0 + w.in_lo;
        w.spec = w.goff >= 0 && w.goff + 16 <= p.genome_len;
        if (w.spec) { load16_any(p.genome, w.goff, w.gw); if (!fused) load16_any(p.skipmask, w.goff, w.mw); }
    };
    K4Item it0{(long long)blockIdx.x * step, j0};
    K4Item it1 = next(it0), it2 = next(it1);
    K4Meta m0, m1, m2;
    K4Win w0, w1;
    fetch_meta(it0, m0); fetch_meta(it1, m1);
    fetch_win(it0, m0, w0);
    while (live(it0)) {
        fetch_meta(it2, m2);
        fetch_win(it1, m1, w1);
        {
            const long long r = it0.rb + slot;
            const int j = it0.j;
            const int n = m0.n, rl = m0.rl;
            const bool f = m0.f != 0;
            u32 ev[4] = {0u, 0u, 0u, 0u}, kv[4] = {0u, 0u, 0u, 0u};
            if (w0.has) {
                const int cnt = w0.cnt, in_lo = w0.in_lo, in_hi = in_lo + cnt;
                const uint8_t* s = p.seq + (size_t)r * p.pitch;
                const long long g0 = m0.g0;
                int readidx = 0, refidx = 0;
                const u32* ops = m0.ops;
                const u32 nc = m0.nc;
                const u32 (&sw)[4] = w0.sw;
                for (u32 c = 0; c < nc; ++c) {
                    const u32 word = c == 0 ? w0.op0 : ops[c];
                    const int op = (int)(word & 15u), l = (int)(word >> 4);
                    if (op == 0 || op == 7 || op == 8) {                       // M = X  (:109-114)
                        if (refidx + l > rl || readidx + l > n) { flag(p.status, ST_RANGE, r); break; }   // shape mismatch: ValueError
                        const int a = readidx > in_lo ? readidx : in_lo, b = readidx + l < in_hi ? readidx + l : in_hi;
                        if (a < b) {
                            // reference window aligned with the CHUNK start (as if the operation began there):
                            // bytes [a - in_lo, b - in_lo) of the comparison belong to this operation
                            const int d = a - in_lo;
                            const long long goff = g0 + refidx + (a - readidx) - d;
                            if (goff >= 0) {
                                u32 gw[4], mw[4];
                                if (w0.spec && goff == w0.goff) {
#pragma unroll
                                    for (int w = 0; w < 4; ++w) { gw[w] = w0.gw[w]; mw[w] = w0.mw[w]; }
                                } else {
                                    load16_upto(p.genome, goff, p.genome_len, gw);
                                    if (!fused) load16_upto(p.skipmask, goff, p.genome_len, mw);
                                }
                                if (fused) {
#pragma unroll
                                    for (int w = 0; w < 4; ++w) { mw[w] = gw[w] & 0x80808080u; gw[w] &= 0x7F7F7F7Fu; }
                                }
#pragma unroll
                                for (int w = 0; w < 4; ++w) {
                                    const u32 rm = range_mask(d, b - in_lo, w);
                                    ev[w] = (ev[w] & ~rm) | (nonzero_bytes(sw[w] ^ gw[w]) & rm);
                                    kv[w] = (kv[w] & ~rm) | (nonzero_bytes(mw[w]) & rm);
                                }
                            } else {                                           // within 15 bytes of the genome's first byte
                                const long long roff = g0 + refidx + (a - readidx);
                                for (int q = a; q < b; ++q) {
                                    set_byte(ev, q - in_lo, (u32)(p.genome[roff + (q - a)] & (fused ? 0x7Fu : 0xFFu)) != (u32)s[q] ? 1u : 0u);
                                    set_byte(kv, q - in_lo, mask_at(roff + (q - a)));
                                }
                            }
                        }
                        readidx += l; refidx += l;
                    } else if (op == 1) {                                      // I      (:115-120)
                        if (rl == 0 || refidx >= rl) { flag(p.status, ST_INDEX, r); break; }   // subset_variable[refidx]
                        const int a = readidx > in_lo ? readidx : in_lo, b = readidx + l < in_hi ? readidx + l : in_hi;
                        if (a < b) {                                           // only the chunks the insertion touches look at the mask
                            const int left = refidx - 1 < 0 ? rl - 1 : refidx - 1;               // Python wraps index -1
                            const u32 both = mask_at(g0 + left) & mask_at(g0 + refidx);
#pragma unroll
                            for (int w = 0; w < 4; ++w) {
                                const u32 rm = range_mask(a - in_lo, b - in_lo, w);
                                kv[w] = (kv[w] & ~rm) | ((both * 0x01010101u) & rm);
                            }
                        }
                        readidx += l;
                    } else if (op == 2 || op == 3) {                           // D N    (:121-125)
                        if (n == 0) { flag(p.status, ST_INDEX, r); break; }
                        const int at = readidx - 1 < 0 ? n + (readidx - 1) : readidx - 1;    // skips[-1]: the last base
                        if (at < 0 || at >= n) { flag(p.status, ST_INDEX, r); break; }
                        if (at >= in_lo && at < in_hi) {
                            u32 any = 0u;
                            for (int i = refidx; i < refidx + l && i < rl; ++i) any |= mask_at(g0 + i);
                            set_byte(kv, at - in_lo, get_byte(kv, at - in_lo) | (any ? 1u : 0u));
                        }
                        refidx += l;
                    } else if (op == 4) {                                      // S      (:126-129)
                        const int a = readidx > in_lo ? readidx : in_lo, b = readidx + l < in_hi ? readidx + l : in_hi;
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const u32 rm = range_mask(a - in_lo, b - in_lo, w);
                            kv[w] = (kv[w] & ~rm) | (0x01010101u & rm);
                        }
                        readidx += l;
                    } else if (op == 5 || op == 6) {                           // H P    (:130-134)
                    } else { flag(p.status, ST_RANGE, r); break; }             // unrecognised operation: ValueError
                }
                if (f) {                                                       // output byte i = input byte cnt-1-i
                    reverse16(ev); reverse16(kv);
                    shr_bytes16(ev, 16 - cnt); shr_bytes16(kv, 16 - cnt);
                }
            }
            const size_t off = (size_t)r * p.pitch + (size_t)16 * j;
            *reinterpret_cast<uint4*>(p.err + off) = make_uint4(ev[0], ev[1], ev[2], ev[3]);
            *reinterpret_cast<uint4*>(p.skip + off) = make_uint4(kv[0], kv[1], kv[2], kv[3]);
        }
        it0 = it1; m0 = m1; w0 = w1;
        it1 = it2; m1 = m2;
        it2 = next(it2);
    }
}

// ---------------------------------------------------------------- K6 (BAM-sourced tally)
// gatk/bqsr.py:52-123 tallies aligned reads with strand-aware covariates: the cycle and the
// dinucleotide context are those of the base in SEQUENCING orientation over the aligned part
// (bqsr.py:23-50), and a base is skipped when K4 flagged its site, when its original quality is
// below minscore, when it lies past the adaptor boundary (bqsr.py:158-206, host: a per-read
// range) or when it is 'N' (bqsr.py:86-88).  K6 rewrites every read into that orientation --
// aligned part only, reverse-strand reads reverse-complemented (unknown letters -> 'N', as
// Dinucleotide.complement.get(x, 'N')), padded with uncounted bases to the common length S,
// skipped bases given quality byte 0, errors expressed as cseq != seq -- and the result goes
// through the SAME tally kernel as the FASTQ path (K1): canonical position = cycle, sidecar
// `second` bit = is_read2 (column 2S-1-c), context from the canonical neighbours.
// lane <-> one 16-byte OUTPUT chunk; input windows are unaligned (load16_any).
struct K6Params {
    const uint8_t* seq; const uint8_t* oq; const uint8_t* err; const uint8_t* skip;   // [nreads, pitch]
    const u32* len;                  // query length (must be S: checked on the host)
    const u32* clip;                 // query_alignment_start | query_alignment_end << 16
    const u32* trim;                 // skipped range lo | hi << 16 (lo == hi: none)
    const u32* flags;                // bit 0 reverse, bit 1 read 2, bits 16.. read group
    long long nreads; int pitch; int S; u32 qlo; u32 dlo;
    uint8_t* out_seq; uint8_t* out_cseq; uint8_t* out_qual; u32* out_meta;
    u64* status;
};

__device__ __forceinline__ u32 complement4(u32 w)
{
    const u32 h = (w >> 1) & 0x07070707u;
    const u32 expect = __builtin_amdgcn_perm(0x4E000000u, 0x47544341u, h);
    const u32 comp = __builtin_amdgcn_perm(0x4E4E4E4Eu, 0x43414754u, h);       // A->T C->G T->A G->C, else N
    const u32 bad = nonzero_bytes(expect ^ w) * 0xFFu;                          // not exactly A C G T N
    return (comp & ~bad) | (0x4E4E4E4Eu & bad);
}

__device__ __forceinline__ bool is_acgt(u32 ch) { return ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T'; }

// Work items of a thread: (read block rb, chunk j) with j = j0, j0 + 256, ... inside a row and rb advancing by the
// grid.  They are software-pipelined two deep -- the per-read fields of item i + 2 and the four 16-byte windows of
// item i + 1 are in flight while item i is computed and stored -- because the chain  fields -> windows -> stores
// is what a wave otherwise waits through once per item.
struct K6Item { long long rb; int j; };
struct K6Meta { u32 fl, clip, trim; bool valid; };
struct K6Win { u32 s[4], q[4], e[4], k[4]; int cnt, i0; bool has; };

// NIB: out_seq / out_cseq are 4-bit planes (pitch / 2 bytes per row, the layout of KBBQ_ROWS_NIBBLES): K6 writes 2 B/base
// instead of 3 and K1 reads 2 instead of 3.  A corrected base differs from its base in the low code bit (an N of a
// reverse-strand read that was another letter IS counted by cycle and quality: its corrected code is 5, which only K1's
// comparison ever sees).  A letter outside ACGTN (only a forward-strand read can carry one into the output) cannot be
// packed: ST_LUT, and the caller repeats the pass with character planes -- which is also where the reference's
// TypeError is decided, so this form does not look for it.
template <bool NIB>
__global__ __launch_bounds__(256) void k6_canonical_reads(K6Params p)
{
    const int cpr = p.pitch >> 4;
    const int rpb = cpr <= 256 ? 256 / cpr : 0;
    const int slot = cpr <= 256 ? (int)threadIdx.x / cpr : 0;
    const int j0 = (int)threadIdx.x - slot * cpr;
    const long long step = rpb ? rpb : 1;
    const long long gstep = (long long)gridDim.x * step;
    const bool idle = (rpb && slot >= rpb) || j0 >= cpr;
    const long long plane = p.nreads * (long long)p.pitch;
    auto next = [&](K6Item it) { it.j += 256; if (it.j >= cpr) { it.j = j0; it.rb += gstep; } return it; };
    auto live = [&](const K6Item& it) { return !idle && it.rb + slot < p.nreads; };
    auto fetch_meta = [&](const K6Item& it, K6Meta& m) {
        m.valid = live(it);
        const long long r = m.valid ? it.rb + slot : 0;
        m.fl = p.flags[r]; m.clip = p.clip[r]; m.trim = p.trim[r];
    };
    auto fetch_win = [&](const K6Item& it, const K6Meta& m, K6Win& w) {
        w.has = false; w.cnt = 0; w.i0 = 0;
        if (!m.valid) return;
        const int qs = (int)(m.clip & 0xFFFFu), qe = (int)(m.clip >> 16);
        const int L = qe - qs, c0 = 16 * it.j;
        if (c0 >= L) return;
        // input window [i0, i0 + 16): the chunk's bases in INPUT order occupy its first cnt bytes
        // (a reverse-strand chunk is byte-reversed afterwards; a partial one then shifted down)
        w.has = true;
        w.cnt = L - c0 < 16 ? L - c0 : 16;
        w.i0 = (m.fl & 1u) ? (w.cnt == 16 ? qe - c0 - 16 : qs) : qs + c0;
        const long long at = (it.rb + slot) * (long long)p.pitch + w.i0;
        load16_upto(p.seq, at, plane, w.s);
        load16_upto(p.oq, at, plane, w.q);
        load16_upto(p.err, at, plane, w.e);
        if (p.skip) load16_upto(p.skip, at, plane, w.k);
        else {                                                     // one plane of flags: bit 0 error, bit 1 skip
#pragma unroll
            for (int x = 0; x < 4; ++x) { w.k[x] = w.e[x] & 0x02020202u; w.e[x] &= 0x01010101u; }
        }
    };
    K6Item it0{(long long)blockIdx.x * step, j0};
    K6Item it1 = next(it0), it2 = next(it1);
    K6Meta m0, m1, m2;
    K6Win w0, w1;
    fetch_meta(it0, m0); fetch_meta(it1, m1);
    fetch_win(it0, m0, w0);
    while (live(it0)) {
        fetch_meta(it2, m2);
        fetch_win(it1, m1, w1);
        {
            const long long r = it0.rb + slot;
            const int j = it0.j;
            const u32 fl = m0.fl;
            const bool rev = (fl & 1u) != 0;
            const int tlo = (int)(m0.trim & 0xFFFFu), thi = (int)(m0.trim >> 16);
            const int c0 = 16 * j;
            u32 os[4] = {0x4E4E4E4Eu, 0x4E4E4E4Eu, 0x4E4E4E4Eu, 0x4E4E4E4Eu};
            u32 oc[4] = {0x4E4E4E4Eu, 0x4E4E4E4Eu, 0x4E4E4E4Eu, 0x4E4E4E4Eu};
            u32 oqv[4] = {0u, 0u, 0u, 0u};
            const size_t row = (size_t)r * p.pitch;
            if (j == 0) p.out_meta[r] = (u32)p.S | ((fl >> 16) << 16) | ((fl & 2u) ? 0x80000000u : 0u);
            if (w0.has) {
                const int cnt = w0.cnt, i0 = w0.i0;
                const u32 (&s)[4] = w0.s; const u32 (&q)[4] = w0.q; const u32 (&e)[4] = w0.e; const u32 (&k)[4] = w0.k;
                bool odd = false;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const u32 below = (~((q[w] | 0x80808080u) - p.qlo * 0x01010101u) >> 7) & 0x01010101u;   // q < minscore
                    const u32 trimmed = range_mask(tlo - i0, thi - i0, w) & 0x01010101u;
                    const u32 isn = nonzero_bytes(s[w] ^ 0x4E4E4E4Eu) ^ 0x01010101u;
                    const u32 sk = (nonzero_bytes(k[w]) | below | trimmed | isn) * 0xFFu;
                    u32 code, code5, expect;
                    decode4x(s[w], code, code5, expect);
                    odd |= ((expect ^ s[w]) & byte_mask(cnt, w)) != 0u;
                    os[w] = rev ? complement4(s[w]) : s[w];
                    oc[w] = os[w] ^ (nonzero_bytes(e[w]) << 7);                    // an error: cseq differs from seq
                    oqv[w] = q[w] & ~sk;
                }
                if (NIB) {
                    if (odd && !rev) flag(p.status, ST_LUT, r);
                } else if (odd && !rev) {
                    // the reference's TypeError (compare_reads.py:281-293 via bqsr.py:43-45) is decided on the
                    // ORIGINAL qualities, before any skipping: a looked-up pair with a letter outside ACGT
                    u32 prev = (c0 >= 1) ? p.seq[row + i0 - 1] : 'N';
                    for (int b = 0; b < cnt; ++b) {
                        const u32 cur = (s[b >> 2] >> (8 * (b & 3))) & 0xFFu, qq = (q[b >> 2] >> (8 * (b & 3))) & 0xFFu;
                        if (c0 + b >= 1 && qq >= p.dlo && cur != 'N' && prev != 'N' && !(is_acgt(cur) && is_acgt(prev)))
                            flag(p.status, ST_TYPE, r);
                        prev = cur;
                    }
                }
                if (rev) {
                    reverse16(os); reverse16(oc); reverse16(oqv);
                    if (cnt < 16) { shr_bytes16(os, 16 - cnt); shr_bytes16(oc, 16 - cnt); shr_bytes16(oqv, 16 - cnt); }
                }
                if (cnt < 16) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) {                                  // past the aligned part: uncounted padding
                        const u32 vm = byte_mask(cnt, w);
                        os[w] = (os[w] & vm) | (0x4E4E4E4Eu & ~vm);
                        oc[w] = (oc[w] & vm) | (0x4E4E4E4Eu & ~vm);
                        oqv[w] &= vm;
                    }
                }
            }
            const size_t off = row + (size_t)16 * j;
            if (NIB) {
                u32 cs[4], cc[4], bad = 0u;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    cs[w] = chars_to_codes(os[w], bad);
                    cc[w] = cs[w] ^ nonzero_bytes(os[w] ^ oc[w]);
                }
                const size_t noff = (row >> 1) + (size_t)8 * j;
                *reinterpret_cast<uint2*>(p.out_seq + noff) = make_uint2(cs[0] | (cs[1] << 4), cs[2] | (cs[3] << 4));
                *reinterpret_cast<uint2*>(p.out_cseq + noff) = make_uint2(cc[0] | (cc[1] << 4), cc[2] | (cc[3] << 4));
            } else {
                *reinterpret_cast<uint4*>(p.out_seq + off) = make_uint4(os[0], os[1], os[2], os[3]);
                *reinterpret_cast<uint4*>(p.out_cseq + off) = make_uint4(oc[0], oc[1], oc[2], oc[3]);
            }
            *reinterpret_cast<uint4*>(p.out_qual + off) = make_uint4(oqv[0], oqv[1], oqv[2], oqv[3]);
        }
        it0 = it1; m0 = m1; w0 = w1;
        it1 = it2; m1 = m2;
        it2 = next(it2);
    }
}

// ---------------------------------------------------------------- mate-pair rows
// Optional device layout for paired reads of one length S (2 x 150 bp): ONE row per pair,
//     [mate 1: S bytes][separator][mate 2: S bytes][padding to a multiple of 16]
// separator and padding are 'N' (seq, cseq) / 0 (qual), i.e. uncounted bases.  2S + 1 = 301 -> pitch 304
// instead of 2 x 160: 5 % fewer bytes through HBM for the same bases, and K1 / K2 run unchanged on the
// rows as if they were single reads: byte offset b IS the stored cycle index (K1's table keeps
// second-in-pair cycles mirrored from S upwards, here from S + 1; K2's pair LUT likewise), and the 'N'
// separator gives mate 2's first base "no previous base" exactly like position 0 of a read.
__host__ __device__ __forceinline__ int pair_pitch(int S2) { return (S2 + 1 + 15) & ~15; }
__host__ __device__ __forceinline__ int pair_lut_row_bytes(int S2)
{
    int rb = pair_pitch(S2) + 32;
    if (((rb >> 2) & 1) == 0) rb += 4;        // odd number of dwords per row
    return rb;
}

struct PairLutParams {
    const int16_t* lut16; int rs16; int R; int Qt; int S2; int minscore;
    int twins;                       // rows of two first-in-pair reads: the second half looks up the forward columns too
    int8_t* out;
};

// one workgroup per row (read group, raw quality byte) of the pair LUT
__global__ __launch_bounds__(256) void k3_fill_pair_lut(PairLutParams p)
{
    const int rb = pair_lut_row_bytes(p.S2), cyc = pair_pitch(p.S2), S = p.S2 >> 1;
    const int NR = 33 + p.Qt;
    const int row = blockIdx.x;
    const int r = row / NR, qb = row - r * NR;
    const bool model = qb >= 33 + p.minscore;
    const int16_t* src = p.lut16 + ((size_t)r * p.Qt + (qb >= 33 ? qb - 33 : 0)) * p.rs16;
    int8_t* dst = p.out + (size_t)row * rb;
    for (int x = threadIdx.x; x < rb; x += blockDim.x) {
        int v = 0;
        if (!model) {
            if (x < cyc) v = qb == 0 ? -33 : qb - 33;                     // padding -> 0 ; uncounted -> unchanged
        } else {
            if (x < S) v = src[x];                                        // mate 1: column = position
            else if (x > S && x <= p.S2) v = src[p.twins ? x - S - 1 : p.S2 - 1 - (x - S - 1)]; // mate 2, position i = x-S-1: column 2S-1-i (twins: i)
            else if (x >= cyc && x < cyc + 25) v = src[p.S2 + (x - cyc)];
        }
        dst[x] = (int8_t)v;
    }
}

struct PairPackParams {
    const uint8_t* src[3]; uint8_t* dst[3]; uint8_t fill[3];
    const u32* meta; u32* pmeta;
    long long npairs; int pitch; int ppitch; int S; int unpack;
};

// lane <-> 16-byte chunk of a destination row.  pack: two reads -> one pair row (three planes);
// unpack: one pair row -> two rows of one plane (the K2 output).
__global__ __launch_bounds__(256) void k7_pack_pairs(PairPackParams p)
{
    const int S = p.S;
    if (!p.unpack) {
        const int cpr = p.ppitch >> 4;
        const long long nchunks = p.npairs * cpr;
        for (long long ch = (long long)blockIdx.x * blockDim.x + threadIdx.x; ch < nchunks;
             ch += (long long)gridDim.x * blockDim.x) {
            const long long pr = ch / cpr;
            const int j = (int)(ch - pr * cpr);
            if (j == 0) p.pmeta[pr] = (u32)(2 * S + 1) | (p.meta[2 * pr] & 0x7FFF0000u);
            const long long limit = 2 * p.npairs * (long long)p.pitch;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                if (!p.src[pl]) continue;
                const u32 f4 = p.fill[pl] * 0x01010101u;
                u32 a[4], b[4], o[4];
                // bytes [16j, 16j+16) of the pair row: mate 1 from offset 16j, mate 2 from offset 16j - S - 1
                load16_upto(p.src[pl], (2 * pr) * (long long)p.pitch + 16 * j, limit, a);
                const long long off2 = (2 * pr + 1) * (long long)p.pitch + (16 * j - S - 1);
                if (16 * j + 15 > S) {
                    if (16 * j - S - 1 >= 0) load16_upto(p.src[pl], off2, limit, b);
                    else {                                                   // the chunk holding the separator
                        b[0] = b[1] = b[2] = b[3] = 0u;
                        for (int k = S + 1 - 16 * j; k < 16; ++k)
                            b[k >> 2] |= (u32)p.src[pl][off2 + k] << (8 * (k & 3));
                    }
                } else { b[0] = b[1] = b[2] = b[3] = 0u; }
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const u32 m1 = byte_mask(S - 16 * j, w);                         // bytes of mate 1
                    const u32 m2 = range_mask(S + 1 - 16 * j, 2 * S + 1 - 16 * j, w);   // bytes of mate 2
                    o[w] = (a[w] & m1) | (b[w] & m2) | (f4 & ~(m1 | m2));
                }
                *reinterpret_cast<uint4*>(p.dst[pl] + pr * (long long)p.ppitch + 16 * j) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
    } else {
        const int cpr = p.pitch >> 4;
        const long long nchunks = 2 * p.npairs * cpr;
        const long long limit = p.npairs * (long long)p.ppitch;
        for (long long ch = (long long)blockIdx.x * blockDim.x + threadIdx.x; ch < nchunks;
             ch += (long long)gridDim.x * blockDim.x) {
            const long long rd = ch / cpr;
            const int j = (int)(ch - rd * cpr);
            const long long pr = rd >> 1;
            const int base = (rd & 1) ? S + 1 : 0;
            u32 a[4];
            load16_upto(p.src[0], pr * (long long)p.ppitch + base + 16 * j, limit, a);
#pragma unroll
            for (int w = 0; w < 4; ++w) a[w] &= byte_mask(S - 16 * j, w);            // past the read: zero, as K2 writes
            *reinterpret_cast<uint4*>(p.dst[0] + rd * (long long)p.pitch + 16 * j) = make_uint4(a[0], a[1], a[2], a[3]);
        }
    }
}

struct K5Params {
    const uint8_t* qual; const uint8_t* err; const uint8_t* skip; const u32* len;
    long long nreads; int pitch; int cpr; u32 cpr_magic; int qoffset;
    u64* counts;                     // [0..255] totals, [256..511] errors
    u64* status;
};

// lane <-> 16-byte chunk.  LDS: [257 bins][16 copies] u32, errs << 16 | total (copy = lane & 15: the 40-odd bins in
// use are hot; copies cut the same-address collisions of a wave's atomic 16-fold), bin 256 = trash (skipped sites,
// bytes past the read, values below the offset) so that every byte costs exactly one unconditional LDS atomic.
// 16-bit halves: a copy receives at most 16 lanes x 16 bytes per workgroup iteration -> flushed every 255 iterations.
#define K5_THREADS 256
#define K5_COPIES 16
#define K5_FLUSH_ITERS (65535 / ((K5_THREADS / K5_COPIES) * 16))
__global__ __launch_bounds__(K5_THREADS) void k5_count_q(K5Params p)
{
    __shared__ u32 h[257 * K5_COPIES];
    for (int i = threadIdx.x; i < 257 * K5_COPIES; i += blockDim.x) h[i] = 0u;
    __syncthreads();
    const long long nchunks = p.nreads * p.cpr;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long iters = (nchunks + stride - 1) / stride;                 // the same for every thread: barriers are safe
    const u32 copy = threadIdx.x & (K5_COPIES - 1);
    auto flush = [&]() {
        __syncthreads();
        for (int b = threadIdx.x; b < 256; b += blockDim.x) {
            u32 t = 0u, e = 0u;
            for (int c = 0; c < K5_COPIES; ++c) { const u32 v = h[b * K5_COPIES + c]; h[b * K5_COPIES + c] = 0u; t += v & 0xFFFFu; e += v >> 16; }
            if (t) atomicAdd(&p.counts[b], (u64)t);
            if (e) atomicAdd(&p.counts[256 + b], (u64)e);
        }
        __syncthreads();
    };
    int since = 0;
    // the loads of the NEXT chunk are issued before the current one is binned (one step of software prefetch)
    struct Chunk { uint4 q, e, s; long long r; int nb; };
    auto fetch = [&](long long ch, Chunk& c) {
        c.nb = 0; c.r = 0;
        if (ch >= nchunks) return;
        c.r = ch / p.cpr;
        const int j = (int)(ch - c.r * p.cpr);
        c.nb = (int)p.len[c.r] - 16 * j;
        if (c.nb <= 0) return;
        const size_t off = (size_t)c.r * p.pitch + (size_t)16 * j;
        c.q = *reinterpret_cast<const uint4*>(p.qual + off);
        c.e = *reinterpret_cast<const uint4*>(p.err + off);
        if (p.skip) c.s = *reinterpret_cast<const uint4*>(p.skip + off);
        else {                                                     // one plane of flags: bit 0 error, bit 1 skip
            c.s = make_uint4(c.e.x & 0x02020202u, c.e.y & 0x02020202u, c.e.z & 0x02020202u, c.e.w & 0x02020202u);
            c.e = make_uint4(c.e.x & 0x01010101u, c.e.y & 0x01010101u, c.e.z & 0x01010101u, c.e.w & 0x01010101u);
        }
    };
    long long ch = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    Chunk cur, nxt;
    fetch(ch, cur);
    for (long long it = 0; it < iters; ++it, ch += stride) {
        fetch(ch + stride, nxt);
        if (cur.nb > 0) {
            const int nb = cur.nb;
            const u32 q[4] = {cur.q.x, cur.q.y, cur.q.z, cur.q.w}, e[4] = {cur.e.x, cur.e.y, cur.e.z, cur.e.w}, s[4] = {cur.s.x, cur.s.y, cur.s.z, cur.s.w};
            bool negative = false;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int sh = 8 * (i & 3);
                const int qq = (int)((q[i >> 2] >> sh) & 0xFFu) - p.qoffset;
                const bool counted = i < nb && ((s[i >> 2] >> sh) & 0xFFu) == 0u;
                negative |= counted && qq < 0;                               // np.bincount rejects negative values: ValueError
                const u32 bin = (counted && qq >= 0) ? (u32)qq : 256u;
                const u32 inc = ((e[i >> 2] >> sh) & 0xFFu) ? 0x10001u : 1u;
                atomicAdd(&h[bin * K5_COPIES + copy], inc);
            }
            if (negative) flag(p.status, ST_RANGE, cur.r);
        }
        cur = nxt;
        if (++since == K5_FLUSH_ITERS) { flush(); since = 0; }
    }
    flush();
}
