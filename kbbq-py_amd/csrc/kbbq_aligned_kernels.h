// kbbq_aligned_kernels.h -- K4, second form: compare_reads.find_read_errors (compare_reads.py:84-139) without the
// wave-lock-step CIGAR walk.
//
// What bounded the first form (kbbq_kernels_v3.h k4_find_errors; profiles/r01_pmc_aligned.md): the
// reference / mask window of a chunk was fetched at the offset that is right when NO insertion or deletion precedes
// the chunk; every chunk behind an indel found out inside the walk that it needed another window and fetched it there,
// a dependent load that parks the whole wave -- 0.61 ms when every read is one M block, 0.78 ms with 5 % indel reads.
//
// Here the first four CIGAR operations of every read sit in one 32-byte record per read WITH their cumulative read /
// reference positions (k4_read_records writes it from the CSR arrays; a host reader can fill it while parsing), so
// they arrive together with the read's other fields, one pipeline stage ahead of the windows.  A lane (<-> one
// 16-byte output chunk) tests those <= 4 read ranges against its own 16 positions -- independent comparisons in
// registers, no walk -- and learns where they sit:
//   * inside ONE M / = / X operation, no D / N operation pointing at one of its positions, nothing odd anywhere in
//     the CIGAR: the chunk is "simple"; its reference window starts at ref_start + refidx + (in_lo - readidx) and
//     is fetched right away -- the correct window, whatever precedes the chunk;
//   * anything else (an operation boundary inside the chunk, more than four operations, a shape error): the chunk
//     takes the sequential walk of the first form, unchanged (k4_walk_chunk), which carries the reference's exact
//     semantics: an insertion's both-neighbours test, a deletion OR-ing into the base before it, Python's [-1]
//     wraps, ValueError / IndexError on malformed input.  That walk is ~500 vector instructions and a lane that takes
//     it holds its whole wave: with one such chunk per indel read (0.5 % of the chunks at 5 % indel reads) 28 % of the
//     wave iterations paid for it (measured: +0.19 ms on 0.46).  Walking them densely at the end of the workgroup (a
//     queue of (read, chunk) in LDS, 64 queued chunks per wave) is no better: everything a chunk needs has to be
//     fetched again through scattered loads (measured: the same +0.19 ms).  So
//   * the COMMON boundary shapes are composed from registers without a walk ("two-window" chunks): at most two
//     M / = / X operations overlap the chunk, with at most one insertion or soft clip between / beside them and at
//     most one short deletion behind one of them.  Stage B fetches the windows of both operations (a simple chunk's
//     second window is 16 idle bytes); stage C masks the two comparisons into their byte ranges, fills the gap
//     (soft clip: skipped; insertion: skipped iff both reference neighbours are -- both sit in the two windows) and
//     ORs the deletion's sites into the base before it (they sit in that base's window when the deletion is short);
//   * what is left (three operations in one chunk, long deletions, > 4 operations, malformed input) is queued in LDS
//     and walked at the end of the workgroup.
// Three pipeline stages per thread: fields + inline operations of item i+2 | register walk, read bytes and window
// of item i+1 | comparison and store of item i.
//
// Output: two byte planes (errors, skips) as before, or -- skip == NULL -- ONE plane of flags, bit 0 = error,
// bit 1 = skip, which K5 / K6 read in place of the two (1 byte per base less to write here and to read there).
#pragma once
#include "kbbq_kernels_v3.h"

#define K4_INLINE_OPS 4
#define K4_NOT_SIMPLE 0x7FFFFFFF

// One 32-byte record per read, written by k4_read_records from the CSR CIGAR arrays (a host reader can fill it while
// parsing): for each of the first four operations
//     word 0 = read range [rs, re) the operation covers (16 bits each; a deletion: [target base, target base + 1))
//     word 1 = kind << 29 | payload      kind 1 M/=/X: payload = refidx - readidx (29-bit two's complement)
//                                        kind 2 insertion, 3 soft clip: -
//                                        kind 4 deletion / reference skip: payload = length | usable << 20
//                                               (usable: the operation before it is an M whose last base is the target,
//                                                and the deleted sites lie inside the read's reference window)
//                                        kind 0 nothing that touches read bases (H, P, or past the last operation)
// kind 7 in operation 0 marks a read every chunk of which takes the sequential walk: more than four operations, none,
// an unknown operation, or a shape the reference raises on (the walk raises it).
struct K4Rec { uint4 lo, hi; };
#define K4_KIND_BAD 7u

struct K4v2Params {
    K4Params base;
    const K4Rec* recs;               // [nreads]
    const uint8_t* idle16;           // any 16 readable bytes: what a chunk without a window of its own loads instead
    // only SOME reads (kbbq_tally_aligned_dev: the reads that are not one M / = / X operation): work item i is read rows[i],
    // *nrows of them (a count k4_read_records left on the device); NULL: every read of the batch
    const u32* rows; const u32* nrows;
};

struct K4RecParams {
    const u32* len; const int* ref_len; const u32* cig_off; const u32* cig_n; const u32* cigar; long long nreads; K4Rec* recs;
    // kbbq_tally_aligned_dev: which reads need k4v2_find_errors at all?  A read that is ONE M / = / X operation over all of its
    // bases, reference window inside the genome for every 16-byte chunk of its row, is compared with the reference by the tally
    // kernel itself (k1v3_aligned_ref): aflags_out = aflags_in | 4 for it; every other read is appended to rows[*nrows].
    const u32* aflags_in; u32* aflags_out; u32* rows; u32* nrows;
    const long long* ref_start; long long genome_len; int pitch;
};

#define K4_LIST_BUF 1024          // listed reads a workgroup collects in LDS before it appends them to rows[] with ONE global atomic
__global__ __launch_bounds__(256) void k4_read_records(K4RecParams p)
{
    // (the list: one global atomic per listed read -- 800 K of them on one address for 16 M reads with 5 % indel reads -- took
    //  2.5 ms; collected per workgroup in LDS and appended K4_LIST_BUF at a time they cost nothing measurable)
    __shared__ u32 buf[K4_LIST_BUF];
    __shared__ u32 nbuf, base;
    const bool listing = p.aflags_out != nullptr;
    if (listing) { if (threadIdx.x == 0) nbuf = 0u; __syncthreads(); }
    auto flush = [&]() {                                       // every thread of the workgroup calls this
        __syncthreads();
        const u32 k = nbuf;
        if (threadIdx.x == 0 && k) base = atomicAdd(p.nrows, k);
        __syncthreads();
        for (u32 i = threadIdx.x; i < k; i += blockDim.x) p.rows[base + i] = buf[i];
        __syncthreads();
        if (threadIdx.x == 0) nbuf = 0u;
        __syncthreads();
    };
    for (long long r0 = (long long)blockIdx.x * blockDim.x; r0 < p.nreads; r0 += (long long)gridDim.x * blockDim.x) {
        const long long r = r0 + threadIdx.x;
        if (r < p.nreads) {
        const u32 nc = p.cig_n[r];
        const int n = (int)p.len[r], rl = p.ref_len[r];
        const u32* ops = p.cigar + p.cig_off[r];
        u32 w0[K4_INLINE_OPS] = {0u, 0u, 0u, 0u}, w1[K4_INLINE_OPS] = {0u, 0u, 0u, 0u};
        bool ok = nc > 0 && nc <= K4_INLINE_OPS && n <= 65535;
        int readidx = 0, refidx = 0;
        bool prev_m = false;
        for (u32 c = 0; c < K4_INLINE_OPS && c < nc && ok; ++c) {
            const u32 word = ops[c];
            const int op = (int)(word & 15u), l = (int)(word >> 4);
            if (op == 0 || op == 7 || op == 8) {
                const int shift = refidx - readidx;
                if (refidx + l > rl || readidx + l > n || shift >= (1 << 28) || shift < -(1 << 28)) { ok = false; break; }
                w0[c] = (u32)readidx | (u32)(readidx + l) << 16;
                w1[c] = (1u << 29) | ((u32)shift & 0x1FFFFFFFu);
                readidx += l; refidx += l; prev_m = l > 0;
            } else if (op == 1 || op == 4) {
                if (op == 1 && (rl == 0 || refidx >= rl)) { ok = false; break; }
                if (readidx + l > 65535) { ok = false; break; }
                w0[c] = (u32)readidx | (u32)(readidx + l) << 16;
                w1[c] = (op == 1 ? 2u : 3u) << 29;
                readidx += l; prev_m = false;
            } else if (op == 2 || op == 3) {
                const int at = readidx - 1 < 0 ? n + (readidx - 1) : readidx - 1;
                if (n == 0 || at < 0 || at >= n) { ok = false; break; }
                const bool usable = prev_m && readidx >= 1 && refidx + l <= rl && l < (1 << 20);
                w0[c] = (u32)at | (u32)(at + 1) << 16;
                w1[c] = (4u << 29) | (usable ? (1u << 20) | (u32)l : 0u);
                refidx += l; prev_m = false;
            } else if (op == 5 || op == 6) {
                // touches nothing
            } else { ok = false; break; }
        }
        if (!ok) { w0[0] = 0u; w1[0] = K4_KIND_BAD << 29; }
        K4Rec rec;
        rec.lo = make_uint4(w0[0], w1[0], w0[1], w1[1]);
        rec.hi = make_uint4(w0[2], w1[2], w0[3], w1[3]);
        bool simple = false;
        if (listing) {
            // one operation, M / = / X, covering exactly the read (the record above: range [0, n), shift 0), and every chunk's
            // window [g0 + 16 j, g0 + 16 j + 16) readable
            const long long g0 = p.ref_start[r];
            simple = ok && nc == 1 && n > 0 && w0[0] == ((u32)n << 16) && w1[0] == (1u << 29)
                     && g0 >= 0 && g0 + p.pitch <= p.genome_len;
            p.aflags_out[r] = (p.aflags_in[r] & ~4u) | (simple ? 4u : 0u);
            if (!simple) buf[atomicAdd(&nbuf, 1u)] = (u32)r;
        }
        if (!simple) p.recs[r] = rec;                          // nobody reads the record of a read the tally kernel compares itself
        }
        if (listing) {                                          // room for the next iteration's 256 reads?  (a uniform decision)
            __syncthreads();
            if (nbuf > K4_LIST_BUF - 256) flush();
        }
    }
    if (listing) flush();
}

// the sequential walk of the first form for ONE chunk (compare_reads.py:100-137), operation by operation in CIGAR order
template <bool FUSED>
__device__ __forceinline__ void k4_walk_chunk(const K4Params& p, long long r, int n, int rl, long long g0, const u32* ops, u32 nc,
                                           const u32 (&sw)[4], int in_lo, int cnt, u32 (&ev)[4], u32 (&kv)[4])
{
    constexpr bool fused = FUSED;
    auto mask_at = [&](long long i) -> u32 { return fused ? (u32)(p.genome[i] >> 7) : (u32)(p.skipmask[i] != 0); };
    const int in_hi = in_lo + cnt;
    const uint8_t* s = p.seq + (size_t)r * p.pitch;
    int readidx = 0, refidx = 0;
    for (u32 c = 0; c < nc; ++c) {
        const u32 word = ops[c];
        const int op = (int)(word & 15u), l = (int)(word >> 4);
        if (op == 0 || op == 7 || op == 8) {                       // M = X  (:109-114)
            if (refidx + l > rl || readidx + l > n) { flag(p.status, ST_RANGE, r); break; }   // shape mismatch: ValueError
            const int a = readidx > in_lo ? readidx : in_lo, b = readidx + l < in_hi ? readidx + l : in_hi;
            if (a < b) {
                const int d = a - in_lo;
                const long long goff = g0 + refidx + (a - readidx) - d;
                if (goff >= 0) {
                    u32 gw[4], mw[4];
                    load16_upto(p.genome, goff, p.genome_len, gw);
                    if (!fused) load16_upto(p.skipmask, goff, p.genome_len, mw);
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
            if (a < b) {
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
}

struct K4v2Meta { int n; u32 f; long long g0; K4Rec rec; bool valid; long long r; };
// a chunk's composition out of two windows: byte ranges (relative to the chunk) of the two M operations and of the gap,
// 4 bits each; kind: 0 simple-or-two-window, 1 queued
struct K4v2Win {
    u32 sw[4], gw[4], mw[4], gw2[4], mw2[4];
    u32 ranges;          // loA | hiA << 5 | loB << 10 | hiB << 15 | glo << 20 | ghi << 25      (0..16 each)
    u32 extra;           // gap kind (0 none, 1 insertion, 2 soft clip) | deletion: present << 2 | window << 3 | at << 4 | len << 9
    int in_lo, cnt; bool has, composed;
};
// 0xFF in the bytes of word w whose position lies in [lo, hi), 0 <= lo, hi <= 16 (cheaper than two byte_masks)
__device__ __forceinline__ u32 span_mask(u32 lo, u32 hi, int w)
{
    const u32 bits = ((1u << hi) - 1u) & ~((1u << lo) - 1u);
    const u32 nib = (bits >> (4 * w)) & 0xFu;
    return ((nib * 0x00204081u) & 0x01010101u) * 0xFFu;
}
#define K4_QUEUE 2048            // queued (read, chunk) pairs per workgroup; beyond that a chunk is walked where it is met

template <bool FUSED>
__global__ __launch_bounds__(256) void k4v2_find_errors(K4v2Params q)
{
    const K4Params& p = q.base;
    const int cpr = p.pitch >> 4;
    const int rpb = cpr <= 256 ? 256 / cpr : 0;
    const int slot = cpr <= 256 ? (int)threadIdx.x / cpr : 0;
    const int j0 = (int)threadIdx.x - slot * cpr;
    const long long step = rpb ? rpb : 1;
    const long long gstep = (long long)gridDim.x * step;
    const bool idle = (rpb && slot >= rpb) || j0 >= cpr;
    constexpr bool fused = FUSED;
    const long long nitems = q.rows ? (long long)*q.nrows : p.nreads;       // work items: reads, or the listed reads
    auto next = [&](K4Item it) { it.j += 256; if (it.j >= cpr) { it.j = j0; it.rb += gstep; } return it; };
    auto live = [&](const K4Item& it) { return !idle && it.rb + slot < nitems; };
    // stage A: the read's fields and its inline operations (independent loads: one round trip)
    auto fetch_meta = [&](const K4Item& it, K4v2Meta& m) {
        m.valid = live(it);
        const long long r = m.valid ? (q.rows ? (long long)q.rows[it.rb + slot] : it.rb + slot) : 0;
        m.r = r;
        m.n = (int)p.len[r]; m.f = p.flip ? p.flip[r] : 0u; m.g0 = p.ref_start[r];
        m.rec = q.recs[r];
    };
    // stage B: where do this chunk's input positions sit?  (registers only)  then the read bytes and THE window.
    // Both loads are UNCONDITIONAL (a chunk without work reads the first 16 bytes of the plane, one without a simple window
    // 16 idle bytes): a load under a branch makes the compiler wait for it at the branch's end, which would park the
    // wave in the middle of the pipeline (the first build of this kernel did exactly that: 2.1 ms instead of 0.6).
    auto fetch_win = [&](const K4Item& it, const K4v2Meta& m, K4v2Win& w) {
        const int n = m.n, out_lo = 16 * it.j, out_hi = out_lo + 16 < n ? out_lo + 16 : n;
        w.has = m.valid && out_lo < n;
        w.cnt = w.has ? out_hi - out_lo : 0;
        w.in_lo = w.has ? (m.f ? n - out_hi : out_lo) : 0;
        const int a = w.in_lo, b = w.in_lo + w.cnt;
        load16_any(p.seq, w.has ? m.r * p.pitch + a : 0ll, w.sw);   // the chunk's read bytes (may run into the next row)
        // the four operation records against [a, b): independent range tests, no branches
        const u32 r0[4] = {m.rec.lo.x, m.rec.lo.z, m.rec.hi.x, m.rec.hi.z}, r1[4] = {m.rec.lo.y, m.rec.lo.w, m.rec.hi.y, m.rec.hi.w};
        const bool ok = w.has && (r1[0] >> 29) != K4_KIND_BAD;
        int nM = 0, shiftA = 0, loA = 0, hiA = 0, shiftB = 0, loB = 0, hiB = 0;     // M operations overlapping [a, b)
        int nG = 0, gkind = 0, glo = 0, ghi = 0;                                    // insertion / soft clip overlapping it
        int nD = 0, dat = 0, dlen = 0, dusable = 0;                                 // deletion whose target base lies in it
        // most reads are ONE operation: when every read of this wave is, only record 0 is looked at (wave-uniform branch)
        const bool more = ((r1[1] | r1[2] | r1[3]) >> 29) != 0u;
        const int nrec = __ballot(w.has && more) == 0ull ? 1 : K4_INLINE_OPS;
#pragma unroll
        for (int c = 0; c < K4_INLINE_OPS; ++c) {
            if (c >= nrec) break;
            const int kind = (int)(r1[c] >> 29);
            const int rs = (int)(r0[c] & 0xFFFFu), re = (int)(r0[c] >> 16);
            const int lo = rs > a ? rs : a, hi = re < b ? re : b;
            const bool ov = lo < hi;
            const bool isM = kind == 1 && ov, isG = (kind == 2 || kind == 3) && ov, isD = kind == 4 && ov;
            const int shift = (int)(r1[c] << 3) >> 3;                               // sign-extend the 29-bit payload
            const bool first = isM && nM == 0, second = isM && nM != 0;
            shiftA = first ? shift : shiftA; loA = first ? lo - a : loA; hiA = first ? hi - a : hiA;
            shiftB = second ? shift : shiftB; loB = second ? lo - a : loB; hiB = second ? hi - a : hiB;
            nM += isM ? 1 : 0;
            gkind = isG ? kind - 1 : gkind; glo = isG ? lo - a : glo; ghi = isG ? hi - a : ghi;
            nG += isG ? 1 : 0;
            dat = isD ? rs - a : dat; dlen = isD ? (int)(r1[c] & 0xFFFFFu) : dlen; dusable = isD ? (int)((r1[c] >> 20) & 1u) : dusable;
            nD += isD ? 1 : 0;
        }
        // the deletion's target is the last base of one of the two M ranges seen here: its sites follow in that window
        const int dwin = (nD && dusable) ? (hiA == dat + 1 ? 1 : (nM == 2 && hiB == dat + 1 ? 2 : 0)) : 0;
        const long long oA = m.g0 + a + shiftA, oB = m.g0 + a + shiftB;
        bool composed = ok && nM >= 1 && nM <= 2 && nG <= 1 && nD <= 1;
        composed = composed && (hiA - loA) + (hiB - loB) + (ghi - glo) == w.cnt;                 // the three ranges tile the chunk
        composed = composed && oA >= 0 && oA + 16 <= p.genome_len && (nM < 2 || (oB >= 0 && oB + 16 <= p.genome_len));
        // an insertion needs both reference neighbours in the windows: M before it and M after it, inside this chunk
        composed = composed && (gkind != 1 || (nM == 2 && hiA == glo && ghi == loB && nD == 0));
        // a deletion's sites must lie in the window of the base before it
        composed = composed && (nD == 0 || (dwin != 0 && dat + 1 + dlen <= 16));
        w.composed = composed;
        w.ranges = (u32)loA | (u32)hiA << 5 | (u32)loB << 10 | (u32)hiB << 15 | (u32)glo << 20 | (u32)ghi << 25;
        w.extra = (u32)(nG ? gkind : 0) | (u32)(nD ? 1 : 0) << 2 | (u32)(dwin == 2 ? 1 : 0) << 3 | (u32)dat << 4 | (u32)(dlen & 31) << 9;
        const bool two = composed && nM == 2;
        load16_any(composed ? p.genome + oA : q.idle16, 0, w.gw);
        load16_any(two ? p.genome + oB : q.idle16, 0, w.gw2);
        if (!fused) {
            load16_any(composed ? p.skipmask + oA : q.idle16, 0, w.mw);
            load16_any(two ? p.skipmask + oB : q.idle16, 0, w.mw2);
        }
    };
    __shared__ u32 queued;
    __shared__ uint2 queue[K4_QUEUE];                                      // (read relative to the launch: low / high word | chunk << 24)
    if (threadIdx.x == 0) queued = 0u;
    __syncthreads();
    // one chunk through the sequential walk, from scratch (fields, read bytes, operations, windows: all fetched here)
    auto walk_and_store = [&](long long r, int j) {
        const int n = (int)p.len[r], rl = p.ref_len[r];
        const bool f = p.flip && p.flip[r] != 0;
        const int out_lo = 16 * j, out_hi = out_lo + 16 < n ? out_lo + 16 : n;
        const int cnt = out_hi - out_lo, in_lo = f ? n - out_hi : out_lo;
        u32 sw[4], ev[4] = {0u, 0u, 0u, 0u}, kv[4] = {0u, 0u, 0u, 0u};
        load16_any(p.seq + (size_t)r * p.pitch, in_lo, sw);
#ifndef K4_ABL_NOWALK
        k4_walk_chunk<FUSED>(p, r, n, rl, p.ref_start[r], p.cigar + p.cig_off[r], p.cig_n[r], sw, in_lo, cnt, ev, kv);
#else
        ev[0] = sw[0]; kv[0] = (u32)rl;
#endif
        if (f) {                                                           // output byte i = input byte cnt-1-i
            reverse16(ev); reverse16(kv);
            shr_bytes16(ev, 16 - cnt); shr_bytes16(kv, 16 - cnt);
        }
        const size_t off = (size_t)r * p.pitch + (size_t)16 * j;
        if (p.skip) {
            *reinterpret_cast<uint4*>(p.err + off) = make_uint4(ev[0], ev[1], ev[2], ev[3]);
            *reinterpret_cast<uint4*>(p.skip + off) = make_uint4(kv[0], kv[1], kv[2], kv[3]);
        } else {
            *reinterpret_cast<uint4*>(p.err + off) = make_uint4(ev[0] | (kv[0] << 1), ev[1] | (kv[1] << 1), ev[2] | (kv[2] << 1), ev[3] | (kv[3] << 1));
        }
    };
    K4Item it0{(long long)blockIdx.x * step, j0};
    K4Item it1 = next(it0), it2 = next(it1);
    K4v2Meta m0, m1, m2;
    K4v2Win w0, w1;
    fetch_meta(it0, m0); fetch_meta(it1, m1);
    fetch_win(it0, m0, w0);
    while (live(it0)) {
        fetch_meta(it2, m2);
        fetch_win(it1, m1, w1);
        {
            const long long r = m0.r;
            const int j = it0.j;
            u32 ev[4] = {0u, 0u, 0u, 0u}, kv[4] = {0u, 0u, 0u, 0u};
            bool store = true;
            if (w0.has) {
                const int cnt = w0.cnt;
                // every composed chunk of this wave inside ONE operation, nothing beside it (the common wave): one comparison
                const bool plain = w0.composed && (w0.ranges & 0x3FFu) == ((u32)cnt << 5) && (w0.ranges >> 10) == 0u && (w0.extra & 7u) == 0u;
                if (__ballot(w0.composed && !plain) == 0ull && w0.composed) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const u32 g = fused ? (w0.gw[w] & 0x7F7F7F7Fu) : w0.gw[w];
                        const u32 mk = fused ? (w0.gw[w] & 0x80808080u) : w0.mw[w];
                        const u32 rm = byte_mask(cnt, w);
                        ev[w] = nonzero_bytes(w0.sw[w] ^ g) & rm;
                        kv[w] = nonzero_bytes(mk) & rm;
                    }
                    if (m0.f != 0) {                                               // output byte i = input byte cnt-1-i
                        reverse16(ev); reverse16(kv);
                        shr_bytes16(ev, 16 - cnt); shr_bytes16(kv, 16 - cnt);
                    }
                } else if (w0.composed) {
                    const u32 rg = w0.ranges, ex = w0.extra;
                    const u32 loA = rg & 31u, hiA = (rg >> 5) & 31u, loB = (rg >> 10) & 31u, hiB = (rg >> 15) & 31u;
                    const u32 glo = (rg >> 20) & 31u, ghi = (rg >> 25) & 31u;
                    u32 gapv = (ex & 3u) == 2u ? 0x01010101u : 0u;                  // soft clip: skipped (:126-129)
                    u32 mA[4], mB[4];
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        mA[w] = fused ? (w0.gw[w] & 0x80808080u) : w0.mw[w];
                        mB[w] = fused ? (w0.gw2[w] & 0x80808080u) : w0.mw2[w];
                    }
                    if ((ex & 3u) == 1u) {                                          // insertion: both neighbours skipped (:115-120)
                        const u32 left = get_byte(mA, (int)hiA - 1), right = get_byte(mB, (int)loB);
                        gapv = (left != 0u && right != 0u) ? 0x01010101u : 0u;
                    }
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const u32 gA = fused ? (w0.gw[w] & 0x7F7F7F7Fu) : w0.gw[w];
                        const u32 gB = fused ? (w0.gw2[w] & 0x7F7F7F7Fu) : w0.gw2[w];
                        const u32 rA = span_mask(loA, hiA, w), rB = span_mask(loB, hiB, w), rG = span_mask(glo, ghi, w);
                        ev[w] = (nonzero_bytes(w0.sw[w] ^ gA) & rA) | (nonzero_bytes(w0.sw[w] ^ gB) & rB);
                        kv[w] = (nonzero_bytes(mA[w]) & rA) | (nonzero_bytes(mB[w]) & rB) | (gapv & rG);
                    }
                    if (ex & 4u) {                                                  // deletion: its sites OR into the base before it (:121-125)
                        const u32 at = (ex >> 4) & 31u, dl = (ex >> 9) & 31u;
                        u32 any = 0u;
#pragma unroll
                        for (int w = 0; w < 4; ++w) any |= ((ex & 8u) ? mB[w] : mA[w]) & span_mask(at + 1u, at + 1u + dl, w);
                        if (any) {
#pragma unroll
                            for (int w = 0; w < 4; ++w) kv[w] |= 0x01010101u & span_mask(at, at + 1u, w);
                        }
                    }
                    if (m0.f != 0) {                                               // output byte i = input byte cnt-1-i
                        reverse16(ev); reverse16(kv);
                        shr_bytes16(ev, 16 - cnt); shr_bytes16(kv, 16 - cnt);
                    }
                } else {
                    // not here: queued for the dense pass below (or, queue full, walked at once)
#ifndef K4_ABL_NOQUEUE
                    const u32 slot_q = atomicAdd(&queued, 1u);
                    if (slot_q < K4_QUEUE) queue[slot_q] = make_uint2((u32)r, (u32)((u64)r >> 32) | ((u32)j << 24));
                    else walk_and_store(r, j);
                    store = false;
#endif
                }
            }
            if (store) {
                const size_t off = (size_t)r * p.pitch + (size_t)16 * j;
                if (p.skip) {
                    *reinterpret_cast<uint4*>(p.err + off) = make_uint4(ev[0], ev[1], ev[2], ev[3]);
                    *reinterpret_cast<uint4*>(p.skip + off) = make_uint4(kv[0], kv[1], kv[2], kv[3]);
                } else {
                    *reinterpret_cast<uint4*>(p.err + off) = make_uint4(ev[0] | (kv[0] << 1), ev[1] | (kv[1] << 1), ev[2] | (kv[2] << 1), ev[3] | (kv[3] << 1));
                }
            }
        }
        it0 = it1; m0 = m1; w0 = w1;
        it1 = it2; m1 = m2;
        it2 = next(it2);
    }
    // the dense pass over the chunks that need the sequential walk
    __syncthreads();
    const u32 nq = queued < K4_QUEUE ? queued : K4_QUEUE;
#ifndef K4_ABL_NOTAIL
    for (u32 i = threadIdx.x; i < nq; i += blockDim.x) {
        const uint2 e = queue[i];
        walk_and_store((long long)(((u64)(e.y & 0x00FFFFFFu) << 32) | e.x), (int)(e.y >> 24));
    }
#endif
}
