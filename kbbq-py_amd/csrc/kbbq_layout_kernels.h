// kbbq_layout_kernels.h -- device passes that turn input-order rows into the layouts K1 / K2 run fastest on.
//
// The packer hands reads over as one row per read (include/kbbq_hip.h "padded SoA").  The kernels prefer
//   * mate-pair rows  -- both mates of a 2 x S pair in ONE row [mate 1][sep][mate 2][pad]  (kbbq_kernels_v3.h),
//   * 4-bit sequence planes -- seq / cseq as one code nibble per base (kbbq_kernels_v3.h "4-bit sequence planes"),
//   * rows grouped by read group -- a stable counting sort of the rows by the read-group id of their sidecar word,
//     so that a K1 / K2 slice walks only its own group's rows; K2 stores through the permutation, straight back
//     into input order.
// k7_meta_stats decides what a batch qualifies for (and measures the shortest / longest read, which K1 needs to
// size its LDS tables); k7_rg_sort builds the permutation; k7_lay_out writes the destination planes in ONE pass
// (pair packing + gather by read-group segment + nibble packing fused: every source byte is read once, every
// destination byte written once).  None of this is on the reference's path -- it has no device -- so there is no
// reference line to cite beyond the sidecar rules of compare_reads.py:304-318.
#pragma once
#include "kbbq_kernels_v3.h"

// ---------------------------------------------------------------- sidecar statistics
// stats[0] shortest non-empty read   stats[1] longest read   stats[2] largest read-group id
// stats[3] number of violations of "uniform first/second pairs": read 2p not first-in-pair, read 2p+1 not second,
//          mates in different read groups, a length different from read 0's (0 = the batch can use mate-pair rows,
//          given an even, non-zero number of reads)
// stats[4] empty reads
// stats[5] violations of "uniform pairs of FIRST-in-pair reads" (single-end input, two reads to a mate-pair row): a read that
//          is second in pair, neighbours 2p / 2p+1 in different read groups, a length different from read 0's, an odd count
#define K7_NSTATS 8
struct MetaStatsParams { const u32* meta; long long n; int* stats; };

__global__ __launch_bounds__(256) void k7_meta_stats(MetaStatsParams p)
{
    int mn = 0x7FFFFFFF, mx = 0, rgmax = 0, viol = 0, empty = 0, tviol = 0;
    const u32 len0 = p.n > 0 ? (p.meta[0] & 0xFFFFu) : 0u;
    const long long npairs = (p.n + 1) >> 1;
    for (long long pr = (long long)blockIdx.x * blockDim.x + threadIdx.x; pr < npairs; pr += (long long)gridDim.x * blockDim.x) {
        const bool has2 = 2 * pr + 1 < p.n;
        const uint2 mm = make_uint2(p.meta[2 * pr], has2 ? p.meta[2 * pr + 1] : 0xFFFFFFFFu);
        const u32 m[2] = {mm.x, mm.y};
        for (int k = 0; k < (has2 ? 2 : 1); ++k) {
            const int len = (int)(m[k] & 0xFFFFu), rg = (int)((m[k] >> 16) & 0x7FFFu);
            if (len) { mn = len < mn ? len : mn; } else ++empty;
            mx = len > mx ? len : mx;
            rgmax = rg > rgmax ? rg : rgmax;
            if ((m[k] & 0xFFFFu) != len0) { ++viol; ++tviol; }
            if ((m[k] >> 31) != 0u) ++tviol;
        }
        if ((mm.x >> 31) != 0u) ++viol;
        if (has2 && ((mm.y >> 31) == 0u || ((mm.x ^ mm.y) & 0x7FFF0000u) != 0u)) ++viol;
        if (!has2) ++viol;                                            // (a lone last single-end read gets an empty partner: no twin violation)
        if (has2 && ((mm.x ^ mm.y) & 0x7FFF0000u) != 0u) ++tviol;
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        int o;
        o = __shfl_xor(mn, off); mn = o < mn ? o : mn;
        o = __shfl_xor(mx, off); mx = o > mx ? o : mx;
        o = __shfl_xor(rgmax, off); rgmax = o > rgmax ? o : rgmax;
        viol += __shfl_xor(viol, off);
        empty += __shfl_xor(empty, off);
        tviol += __shfl_xor(tviol, off);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&p.stats[0], mn); atomicMax(&p.stats[1], mx); atomicMax(&p.stats[2], rgmax);
        if (viol) atomicAdd(&p.stats[3], viol);
        if (empty) atomicAdd(&p.stats[4], empty);
        if (tviol) atomicAdd(&p.stats[5], tviol);
    }
}

// ---------------------------------------------------------------- stable counting sort of the rows by read group
// A row is a read, or (pairs) a pair of reads whose read group is the first mate's.  Workgroup b owns rows
// [b * K7_SORT_ROWS, +K7_SORT_ROWS); wave w of it the contiguous sub-range [w * 256, +256), walked 64 rows at a time,
// so "earlier row" = (earlier workgroup, earlier wave, earlier step, lower lane) and the sort is stable.
//   pass 1 (SCATTER = false): hist[rg * nblocks + b] = rows of group rg in workgroup b
//   k7_rg_scan:               exclusive scan of hist in that (rg-major) order -> the first output slot of every
//                             (rg, b); seg[rg] = first slot of group rg, seg[R] = nrows
//   pass 2 (SCATTER = true):  perm[slot] = row
#define K7_SORT_THREADS 1024
#define K7_SORT_ROWS 4096
#define K7_SORT_MAXR 256
struct RgSortParams {
    const u32* meta; long long nrows; int pairs; int R; long long nblocks;
    u32* hist;                 // [R][nblocks]
    long long* perm; u64* status;
};

template <bool SCATTER>
__global__ __launch_bounds__(K7_SORT_THREADS) void k7_rg_sort(RgSortParams p)
{
    __shared__ u32 cnt_[(K7_SORT_THREADS / 64) * K7_SORT_MAXR];       // [wave][rg]
    volatile u32* cnt = cnt_;                                         // written by a wave's leader lane, read by its other lanes
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = K7_SORT_THREADS / 64;
    const int R = p.R;
    for (int i = threadIdx.x; i < nwaves * R; i += blockDim.x) cnt[i] = 0u;
    __syncthreads();
    const long long b = blockIdx.x;
    const long long row0 = b * K7_SORT_ROWS + (long long)wave * (K7_SORT_ROWS / nwaves);
    const int steps = K7_SORT_ROWS / nwaves / 64;
    const u64 below = (1ull << lane) - 1ull;
    int rgs[K7_SORT_ROWS / K7_SORT_THREADS];
    // count (both passes): cnt[wave][rg]
#pragma unroll
    for (int t = 0; t < steps; ++t) {
        const long long row = row0 + 64 * t + lane;
        const bool valid = row < p.nrows;
        int rg = valid ? (int)((p.meta[p.pairs ? 2 * row : row] >> 16) & 0x7FFFu) : -1;
        if (valid && rg >= R) { flag(p.status, ST_RANGE, row); rg = -1; }
        rgs[t] = rg;
        u64 todo = __ballot(rg >= 0);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int v = __builtin_amdgcn_readlane(rg, leader);
            const u64 m = __ballot(rg == v);
            if (lane == leader) cnt[wave * R + v] += (u32)__popcll(m);
            todo &= ~m;
        }
    }
    __syncthreads();
    if (!SCATTER) {
        for (int v = threadIdx.x; v < R; v += blockDim.x) {
            u32 s = 0u;
            for (int w = 0; w < nwaves; ++w) s += cnt[w * R + v];
            p.hist[(size_t)v * p.nblocks + b] = s;
        }
        return;
    }
    // first slot of (wave, rg): the scanned histogram entry of (rg, b) + the counts of the earlier waves
    for (int v = threadIdx.x; v < R; v += blockDim.x) {
        u32 run = p.hist[(size_t)v * p.nblocks + b];
        for (int w = 0; w < nwaves; ++w) { const u32 c = cnt[w * R + v]; cnt[w * R + v] = run; run += c; }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < steps; ++t) {
        const long long row = row0 + 64 * t + lane;
        const int rg = rgs[t];
        u64 todo = __ballot(rg >= 0);
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const int v = __builtin_amdgcn_readlane(rg, leader);
            const u64 m = __ballot(rg == v);
            const u32 base = cnt[wave * R + v];
            if (rg == v) p.perm[base + (u32)__popcll(m & below)] = row;
            if (lane == leader) cnt[wave * R + v] = base + (u32)__popcll(m);
            todo &= ~m;
        }
    }
}

struct RgScanParams { u32* hist; long long count; long long nblocks; int R; long long nrows; long long* seg; };

// one workgroup: exclusive scan of `count` = R * nblocks entries (rows < 2^32), thread t owning a contiguous slice
__global__ __launch_bounds__(1024) void k7_rg_scan(RgScanParams p)
{
    __shared__ u64 part[1024];
    const long long per = (p.count + blockDim.x - 1) / blockDim.x;
    const long long lo = (long long)threadIdx.x * per, hi = lo + per < p.count ? lo + per : p.count;
    u64 s = 0;
    for (long long i = lo; i < hi; ++i) s += p.hist[i];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 run = 0;
        for (int t = 0; t < (int)blockDim.x; ++t) { const u64 c = part[t]; part[t] = run; run += c; }
    }
    __syncthreads();
    u64 run = part[threadIdx.x];
    for (long long i = lo; i < hi; ++i) {
        const u32 c = p.hist[i];
        p.hist[i] = (u32)run;
        run += c;
    }
    __threadfence_block();
    __syncthreads();
    for (int v = threadIdx.x; v < p.R; v += blockDim.x) p.seg[v] = (long long)p.hist[(size_t)v * p.nblocks];
    if (threadIdx.x == 0) p.seg[p.R] = p.nrows;
}

// ---------------------------------------------------------------- the layout pass
// lane <-> one 16-byte chunk of a DESTINATION row (16 bases: 16 bytes of the qual plane, 16 bytes or -- nibble
// planes -- 8 bytes of seq / cseq).  Destination row d is source row perm[d] (rows grouped by read group) or d.
// pairs: source rows 2r and 2r + 1 (S bases each) -> [mate 1][sep][mate 2][pad] as k7_pack_pairs.
// Bytes past a read's length are rewritten as the layout contract wants them ('N' / quality 0) whatever the source
// holds there.  nib: a seq or cseq character outside ACGTN cannot be packed -> status ST_LUT (the caller keeps byte
// planes for this batch).
struct LayOutParams {
    const uint8_t* src[3]; uint8_t* dst[3];           // seq, cseq (may be NULL), qual
    const u32* meta; u32* dmeta;
    const long long* perm;
    long long nrows;                                  // destination rows
    long long nsrc;                                   // source reads (pairs: 2 * nrows, or one less -- the last row's second half stays empty)
    int pitch, dpitch, S, pairs, nib;
    int parts;                                        // sequential fronts of the traversal (k7_lay_out; KBBQ_K7_PARTS, default 8)
    u64* status;
};

// byte i <- byte i - nb (0 <= nb <= 15), zero fill: the inverse of shr_bytes16
__device__ __forceinline__ void shl_bytes16(u32 v[4], int nb)
{
    const int ws = nb >> 2; const u32 bs = (u32)(nb & 3) * 8u;
    const u32 a3 = ws == 0 ? v[3] : ws == 1 ? v[2] : ws == 2 ? v[1] : v[0];
    const u32 a2 = ws == 0 ? v[2] : ws == 1 ? v[1] : ws == 2 ? v[0] : 0u;
    const u32 a1 = ws == 0 ? v[1] : ws == 1 ? v[0] : 0u;
    const u32 a0 = ws == 0 ? v[0] : 0u;
    // (hi:lo) >> (32 - bs) for bs in {0, 8, 16, 24}; bs == 0 keeps the word
    v[3] = bs ? __builtin_amdgcn_alignbit(a3, a2, 32u - bs) : a3;
    v[2] = bs ? __builtin_amdgcn_alignbit(a2, a1, 32u - bs) : a2;
    v[1] = bs ? __builtin_amdgcn_alignbit(a1, a0, 32u - bs) : a1;
    v[0] = a0 << bs;
}

// 16 bytes from plane[at .. at + 16) where `at` may be up to 15 bytes before the plane's start or run up to 15 bytes past
// its end (`limit` bytes): ONE unconditional 16-byte load at the clamped offset, then a byte shift -- bytes outside the
// plane come back zero.  (A load under a branch parks the wave at the branch's end; see kbbq_aligned_kernels.h.)
__device__ __forceinline__ void load16_clamped(const uint8_t* plane, long long at, long long limit, u32 out[4])
{
    const long long hi = limit - 16;                                   // limit >= 16: the caller's planes are at least one chunk
    const long long c = at < 0 ? 0 : (at > hi ? hi : at);
    load16_any(plane, c, out);
    if (at < 0) shl_bytes16(out, (int)(-at));
    else if (at > hi) shr_bytes16(out, (int)(at - hi));
}

// a workgroup takes 256 / cpr whole destination rows per iteration: (row slot, chunk) of a thread are fixed, no division
// per chunk; every load is unconditional
__global__ __launch_bounds__(256) void k7_lay_out(LayOutParams p)
{
    const int cpr = p.dpitch >> 4;
    const int S = p.S;
    const int rpb = cpr <= 256 ? 256 / cpr : 1;
    const int slot = cpr <= 256 ? (int)threadIdx.x / cpr : 0;
    const int j0 = (int)threadIdx.x - slot * cpr;
    const bool idle = cpr <= 256 && slot >= rpb;
    const long long limit = p.nsrc * (long long)p.pitch;                // bytes in a source plane (perm is a permutation)
    // The workgroups walk the destination rows as `parts` sequential fronts, workgroup b in part b % parts (round 3): neighbouring blocks of
    // rows -- 13 rows of 304 bytes do not end on a cache line -- are then written by workgroups of the SAME XCD (numbers 8 apart), whose partial
    // lines meet in one L2, and a read + write traversal likes several fronts better than one (kbbq_k2_tile.h).  parts <= 1: one front.
    long long rb0 = (long long)blockIdx.x * rpb, rb_step = (long long)gridDim.x * rpb, rb_end = p.nrows;
    if (p.parts > 1 && (int)gridDim.x % p.parts == 0) {
        const long long per = ((p.nrows + p.parts - 1) / p.parts + rpb - 1) / rpb * rpb;      // rows of a part: whole blocks
        const int part = (int)blockIdx.x % p.parts;
        rb0 = part * per + (long long)((int)blockIdx.x / p.parts) * rpb;
        rb_step = (long long)((int)gridDim.x / p.parts) * rpb;
        rb_end = (part + 1) * per < p.nrows ? (part + 1) * per : p.nrows;
    }
    for (long long rb = rb0; rb < rb_end; rb += rb_step) {
        const long long d = rb + slot;
        if (idle || d >= rb_end) continue;
        const long long r = p.perm ? p.perm[d] : d;
        for (int j = j0; j < cpr; j += 256) {
            u32 o[3][4];
            if (p.pairs) {
                if (j == 0) p.dmeta[d] = (u32)(2 * S + 1) | (p.meta[2 * r] & 0x7FFF0000u);
                // bytes [16j, 16j+16) of the pair row: mate 1 from its offset 16j, mate 2 from its offset 16j - S - 1
                const long long at1 = (2 * r) * (long long)p.pitch + (16 * j < p.pitch ? 16 * j : p.pitch - 16);
                const bool lone = 2 * r + 1 >= p.nsrc;                  // an odd number of single-end reads: the last row's second half is padding
                const long long at2 = lone ? 0 : (2 * r + 1) * (long long)p.pitch + (16 * j - S - 1 < -15 ? 0 : 16 * j - S - 1);
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    if (!p.src[pl]) continue;
                    const u32 f4 = pl == 2 ? 0u : 0x4E4E4E4Eu;
                    u32 a[4], b[4];
                    load16_any(p.src[pl], at1, a);
                    load16_clamped(p.src[pl], at2, limit, b);
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const u32 m1 = 16 * j < p.pitch ? byte_mask(S - 16 * j, w) : 0u;
                        const u32 m2 = lone ? 0u : range_mask(S + 1 - 16 * j, 2 * S + 1 - 16 * j, w);
                        o[pl][w] = (a[w] & m1) | (b[w] & m2) | (f4 & ~(m1 | m2));
                    }
                }
            } else {
                const u32 m = p.meta[r];
                const int len = (int)(m & 0xFFFFu);
                if (j == 0) p.dmeta[d] = m;
                const size_t off = (size_t)r * p.pitch + (size_t)16 * j;
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    if (!p.src[pl]) continue;
                    const u32 f4 = pl == 2 ? 0u : 0x4E4E4E4Eu;
                    const uint4 v = *reinterpret_cast<const uint4*>(p.src[pl] + off);
                    const u32 a[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        const u32 m1 = byte_mask(len - 16 * j, w);
                        o[pl][w] = (a[w] & m1) | (f4 & ~m1);
                    }
                }
            }
            *reinterpret_cast<uint4*>(p.dst[2] + (size_t)d * p.dpitch + (size_t)16 * j) = make_uint4(o[2][0], o[2][1], o[2][2], o[2][3]);
#pragma unroll
            for (int pl = 0; pl < 2; ++pl) {
                if (!p.src[pl]) continue;
                if (p.nib) {
                    u32 bad = 0u;
                    const u32 c0 = chars_to_codes(o[pl][0], bad), c1 = chars_to_codes(o[pl][1], bad);
                    const u32 c2 = chars_to_codes(o[pl][2], bad), c3 = chars_to_codes(o[pl][3], bad);
                    if (bad) flag(p.status, ST_LUT, 0);
                    *reinterpret_cast<uint2*>(p.dst[pl] + (size_t)d * (p.dpitch >> 1) + (size_t)8 * j) = make_uint2(c0 | (c1 << 4), c2 | (c3 << 4));
                } else {
                    *reinterpret_cast<uint4*>(p.dst[pl] + (size_t)d * p.dpitch + (size_t)16 * j) = make_uint4(o[pl][0], o[pl][1], o[pl][2], o[pl][3]);
                }
            }
        }
    }
}

// nibble planes back to characters (tests, and callers that want to look at a packed batch): lane <-> 16 bases
struct UnNibParams { const uint8_t* src; uint8_t* dst; long long nchunks; };
__global__ __launch_bounds__(256) void k7_unpack_nibbles(UnNibParams p)
{
    for (long long ch = (long long)blockIdx.x * blockDim.x + threadIdx.x; ch < p.nchunks; ch += (long long)gridDim.x * blockDim.x) {
        const uint2 v = *reinterpret_cast<const uint2*>(p.src + 8 * ch);
        *reinterpret_cast<uint4*>(p.dst + 16 * ch) = make_uint4(codes_to_chars(nib_lo(v.x)), codes_to_chars(nib_hi(v.x)),
                                                                 codes_to_chars(nib_lo(v.y)), codes_to_chars(nib_hi(v.y)));
    }
}

// dst[i] += src[i] over int64 count tables: a length band's tables into the file's (the tables of several bands, or of a
// layout tried first, add up: kbbq_accumulate* itself adds, this is for tables already tallied apart)
struct AddTablesParams { long long* dst; const long long* src; long long n; };
__global__ __launch_bounds__(256) void k7_add_tables(AddTablesParams p)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.n; i += (long long)gridDim.x * blockDim.x) p.dst[i] += p.src[i];
}
