// kbbq_k2_tile.h -- K2 with SHORT-LIVED workgroups: what 4-bit planes run since round 2 (compare_reads.py:320-328 as kbbq_kernels_v3.h k2v3_apply).
//
// Why: the 2 read + 1 written traversal of persistent waves stops at 5.25-5.45 TB/s on this device, one short-lived wave per
// KiB reaches 6.0-6.1 (profiles/r01_traversal_microbench.md); K2 is bound by that traversal (L1 request queue full 85 % of
// the time, profiles/r02_pmc_packed.md).  Here a workgroup of 16 waves stages the pair LUT once, every wave takes K2T_STEPS
// consecutive KiB-steps of chunks (all loads issued up front, counted waits), stores, and the workgroup ends.
// Shapes: 4-bit sequence planes -- mate-pair rows (what the layout pass writes for paired reads of one length) or one read
// per row with the LUT narrowed to the row's pitch (k3_fill_row_lut) as long as that LUT stays small --, one read
// group or rows grouped by read group (a workgroup never straddles two groups: k2t_plan gives every group its own run of
// workgroups), optionally stored through the permutation.  Anything the fast path cannot serve is reported (ST_LUT)
// exactly as k2v3_apply does for pair rows.  Measured (50 M reads, same device, `KBBQ_K2_TILE=0` A/B): 3.83 -> 3.53 ms;
// 2 / 6 / 8 steps per wave 3.79 / 4.65 / 4.87 ms, 512-thread workgroups 3.56 ms.
#pragma once
#include "kbbq_kernels_v3.h"

#ifndef K2T_THREADS
#define K2T_THREADS 1024
#endif
#ifndef K2T_STEPS
#define K2T_STEPS 4
#endif

struct K2tParams {
    const uint8_t* seq; const uint8_t* qual; const u32* meta;
    int xcd_tiles;                 // workgroup b of a batch's T workgroups takes tile (b % 8) * (T / 8) + b / 8: every XCD (workgroup number mod 8) walks a
                                   // contiguous eighth of the planes instead of every eighth tile -- 2.5-3 % on the headline's K2 (KBBQ_K2_XCD_TILES=0: tile b)
    long long nchunks;             // rows * cpr
    int cpr; u32 cpr_magic;        // ceil(2^32 / cpr): exact quotients for the small numerators used below
    int Qt; int S2; int maxlen;
    const int8_t* lut; int lut_bytes; u32 rb; u32 ctx_off;     // lut_bytes: ONE read group's rows
    u32 W;                         // one read per row: offset of the mirrored cycle entries second-in-pair reads use (0 on mate-pair rows)
    const long long* seg;          // rows grouped by read group (NULL: one group, all rows)
    const int* wg_start;           // [R + 1]: first workgroup of every group (k2t_plan); workgroups >= wg_start[R] have nothing to do
    const int2* order;             // optional (stores through perm): workgroup b works as order[b] = (group, workgroup of the group), k2t_order
    int R;
    const long long* perm;         // optional: row i is stored as row perm[i]
    int pitch;
    uint8_t* out; u64* status;
};

// first workgroup of every read group: group g owns ceil(its chunks / chunks per workgroup) workgroups
struct K2tPlanParams { const long long* seg; int R; int cpr; int* wg_start; };
__global__ void k2t_plan(K2tPlanParams p)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const long long per = (long long)(K2T_THREADS / 64) * 64 * K2T_STEPS;
    long long run = 0;
    for (int g = 0; g < p.R; ++g) {
        p.wg_start[g] = (int)run;
        run += ((p.seg[g + 1] - p.seg[g]) * p.cpr + per - 1) / per;
    }
    p.wg_start[p.R] = (int)run;
}

// Stores through the permutation put the rows of every group back between the rows of the others: a group's rows are
// 1 / R of every stretch of the output.  Launched group after group, the output's cache lines are written R times, each
// time in part, long after the previous part has left the memory-side cache.  k2t_order lets the groups advance
// TOGETHER: workgroup (g, i) gets the rank of its relative position (2 i + 1) / (2 n_g) among all workgroups (ties by
// group), so that at any moment the running workgroups of all groups store into the same stretch of the output and
// their partial lines meet in the cache.  One thread per workgroup, R comparisons each.
// ... and on the same XCD: workgroups are handed to the 8 XCDs round-robin by their number (number mod 8), each XCD has an L2 of its
// own, and partial lines of two XCDs cannot meet before the memory side.  Adjacent rows of the output belong to DIFFERENT groups at
// the SAME relative position, i.e. to R consecutive ranks: inside every window of 8 R ranks, rank R q + g (q = 0..7) becomes
// workgroup number q + 8 g of the window, so the R groups' workgroups of one position share XCD q (`xcd` = 8; 0: ranks as they are).
struct K2tOrderParams { const int* wg_start; int R; int2* order; int xcd; };
__global__ __launch_bounds__(256) void k2t_order(K2tOrderParams p)
{
    const int b = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (b >= p.wg_start[p.R]) return;
    int lo = 0, hi = p.R;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (p.wg_start[mid] <= b) lo = mid; else hi = mid; }
    const int g = lo, i = b - p.wg_start[g];
    const long long ng = p.wg_start[g + 1] - p.wg_start[g];
    long long rank = 0;
    for (int h = 0; h < p.R; ++h) {
        const long long nh = p.wg_start[h + 1] - p.wg_start[h];
        if (nh == 0) continue;
        const long long A = (2ll * i + 1) * nh;                       // odd k = 2 j + 1 of group h with k * ng < A (<= A when h < g)
        const long long M = h < g ? A / ng : (A + ng - 1) / ng - 1;   // largest admissible k
        long long cnt = (M + 1) / 2;
        rank += cnt < nh ? cnt : nh;
    }
    if (p.xcd > 0) {
        const long long total = p.wg_start[p.R], win = (long long)p.xcd * p.R;
        const long long base = rank - rank % win;
        if (base + win <= total) { const long long r = rank - base; rank = base + r / p.R + (long long)p.xcd * (r % p.R); }
    }
    p.order[rank] = make_int2(g, i);
}

// the kernel's body for workgroup `bx` of the batch `p` describes (k2t_apply: blockIdx.x; k2t_bands: the workgroup's
// number within its length band)
// NIB = false (round 4): CHARACTER planes -- what rows a caller already holds on the device get (kbbq_apply_dev, one read group)
// and what a batch with a letter outside ACGTN keeps; 16 bytes of sequence per chunk instead of 8, decoded with decode4x as in
// k2v3_apply.  The kernel still only REPORTS what it cannot serve (a foreign letter included: the reference's TypeError is
// decided by the checked kernel the caller runs next), so the persistent kernel remains the form that handles such chunks in
// place -- rows of several read groups that are not grouped take it, as before.
template <bool NIB>
__device__ __forceinline__ void k2t_body(const K2tParams& p, u32* lds, const int bx, const int ntiles)
{
    const int lane = lane_id();
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = K2T_THREADS / 64;
    // this workgroup's read group and chunk range (grouped rows), or everything
    int g = 0;
    long long chunk_lo = 0, chunk_hi = p.nchunks, wg = bx;
    long long tiles = ntiles;                                       // workgroups that share this batch (group): the XCDs' eighths are cut from them
    if (p.seg) {
        if (bx >= p.wg_start[p.R]) return;
        if (p.order) {
            const int2 o = p.order[bx];
            g = o.x; wg = o.y; tiles = 0;                           // (stores through the permutation: k2t_order places the workgroups)
        } else {
            int lo = 0, hi = p.R;                                   // largest g with wg_start[g] <= bx
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (p.wg_start[mid] <= bx) lo = mid; else hi = mid; }
            g = lo;
            wg = (long long)bx - p.wg_start[g];
            tiles = p.wg_start[g + 1] - p.wg_start[g];
        }
        chunk_lo = p.seg[g] * p.cpr; chunk_hi = p.seg[g + 1] * p.cpr;
    }
    if (p.xcd_tiles) {
        const long long P = p.xcd_tiles == 1 ? 8 : p.xcd_tiles;     // parts (1: one per XCD; other counts for the experiment: multiples of 8 give an XCD several)
        tiles -= tiles % P;                                         // whole parts only; the last few workgroups keep their tiles
        if (wg < tiles) wg = (wg % P) * (tiles / P) + wg / P;
    }
    const long long base = chunk_lo + (wg * nwaves + wave) * (64 * K2T_STEPS);
    // chunk -> (row, chunk in row): ONE wave-uniform division for the wave's first chunk, small numerators per lane
    const long long base_c = base < chunk_hi ? base : chunk_hi - 1;
    const long long row0 = base_c / p.cpr;
    const u32 rem0 = (u32)(base_c - row0 * p.cpr);
    // 1. the wave's data loads, all of them, before anything else (clamped at the end of the planes: re-read the last chunk)
    u32 sq[K2T_STEPS][NIB ? 2 : 4], ql[K2T_STEPS][4], mk[K2T_STEPS];
    long long cidx[K2T_STEPS];
#pragma unroll
    for (int s = 0; s < K2T_STEPS; ++s) {
        const long long c = base + 64 * s + lane;
        const long long cc = c < chunk_hi ? c : chunk_hi - 1;
        cidx[s] = c;
        if constexpr (NIB) {
            const uint2 sv = *reinterpret_cast<const uint2*>(p.seq + 8 * cc);
            sq[s][0] = sv.x; sq[s][1] = sv.y;
        } else {
            const uint4 sv = *reinterpret_cast<const uint4*>(p.seq + 16 * cc);
            sq[s][0] = sv.x; sq[s][1] = sv.y; sq[s][2] = sv.z; sq[s][3] = sv.w;
        }
#ifdef KBBQ_Q6_PROBE
        // TIMING ONLY (wrong results): what would K2 gain from qualities packed 16 to 12 bytes?  The loads and the unpacking work of
        // such a plane (three words of 6-bit fields + the fourth word's fields in their spare bit pairs), on whatever bytes lie there
        uint4 qv;
        {
            const uint3 q3 = *reinterpret_cast<const uint3*>(p.qual + 12 * cc);
            qv.x = (q3.x & 0x1F1F1F1Fu) + 0x27272727u; qv.y = (q3.y & 0x1F1F1F1Fu) + 0x27272727u; qv.z = (q3.z & 0x1F1F1F1Fu) + 0x27272727u;
            qv.w = (((q3.x >> 6) & 0x03030303u) | ((q3.y >> 4) & 0x0C0C0C0Cu) | ((q3.z >> 2) & 0x10101010u)) + 0x27272727u;
        }
#else
        const uint4 qv = *reinterpret_cast<const uint4*>(p.qual + 16 * cc);
#endif
        ql[s][0] = qv.x; ql[s][1] = qv.y; ql[s][2] = qv.z; ql[s][3] = qv.w;
        const u32 rel = rem0 + (u32)(cc - base_c);                    // < cpr + 64 * K2T_STEPS
        mk[s] = p.meta[row0 + __umulhi(rel, p.cpr_magic)];
    }
    // the base before the wave's first chunk (its last nibble), for the context of the first base
    const long long c0 = base < chunk_hi ? base : chunk_hi - 1;
    u32 before = 4u;
    if (c0 > 0) {
        if constexpr (NIB) before = (u32)p.seq[8 * c0 - 1] >> 4;
        else {                                                                  // the character's code (A0 T1 G2 C3, anything else 4:
            u32 cd, cd5, expect;                                                // a foreign letter is reported by the lane that owns it)
            const u32 ch4 = (u32)p.seq[16 * c0 - 1] * 0x01010101u;
            decode4x(ch4, cd, cd5, expect);
            before = expect == ch4 ? (cd & 0xFFu) : 4u;
        }
    }
    // 2. the LUT (L2-resident after the first workgroups), then one barrier
    {
        const uint4* src = reinterpret_cast<const uint4*>(p.lut + (size_t)g * (size_t)(33 + p.Qt) * p.rb);
        const int n16 = p.lut_bytes >> 4;
        for (int i = threadIdx.x; i < n16; i += K2T_THREADS) reinterpret_cast<uint4*>(lds)[i] = src[i];
        __syncthreads();
    }
    const u32 rb = p.rb;
    const u32 hi_add = (u32)(0x80 - (p.Qt + 33)) * 0x01010101u;
    u32 carry_code = 5u * before;
#pragma unroll
    for (int s = 0; s < K2T_STEPS; ++s) {
        const long long c = cidx[s];
        const bool act0 = c < chunk_hi;
        const u32 rel = rem0 + (u32)((act0 ? c : chunk_hi - 1) - base_c);
        const u32 drow = __umulhi(rel, p.cpr_magic);
        const long long row = row0 + drow;
        const int j = (int)(rel - drow * (u32)p.cpr);
        const int len = (int)(mk[s] & 0xFFFFu);
        const int nb = act0 ? len - 16 * j : 0;
        u32 code[4], code5[4], hiq = 0u, badbits = 0u;
        if constexpr (NIB) {
            code[0] = nib_lo(sq[s][0]); code[1] = nib_hi(sq[s][0]); code[2] = nib_lo(sq[s][1]); code[3] = nib_hi(sq[s][1]);
            badbits = nib_invalid(sq[s][0]) | nib_invalid(sq[s][1]);
        }
#pragma unroll
        for (int wd = 0; wd < 4; ++wd) {
            if constexpr (NIB) code5[wd] = (code[wd] << 2) + code[wd];
            else {
                u32 expect;
                decode4x(sq[s][wd], code[wd], code5[wd], expect);
                badbits |= expect ^ sq[s][wd];                                   // a letter outside ACGTN (padding is N by the layout contract)
            }
            hiq |= ((ql[s][wd] & 0x7F7F7F7Fu) + hi_add) | ql[s][wd];
        }
        hiq &= 0x80808080u;
        const u32 last_code5 = code5[3] >> 24;
        u32 prev_code5 = wave_shr1(last_code5, carry_code);
        carry_code = (u32)__builtin_amdgcn_readlane((int)last_code5, 63);
        if (j == 0) prev_code5 = 20u;
        if (act0) {
            u32 o[4] = {0u, 0u, 0u, 0u};
            if (nb > 0) {
                const int rg = (int)((mk[s] >> 16) & 0x7FFFu);
                if (hiq != 0u || rg != g || len > p.maxlen || badbits) {
                    flag(p.status, ST_LUT, 0);                       // the caller re-runs on one-read-per-row planes
                } else {
                    u32 d5[4];
                    u32 pc5 = prev_code5 << 24;
#pragma unroll
                    for (int wd = 0; wd < 4; ++wd) {
                        d5[wd] = __builtin_amdgcn_alignbyte(code5[wd], pc5, 3) + code[wd];
                        pc5 = code5[wd];
                    }
                    const u32 A = ((mk[s] >> 31) ? p.W : 0u) + (u32)(16 * j), C = p.ctx_off;
#pragma unroll
                    for (int wd = 0; wd < 4; ++wd) {
                        int v1[4], v2[4];
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const u32 qb = (ql[s][wd] >> (8 * b)) & 0xFFu;
                            const u32 rowq = __umul24(qb, rb);
                            const u32 dd = (d5[wd] >> (8 * b)) & 0xFFu;
                            v1[b] = *reinterpret_cast<const int8_t*>(reinterpret_cast<const char*>(lds) + (rowq + A + (u32)(4 * wd + b)));
                            v2[b] = *reinterpret_cast<const int8_t*>(reinterpret_cast<const char*>(lds) + (rowq + C + dd));
                        }
#pragma unroll
                        for (int b = 0; b < 4; ++b) o[wd] |= (u32)(v1[b] + v2[b] + 33) << (8 * b);
                    }
                }
            }
            uint8_t* dst = p.perm ? p.out + p.perm[row] * p.pitch + 16 * j : p.out + 16 * c;
            *reinterpret_cast<uint4*>(dst) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
}

template <bool NIB = true>
__global__ __launch_bounds__(K2T_THREADS) void k2t_apply(K2tParams p)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    k2t_body<NIB>(p, lds, (int)blockIdx.x, (int)gridDim.x);
}

// ONE launch over all length bands of a mixed-length input (k1v3_bands' counterpart): workgroup blockIdx.x belongs to the
// band whose run of workgroups holds it and stages THAT band's LUT (narrowed to the band's row pitch); the launch's LDS
// size is the largest band's.  Rows not grouped by read group (one read group); anything else keeps a launch per band.
#define K2T_MAX_BANDS 16
struct K2tBandsParams {
    int nbands;
    int wg_start[K2T_MAX_BANDS + 1];
    K2tParams band[K2T_MAX_BANDS];
};

__global__ __launch_bounds__(K2T_THREADS) void k2t_bands(K2tBandsParams t)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    int lo = 0, hi = t.nbands;                                      // largest b with wg_start[b] <= blockIdx.x
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (t.wg_start[mid] <= (int)blockIdx.x) lo = mid; else hi = mid; }
    k2t_body<true>(t.band[lo], lds, (int)blockIdx.x - t.wg_start[lo], t.wg_start[lo + 1] - t.wg_start[lo]);
}
