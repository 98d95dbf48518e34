// kbbq_kernels.h -- gfx950 (CDNA4, wave64) kernels of the kbbq recalibrate hot path.
//
// K1  k1_accumulate : error flagging + covariate binning   (recalibrate.py:57-119)
// K2  k2_apply      : delta-Q LUT lookup, new quality bytes (compare_reads.py:320-328)
// KS  ks_synth      : synthetic reads (bench/tests)
//
// Work decomposition shared by K1/K2 ("read blocks"): a wave owns 64 consecutive
// reads at a time (lane <-> read for the 4-byte sidecar), then walks their rows
// as 16-byte chunks, lane <-> chunk, 64 chunks (1 KiB per plane) per step, so
// every global access is a full-width coalesced dwordx4 load/store.  The chunk's
// read is found with one ds_bpermute of the sidecar; within a chunk the 16 bases
// are handled 4 at a time with byte-parallel (SWAR / v_perm_b32) arithmetic.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define KQ      43          // maxscore + 1 (recalibrate.py:36)
#define KND     16          // dinucleotides
#define K1_THREADS 512
#define K2_THREADS 1024     // launch bound; launched with 256 or 1024 threads depending on the LUT's LDS footprint
#define K1_FLUSH_ITERS 96   // WG iterations between LDS flushes: 96 * 8 waves * 64 reads
                            // = 49,152 reads < 65,535 (16-bit packed LDS counters)

// status words (device, unsigned long long[4]): min read index per error class
#define ST_INDEX 0
#define ST_TYPE  1
#define ST_RANGE 2
#define ST_LUT   3   // set (to 0) when a device-built LUT needs the checked apply kernel

typedef unsigned long long u64;
typedef unsigned int u32;

struct K1Params {
    const uint8_t* seq; const uint8_t* cseq; const uint8_t* qual; const u32* meta;
    long long nreads; int pitch; int cpr; u32 cpr_magic; int R; int S2; int minscore;
    u32 qlo;                   // 33 + minscore: quality byte of the lowest counted score
    u32 dlo;                   // 33 + dinucleotide minscore (only read by the SPLIT variant)
    int pos_stride;            // LDS row stride of the pos table, in 32-bit words
    u64* tables; u64* status;
};

struct K2Params {
    const uint8_t* seq; const uint8_t* qual; const u32* meta;
    long long nreads; int pitch; int cpr; u32 cpr_magic; int R; int Qt; int S2; int minscore;
    u32 qlo;
    const int16_t* lut; int lut_in_lds; int lut_count;
    const void* stage; int stage_bytes;    // what the LDS variant stages: the int16 LUT or its int8 copy
    uint8_t* out; u64* status;
};

struct KSParams {
    uint8_t* seq; uint8_t* cseq; uint8_t* qual; u32* meta;
    long long first; long long nreads; long long total; int pitch; int cpr;
    u64 seed; int len_lo; int len_hi; int nrg; int qlo; int qhi; u32 thr[KQ];
};

// ---------------------------------------------------------------- helpers
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// wave_shr:1 -- every lane receives lane-1's value, lane 0 receives `lane0`.
__device__ __forceinline__ u32 wave_shr1(u32 v, u32 lane0)
{
    return (u32)__builtin_amdgcn_update_dpp((int)lane0, (int)v, 0x138, 0xF, 0xF, false);
}

__device__ __forceinline__ u32 bperm(u32 v, int src_lane)
{
    return (u32)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v);
}

// byte-parallel nucleotide decoding of 4 bases (one 32-bit word of the seq plane).
//   h      = bits 1..3 of every byte:  A->0 C->1 T->2 G->3 N->7   (others 4,5,6 or aliases)
//   expect = the character that h stands for; a byte is in the alphabet iff expect == byte
//   code   = reference order A0 T1 G2 C3 (compare_reads.py:199), 0x10 for N / other
__device__ __forceinline__ void decode4(u32 w, u32& expect, u32& code)
{
    const u32 h = (w >> 1) & 0x07070707u;
    expect = __builtin_amdgcn_perm(0x4E000000u, 0x47544341u, h);
    code   = __builtin_amdgcn_perm(0x10101010u, 0x02010300u, h);
}

// mask with 0xFF in byte k iff (4*wd + k) < nb
__device__ __forceinline__ u32 byte_mask(int nb, int wd)
{
    const int m = nb - 4 * wd;
    return m >= 4 ? 0xFFFFFFFFu : (m <= 0 ? 0u : ((1u << (8 * m)) - 1u));
}

// (the plain load first: a status word only ever decreases, so a value already <= `read` -- however stale the copy this lane sees --
//  means the atomic could not change it.  Without it a batch in which EVERY chunk is flagged -- qualities in another encoding, a
//  kernel run twice over its own output -- spends a second in same-address atomics: 1 010 ms for 50 M reads, measured on K2.)
__device__ __forceinline__ void flag(u64* status, int which, long long read)
{
    if (__hip_atomic_load(&status[which], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > (u64)read) atomicMin(&status[which], (u64)read);
}

// Exact restatement of the reference's TypeError condition for one chunk
// (compare_reads.py:281-293): a dinucleotide (i-1, i) is looked up iff i >= 1,
// q[i] >= minscore and neither base is 'N'; the lookup fails when either base is
// outside ACGT.  Only reached when the cheap alphabet test fired.
__device__ __noinline__ bool chunk_type_error(u32 s0, u32 s1, u32 s2, u32 s3,
                                              u32 q0, u32 q1, u32 q2, u32 q3, u32 prevchar,
                                              int nb, int pos0, int minscore)
{
    const u32 s[4] = {s0, s1, s2, s3};
    const u32 q[4] = {q0, q1, q2, q3};
    u32 prev = prevchar;
    for (int i = 0; i < nb && i < 16; ++i) {
        const u32 ch = (s[i >> 2] >> (8 * (i & 3))) & 0xFFu;
        const int qq = (int)((q[i >> 2] >> (8 * (i & 3))) & 0xFFu) - 33;
        const bool looked = (pos0 + i) >= 1 && qq >= minscore && ch != 'N' && prev != 'N';
        const bool okc = ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T';
        const bool okp = prev == 'A' || prev == 'C' || prev == 'G' || prev == 'T';
        if (looked && !(okc && okp)) return true;
        prev = ch;
    }
    return false;
}

// ---------------------------------------------------------------- K1
// LDS: pos[KQ][pos_stride] u32, (errs << 16 | total) per (q, column)   -- ds_add_u32
//      dn [KQ][16]         u64, (errs << 32 | total) per (q, dinuc)    -- ds_add_u64
// One read group per workgroup (blockIdx.y): reads of other groups are compacted
// away per 64-read block with a ballot + ds_permute.
// SPLIT: the dinucleotide context has its own quality threshold (ReadData semantics,
// read.py:336-369: `skips` decide what is counted, minscore only what has a context).
template <bool SPLIT>
__global__ __launch_bounds__(K1_THREADS) void k1_accumulate(K1Params p)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    const int pos_words = KQ * p.pos_stride;
    const int pos_words_al = (pos_words + 1) & ~1;
    u32* pos = lds;
    u64* dn = reinterpret_cast<u64*>(lds + pos_words_al);

    for (int i = threadIdx.x; i < pos_words_al + 2 * KQ * KND; i += blockDim.x) lds[i] = 0u;
    __syncthreads();

    const int g = blockIdx.y;
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const long long nblocks = (p.nreads + 63) >> 6;
    const long long iters = (nblocks + nwaves - 1) / nwaves;
    const u32 row_bytes = (u32)p.pos_stride * 4u;
    const u32 qlo = p.qlo;
    const u32 qspan = (u32)(KQ - 1 + 33) - qlo;       // counted quality bytes are qlo .. 'K' (q = 42)
    const u32 dn_base = (u32)pos_words_al * 4u;
    int since_flush = 0;

    u64* pos_errs = p.tables;
    u64* pos_total = p.tables + (size_t)p.R * KQ * p.S2;
    u64* dn_errs = p.tables + 2 * (size_t)p.R * KQ * p.S2;
    u64* dn_total = dn_errs + (size_t)p.R * KQ * KND;

    auto flush = [&]() {
        for (int q = wave; q < KQ; q += nwaves) {
            const size_t grow = ((size_t)g * KQ + q) * p.S2;
            for (int col = lane; col < p.S2; col += 64) {
                const u32 v = pos[q * p.pos_stride + col];
                if (v) {
                    pos[q * p.pos_stride + col] = 0u;
                    atomicAdd(&pos_total[grow + col], (u64)(v & 0xFFFFu));
                    if (v >> 16) atomicAdd(&pos_errs[grow + col], (u64)(v >> 16));
                }
            }
        }
        for (int e = threadIdx.x; e < KQ * KND; e += blockDim.x) {
            const u64 v = dn[e];
            if (v) {
                dn[e] = 0ull;
                atomicAdd(&dn_total[(size_t)g * KQ * KND + e], v & 0xFFFFFFFFull);
                if (v >> 32) atomicAdd(&dn_errs[(size_t)g * KQ * KND + e], v >> 32);
            }
        }
    };

    for (long long it = blockIdx.x; it < iters; it += gridDim.x) {
        const long long blk = it * nwaves + wave;
        if (blk < nblocks) {
            const long long read0 = blk << 6;
            const long long myread = read0 + lane;
            const u32 m = myread < p.nreads ? p.meta[myread] : 0u;
            const bool match = myread < p.nreads && (int)((m >> 16) & 0x7FFFu) == g && (m & 0xFFFFu) != 0u;
            const u64 mask = __ballot(match);
            const int n = __popcll(mask);
            // compact the matching reads' (meta, lane) to lanes 0..n-1
            const int rank = __popcll(mask & ((1ull << lane) - 1ull));
            u32 cm = 0u, coff = 0u;
            if (n == 64) { cm = m; coff = (u32)lane; }
            else if (n > 0) {
                const int dst = match ? rank : 63;     // non-matching lanes park on lane 63 ...
                const u32 pm = (u32)__builtin_amdgcn_ds_permute(dst << 2, (int)(match ? m : 0u));
                const u32 po = (u32)__builtin_amdgcn_ds_permute(dst << 2, (int)(match ? (u32)lane : 0u));
                // ... which is only a real slot when n == 64 (handled above)
                cm = pm; coff = po;
            }
            const int total = n * p.cpr;
            u32 carry_code = 0x10u, carry_char = 0u;
            for (int w0 = 0; w0 < total; w0 += 64) {
                const int w = w0 + lane;
                const bool act0 = w < total;
                const int k = act0 ? (p.cpr == 1 ? w : (int)__umulhi((u32)w, p.cpr_magic)) : 0;
                const int j = w - k * p.cpr;
                const u32 mk = bperm(cm, k);
                const u32 off = bperm(coff, k);
                const int len = (int)(mk & 0xFFFFu);
                const bool second = (mk >> 31) != 0u;
                const int pos0 = 16 * j;
                const int nb = act0 ? (len - pos0) : 0;
                const bool act = nb > 0;
                u32 s[4] = {0u, 0u, 0u, 0u}, c[4] = {0u, 0u, 0u, 0u}, q[4] = {0u, 0u, 0u, 0u};
                const long long read = read0 + off;
                if (act) {
                    const size_t rowoff = (size_t)read * p.pitch + (size_t)pos0;
                    const uint4 sv = *reinterpret_cast<const uint4*>(p.seq + rowoff);
                    const uint4 cv = *reinterpret_cast<const uint4*>(p.cseq + rowoff);
                    const uint4 qv = *reinterpret_cast<const uint4*>(p.qual + rowoff);
                    s[0] = sv.x; s[1] = sv.y; s[2] = sv.z; s[3] = sv.w;
                    c[0] = cv.x; c[1] = cv.y; c[2] = cv.z; c[3] = cv.w;
                    q[0] = qv.x; q[1] = qv.y; q[2] = qv.z; q[3] = qv.w;
                }
                // byte-parallel decode; alphabet and q-range screening
                u32 code[4], badbits = 0u, hiq = 0u;
#pragma unroll
                for (int wd = 0; wd < 4; ++wd) {
                    u32 expect;
                    decode4(s[wd], expect, code[wd]);
                    const u32 bm = byte_mask(nb, wd);
                    badbits |= (expect ^ s[wd]) & bm;
                    // any quality byte > 'K' (q > 42)?  (bytes beyond the read are zero)
                    hiq |= (((q[wd] & 0x7F7F7F7Fu) + 0x34343434u) | q[wd]) & 0x80808080u;
                }
                // previous base (code and character) for the chunk's first dinucleotide
                const u32 last_code = code[3] >> 24;
                const u32 last_char = s[3] >> 24;
                u32 prev_code = wave_shr1(last_code, carry_code);
                u32 prev_char = wave_shr1(last_char, carry_char);
                carry_code = (u32)__builtin_amdgcn_readlane((int)last_code, 63);
                carry_char = (u32)__builtin_amdgcn_readlane((int)last_char, 63);
                if (j == 0) { prev_code = 0x10u; prev_char = 0u; }   // dinuc[0] = -1
                if (act) {
                    if (hiq || 2 * len > p.S2) flag(p.status, ST_INDEX, read);   // recalibrate.py:114-115; read longer than the tables
                    if (badbits && chunk_type_error(s[0], s[1], s[2], s[3], q[0], q[1], q[2], q[3], prev_char, nb, pos0, p.minscore))
                        flag(p.status, ST_TYPE, read);                // compare_reads.py:224,292
                    if (2 * len <= p.S2) {
                    // LDS byte address of (q, column): qb*row_bytes + colbase, colbase steps by +-4
                    const int col0 = second ? (2 * len - 1 - pos0) : pos0;
                    const int dcol = second ? -4 : 4;
                    int colbase = col0 * 4 - 33 * (int)row_bytes;
                    const int dnb = (int)dn_base - 33 * 128;
                    u32 pc = prev_code << 24;        // alignbyte below takes byte 3 of the previous word
#pragma unroll
                    for (int wd = 0; wd < 4; ++wd) {
                        const u32 pw = __builtin_amdgcn_alignbyte(code[wd], pc, 3);  // codes of bases i-1
                        const u32 dw = (pw << 2) + code[wd];                         // 4*prev + cur, >= 16: no context
                        pc = code[wd];
                        const u32 xw = s[wd] ^ c[wd];
#pragma unroll
                        for (int b = 0; b < 4; ++b) {
                            const u32 qb = (q[wd] >> (8 * b)) & 0xFFu;
                            const bool valid = (qb - qlo) <= qspan;          // minscore <= q <= 42
                            const bool err = ((xw >> (8 * b)) & 0xFFu) != 0u;  // recalibrate.py:13-20
                            const u32 d = (dw >> (8 * b)) & 0xFFu;
                            if (valid) {
                                const u32 a = __umul24(qb, row_bytes) + (u32)colbase;
                                atomicAdd(reinterpret_cast<u32*>(reinterpret_cast<char*>(lds) + a),
                                          err ? 0x10001u : 1u);              // recalibrate.py:116-117
                                if (d < 16u && (!SPLIT || qb >= p.dlo)) {
                                    const u32 ad = (qb << 7) + (d << 3) + (u32)dnb;
                                    atomicAdd(reinterpret_cast<u64*>(reinterpret_cast<char*>(lds) + ad),
                                              err ? 0x100000001ull : 1ull);  // recalibrate.py:118-119
                                }
                            }
                            colbase += dcol;
                        }
                    }
                    }
                }
            }
        }
        if (++since_flush == K1_FLUSH_ITERS) {
            __syncthreads(); flush(); __syncthreads();
            since_flush = 0;
        }
    }
    __syncthreads();
    flush();
}

// ---------------------------------------------------------------- K2
// LUT (int16), one row of `rs` entries per (read group, quality):
//     row[0 .. S2-1]      = meanq + rgdq + qdq + posdq[cycle]          (lut1)
//     row[S2 .. S2+24]    = dinucdq[5 * code(prev) + code(cur)]        (lut2, code N/none = 4:
//                           every entry that involves code 4 holds dinucdq[..., -1])
// rs = lut_row_stride(S2): rs / 2 is odd so that consecutive rows start on different banks.
__host__ __device__ __forceinline__ int lut_row_stride(int S2)
{
    int rs = (S2 + 25 + 1) & ~1;
    if (((rs >> 1) & 1) == 0) rs += 2;
    return rs;
}

// Exact per-base restatement of compare_reads.py:320-328 for ONE chunk, used whenever the
// fast path cannot be taken (read group / quality / cycle beyond the tables).  Reads the LUT
// from global memory.  Returns the 16 output bytes through o[4].
__device__ __noinline__ uint4 chunk_apply_exact(const int16_t* lut, int rs, int R, int Qt, int S2,
                                               u32 qlo, int rg, bool second, int pos0, int nb,
                                               u32 q0, u32 q1, u32 q2, u32 q3,
                                               u32 d0, u32 d1, u32 d2, u32 d3,
                                               u64* status, long long read)
{
    const u32 q[4] = {q0, q1, q2, q3};
    const u32 d5[4] = {d0, d1, d2, d3};
    u32 o[4] = {0u, 0u, 0u, 0u};
    for (int i = 0; i < 16; ++i) {
        const u32 qb = (q[i >> 2] >> (8 * (i & 3))) & 0xFFu;
        u32 ob = qb;
        if (i < nb && qb >= qlo) {
            const int qq = (int)qb - 33;
            const int pos = pos0 + i;
            const int col = second ? (S2 - 1 - pos) : pos;           // Python wrap of -(i+1) on S2
            if (rg >= R || qq >= Qt || col < 0 || col >= S2) flag(status, ST_INDEX, read);
            else {
                const int16_t* row = lut + ((size_t)rg * Qt + qq) * rs;
                const int v = (int)row[col] + (int)row[S2 + (int)((d5[i >> 2] >> (8 * (i & 3))) & 0xFFu)] + 33;
                if (v < 0 || v > 255) flag(status, ST_RANGE, read);
                ob = (u32)v & 0xFFu;
            }
        }
        if (i >= nb) ob = 0u;
        o[i >> 2] |= ob << (8 * (i & 3));
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

// ELEM: element type of the staged LUT (int16_t canonical, or int8_t copy when every value fits:
// half the LDS, which is what lets 8 read groups of 2x150 tables stay on chip).
template <typename ELEM, bool LDS_LUT, bool CHECK_RANGE>
__global__ __launch_bounds__(K2_THREADS) void k2_apply(K2Params p)
{
    extern __shared__ __attribute__((aligned(16))) u32 lds[];
    if (LDS_LUT) {
        const u32* src = reinterpret_cast<const u32*>(p.stage);
        const int nw = (p.stage_bytes + 3) >> 2;
        for (int i = threadIdx.x; i < nw; i += blockDim.x) lds[i] = src[i];
        __syncthreads();
    }
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    const int nwaves = blockDim.x >> 6;
    const long long nblocks = (p.nreads + 63) >> 6;
    const u32 qlo = p.qlo;
    const int rs = lut_row_stride(p.S2);
    const u32 E = (u32)sizeof(ELEM);
    const u32 rs2 = (u32)rs * E;                                // row stride in bytes
    const u32 hi_add = (u32)(0x80 - (p.Qt + 33)) * 0x01010101u; // byte >= Qt+33  <=>  bit 7 of byte + hi_add

    for (long long blk = (long long)blockIdx.x * nwaves + wave; blk < nblocks;
         blk += (long long)gridDim.x * nwaves) {
        const long long read0 = blk << 6;
        const long long myread = read0 + lane;
        const u32 m = myread < p.nreads ? p.meta[myread] : 0u;
        const int n = (int)((p.nreads - read0) < 64 ? (p.nreads - read0) : 64);
        const int total = n * p.cpr;
        u32 carry_code = 4u, carry_char = 0u;
        for (int w0 = 0; w0 < total; w0 += 64) {
            const int w = w0 + lane;
            const bool act0 = w < total;
            const int k = act0 ? (p.cpr == 1 ? w : (int)__umulhi((u32)w, p.cpr_magic)) : 0;
            const int j = w - k * p.cpr;
            const u32 mk = bperm(m, k);
            const int len = (int)(mk & 0xFFFFu);
            const int rg = (int)((mk >> 16) & 0x7FFFu);
            const bool second = (mk >> 31) != 0u;
            const int pos0 = 16 * j;
            const int nb = act0 ? (len - pos0) : 0;
            const bool act = nb > 0;
            const long long read = read0 + k;
            const size_t rowoff = (size_t)read * p.pitch + (size_t)pos0;
            u32 s[4] = {0u, 0u, 0u, 0u}, q[4] = {0u, 0u, 0u, 0u};
            if (act) {
                const uint4 sv = *reinterpret_cast<const uint4*>(p.seq + rowoff);
                const uint4 qv = *reinterpret_cast<const uint4*>(p.qual + rowoff);
                s[0] = sv.x; s[1] = sv.y; s[2] = sv.z; s[3] = sv.w;
                q[0] = qv.x; q[1] = qv.y; q[2] = qv.z; q[3] = qv.w;
            }
            u32 code[4], badbits = 0u, hiq = 0u;
#pragma unroll
            for (int wd = 0; wd < 4; ++wd) {
                const u32 h = (s[wd] >> 1) & 0x07070707u;
                const u32 expect = __builtin_amdgcn_perm(0x4E000000u, 0x47544341u, h);
                code[wd] = __builtin_amdgcn_perm(0x04040404u, 0x02010300u, h);   // A0 T1 G2 C3, N/other 4
                badbits |= (expect ^ s[wd]) & byte_mask(nb, wd);
                hiq |= (((q[wd] & 0x7F7F7F7Fu) + hi_add) | q[wd]) & 0x80808080u;   // some q >= Qt
            }
            const u32 last_code = code[3] >> 24;
            const u32 last_char = s[3] >> 24;
            u32 prev_code = wave_shr1(last_code, carry_code);
            u32 prev_char = wave_shr1(last_char, carry_char);
            carry_code = (u32)__builtin_amdgcn_readlane((int)last_code, 63);
            carry_char = (u32)__builtin_amdgcn_readlane((int)last_char, 63);
            if (j == 0) { prev_code = 4u; prev_char = 0u; }           // dinuc[0] = -1
            if (act0) {
                u32 o[4] = {0u, 0u, 0u, 0u};
                if (act) {
                    if (badbits && chunk_type_error(s[0], s[1], s[2], s[3], q[0], q[1], q[2], q[3],
                                                    prev_char, nb, pos0, p.minscore))
                        flag(p.status, ST_TYPE, read);
                    // 5 * code(prev) + code(cur) for the 16 bases
                    u32 d5[4];
                    u32 pc = prev_code << 24;
#pragma unroll
                    for (int wd = 0; wd < 4; ++wd) {
                        const u32 pw = __builtin_amdgcn_alignbyte(code[wd], pc, 3);
                        d5[wd] = (pw << 2) + pw + code[wd];
                        pc = code[wd];
                    }
                    const bool trouble = hiq != 0u || rg >= p.R || len > p.S2;
                    if (trouble || !LDS_LUT) {
                        const uint4 e = chunk_apply_exact(p.lut, rs, p.R, p.Qt, p.S2, qlo, rg, second, pos0, nb,
                                                          q[0], q[1], q[2], q[3], d5[0], d5[1], d5[2], d5[3],
                                                          p.status, read);
                        o[0] = e.x; o[1] = e.y; o[2] = e.z; o[3] = e.w;
                    } else {
                        // fast path: every index is in range; bytes beyond the read are 0 (< qlo) and pass through
                        const u32 rowbase = (u32)rg * (u32)p.Qt * rs2 - 33u * rs2;
                        u32 colb = rowbase + (u32)(second ? (p.S2 - 1 - pos0) : pos0) * E;
                        const u32 dcol = second ? 0u - E : E;
                        const u32 dnb = rowbase + (u32)p.S2 * E;
                        bool range_err = false;
#pragma unroll
                        for (int wd = 0; wd < 4; ++wd) {
#pragma unroll
                            for (int b = 0; b < 4; ++b) {
                                const u32 qb = (q[wd] >> (8 * b)) & 0xFFu;
                                const u32 qc = qb > 33u ? qb : 33u;               // keep the address inside the LUT
                                const u32 rowq = __umul24(qc, rs2);
                                const u32 dd = (d5[wd] >> (8 * b)) & 0xFFu;
                                const int v1 = *reinterpret_cast<const ELEM*>(reinterpret_cast<const char*>(lds) + (rowq + colb));
                                const int v2 = *reinterpret_cast<const ELEM*>(reinterpret_cast<const char*>(lds) + (rowq + dnb + dd * E));
                                const int v = v1 + v2 + 33;
                                const bool counted = qb >= qlo;
                                if (CHECK_RANGE) range_err |= counted && (u32)v > 255u;
                                const u32 ob = counted ? ((u32)v & 0xFFu) : qb;
                                o[wd] |= ob << (8 * b);
                                colb += dcol;
                            }
                        }
                        if (CHECK_RANGE && range_err) flag(p.status, ST_RANGE, read);
                    }
                }
                *reinterpret_cast<uint4*>(p.out + rowoff) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
    }
}

// ---------------------------------------------------------------- KS
__device__ __forceinline__ u64 mix64(u64 x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// one lane per 16-byte chunk; spec in oracle/kbbq_oracle.c (written independently here)
__global__ __launch_bounds__(256) void ks_synth(KSParams p)
{
    const long long nchunks = p.nreads * p.cpr;
    for (long long ch = (long long)blockIdx.x * blockDim.x + threadIdx.x; ch < nchunks;
         ch += (long long)gridDim.x * blockDim.x) {
        const long long k = ch / p.cpr;
        const int j = (int)(ch - k * p.cpr);
        const u64 read = (u64)(p.first + k);
        const int len = p.len_lo + (int)((read * (u64)(p.len_hi - p.len_lo + 1)) / (u64)p.total);
        if (j == 0) {
            const u32 rg = (u32)((read >> 1) % (u64)p.nrg);
            p.meta[k] = (u32)len | (rg << 16) | ((u32)(read & 1ull) << 31);
        }
        const u64 hr = mix64(p.seed + read);
        const u64 nq = (u64)(p.qhi - p.qlo + 1);
        const u32 NNNN = 0x4E4E4E4Eu;      // bytes past the read: 'N' in seq / cseq, 0 in qual
        u32 s[4] = {NNNN, NNNN, NNNN, NNNN}, c[4] = {NNNN, NNNN, NNNN, NNNN}, q[4] = {0u, 0u, 0u, 0u};
        for (int b = 0; b < 16; ++b) {
            const int i = 16 * j + b;
            if (i >= len) break;
            const u64 r = mix64(hr ^ (u64)i);
            const u32 bb = (u32)(r & 3ull);
            const bool isn = ((r >> 2) & 1023ull) == 0ull;
            const int qq = (int)((((r >> 12) & 0xFFFFull) * nq) >> 16) + p.qlo;
            const bool err = (u32)(r >> 32) < p.thr[qq];
            const u32 sub = (bb + 1u + (u32)(((r >> 28) & 15ull) % 3ull)) & 3u;
            const u32 acgt = 0x54474341u;   // 'A','C','G','T'
            const u32 sc = isn ? (u32)'N' : ((acgt >> (8 * bb)) & 0xFFu);
            const u32 cc = err ? (isn ? ((acgt >> (8 * bb)) & 0xFFu) : ((acgt >> (8 * sub)) & 0xFFu)) : sc;
            s[b >> 2] = (s[b >> 2] & ~(0xFFu << (8 * (b & 3)))) | (sc << (8 * (b & 3)));
            c[b >> 2] = (c[b >> 2] & ~(0xFFu << (8 * (b & 3)))) | (cc << (8 * (b & 3)));
            q[b >> 2] |= (u32)(qq + 33) << (8 * (b & 3));
        }
        const size_t off = (size_t)k * p.pitch + (size_t)16 * j;
        *reinterpret_cast<uint4*>(p.seq + off) = make_uint4(s[0], s[1], s[2], s[3]);
        *reinterpret_cast<uint4*>(p.cseq + off) = make_uint4(c[0], c[1], c[2], c[3]);
        *reinterpret_cast<uint4*>(p.qual + off) = make_uint4(q[0], q[1], q[2], q[3]);
    }
}
