// parallel_gunzip.cpp -- a gzip member inflated on many threads (host only, no GPU code).
//
// Why: the reference reads `.fastq.gz` through pysam (recalibrate.py:56,141), and that is what sequencing data usually is.  A gzip
// member is ONE DEFLATE stream: zlib inflates it at 0.2-0.4 GB/s of text on one thread, libdeflate (fast_inflate.h) at 2-2.5x that,
// still one thread -- against 10 GB/s and more for the packer on plain text.  The stream can be cut, though (the two-pass scheme of
// Kerbiriou & Chikhi's pugz and Knespel & Brunst's rapidgzip, written from their published descriptions):
//
//   1. the compressed bytes are cut into chunks; for every chunk but the first a thread SEARCHES the bit position of a DEFLATE block
//      header at or behind the cut: a position where a dynamic-Huffman header parses into complete, non-over-subscribed codes with an
//      end-of-block symbol and the whole block decodes, followed by another plausible header.  Chance positions fail these tests
//      within a few bits; the rare survivor is caught in step 3;
//   2. every chunk is decoded from its block start WITHOUT knowing the 32 KiB of text before it: the output is 16-bit symbols, a
//      literal byte or a MARKER "byte k of the unknown window"; back-references copy symbols, markers included.  A chunk stops at a
//      block boundary that must be EXACTLY the next chunk's start (else the whole call fails and zlib takes over);
//   3. in file order (32 KiB per chunk, sequential), then side by side (everything else), the markers are replaced through the now
//      known window -- one 64 K-entry table per chunk -- while the bytes go to their place in the output and their CRC-32 is taken;
//      the chunks' CRCs are combined and compared with the member's trailer, as is the size.
//
// Memory is bounded: a WINDOW of 2 x threads chunks is in flight (kbbq_pgz_next), whatever the file's size.  Everything that is
// not plain success -- damaged data, a boundary that did not meet, a trailer that does not match -- is reported as "not taken": the
// callers fall back to zlib from the start of the input, which also words the error of a damaged file.
#include "parallel_gunzip.h"
#include "fast_inflate.h"
#include "host_threads.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#include <zlib.h>

namespace {

constexpr size_t WIN = 32768;
constexpr uint16_t MARK = 0x8000;                   // symbol >= MARK: byte (symbol - MARK) of the window before the chunk
constexpr size_t NONE = ~(size_t)0;

// ---------------------------------------------------------------- bits, LSB first
struct Bits {
    const uint8_t* base; size_t nbytes; size_t pos; uint64_t buf; unsigned cnt; bool over;
    void init(const uint8_t* b, size_t n, size_t bitpos)
    {
        base = b; nbytes = n; pos = bitpos >> 3; buf = 0; cnt = 0; over = false;
        refill();
        drop((unsigned)(bitpos & 7));
    }
    inline void refill()
    {
        if (pos + 8 <= nbytes) {
            uint64_t w; memcpy(&w, base + pos, 8);
            buf |= w << cnt;                         // bits above cnt are OR-ed again, with the same values, by the next refill
            const unsigned adv = (63 - cnt) >> 3;
            pos += adv; cnt += adv << 3;
        } else {
            while (cnt <= 56 && pos < nbytes) { buf |= (uint64_t)base[pos++] << cnt; cnt += 8; }
        }
    }
    inline uint32_t peek(unsigned n) const { return (uint32_t)(buf & (((uint64_t)1 << n) - 1)); }
    inline void drop(unsigned n) { if (n > cnt) { over = true; cnt = 0; buf = 0; return; } buf >>= n; cnt -= n; }
    inline uint32_t take(unsigned n) { const uint32_t v = peek(n); drop(n); return v; }
    size_t bitpos() const { return pos * 8 - cnt; }
    void align_byte() { drop(cnt & 7); }
};

// ---------------------------------------------------------------- canonical Huffman, two-level tables
// entry: symbol << 16 | code length (1..15); 0 = no such code; a primary entry with bit 15 set links a subtable:
// start << 16 | 0x8000 | subtable bits
struct Huff {
    std::vector<uint32_t> t; unsigned pbits = 0;
    std::vector<uint8_t> sub;                         // scratch: longest (len - pbits) per primary slot
};

inline unsigned rev_bits(unsigned code, unsigned len)
{
    unsigned r = 0;
    for (unsigned i = 0; i < len; ++i) { r = (r << 1) | (code & 1); code >>= 1; }
    return r;
}

// false: over-subscribed, or incomplete where zlib does not accept it (it accepts an incomplete code only when its longest length is
// 1, and never for the code-length code: inftrees.c)
bool build(const uint8_t* lens, unsigned n, unsigned pbits, Huff& h, bool strict)
{
    unsigned count[16] = {0}, next[16];
    for (unsigned i = 0; i < n; ++i) ++count[lens[i]];
    count[0] = 0;
    unsigned maxlen = 0;
    for (unsigned l = 1; l <= 15; ++l) if (count[l]) maxlen = l;
    h.pbits = pbits;
    const unsigned psize = 1u << pbits;
    if (maxlen == 0) {                                // no codes at all: fine for distances (a block of literals only)
        if (strict) return false;
        h.t.assign(psize, 0);
        return true;
    }
    int left = 1;
    for (unsigned l = 1; l <= 15; ++l) { left <<= 1; left -= (int)count[l]; if (left < 0) return false; }
    if (left > 0 && (strict || maxlen != 1)) return false;
    unsigned code = 0;
    for (unsigned l = 1; l <= 15; ++l) { code = (code + count[l - 1]) << 1; next[l] = code; }
    size_t total = psize;
    if (maxlen > pbits) {
        h.sub.assign(psize, 0);
        unsigned nx[16]; memcpy(nx, next, sizeof nx);
        for (unsigned s = 0; s < n; ++s) {
            const unsigned l = lens[s];
            if (!l) continue;
            const unsigned c = nx[l]++;
            if (l > pbits) { const unsigned slot = rev_bits(c, l) & (psize - 1); h.sub[slot] = std::max<uint8_t>(h.sub[slot], (uint8_t)(l - pbits)); }
        }
        for (unsigned p = 0; p < psize; ++p) if (h.sub[p]) total += (size_t)1 << h.sub[p];
    }
    h.t.assign(total, 0);
    size_t cur = psize;
    if (maxlen > pbits)
        for (unsigned p = 0; p < psize; ++p)
            if (h.sub[p]) { h.t[p] = (uint32_t)(cur << 16) | 0x8000u | h.sub[p]; cur += (size_t)1 << h.sub[p]; }
    for (unsigned s = 0; s < n; ++s) {
        const unsigned l = lens[s];
        if (!l) continue;
        const unsigned r = rev_bits(next[l]++, l);
        const uint32_t e = (uint32_t)s << 16 | l;
        if (l <= pbits) {
            for (unsigned k = r; k < psize; k += 1u << l) h.t[k] = e;
        } else {
            const uint32_t link = h.t[r & (psize - 1)];
            const unsigned sb = link & 15, hi = r >> pbits, hl = l - pbits;
            uint32_t* st = h.t.data() + (link >> 16);
            for (unsigned k = hi; k < (1u << sb); k += 1u << hl) st[k] = e;
        }
    }
    return true;
}

inline uint32_t decode_sym(const Huff& h, const Bits& b)
{
    uint32_t e = h.t[b.buf & ((1u << h.pbits) - 1)];
    if (e & 0x8000u) e = h.t[(e >> 16) + ((b.buf >> h.pbits) & ((1u << (e & 15)) - 1))];
    return e;
}

const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
const uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
const uint8_t kPreOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct Codes { Huff lit, dist, pre; };

// the header of a dynamic block (behind its 3 type bits) -> c.lit / c.dist
bool read_dynamic(Bits& b, Codes& c)
{
    b.refill();
    const unsigned hlit = b.take(5) + 257, hdist = b.take(5) + 1, hclen = b.take(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    uint8_t pl[19] = {0};
    for (unsigned i = 0; i < hclen; ++i) { if (b.cnt < 3) b.refill(); pl[kPreOrder[i]] = (uint8_t)b.take(3); }
    if (b.over || !build(pl, 19, 7, c.pre, true)) return false;
    uint8_t lens[286 + 30];
    unsigned i = 0;
    const unsigned total = hlit + hdist;
    while (i < total) {
        b.refill();
        const uint32_t e = decode_sym(c.pre, b);
        const unsigned l = e & 0xFF;
        if (!l) return false;
        b.drop(l);
        const unsigned s = e >> 16;
        if (s < 16) { lens[i++] = (uint8_t)s; continue; }
        unsigned rep, val = 0;
        if (s == 16) { if (i == 0) return false; val = lens[i - 1]; rep = 3 + b.take(2); }
        else if (s == 17) rep = 3 + b.take(3);
        else rep = 11 + b.take(7);
        if (i + rep > total) return false;
        memset(lens + i, (int)val, rep); i += rep;
    }
    if (b.over || lens[256] == 0) return false;
    return build(lens, hlit, 11, c.lit, false) && build(lens + hlit, hdist, 8, c.dist, false);
}

void fixed_codes(Codes& c)
{
    uint8_t l[288];
    for (unsigned i = 0; i < 144; ++i) l[i] = 8;
    for (unsigned i = 144; i < 256; ++i) l[i] = 9;
    for (unsigned i = 256; i < 280; ++i) l[i] = 7;
    for (unsigned i = 280; i < 288; ++i) l[i] = 8;
    build(l, 288, 11, c.lit, false);
    uint8_t d[32]; memset(d, 5, sizeof d);
    build(d, 32, 8, c.dist, false);
}

// ---------------------------------------------------------------- one chunk
struct Chunk {
    size_t start_bit = NONE;                 // where its first block header is (NONE: none found, the chunk before it runs through)
    raw_vector<uint16_t> sym;                // [WIN of window | output symbols]
    size_t n = 0;                            // output symbols (behind the WIN of window)
    size_t end_bit = 0;                      // where decoding stopped (a block boundary)
    int end = -1;                            // 0: at the requested stop / limit; 1: the member's last block ended; -1: failed
    bool markers = true;                     // decoded against an unknown window
    size_t limit = ~(size_t)0;               // symbols a chunk may become: text that inflates 100-fold is left to zlib's constant memory
    uint32_t crc = 0;
};

// One block's symbols through the codes c into ch (or nowhere: `count_only`, the finder's trial).  false: invalid data.
template <bool COUNT_ONLY>
inline bool inflate_block(Bits& b, const Codes& c, Chunk& ch, size_t& o, size_t floor_o, size_t& counted)
{
    uint16_t* out = COUNT_ONLY ? nullptr : ch.sym.data();
    size_t cap = COUNT_ONLY ? 0 : ch.sym.size();
    for (;;) {
        if (b.over) return false;                                 // the input ended inside the block
        b.refill();
        uint32_t e = decode_sym(c.lit, b);
        unsigned l = e & 0xFF;
        if (!l) return false;
        b.drop(l);
        unsigned s = e >> 16;
        if (s < 256) {
            if (COUNT_ONLY) { ++counted; continue; }
            if (o + 1 > cap) { if (cap > ch.limit) return false; ch.sym.resize(cap * 2); out = ch.sym.data(); cap = ch.sym.size(); }
            out[o++] = (uint16_t)s;
            // a second literal from the same refill (two litlen codes are at most 30 of the >= 56 bits)
            e = decode_sym(c.lit, b); l = e & 0xFF;
            if (!l) return false;
            s = e >> 16;
            if (s >= 256) { b.drop(l); goto not_literal; }
            b.drop(l);
            if (o + 1 > cap) { if (cap > ch.limit) return false; ch.sym.resize(cap * 2); out = ch.sym.data(); cap = ch.sym.size(); }
            out[o++] = (uint16_t)s;
            continue;
        }
    not_literal:
        if (s == 256) return !b.over;
        if (s > 285) return false;
        if (b.cnt < 5 + 15 + 13) b.refill();
        const unsigned len = kLenBase[s - 257] + b.take(kLenExtra[s - 257]);
        const uint32_t de = decode_sym(c.dist, b);
        const unsigned dl = de & 0xFF;
        if (!dl) return false;
        b.drop(dl);
        const unsigned ds = de >> 16;
        if (ds > 29) return false;
        const size_t dist = kDistBase[ds] + b.take(kDistExtra[ds]);
        if (b.over) return false;
        if (COUNT_ONLY) { counted += len; continue; }
        if (dist > o - floor_o) return false;                    // further back than the member's own text (only known for its first chunk)
        if (o + len > cap) { if (cap > ch.limit) return false; ch.sym.resize(std::max(cap * 2, o + len)); out = ch.sym.data(); cap = ch.sym.size(); }
        const uint16_t* from = out + o - dist;
        uint16_t* to = out + o;
        if (dist >= len) memcpy(to, from, (size_t)len * 2);
        else for (unsigned i = 0; i < len; ++i) to[i] = from[i];
        o += len;
    }
}

// Decode from start_bit, block after block, until the boundary `stop_bit` (it must be met exactly), or -- stop_bit NONE -- the first
// boundary at or behind `limit_bit`, or the member's last block.  window: the WIN bytes of text before the chunk (nullptr: unknown,
// markers), of which only the last `valid` are the member's own.
void decode_chunk(const uint8_t* src, size_t nbytes, size_t stop_bit, size_t limit_bit, const uint8_t* window, size_t valid, Chunk& ch, size_t guess)
{
    ch.end = -1; ch.n = 0;
    ch.markers = window == nullptr;
    ch.limit = std::max<size_t>(guess * 8, (size_t)32 << 20);      // (a chunk's share of the input inflating more than 40-fold: not sequencing text)
    if (ch.sym.size() < WIN + std::max<size_t>(guess, 1 << 16)) kbbq_resize_fresh(ch.sym, WIN + std::max<size_t>(guess, 1 << 16));   // (kept from window to window: fresh pages cost more than the decoding)
    uint16_t* w = ch.sym.data();
    if (window) for (size_t i = 0; i < WIN; ++i) w[i] = window[i];
    else for (size_t i = 0; i < WIN; ++i) w[i] = (uint16_t)(MARK | i);
    const size_t floor_o = window ? WIN - valid : 0;
    Bits b; b.init(src, nbytes, ch.start_bit);
    Codes c;
    size_t o = WIN, dummy = 0;
    for (;;) {
        b.refill();
        const unsigned final_block = b.take(1), type = b.take(2);
        if (b.over) return;
        if (type == 0) {
            b.align_byte(); b.refill();
            const unsigned len = b.take(16), nlen = b.take(16);
            if (b.over || (len ^ nlen) != 0xFFFF) return;
            if (o + len > ch.sym.size()) { if (ch.sym.size() > ch.limit) return; ch.sym.resize(std::max(ch.sym.size() * 2, o + len)); }
            uint16_t* out = ch.sym.data();
            // the stored bytes: first what the bit buffer holds, then straight from the input
            unsigned left = len;
            while (left && b.cnt >= 8) { out[o++] = (uint16_t)b.take(8); --left; }
            if (left) {
                if (b.cnt != 0 || b.pos + left > nbytes) return;
                for (unsigned i = 0; i < left; ++i) out[o + i] = src[b.pos + i];
                o += left; b.pos += left;
                b.buf = 0;                                       // (what it held above its count belonged to the old position)
            }
        } else if (type == 1) {
            fixed_codes(c);
            if (!inflate_block<false>(b, c, ch, o, floor_o, dummy)) return;
        } else if (type == 2) {
            if (!read_dynamic(b, c)) return;
            if (!inflate_block<false>(b, c, ch, o, floor_o, dummy)) return;
        } else return;
        const size_t at = b.bitpos();
        if (final_block) { ch.n = o - WIN; ch.end_bit = at; ch.end = 1; return; }
        if (stop_bit != NONE) {
            if (at == stop_bit) { ch.n = o - WIN; ch.end_bit = at; ch.end = 0; return; }
            if (at > stop_bit) return;
        } else if (at >= limit_bit) { ch.n = o - WIN; ch.end_bit = at; ch.end = 0; return; }
    }
}

// The first bit position in [from_bit, until_bit) where a non-final dynamic block starts -- by every test short of knowing the text
// before it (see the head of this file); NONE if there is none.
size_t find_block(const uint8_t* src, size_t nbytes, size_t from_bit, size_t until_bit)
{
    Codes c; Chunk none;
    until_bit = std::min(until_bit, nbytes >= 24 ? (nbytes - 24) * 8 : 0);      // (the tests below load words ahead)
    for (size_t bit = from_bit; bit < until_bit; ++bit) {
        uint32_t v; memcpy(&v, src + (bit >> 3), 4);
        v >>= bit & 7;
        if ((v & 7) != 4) continue;                                           // BFINAL = 0, BTYPE = 2 (LSB first: 0, then 01 -> bits 0b100)
        if (((v >> 3) & 31) > 29 || ((v >> 8) & 31) > 29) continue;           // HLIT <= 286 - 257, HDIST <= 30 - 1
        {   // the code-length code must be complete: sum of 2^-len over its HCLEN + 4 three-bit lengths == 1 (here in units of 2^-7)
            uint64_t w; memcpy(&w, src + ((bit + 13) >> 3), 8);
            w >>= (bit + 13) & 7;
            const unsigned hclen = (unsigned)(w & 15) + 4;
            w >>= 4;                                                            // 57 - 4 >= 53 bits left... 19 x 3 = 57: the last length may need a second word
            unsigned kraft = 0, i = 0;
            for (; i < hclen && i < 17; ++i) { const unsigned l = (unsigned)(w & 7); w >>= 3; if (l) kraft += 128u >> l; }
            if (i < hclen) {
                uint64_t w2; memcpy(&w2, src + ((bit + 17 + 51) >> 3), 8);
                w2 >>= (bit + 17 + 51) & 7;
                for (; i < hclen; ++i) { const unsigned l = (unsigned)(w2 & 7); w2 >>= 3; if (l) kraft += 128u >> l; }
            }
            if (kraft != 128) continue;
        }
        Bits b; b.init(src, nbytes, bit + 3);
        if (!read_dynamic(b, c)) continue;
        size_t o = 0, counted = 0;
        if (!inflate_block<true>(b, c, none, o, 0, counted) || counted == 0) continue;
        // what follows must look like a block header too
        b.refill();
        const unsigned nf = b.take(1), nt = b.take(2); (void)nf;
        if (b.over || nt == 3) continue;
        if (nt == 0) {
            b.align_byte(); b.refill();
            const unsigned len = b.take(16), nlen = b.take(16);
            if (b.over || (len ^ nlen) != 0xFFFF) continue;
        } else if (nt == 2) {
            if (b.peek(5) > 29 || ((b.buf >> 5) & 31) > 29) continue;
        }
        return bit;
    }
    return NONE;
}

uint32_t crc_of(const uint8_t* p, size_t n)
{
    const kbbq_libdeflate* l = kbbq_libdeflate_get();
    if (l) return l->crc32(0, p, n);
    uLong c = crc32(0L, Z_NULL, 0);
    while (n) { const uInt k = (uInt)std::min<size_t>(n, 1u << 30); c = crc32(c, p, k); p += k; n -= k; }
    return (uint32_t)c;
}

}  // namespace

struct kbbq_pgz {
    const uint8_t* src; size_t n; unsigned threads;
    size_t chunk_bytes;
    size_t pos = 0;                       // byte position between members; bit position / 8 inside one is in `bit`
    bool in_member = false;
    size_t bit = 0;
    uint8_t win[WIN]; size_t valid = 0;   // the member's last WIN bytes of text (the last `valid` of them exist)
    uint32_t crc = 0; uint64_t isize = 0;
    size_t delivered = 0;
    bool failed = false;
    long calls = 0, fail_after = -1;      // KBBQ_PGZ_TEST_FAIL_AFTER: the (n + 1)-th window is "not taken" (the callers' way back to zlib, for the tests)
    size_t k_cap = ~(size_t)0 >> 2;       // chunks per window at most
    std::vector<Chunk> chunks;
    // a decoded window waiting to be written out (kbbq_pgz_prepare -> kbbq_pgz_emit)
    bool ready = false;
    std::vector<size_t> order; std::vector<std::vector<uint8_t>> before;
    size_t total = 0, end_bit = 0; bool member_ends = false; unsigned nt = 1;
    bool trace = false; double ms[4] = {0, 0, 0, 0};      // KBBQ_PGZ_TRACE: search / decode / chain / emit milliseconds of the window (stderr)
};

kbbq_pgz* kbbq_pgz_open(const uint8_t* src, size_t n, unsigned threads)
{
    kbbq_pgz* z = new kbbq_pgz();
    z->src = src; z->n = n;
    z->threads = std::min(threads ? threads : kbbq_host_thread_ceiling(), 32u);     // (a window is 2 x threads chunks of ~25 MB of symbols: 1.6 GB at 32)
    const char* e = getenv("KBBQ_PGZ_CHUNK");                    // compressed bytes per chunk (tests: small chunks on small files)
    z->chunk_bytes = e && atoll(e) >= 64 ? (size_t)atoll(e) : (size_t)1 << 20;
    z->trace = getenv("KBBQ_PGZ_TRACE") != nullptr;
    const char* f = getenv("KBBQ_PGZ_TEST_FAIL_AFTER");
    if (f) z->fail_after = atol(f);
    return z;
}

void kbbq_pgz_close(kbbq_pgz* z) { delete z; }
size_t kbbq_pgz_delivered(const kbbq_pgz* z) { return z->delivered; }

// the gzip header at src[pos]: its length, or 0 if it is not one (RFC 1952)
static size_t gzip_header(const uint8_t* s, size_t n)
{
    if (n < 18 || s[0] != 0x1f || s[1] != 0x8b || s[2] != 8 || (s[3] & 0xE0)) return 0;
    const unsigned flg = s[3];
    size_t at = 10;
    if (flg & 4) { if (at + 2 > n) return 0; at += 2 + ((size_t)s[at] | (size_t)s[at + 1] << 8); }
    if (flg & 8) { while (at < n && s[at]) ++at; ++at; }
    if (flg & 16) { while (at < n && s[at]) ++at; ++at; }
    if (flg & 2) at += 2;
    return at + 8 <= n ? at : 0;
}

int kbbq_pgz_prepare(kbbq_pgz* z, size_t* total_out)
{
    if (z->failed) return -1;
    if (z->ready) { *total_out = z->total; return 1; }
    if (z->fail_after >= 0 && z->calls++ >= z->fail_after) { z->failed = true; return -1; }
    const uint8_t* src = z->src; const size_t n = z->n;
    if (!z->in_member) {
        while (z->pos < n && src[z->pos] == 0) ++z->pos;          // NUL padding behind a member is tolerated, as gzip(1) does
        if (z->pos >= n) return 0;
        const size_t h = gzip_header(src + z->pos, n - z->pos);
        if (!h) { z->failed = true; return -1; }
        z->bit = (z->pos + h) * 8; z->in_member = true; z->valid = 0; z->crc = 0; z->isize = 0;
    }
    // this window's chunks: the first at the known position, the others cut every chunk_bytes behind it
    const size_t first_byte = z->bit >> 3, C = z->chunk_bytes;
    const size_t room = n - first_byte;
    size_t K = std::min<size_t>(std::max<size_t>(2 * z->threads, 2), (room + C - 1) / C);
    K = std::max<size_t>(std::min(K, z->k_cap), 1);              // (a file of many short members: no wider than the last ones were long)
    z->chunks.resize(K);
    std::vector<Chunk>& ch = z->chunks;
    const size_t limit_bit = (first_byte + K * C) * 8;
    ch[0].start_bit = z->bit;
    const unsigned nt = (unsigned)std::min<size_t>(z->threads, K);
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](int i) { if (!z->trace) return; const auto t = std::chrono::steady_clock::now(); z->ms[i] = std::chrono::duration<double, std::milli>(t - t_prev).count(); t_prev = t; };
    {
        std::atomic<size_t> nextk(1);
        kbbq_parallel(nt, [&](unsigned) {
            for (size_t k; (k = nextk.fetch_add(1)) < K;)
                ch[k].start_bit = find_block(src, n, (first_byte + k * C) * 8, (first_byte + (k + 1) * C) * 8);
        });
    }
    lap(0);
    std::vector<size_t> stop(K, NONE);
    { size_t later = NONE; for (size_t k = K; k-- > 0;) { stop[k] = later; if (ch[k].start_bit != NONE) later = ch[k].start_bit; } }
    {
        std::atomic<size_t> nextk(0);
        const size_t guess = C * 5;
        kbbq_parallel(nt, [&](unsigned) {
            for (size_t k; (k = nextk.fetch_add(1)) < K;) {
                if (ch[k].start_bit == NONE) { ch[k].end = -2; continue; }
                decode_chunk(src, n, stop[k], limit_bit, k == 0 ? z->win : nullptr, k == 0 ? z->valid : 0, ch[k], guess);
            }
        });
    }
    lap(1);
    // the chain: every chunk must have ended where the next one began
    std::vector<size_t>& order = z->order; order.clear();
    bool member_ends = false; size_t end_bit = 0;
    for (size_t k = 0; k < K;) {
        if (ch[k].end < 0) { z->failed = true; return -1; }
        order.push_back(k);
        end_bit = ch[k].end_bit;
        if (ch[k].end == 1) { member_ends = true; break; }
        size_t nx = k + 1;
        while (nx < K && ch[nx].start_bit == NONE) ++nx;
        if (nx < K && ch[nx].start_bit != end_bit) { z->failed = true; return -1; }
        k = nx;
    }
    // windows in file order: chunk j's markers point into the WIN bytes before it
    size_t total = 0;
    for (size_t k : order) total += ch[k].n;
    std::vector<std::vector<uint8_t>>& before = z->before;       // the window before every chunk that has markers
    before.assign(order.size(), std::vector<uint8_t>());
    {
        std::vector<uint8_t> w(z->win, z->win + WIN); size_t valid = z->valid;
        for (size_t j = 0; j < order.size(); ++j) {
            const Chunk& c = ch[order[j]];
            if (c.markers) {
                if (valid < WIN) {                                // near the member's start: a marker may point before its first byte
                    const uint16_t* s = c.sym.data() + WIN;
                    for (size_t i = 0; i < c.n; ++i) if (s[i] >= MARK && (size_t)(s[i] - MARK) < WIN - valid) { z->failed = true; return -1; }
                }
                before[j] = w;
            }
            // the window behind this chunk: its last WIN bytes (fewer: the tail of the old window moves up)
            const size_t take = std::min(c.n, WIN);
            std::vector<uint8_t> nw(WIN);
            if (take < WIN) memcpy(nw.data(), w.data() + take, WIN - take);
            const uint16_t* s = c.sym.data() + WIN + c.n - take;
            for (size_t i = 0; i < take; ++i) nw[WIN - take + i] = s[i] < MARK ? (uint8_t)s[i] : w[s[i] - MARK];
            w.swap(nw);
            valid = std::min(WIN, valid + c.n);
        }
        before.emplace_back(std::move(w));                       // the window behind the last chunk: committed below, once the trailer agrees
    }
    lap(2);
    z->total = total; z->end_bit = end_bit; z->member_ends = member_ends; z->nt = nt; z->ready = true;
    *total_out = total;
    return 1;
}

// The prepared window's text to out[0, total): markers resolved, CRC-32 taken, the member's trailer checked when it ends here.
// 1 = written and committed; -1 = not taken (what was written is to be ignored).
int kbbq_pgz_emit(kbbq_pgz* z, uint8_t* out)
{
    if (z->failed || !z->ready) return -1;
    z->ready = false;
    const auto t_emit = std::chrono::steady_clock::now();
    const uint8_t* src = z->src; const size_t n = z->n;
    std::vector<Chunk>& ch = z->chunks;
    const std::vector<size_t>& order = z->order; const std::vector<std::vector<uint8_t>>& before = z->before;
    const size_t total = z->total, end_bit = z->end_bit; const bool member_ends = z->member_ends; const unsigned nt = z->nt;
    std::vector<size_t> at(order.size() + 1, 0);
    for (size_t j = 0; j < order.size(); ++j) at[j + 1] = at[j] + ch[order[j]].n;
    {
        std::atomic<size_t> nextj(0);
        kbbq_parallel((unsigned)std::min<size_t>(nt, order.size()), [&](unsigned) {
            std::vector<uint8_t> lut;
            for (size_t j; (j = nextj.fetch_add(1)) < order.size();) {
                Chunk& c = ch[order[j]];
                uint8_t* dst = out + at[j];
                const uint16_t* s = c.sym.data() + WIN;
                if (c.markers) {
                    lut.resize(65536);
                    for (unsigned i = 0; i < 256; ++i) lut[i] = (uint8_t)i;
                    memcpy(lut.data() + MARK, before[j].data(), WIN);
                    for (size_t i = 0; i < c.n; ++i) dst[i] = lut[s[i]];
                } else {
                    for (size_t i = 0; i < c.n; ++i) dst[i] = (uint8_t)s[i];
                }
                c.crc = crc_of(dst, c.n);
            }
        });
    }
    uint32_t crc = z->crc; uint64_t isize = z->isize;
    for (size_t j = 0; j < order.size(); ++j) {
        const Chunk& c = ch[order[j]];
        crc = (uint32_t)crc32_combine(crc, c.crc, (z_off_t)c.n);
        isize += c.n;
    }
    size_t next_pos = 0;
    if (member_ends) {
        const size_t p = (end_bit + 7) >> 3;
        if (p + 8 > n) { z->failed = true; return -1; }
        const uint32_t want_crc = (uint32_t)src[p] | (uint32_t)src[p + 1] << 8 | (uint32_t)src[p + 2] << 16 | (uint32_t)src[p + 3] << 24;
        const uint32_t want_size = (uint32_t)src[p + 4] | (uint32_t)src[p + 5] << 8 | (uint32_t)src[p + 6] << 16 | (uint32_t)src[p + 7] << 24;
        if (want_crc != crc || want_size != (uint32_t)isize) { z->failed = true; return -1; }
        next_pos = p + 8;
        // a FIRST member that is over within one chunk and is followed by more: a file of many short members (concatenated files,
        // blocked formats other than BGZF) -- nothing to spread over threads, and a window's set-up per member costs more than zlib
        if (z->delivered == 0 && order.size() <= 1 && next_pos + 18 < n) { z->failed = true; return -1; }
    }
    // commit
    const std::vector<uint8_t>& w = before.back();
    memcpy(z->win, w.data(), WIN);
    z->valid = std::min<size_t>(WIN, z->valid + total);
    z->crc = crc; z->isize = isize;
    z->delivered += total;
    if (z->trace)
        fprintf(stderr, "[pgz] window: %zu chunks, %zu bytes; search %.1f decode %.1f chain %.1f emit %.1f ms\n", order.size(), total, z->ms[0], z->ms[1], z->ms[2],
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_emit).count());
    if (member_ends) { z->in_member = false; z->pos = next_pos; z->k_cap = order.size() <= 1 ? 1 : 2 * order.size(); }     // (short members one after the other: no search ahead)
    else { z->bit = end_bit; z->k_cap = std::max<size_t>(z->k_cap * 2, 2); }
    return 1;
}

int kbbq_pgz_next(kbbq_pgz* z, kbbq_bytes& out)
{
    size_t total = 0;
    const int rc = kbbq_pgz_prepare(z, &total);
    if (rc != 1) return rc;
    const size_t old = out.size();
    out.resize(old + total);
    if (kbbq_pgz_emit(z, out.data() + old) != 1) { out.resize(old); return -1; }
    return 1;
}

// smallest input the callers hand to this decoder (compressed bytes; KBBQ_PGZ_MIN_BYTES): below it one thread is as good
size_t kbbq_pgz_min_bytes()
{
    const char* e = getenv("KBBQ_PGZ_MIN_BYTES");
    return e && atoll(e) >= 0 ? (size_t)atoll(e) : (size_t)8 << 20;
}

bool kbbq_parallel_gunzip(const uint8_t* src, size_t n, kbbq_bytes& out, unsigned threads)
{
    kbbq_pgz* z = kbbq_pgz_open(src, n, threads);
    const size_t old = out.size();
    int rc;
    while ((rc = kbbq_pgz_next(z, out)) == 1) {}
    kbbq_pgz_close(z);
    if (rc < 0) { out.resize(old); return false; }
    return true;
}
