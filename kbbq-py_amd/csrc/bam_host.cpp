// bam_host.cpp -- see bam_host.h.  Host only, no GPU code.
#include "bam_host.h"
#include "fast_inflate.h"
#include "host_threads.h"
#include "parallel_gunzip.h"

#include <algorithm>
#include <atomic>
#include <charconv>
#include <cstdio>
#include <cstring>
#include <thread>

#include <zlib.h>

namespace {

inline uint16_t le16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline int32_t les32(const uint8_t* p) { return (int32_t)le32(p); }

template <typename F> void threads_over(size_t n, unsigned nt, F f)
{
    if (nt <= 1 || n < 2) { f(0, n); return; }
    std::vector<std::thread> th;
    const size_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        const size_t lo = std::min(n, t * per), hi = std::min(n, lo + per);
        if (lo < hi) th.emplace_back([=]() { f(lo, hi); });
    }
    for (auto& t : th) t.join();
}

struct Block { size_t src, csize; size_t dst; uint32_t isize, crc; };

// BGZF: every member is  1f 8b 08 04 | mtime xfl os (6) | xlen (2) | subfields ... 'B' 'C' 02 00 BSIZE(2) ... | deflate | crc32 isize
bool index_bgzf(const uint8_t* s, size_t n, std::vector<Block>& blocks)
{
    size_t at = 0, out = 0;
    while (at < n) {
        if (n - at < 18 || s[at] != 0x1f || s[at + 1] != 0x8b || s[at + 2] != 8 || !(s[at + 3] & 4)) return false;
        const size_t xlen = le16(s + at + 10);
        if (n - at < 12 + xlen + 8) return false;
        size_t bsize = 0;
        for (size_t x = at + 12; x + 4 <= at + 12 + xlen;) {
            const size_t slen = le16(s + x + 2);
            if (s[x] == 'B' && s[x + 1] == 'C' && slen == 2 && x + 6 <= at + 12 + xlen) bsize = (size_t)le16(s + x + 4) + 1;
            x += 4 + slen;
        }
        if (bsize < 12 + xlen + 8 || bsize > n - at) return false;
        Block b;
        b.src = at + 12 + xlen; b.csize = bsize - (12 + xlen) - 8;
        b.crc = le32(s + at + bsize - 8); b.isize = le32(s + at + bsize - 4); b.dst = out;
        if (b.isize > (1u << 16)) return false;                               // BGZF blocks hold at most 64 KiB
        out += b.isize; at += bsize;
        blocks.push_back(b);
    }
    return true;
}

// gzip members one after the other through libdeflate (fast_inflate.h); false: not available or anything but success -- zlib
// takes the input from its start and decides what is wrong with it
bool inflate_members_fast(const uint8_t* src, size_t n, kbbq_bytes& out)
{
    const kbbq_libdeflate* l = kbbq_libdeflate_get();
    if (!l || n < 18) return false;
    void* d = l->alloc_decompressor();
    if (!d) return false;
    // a single member's trailer names its inflated size (mod 2^32); several members, or more than 4 GB: the buffer grows
    out.resize(std::max<size_t>({n * 4, (size_t)le32(src + n - 4) + 64, (size_t)1 << 16}));
    size_t used = 0, at = 0;
    bool ok = true;
    while (at < n && ok) {
        size_t in_used = 0, out_used = 0;
        const int rc = l->gzip_decompress_ex(d, src + at, n - at, out.data() + used, out.size() - used, &in_used, &out_used);
        if (rc == 3) { out.resize(out.size() * 2); continue; }                 // insufficient space: the member again, into twice the room
        if (rc != 0 || in_used == 0) { ok = false; break; }
        at += in_used; used += out_used;
    }
    l->free_decompressor(d);
    if (ok) out.resize(used);
    return ok;
}

bool inflate_serial(const uint8_t* src, size_t n, kbbq_bytes& out, std::string& err)
{
    // large members on all threads (parallel_gunzip.cpp), else one thread through libdeflate, else -- and for whatever those two
    // do not report as success -- zlib
    if (n >= kbbq_pgz_min_bytes() && n >= 18 && kbbq_host_thread_ceiling() > 1) {
        out.clear();
        const size_t hint = le32(src + n - 4);                                 // one member below 4 GB: its size; else the vector grows
        out.reserve(std::max<size_t>(hint >= n ? hint : 0, n * 3));
        kbbq_advise_huge(out.data(), out.capacity());
        if (kbbq_parallel_gunzip(src, n, out, 0)) return true;
        out.clear();
    }
    if (inflate_members_fast(src, n, out)) return true;
    z_stream z; memset(&z, 0, sizeof z);
    if (inflateInit2(&z, 15 + 32) != Z_OK) { err = "zlib: inflateInit2 failed"; return false; }
    out.resize(std::max<size_t>(n * 4, 1 << 16));
    size_t used = 0, at = 0;
    while (at < n) {
        z.next_in = const_cast<Bytef*>(src + at); z.avail_in = (uInt)std::min<size_t>(n - at, 1u << 30);
        const size_t in0 = z.avail_in;
        for (;;) {
            if (used == out.size()) out.resize(out.size() * 2);
            z.next_out = out.data() + used; z.avail_out = (uInt)std::min<size_t>(out.size() - used, 1u << 30);
            const size_t out0 = z.avail_out;
            const int rc = inflate(&z, Z_NO_FLUSH);
            used += out0 - z.avail_out;
            if (rc == Z_STREAM_END) {                                          // next member, if any
                at += in0 - z.avail_in;
                if (at < n && inflateReset(&z) != Z_OK) { inflateEnd(&z); err = "zlib: inflateReset failed"; return false; }
                break;
            }
            if (rc != Z_OK && rc != Z_BUF_ERROR) { inflateEnd(&z); err = "corrupt gzip stream"; return false; }
            if (z.avail_in == 0 && z.avail_out != 0) {                        // input exhausted inside a member
                at += in0;
                if (at >= n) { inflateEnd(&z); err = "truncated gzip stream"; return false; }
                break;
            }
        }
    }
    inflateEnd(&z);
    out.resize(used);
    return true;
}

const char kSeqCodes[] = "=ACMGRSVTWYHKDBN";
const char kCigarOps[] = "MIDNSHP=X???????";

inline void put(kbbq_bytes& o, const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; o.insert(o.end(), b, b + n); }
inline void put(kbbq_bytes& o, char c) { o.push_back((uint8_t)c); }
template <typename I> inline void put_int(kbbq_bytes& o, I v)
{
    char tmp[24];
    auto r = std::to_chars(tmp, tmp + sizeof tmp, v);
    put(o, tmp, (size_t)(r.ptr - tmp));
}
inline void put_float(kbbq_bytes& o, float v)
{
    char tmp[40];
    const int k = snprintf(tmp, sizeof tmp, "%g", (double)v);
    put(o, tmp, (size_t)std::max(k, 0));
}

// Where a thread renders its records: room() once per record for the longest line the record can become, then unchecked writes
// (vector::insert per field -- a capacity check and an iterator dance for every few bytes -- ran at 130 MB/s per thread).
struct Sink {
    kbbq_bytes buf; size_t n = 0;
    void room(size_t k) { if (buf.size() - n < k) buf.resize(std::max(buf.size() * 2, n + k)); }
    uint8_t* at() { return buf.data() + n; }
};
inline void put(Sink& o, const void* p, size_t n) { memcpy(o.at(), p, n); o.n += n; }
inline void put(Sink& o, char c) { *o.at() = (uint8_t)c; ++o.n; }
template <typename I> inline void put_int(Sink& o, I v)
{
    auto r = std::to_chars((char*)o.at(), (char*)o.at() + 24, v);
    o.n = (size_t)((uint8_t*)r.ptr - o.buf.data());
}
inline void put_float(Sink& o, float v)
{
    const int k = snprintf((char*)o.at(), 40, "%g", (double)v);
    o.n += (size_t)std::max(k, 0);
}

// one record (without its block_size word) -> one SAM line; false if the record is inconsistent.  longest_ref: of the reference names.
// No field expands more than 5x (an element of a B:c array: one byte -> ",-128"), the fixed fields are < 100 characters.
bool format_record(const uint8_t* r, size_t len, const std::vector<std::string>& refs, size_t longest_ref, Sink& o)
{
    if (len < 32) return false;
    o.room(6 * len + 2 * longest_ref + 256);
    const int32_t ref_id = les32(r), pos = les32(r + 4);
    const uint32_t l_name = r[8], mapq = r[9], n_cig = le16(r + 12), flag = le16(r + 14);
    const int32_t l_seq = les32(r + 16), next_ref = les32(r + 20), next_pos = les32(r + 24), tlen = les32(r + 28);
    if (l_seq < 0 || l_name == 0) return false;
    const size_t need = 32 + (size_t)l_name + 4 * (size_t)n_cig + ((size_t)l_seq + 1) / 2 + (size_t)l_seq;
    if (need > len) return false;
    const uint8_t* name = r + 32; const uint8_t* cig = name + l_name; const uint8_t* seq = cig + 4 * (size_t)n_cig;
    const uint8_t* qual = seq + ((size_t)l_seq + 1) / 2; const uint8_t* aux = qual + l_seq; const uint8_t* end = r + len;
    if (name[l_name - 1] != 0) return false;
    put(o, name, l_name - 1); put(o, '\t'); put_int(o, flag); put(o, '\t');
    if (ref_id < 0) put(o, '*'); else if ((size_t)ref_id < refs.size()) put(o, refs[(size_t)ref_id].data(), refs[(size_t)ref_id].size()); else return false;
    put(o, '\t'); put_int(o, (int64_t)pos + 1); put(o, '\t'); put_int(o, mapq); put(o, '\t');
    if (n_cig == 0) put(o, '*');
    for (uint32_t c = 0; c < n_cig; ++c) { const uint32_t w = le32(cig + 4 * c); put_int(o, w >> 4); put(o, kCigarOps[w & 15u]); }
    put(o, '\t');
    if (next_ref < 0) put(o, '*'); else if (next_ref == ref_id) put(o, '='); else if ((size_t)next_ref < refs.size()) put(o, refs[(size_t)next_ref].data(), refs[(size_t)next_ref].size()); else return false;
    put(o, '\t'); put_int(o, (int64_t)next_pos + 1); put(o, '\t'); put_int(o, tlen); put(o, '\t');
    if (l_seq == 0) put(o, '*');
    else {
        uint8_t* w = o.at();
        for (int32_t i = 0; i + 1 < l_seq; i += 2) { const uint8_t b = seq[i >> 1]; w[i] = (uint8_t)kSeqCodes[b >> 4]; w[i + 1] = (uint8_t)kSeqCodes[b & 15]; }
        if (l_seq & 1) w[l_seq - 1] = (uint8_t)kSeqCodes[seq[(l_seq - 1) >> 1] >> 4];
        o.n += (size_t)l_seq;
    }
    put(o, '\t');
    if (l_seq == 0 || qual[0] == 0xFF) put(o, '*');
    else { uint8_t* w = o.at(); for (int32_t i = 0; i < l_seq; ++i) w[i] = (uint8_t)(qual[i] + 33); o.n += (size_t)l_seq; }
    // optional fields: tag(2) type(1) value
    const uint8_t* p = aux;
    while (p < end) {
        if (end - p < 3) return false;
        put(o, '\t'); put(o, p, 2); put(o, ':');
        const char type = (char)p[2]; p += 3;
        auto ints = [&](char t, const uint8_t*& q) -> bool {                  // one integer / float of BAM type t
            switch (t) {
            case 'c': if (end - q < 1) return false; put_int(o, (int)(int8_t)q[0]); q += 1; return true;
            case 'C': if (end - q < 1) return false; put_int(o, (unsigned)q[0]); q += 1; return true;
            case 's': if (end - q < 2) return false; put_int(o, (int)(int16_t)le16(q)); q += 2; return true;
            case 'S': if (end - q < 2) return false; put_int(o, (unsigned)le16(q)); q += 2; return true;
            case 'i': if (end - q < 4) return false; put_int(o, les32(q)); q += 4; return true;
            case 'I': if (end - q < 4) return false; put_int(o, le32(q)); q += 4; return true;
            case 'f': { if (end - q < 4) return false; float v; const uint32_t w = le32(q); memcpy(&v, &w, 4); put_float(o, v); q += 4; return true; }
            default: return false;
            }
        };
        if (type == 'A') { if (end - p < 1) return false; put(o, "A:", 2); put(o, (char)p[0]); p += 1; }
        else if (type == 'Z' || type == 'H') {
            const uint8_t* z = (const uint8_t*)memchr(p, 0, (size_t)(end - p));
            if (!z) return false;
            put(o, type); put(o, ':'); put(o, p, (size_t)(z - p)); p = z + 1;
        } else if (type == 'B') {
            if (end - p < 5) return false;
            const char sub = (char)p[0]; const uint32_t count = le32(p + 1); p += 5;
            put(o, "B:", 2); put(o, sub);
            for (uint32_t k = 0; k < count; ++k) { put(o, ','); if (!ints(sub, p)) return false; }
        } else if (type == 'f') { put(o, "f:", 2); if (!ints('f', p)) return false; }
        else { put(o, "i:", 2); if (!ints(type, p)) return false; }
    }
    put(o, '\n');
    return true;
}

}  // namespace

bool kbbq_inflate_all(const uint8_t* src, size_t n, kbbq_bytes& out, std::string& err)
{
    std::vector<Block> blocks;
    if (!index_bgzf(src, n, blocks)) return inflate_serial(src, n, out, err);   // plain gzip (or a damaged BGZF: zlib decides)
    const size_t total = blocks.empty() ? 0 : blocks.back().dst + blocks.back().isize;
    kbbq_resize_fresh(out, total);
    std::atomic<int> bad(0);
    threads_over(blocks.size(), kbbq_threads_for(n), [&](size_t lo, size_t hi) {
        kbbq_block_inflater fast;
        z_stream z; memset(&z, 0, sizeof z);
        if (inflateInit2(&z, -15) != Z_OK) { bad = 1; return; }
        for (size_t b = lo; b < hi && !bad.load(); ++b) {
            const Block& k = blocks[b];
            if (fast.block(src + k.src, k.csize, out.data() + k.dst, k.isize, k.crc)) continue;
            Bytef nothing = 0;                                   // an empty block (bgzip's EOF marker; an empty file is ONLY that):
            z.next_in = const_cast<Bytef*>(src + k.src); z.avail_in = (uInt)k.csize;   // zlib rejects a NULL next_out
            z.next_out = k.isize ? out.data() + k.dst : &nothing; z.avail_out = k.isize;
            const int rc = k.isize || k.csize ? inflate(&z, Z_FINISH) : Z_STREAM_END;
            if ((rc != Z_STREAM_END && !(rc == Z_OK && z.avail_out == 0)) || z.avail_out != 0
                || crc32(crc32(0L, Z_NULL, 0), k.isize ? out.data() + k.dst : &nothing, k.isize) != k.crc) { bad = 2; break; }
            inflateReset(&z);
        }
        inflateEnd(&z);
    });
    if (bad.load()) { err = bad.load() == 1 ? "zlib: inflateInit2 failed" : "corrupt BGZF block (inflate / CRC mismatch)"; return false; }
    return true;
}

bool kbbq_bam_to_sam(const uint8_t* bam, size_t n, kbbq_bytes& text, std::string& err)
{
    if (n < 12 || memcmp(bam, "BAM\1", 4) != 0) { err = "not a BAM file"; return false; }
    size_t at = 4;
    const int32_t l_text = les32(bam + at); at += 4;
    if (l_text < 0 || (size_t)l_text > n - at - 4) { err = "BAM header text runs past the file"; return false; }
    const uint8_t* htext = bam + at; at += (size_t)l_text;
    size_t hlen = (size_t)l_text; while (hlen && htext[hlen - 1] == 0) --hlen;          // NUL padding
    const int32_t n_ref = les32(bam + at); at += 4;
    if (n_ref < 0) { err = "BAM: negative reference count"; return false; }
    std::vector<std::string> refs; std::vector<int32_t> ref_len;
    for (int32_t i = 0; i < n_ref; ++i) {
        if (n - at < 4) { err = "BAM reference list runs past the file"; return false; }
        const int32_t l_name = les32(bam + at); at += 4;
        if (l_name <= 0 || (size_t)l_name + 4 > n - at) { err = "BAM reference list runs past the file"; return false; }
        refs.emplace_back((const char*)bam + at, (size_t)l_name - 1); at += (size_t)l_name;
        ref_len.push_back(les32(bam + at)); at += 4;
    }
    text.clear();
    put(text, htext, hlen);
    if (hlen && text.back() != '\n') put(text, '\n');
    bool has_sq = false;
    for (size_t i = 0; i + 3 <= hlen; ++i) if ((i == 0 || htext[i - 1] == '\n') && !memcmp(htext + i, "@SQ", 3)) { has_sq = true; break; }
    if (!has_sq)                                                               // the binary list is authoritative
        for (size_t i = 0; i < refs.size(); ++i) { put(text, "@SQ\tSN:", 7); put(text, refs[i].data(), refs[i].size()); put(text, "\tLN:", 4); put_int(text, ref_len[i]); put(text, '\n'); }
    // record offsets (a chain of block_size words), then the lines in parallel
    std::vector<size_t> rec;
    while (at < n) {
        if (n - at < 4) { err = "BAM: truncated record"; return false; }
        const int32_t bs = les32(bam + at);
        if (bs < 32 || (size_t)bs > n - at - 4) { err = "BAM: truncated record"; return false; }
        rec.push_back(at + 4); at += 4 + (size_t)bs;
    }
    rec.push_back(n + 4);
    const unsigned nt = kbbq_threads_for(n);
    std::vector<Sink> parts(nt);
    std::atomic<long long> bad(-1);
    const size_t nrec = rec.size() - 1, per = (nrec + nt - 1) / std::max(1u, nt);
    size_t longest_ref = 0;
    for (const auto& nm : refs) longest_ref = std::max(longest_ref, nm.size());
    kbbq_parallel(nt, [&](unsigned t) {
        const size_t lo = std::min(nrec, t * per), hi = std::min(nrec, lo + per);
        if (lo >= hi) return;
        Sink& o = parts[t];
        kbbq_resize_fresh(o.buf, (rec[hi] - rec[lo]) * 2 + 4096);
        for (size_t i = lo; i < hi; ++i)
            if (!format_record(bam + rec[i], rec[i + 1] - rec[i] - 4, refs, longest_ref, o)) {
                long long cur = bad.load();
                while ((cur < 0 || (long long)i < cur) && !bad.compare_exchange_weak(cur, (long long)i)) {}
                return;
            }
    });
    if (bad.load() >= 0) { err = "BAM: malformed alignment record " + std::to_string(bad.load()); return false; }
    std::vector<size_t> at_part(nt + 1, text.size());                          // the parts side by side into their places
    for (unsigned t = 0; t < nt; ++t) at_part[t + 1] = at_part[t] + parts[t].n;
    kbbq_resize_fresh(text, at_part[nt]);
    kbbq_parallel(nt, [&](unsigned t) {
        if (parts[t].n) memcpy(text.data() + at_part[t], parts[t].buf.data(), parts[t].n);
        kbbq_bytes().swap(parts[t].buf);
    });
    return true;
}
