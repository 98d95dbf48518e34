// fastq_stream.cpp -- a FASTQ file read SEQUENTIALLY, segment by segment (no GPU code).
//
// The reference walks its two input files read by read in constant memory (recalibrate.py:56-57 zip(FastxFile, FastxFile),
// :141-156), so it takes inputs of any size and inputs that cannot be mapped or sought -- `-f <(zcat a.fq.gz) <(zcat b.fq.gz)`.
// The mapped reader of fastq_host.cpp indexes a whole file (or a rank's byte range of it).  This reader hands out SEGMENTS: a
// few hundred MB of whole records at a time, each an ordinary kbbq_fastq over memory of its own, indexed by the same code
// (kbbq_fastq_index_range_), so that scan / fill / format work on it unchanged; what the reference's walk carries from read to
// read -- read groups met so far, the longest read so far, the record number -- the caller carries from segment to segment
// (kbbq_fastq_scan_next).  File A leads (segments of about max_bytes), file B follows with the same NUMBER of records per
// segment; a shorter file B ends pass 1 where zip() would.  Pass 2 needs file A a second time: a regular file is
// simply read again, a pipe is copied to a spool file as it is read (kbbq_fastq_stream_tee) and pass 2 reads the spool.
// gzip / bgzip bytes -- a .fq.gz file of any size, or a pipe that carries them -- are inflated as they are read, member after member.
#include "../../include/kbbq_hip.h"
#include "fast_inflate.h"
#include "fastq_host.h"
#include "host_threads.h"
#include "parallel_gunzip.h"

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

int kbbq_set_error_(int code, const char* msg);      // defined in kbbq_hip.hip

struct kbbq_fastq_stream {
    int fd = -1; bool own_fd = false;
    bool regular = false; int64_t size = -1;          // regular files: size known, positioned parallel reads
    int64_t pos = 0;                                   // bytes consumed from the descriptor
    bool eof = false;
    raw_vector<uint8_t> carry;                         // read but not handed out yet (a partial record; records beyond a follower's count)
    int tee_fd = -1;                                   // every byte handed out is also written here (the spool of a pipe)
    int64_t records = 0, bytes = 0;                    // handed out so far
    std::string path;
    // gzip / bgzip input (what pysam.FastxFile reads transparently): inflated as it is read, member after member, in constant memory
    bool gz = false, zs_open = false, in_member = false, fd_eof = false;
    z_stream zs;
    raw_vector<uint8_t> zin; size_t zin_pos = 0, zin_len = 0;
    std::string zerr;
    // a large gzip member of a regular file: inflated on all threads from a mapping of the file (parallel_gunzip.cpp), a window of
    // chunks at a time into `pend`; anything that decoder does not take on sends the stream back to byte 0 and zlib, which skips
    // what has been handed out already
    kbbq_pgz* pgz = nullptr; const uint8_t* map = nullptr; size_t map_n = 0;
    kbbq_bytes pend; size_t pend_at = 0;
    uint64_t served = 0, skip = 0;
    bool reread = false;                               // compressed, but pass 2 may simply read the file again (no spool)
    void drop_pgz() { if (pgz) kbbq_pgz_close(pgz); pgz = nullptr; if (map) munmap((void*)map, map_n); map = nullptr; kbbq_bytes().swap(pend); pend_at = 0; }
    ~kbbq_fastq_stream() { drop_pgz(); if (zs_open) inflateEnd(&zs); if (own_fd && fd >= 0) close(fd); }
};

// ---- the pool of segment buffers: kept while a stream is open (at most kPoolMax), dropped with the last one
namespace {
std::mutex g_pool_mutex;
std::vector<raw_vector<uint8_t>> g_pool;
int g_live_streams = 0;
constexpr size_t kPoolMax = 6;
}

void kbbq_text_pool_give(raw_vector<uint8_t>&& v)
{
    raw_vector<uint8_t> mine(std::move(v));
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    if (g_live_streams > 0 && g_pool.size() < kPoolMax) g_pool.emplace_back(std::move(mine));
}                                                                  // (else freed here)

raw_vector<uint8_t> kbbq_text_pool_take(size_t capacity)
{
    raw_vector<uint8_t> out;
    {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        size_t best = g_pool.size();
        for (size_t i = 0; i < g_pool.size(); ++i)
            if (g_pool[i].capacity() >= capacity && (best == g_pool.size() || g_pool[i].capacity() < g_pool[best].capacity())) best = i;
        if (best < g_pool.size()) { out = std::move(g_pool[best]); g_pool.erase(g_pool.begin() + (long)best); }
        else if (!g_pool.empty()) g_pool.pop_back();                // none is large enough: one of the small ones makes room
    }
    out.clear();
    if (out.capacity() < capacity) { out.reserve(capacity); kbbq_advise_huge(out.data(), out.capacity()); }
    return out;
}

namespace {

template <typename F> void over_threads(size_t n, unsigned nt, F f)
{
    if (nt <= 1 || n < ((size_t)1 << 20)) { f(0u, (size_t)0, n); return; }
    const size_t per = (n + nt - 1) / nt;
    kbbq_parallel(nt, [&](unsigned t) {
        const size_t lo = std::min(n, (size_t)t * per), hi = std::min(n, lo + per);
        f(t, lo, hi);
    });
}

size_t count_newlines(const uint8_t* p, size_t n)
{
    size_t c = 0;
    for (size_t i = 0; i < n; ++i) c += (p[i] == '\n');          // vectorised by the compiler
    return c;
}

// number of '\n' in buf[0, n), per-thread counts in `parts` (for kth_newline)
size_t count_lines(const uint8_t* buf, size_t n, std::vector<size_t>& parts, std::vector<size_t>& starts)
{
    const unsigned nt = kbbq_threads_for(n);
    parts.assign(nt, 0); starts.assign(nt + 1, n);
    over_threads(n, nt, [&](unsigned t, size_t lo, size_t hi) { parts[t] = count_newlines(buf + lo, hi - lo); starts[t] = lo; });
    if (nt <= 1 || n < ((size_t)1 << 20)) { starts[0] = 0; for (unsigned t = 1; t <= nt; ++t) starts[t] = n; }
    size_t total = 0;
    for (size_t c : parts) total += c;
    return total;
}

// offset just behind the k-th '\n' (k >= 1, k <= the number counted) of buf
size_t behind_kth_newline(const uint8_t* buf, size_t n, size_t k, const std::vector<size_t>& parts, const std::vector<size_t>& starts)
{
    size_t before = 0;
    for (size_t t = 0; t < parts.size(); ++t) {
        if (before + parts[t] >= k) {
            const uint8_t* p = buf + starts[t]; const uint8_t* e = buf + (t + 1 < starts.size() ? std::max(starts[t + 1], starts[t]) : n);
            if (t + 1 == parts.size()) e = buf + n;
            size_t need = k - before;
            while (p < e) {
                const uint8_t* q = (const uint8_t*)memchr(p, '\n', (size_t)(e - p));
                if (!q) break;
                if (--need == 0) return (size_t)(q - buf) + 1;
                p = q + 1;
            }
            return n;                                            // cannot happen: the counts said it is here
        }
        before += parts[t];
    }
    return n;
}

// compressed bytes from the descriptor into s->zin (behind what is still unread there); false on a read error
bool refill_compressed(kbbq_fastq_stream* s)
{
    if (s->zin_pos == s->zin_len) s->zin_pos = s->zin_len = 0;
    if (s->fd_eof || s->zin_len == s->zin.size()) return true;
    for (;;) {
        const ssize_t k = read(s->fd, s->zin.data() + s->zin_len, s->zin.size() - s->zin_len);
        if (k < 0 && errno == EINTR) continue;
        if (k < 0) return false;
        if (k == 0) s->fd_eof = true;
        s->zin_len += (size_t)k; s->pos += k;
        return true;
    }
}

// BGZF (bgzip): every member is a block of at most 64 KB that says its own compressed size (extra field 'B' 'C') and, in its last
// four bytes, its inflated size -- so the whole blocks waiting in s->zin can be inflated AT THE SAME TIME, each straight to its
// place in dst.  Returns the bytes written (0: the next member is not such a block, or does not fit dst any more: the caller's
// serial loop takes it), -1 on a damaged block.
int64_t inflate_bgzf_blocks(kbbq_fastq_stream* s, uint8_t* dst, size_t room)
{
    struct Blk { size_t src, csize, isize, dst; uint32_t crc; };
    std::vector<Blk> blocks;
    const uint8_t* z = s->zin.data();
    size_t at = s->zin_pos, out = 0;
    auto le16 = [](const uint8_t* p) { return (size_t)p[0] | ((size_t)p[1] << 8); };
    auto le32 = [](const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); };
    while (s->zin_len - at >= 18) {
        if (z[at] != 0x1f || z[at + 1] != 0x8b || z[at + 2] != 8 || !(z[at + 3] & 4)) break;
        const size_t xlen = le16(z + at + 10);
        if (s->zin_len - at < 12 + xlen) break;
        size_t bsize = 0;
        for (size_t x = at + 12; x + 4 <= at + 12 + xlen;) {
            const size_t slen = le16(z + x + 2);
            if (z[x] == 'B' && z[x + 1] == 'C' && slen == 2 && x + 6 <= at + 12 + xlen) bsize = le16(z + x + 4) + 1;
            x += 4 + slen;
        }
        if (bsize < 12 + xlen + 8) break;                       // not a BGZF block: the serial loop decides what it is
        if (s->zin_len - at < bsize) break;                     // not all of it has been read yet
        Blk b;
        b.src = at + 12 + xlen; b.csize = bsize - (12 + xlen) - 8; b.crc = le32(z + at + bsize - 8); b.isize = le32(z + at + bsize - 4);
        if (b.isize > (1u << 16) || out + b.isize > room) break;
        b.dst = out; out += b.isize; at += bsize;
        blocks.push_back(b);
    }
    if (blocks.size() < 2) return 0;                             // (one block: the serial loop is as good)
    std::atomic<int> bad(0);
    const unsigned nt = (unsigned)std::min<size_t>(kbbq_threads_for(out), blocks.size());
    kbbq_parallel(nt, [&](unsigned t) {
        kbbq_block_inflater fast;
        z_stream zz; memset(&zz, 0, sizeof zz);
        if (inflateInit2(&zz, -15) != Z_OK) { bad = 1; return; }
        for (size_t b = t; b < blocks.size() && !bad.load(); b += nt) {
            const Blk& k = blocks[b];
            if (fast.block(z + k.src, k.csize, dst + k.dst, k.isize, k.crc)) continue;
            Bytef nothing = 0;
            zz.next_in = const_cast<Bytef*>(z + k.src); zz.avail_in = (uInt)k.csize;
            zz.next_out = k.isize ? dst + k.dst : &nothing; zz.avail_out = (uInt)k.isize;
            const int rc = (k.isize || k.csize) ? inflate(&zz, Z_FINISH) : Z_STREAM_END;
            if ((rc != Z_STREAM_END && !(rc == Z_OK && zz.avail_out == 0)) || zz.avail_out != 0
                || crc32(crc32(0L, Z_NULL, 0), k.isize ? dst + k.dst : &nothing, (uInt)k.isize) != k.crc) { bad = 2; break; }
            inflateReset(&zz);
        }
        inflateEnd(&zz);
    });
    if (bad.load()) { s->zerr = "damaged BGZF block (inflate / CRC mismatch)"; return -1; }
    s->zin_pos = at;
    return (int64_t)out;
}

// up to `want` INFLATED bytes into dst (fewer only at the end of the input); -1 on damaged input (s->zerr says what)
int64_t inflate_some_zlib(kbbq_fastq_stream* s, uint8_t* dst, size_t want)
{
    size_t got = 0;
    while (got < want) {
        if (!s->in_member && s->zin_pos < s->zin_len) {
            if (s->zin_len - s->zin_pos < ((size_t)1 << 20) && !s->fd_eof) {                   // keep whole blocks coming: top the buffer up
                memmove(s->zin.data(), s->zin.data() + s->zin_pos, s->zin_len - s->zin_pos);
                s->zin_len -= s->zin_pos; s->zin_pos = 0;
                if (!refill_compressed(s)) { s->zerr = "read failed"; return -1; }
            }
            const int64_t k = inflate_bgzf_blocks(s, dst + got, want - got);
            if (k < 0) return -1;
            got += (size_t)k;
            if (k > 0) continue;
        }
        if (s->zin_pos == s->zin_len) {
            if (!refill_compressed(s)) { s->zerr = "read failed"; return -1; }
            if (s->zin_pos == s->zin_len) {                                       // the compressed input has ended
                if (s->in_member) { s->zerr = "truncated gzip data"; return -1; }
                s->eof = true;
                break;
            }
        }
        if (!s->in_member) {
            // a new member (bgzip files are thousands of them); NUL padding behind the last member is tolerated as gzip(1) does
            while (s->zin_pos < s->zin_len && s->zin[s->zin_pos] == 0) ++s->zin_pos;
            if (s->zin_pos == s->zin_len) continue;
            if (inflateReset2(&s->zs, 15 + 32) != Z_OK) { s->zerr = "inflateReset failed"; return -1; }
            s->in_member = true;
        }
        s->zs.next_in = s->zin.data() + s->zin_pos;
        s->zs.avail_in = (uInt)std::min<size_t>(s->zin_len - s->zin_pos, 1u << 30);
        s->zs.next_out = dst + got;
        s->zs.avail_out = (uInt)std::min<size_t>(want - got, 1u << 30);
        const uInt in0 = s->zs.avail_in, out0 = s->zs.avail_out;
        const int rc = inflate(&s->zs, Z_NO_FLUSH);
        s->zin_pos += in0 - s->zs.avail_in;
        got += out0 - s->zs.avail_out;
        if (rc == Z_STREAM_END) { s->in_member = false; continue; }
        if (rc != Z_OK && rc != Z_BUF_ERROR) { s->zerr = std::string("damaged gzip data (") + (s->zs.msg ? s->zs.msg : "inflate failed") + ")"; return -1; }
        if (rc == Z_BUF_ERROR && in0 == s->zs.avail_in && out0 == s->zs.avail_out && s->zin_pos < s->zin_len) {
            s->zerr = "damaged gzip data (no progress)"; return -1;
        }
    }
    return (int64_t)got;
}

int64_t inflate_some(kbbq_fastq_stream* s, uint8_t* dst, size_t want)
{
    size_t got = 0;
    while (s->pgz && got < want) {
        if (s->pend_at < s->pend.size()) {
            const size_t k = std::min(want - got, s->pend.size() - s->pend_at);
            memcpy(dst + got, s->pend.data() + s->pend_at, k);
            s->pend_at += k; got += k; s->served += k;
            continue;
        }
        s->pend.clear(); s->pend_at = 0;
        size_t total = 0;
        int rc = kbbq_pgz_prepare(s->pgz, &total);
        if (rc == 1) {
            if (total <= want - got) {                                        // straight to the caller's memory
                rc = kbbq_pgz_emit(s->pgz, dst + got);
                if (rc == 1) { got += total; s->served += total; continue; }
            } else {
                s->pend.resize(total);
                rc = kbbq_pgz_emit(s->pgz, s->pend.data());
                if (rc == 1) continue;
                s->pend.clear();
            }
        }
        if (rc == 0) { s->eof = true; return (int64_t)got; }
        // not taken: back to the file's first byte with zlib, past what has been handed out
        s->drop_pgz();
        if (lseek(s->fd, 0, SEEK_SET) < 0) { s->zerr = "lseek failed"; return -1; }
        s->pos = 0; s->zin_pos = s->zin_len = 0; s->fd_eof = false; s->in_member = false;
        s->skip = s->served;
    }
    if (s->skip) {
        raw_vector<uint8_t> scratch((size_t)1 << 20);
        while (s->skip) {
            const int64_t k = inflate_some_zlib(s, scratch.data(), (size_t)std::min<uint64_t>(s->skip, scratch.size()));
            if (k < 0) return -1;
            if (k == 0) { s->zerr = "damaged gzip data"; return -1; }        // (the input ended before the text already handed out did)
            s->skip -= (uint64_t)k;
        }
    }
    if (got == want || s->eof) return (int64_t)got;
    const int64_t k = inflate_some_zlib(s, dst + got, want - got);
    return k < 0 ? -1 : (int64_t)got + k;
}

// read up to `want` bytes at the stream's position into dst; returns the bytes read (< want only at the end of the input)
int64_t read_some(kbbq_fastq_stream* s, uint8_t* dst, size_t want)
{
    if (s->eof || want == 0) return 0;
    if (s->gz) return inflate_some(s, dst, want);
    if (s->regular) {
        const size_t have = (size_t)std::max<int64_t>(0, s->size - s->pos);
        const size_t take = std::min(want, have);
        std::atomic<int> bad(0);
        const int64_t base = s->pos;
        over_threads(take, kbbq_threads_for(take / 4), [&](unsigned, size_t lo, size_t hi) {
            size_t at = lo;
            while (at < hi) {
                const ssize_t k = pread(s->fd, dst + at, hi - at, (off_t)(base + (int64_t)at));
                if (k < 0 && errno == EINTR) continue;
                if (k <= 0) { bad = 1; return; }
                at += (size_t)k;
            }
        });
        if (bad.load()) return -1;
        s->pos += (int64_t)take;
        if (s->pos >= s->size) s->eof = true;
        return (int64_t)take;
    }
    size_t got = 0;
    while (got < want) {
        const ssize_t k = read(s->fd, dst + got, want - got);
        if (k < 0 && errno == EINTR) continue;
        if (k < 0) return -1;
        if (k == 0) { s->eof = true; break; }
        got += (size_t)k;
    }
    s->pos += (int64_t)got;
    return (int64_t)got;
}

bool write_all(int fd, const uint8_t* p, size_t n)
{
    while (n) {
        const ssize_t k = write(fd, p, n);
        if (k < 0 && errno == EINTR) continue;
        if (k <= 0) return false;
        p += k; n -= (size_t)k;
    }
    return true;
}

} // namespace

extern "C" {

// path: a regular file, a named pipe / process substitution (/dev/fd/N), a character device, or "-" for standard input
int kbbq_fastq_stream_open(const char* path, kbbq_fastq_stream** out)
{
    if (!path || !out) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_stream_open: NULL argument");
    *out = nullptr;
    kbbq_fastq_stream* s = new kbbq_fastq_stream();
    s->path = path;
    if (!strcmp(path, "-")) s->fd = 0;
    else { s->fd = open(path, O_RDONLY); s->own_fd = true; }
    if (s->fd < 0) { delete s; return kbbq_set_error_(KBBQ_E_ARG, (std::string("cannot open ") + path).c_str()); }
    struct stat st;
    if (fstat(s->fd, &st) != 0) { delete s; return kbbq_set_error_(KBBQ_E_ARG, "fstat failed"); }
    s->regular = S_ISREG(st.st_mode);
    if (s->regular) { s->size = (int64_t)st.st_size; s->eof = s->size == 0; }
#ifdef F_SETPIPE_SZ
    else if (S_ISFIFO(st.st_mode)) (void)fcntl(s->fd, F_SETPIPE_SZ, 1 << 20);      // 64 KB by default: fewer wake-ups per GB
#endif
    // gzip / bgzip bytes?  The first two bytes decide (a pipe's are read here and kept)
    s->zin.resize((size_t)4 << 20);
    if (!s->eof) {
        while (s->zin_len < 2 && !s->fd_eof) {
            const ssize_t k = read(s->fd, s->zin.data() + s->zin_len, 2 - s->zin_len);
            if (k < 0 && errno == EINTR) continue;
            if (k < 0) { delete s; return kbbq_set_error_(KBBQ_E_ARG, (std::string(path) + ": read failed").c_str()); }
            if (k == 0) s->fd_eof = true;
            s->zin_len += (size_t)k;
        }
        if (s->zin_len == 2 && s->zin[0] == 0x1f && s->zin[1] == 0x8b) {
            memset(&s->zs, 0, sizeof s->zs);
            if (inflateInit2(&s->zs, 15 + 32) != Z_OK) { delete s; return kbbq_set_error_(KBBQ_E_ARG, "inflateInit failed"); }
            s->zs_open = true; s->gz = true; s->pos = (int64_t)s->zin_len;
            s->reread = s->regular && s->size < ((int64_t)64 << 20);          // (a small file: inflating it twice is nothing)
            if (s->regular && s->size >= 18) {
                void* m = mmap(nullptr, (size_t)s->size, PROT_READ, MAP_PRIVATE, s->fd, 0);
                if (m != MAP_FAILED) {
                    const uint8_t* z = (const uint8_t*)m;
                    bool bgzf = false;                                        // (blocks that say their sizes go side by side as they are read)
                    if (z[3] & 4) {
                        const size_t xlen = (size_t)z[10] | (size_t)z[11] << 8;
                        for (size_t x = 12; x + 4 <= 12 + xlen && x + 4 <= (size_t)s->size;) {
                            const size_t slen = (size_t)z[x + 2] | (size_t)z[x + 3] << 8;
                            if (z[x] == 'B' && z[x + 1] == 'C' && slen == 2) bgzf = true;
                            x += 4 + slen;
                        }
                    }
                    const bool wide = (size_t)s->size >= kbbq_pgz_min_bytes() && kbbq_host_thread_ceiling() > 1;
                    if (bgzf || !wide) munmap(m, (size_t)s->size);
                    else { s->map = z; s->map_n = (size_t)s->size; madvise(m, s->map_n, MADV_SEQUENTIAL); s->pgz = kbbq_pgz_open(z, s->map_n, 0); }
                    if (bgzf || s->pgz) s->reread = true;                     // inflated on all threads: cheaper to do again than to keep a spool of the text
                }
            }
        } else if (s->regular) {
            s->zin_len = 0;                                                          // plain text of a regular file: positioned reads from byte 0
            if (lseek(s->fd, 0, SEEK_SET) < 0) { delete s; return kbbq_set_error_(KBBQ_E_ARG, "lseek failed"); }
        } else {
            s->carry.assign(s->zin.data(), s->zin.data() + s->zin_len);             // plain text of a pipe: the bytes read belong to the first segment
            s->pos = (int64_t)s->zin_len; s->zin_len = 0;
            if (s->fd_eof) s->eof = true;
        }
    }
    { std::lock_guard<std::mutex> lk(g_pool_mutex); ++g_live_streams; }
    *out = s;
    return KBBQ_OK;
}

// 1: a regular file that pass 2 reads a second time as it is -- plain text, or compressed bytes that inflate on all threads (bgzip
// blocks, large gzip members: parallel_gunzip.cpp) or are few; 0: a pipe, standard input, a large gzip file on a one-thread host --
// pass 2 reads a spool of the text
int kbbq_fastq_stream_is_regular(const kbbq_fastq_stream* s) { return s && s->regular && (!s->gz || s->reread) ? 1 : 0; }

// every byte handed out from now on is appended to `fd` (the caller's spool file; not closed here); -1 stops it
int kbbq_fastq_stream_tee(kbbq_fastq_stream* s, int fd)
{
    if (!s) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_stream_tee: NULL stream");
    s->tee_fd = fd;
    return KBBQ_OK;
}

// Read ahead of the next kbbq_fastq_stream_next until `bytes` bytes wait in the stream: what a caller runs on a second thread
// for the FOLLOWING file while the leading file's segment is being read -- two pipes fed by two decompressors are then
// drained at the same time instead of one after the other.  Not to be called while a _next of the same stream runs.
int kbbq_fastq_stream_prefetch(kbbq_fastq_stream* s, size_t bytes)
{
    if (!s) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_stream_prefetch: NULL stream");
    const size_t have = s->carry.size();
    if (s->eof || bytes <= have) return KBBQ_OK;
    bytes -= have;
    s->carry.resize(have + bytes);
    const int64_t got = read_some(s, s->carry.data() + have, bytes);
    s->carry.resize(have + (size_t)std::max<int64_t>(got, 0));
    if (got < 0) return kbbq_set_error_(KBBQ_E_ARG, (s->path + ": " + (s->zerr.empty() ? "read failed" : s->zerr)).c_str());
    return KBBQ_OK;
}

// The next segment: whole records only.
//   records == 0  (the leading file): about max_bytes of text (at least one record, however long);
//   records  > 0  (the following file): exactly that many records, fewer only when the input ends first.
// *segment: a reader over memory of its own (close it with kbbq_fastq_close), NULL when the input has ended and nothing is
// left; *at_end: 1 when the input has ended behind this segment.
int kbbq_fastq_stream_next(kbbq_fastq_stream* s, size_t max_bytes, int64_t records, kbbq_fastq** segment, int* at_end)
{
    if (!s || !segment || records < 0) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_stream_next: bad argument");
    *segment = nullptr;
    if (at_end) *at_end = 0;
    max_bytes = std::max<size_t>(max_bytes, 1 << 16);
    kbbq_fastq* f = new kbbq_fastq();
    raw_vector<uint8_t>& buf = f->text;
    size_t have = s->carry.size();
    size_t cap = std::max(max_bytes, have) + (1 << 16);
    buf = kbbq_text_pool_take(cap);
    buf.resize(cap);
    if (have) memcpy(buf.data(), s->carry.data(), have);
    s->carry.clear();
    const size_t target = (size_t)records * 4;
    size_t cut = 0;
    std::vector<size_t> parts, starts;
    auto behind = [&](size_t lines, size_t nlines) {                            // offset behind `lines` whole lines
        return lines == 0 ? (size_t)0 : (lines <= nlines ? behind_kth_newline(buf.data(), have, lines, parts, starts) : have);
    };
    for (;;) {
        if (!s->eof && have < cap) {
            const int64_t got = read_some(s, buf.data() + have, cap - have);
            if (got < 0) { delete f; return kbbq_set_error_(KBBQ_E_ARG, (s->path + ": " + (s->zerr.empty() ? "read failed" : s->zerr)).c_str()); }
            have += (size_t)got;
        }
        const size_t nlines = count_lines(buf.data(), have, parts, starts);
        // at the end of the input the last line needs no line end (as in the mapped reader)
        const size_t whole = nlines + ((s->eof && have > 0 && buf[have - 1] != '\n') ? 1 : 0);
        if (target && whole >= target) { cut = behind(target, nlines); break; }               // the follower's count is there
        if (s->eof) {
            // whatever is there: whole records only, and nothing may be left behind them (the mapped reader refuses such a file too)
            if (whole & 3) { delete f; return kbbq_set_error_(KBBQ_E_ARG, (s->path + ": not a 4-line-per-record FASTQ file").c_str()); }
            cut = have;
            break;
        }
        if (!target && have >= max_bytes && nlines >= 4) { cut = behind(nlines & ~(size_t)3, nlines); break; }
        cap = have + std::max<size_t>(have / 4, 1 << 20);                      // not enough yet (a follower's longer records, one huge record): read more
        buf.resize(cap);
    }
    if (cut < have) { s->carry.resize(have - cut); memcpy(s->carry.data(), buf.data() + cut, have - cut); }
    if (at_end) *at_end = (s->eof && s->carry.empty()) ? 1 : 0;
    if (cut == 0) { delete f; return KBBQ_OK; }                                // the input has ended: no segment
    buf.resize(cut);
    f->buf = buf.data(); f->size = cut; f->mapped = false;
    const int bad = kbbq_fastq_index_range_(f, 0, cut);
    if (bad) {
        delete f;
        return kbbq_set_error_(KBBQ_E_ARG, bad == 4 ? (s->path + ": not a 4-line-per-record FASTQ file").c_str()
                                          : bad == 1 ? "record header does not start with @"
                                          : bad == 2 ? "sequence and quality lengths differ" : "read longer than 65535 bases");
    }
    if (s->tee_fd >= 0 && !write_all(s->tee_fd, buf.data(), cut)) {
        delete f;
        return kbbq_set_error_(KBBQ_E_ARG, (s->path + ": writing the spool file failed (KBBQ_SPOOL_DIR selects another directory)").c_str());
    }
    s->records += (int64_t)f->h0.size(); s->bytes += (int64_t)cut;
    *segment = f;
    return KBBQ_OK;
}

int kbbq_fastq_stream_close(kbbq_fastq_stream* s)
{
    if (s) {
        std::lock_guard<std::mutex> lk(g_pool_mutex);
        if (--g_live_streams <= 0) { g_live_streams = 0; g_pool.clear(); g_pool.shrink_to_fit(); }
    }
    delete s;
    return KBBQ_OK;
}

} // extern "C"
