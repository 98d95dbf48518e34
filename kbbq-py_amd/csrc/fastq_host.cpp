// fastq_host.cpp -- host-side FASTQ ingest / egress for the hot path (no GPU code).
//
// Replaces, for this path, what the reference gets from pysam (FastxFile iteration,
// recalibrate.py:56-57,141-142), its per-read name parsing (compare_reads.py:304-318,
// recalibrate.py:59-64) and the print() calls of recalibrate.py:153-156.  4-line FASTQ
// records; `name` is the header up to the first whitespace (pysam/kseq semantics).
//
// Error ORDER is part of the behaviour: the reference stops at the first offending read,
// and within one read checks run in the order  read-group inference (:59) -> name prefix
// (:91 -> :17) -> sequence lengths (:20) -> [device: alphabet :94] -> shorter than the
// running maximum (:97) -> [device: q > 42 :114].  kbbq_fastq_scan_pair reports the first
// host-detectable offender and its kind; the caller lets the device look at the reads before
// it (and at it, for kind 5) before raising.
#include "../../include/kbbq_hip.h"
#include "host_threads.h"
#include "bam_host.h"
#include "fastq_host.h"

#include <algorithm>
#include <atomic>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

extern "C" const char* kbbq_last_error(void);
int kbbq_set_error_(int code, const char* msg);      // defined in kbbq_hip.hip

static unsigned nthreads_for(size_t work) { return kbbq_threads_for(work); }

// offsets of every '\n' in buf[lo, hi), appended to v (an AVX2 walk, 32 bytes per step, was measured against this memchr per line: no
// difference -- opening a file is bound by populating the page tables of its mapping, not by finding the line ends)
static void newline_offsets(const uint8_t* buf, size_t lo, size_t hi, raw_vector<uint64_t>& v)
{
    const uint8_t* p = buf + lo; const uint8_t* e = buf + hi;
    while (p < e) {
        const uint8_t* q = (const uint8_t*)memchr(p, '\n', (size_t)(e - p));
        if (!q) break;
        v.push_back((uint64_t)(q - buf));
        p = q + 1;
    }
}

template <typename F> static void parallel_for(int64_t n, unsigned nt, F f)
{
    if (nt <= 1 || n < 4096) { f(0, n); return; }
    kbbq_parallel_parts((size_t)n, nt, [&](unsigned, size_t lo, size_t hi) { f((int64_t)lo, (int64_t)hi); });     // parked workers (host_threads.h)
}

// A FASTQ file whose sequences and qualities run over several lines, as kseq (pysam.FastxFile) reads it: a record starts at a line
// beginning with '@' (name up to the first whitespace, the rest is the comment); sequence lines follow until a line that begins with
// '+'; quality lines follow until they hold at least as many characters as the sequence -- exactly as many, or the file is refused.
// Only printable characters of a sequence / quality line count.  `out` receives the same records four lines each.  False when the
// text is not such a file (no record at all, a record without '+' line, qualities of another length).
static bool unwrap_fastq(const uint8_t* buf, size_t size, kbbq_bytes& out)
{
    out.clear();
    out.reserve(size + 16);
    size_t at = 0, records = 0;
    auto line_end = [&](size_t from) { const void* q = memchr(buf + from, '\n', size - from); return q ? (size_t)((const uint8_t*)q - buf) : size; };
    auto graph = [](uint8_t c) { return c > 32 && c < 127; };
    while (at < size) {
        while (at < size && (buf[at] == '\n' || buf[at] == '\r')) ++at;              // blank lines between records
        if (at >= size) break;
        if (buf[at] != '@') return false;
        size_t e = line_end(at);
        size_t he = e; while (he > at && buf[he - 1] == '\r') --he;
        out.insert(out.end(), buf + at, buf + he); out.push_back('\n');
        at = e < size ? e + 1 : size;
        const size_t seq_at = out.size();
        bool plus = false;
        while (at < size) {                                                           // sequence lines up to the '+' line
            if (buf[at] == '+') { plus = true; break; }
            e = line_end(at);
            for (size_t i = at; i < e; ++i) if (graph(buf[i])) out.push_back(buf[i]);
            at = e < size ? e + 1 : size;
        }
        if (!plus) return false;
        const size_t seqlen = out.size() - seq_at;
        if (seqlen > 65535) return false;
        at = line_end(at); at = at < size ? at + 1 : size;                            // the rest of the '+' line is ignored
        out.push_back('\n'); out.push_back('+'); out.push_back('\n');
        const size_t q_at = out.size();
        while (at < size && out.size() - q_at < seqlen) {                             // quality lines until the sequence's length is reached
            e = line_end(at);
            for (size_t i = at; i < e; ++i) if (graph(buf[i])) out.push_back(buf[i]);
            at = e < size ? e + 1 : size;
        }
        if (out.size() - q_at != seqlen) return false;
        out.push_back('\n');
        ++records;
    }
    return records > 0;
}

int kbbq_fastq_index_range_(kbbq_fastq* f, size_t r0, size_t r1)
{
    // Line index, in parallel and without a merged list of line ends: (1) every thread collects the '\n' offsets of
    // its byte range; (2) a prefix sum of the counts numbers the lines; (3) every thread turns ITS line ends into
    // record fields -- line g is field g % 4 of record g / 4, and starts after the previous line end, which is the
    // previous entry of the same list or the last entry of an earlier thread's.
    const unsigned nt = nthreads_for(r1 - r0);
    std::vector<raw_vector<uint64_t>> parts(nt);
    const size_t per = (r1 - r0 + nt - 1) / nt;
    kbbq_parallel(nt, [&](unsigned t) {
        const size_t lo = std::min(r1, r0 + t * per), hi = std::min(r1, lo + per);
        auto& v = parts[t];
        v.reserve((hi - lo) / 64 + 16);
        newline_offsets(f->buf, lo, hi, v);
    });
    if (r1 == f->size && r1 > r0 && f->buf[f->size - 1] != '\n') parts[nt - 1].push_back(f->size);   // last line without '\n'
    std::vector<uint64_t> base(nt + 1, 0), before(nt, 0);      // first line number of a part; line end before its first
    {
        uint64_t last_end = (uint64_t)r0 - 1;                   // "line end" before the range (before offset 0: -1)
        for (unsigned t = 0; t < nt; ++t) {
            base[t + 1] = base[t] + parts[t].size();
            before[t] = last_end;
            if (!parts[t].empty()) last_end = parts[t].back();
        }
    }
    const uint64_t nlines = base[nt];
    if (nlines % 4 != 0) return 4;
    const int64_t n = (int64_t)(nlines / 4);
    f->range_end = r1;
    kbbq_resize_fresh(f->h0, n); kbbq_resize_fresh(f->s0, n); kbbq_resize_fresh(f->q0, n); kbbq_resize_fresh(f->hlen, n); kbbq_resize_fresh(f->slen, n);
    raw_vector<uint32_t> qlen((size_t)n);
    std::atomic<int> bad(0);
    {
        kbbq_parallel(nt, [&](unsigned t) {
            {
                if (parts[t].empty()) return;
                const auto& v = parts[t];
                uint64_t start = before[t] + 1;                  // (uint64_t)-1 + 1 == 0 for the file's first line
                for (size_t j = 0; j < v.size(); ++j) {
                    const uint64_t g = base[t] + j, i = g >> 2;
                    uint64_t end = v[j];
                    if (end > start && f->buf[end - 1] == '\r') --end;
                    switch (g & 3) {
                    case 0: {
                        if (end <= start || f->buf[start] != '@') { bad = 1; break; }
                        uint64_t ne = start + 1;                 // name: up to the first whitespace
                        while (ne < end && f->buf[ne] != ' ' && f->buf[ne] != '\t') ++ne;
                        f->h0[i] = start + 1; f->hlen[i] = (uint32_t)(ne - start - 1);
                        break;
                    }
                    case 1:
                        if (end - start > 65535) { bad = 3; f->s0[i] = start; f->slen[i] = 0; break; }
                        f->s0[i] = start; f->slen[i] = (uint32_t)(end - start);
                        break;
                    case 2: break;                               // the '+' line
                    default:
                        f->q0[i] = start; qlen[i] = (uint32_t)std::min<uint64_t>(end - start, 0xFFFFFFFFu);
                    }
                    start = v[j] + 1;
                }
            }
        });
    }
    if (!bad.load())
        parallel_for(n, nt, [&](int64_t lo, int64_t hi) {
            for (int64_t i = lo; i < hi; ++i) if (qlen[(size_t)i] != f->slen[i]) { bad = 2; break; }
        });
    return bad.load();
}

extern "C" {

// byte range [range_lo, range_hi) of the file (range_hi < 0: to the end): only the records inside it are indexed
static int open_impl(const char* path, int64_t range_lo, int64_t range_hi, kbbq_fastq** out)
{
    if (!path || !out) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_open: NULL argument");
    *out = nullptr;
    int fd = open(path, O_RDONLY);
    if (fd < 0) return kbbq_set_error_(KBBQ_E_ARG, (std::string("cannot open ") + path).c_str());
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return kbbq_set_error_(KBBQ_E_ARG, "fstat failed"); }
    kbbq_fastq* f = new kbbq_fastq();
    f->size = (size_t)st.st_size;
    unsigned char magic[2] = {0, 0};
    const bool gz = f->size >= 2 && pread(fd, magic, 2, 0) == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
    if (gz) {
        // gzip / bgzip (what pysam.FastxFile reads transparently): inflate all members into memory -- bgzip's
        // independent blocks in parallel (bam_host.cpp), plain gzip on one thread
        void* m = mmap(nullptr, f->size, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (m == MAP_FAILED) { delete f; return kbbq_set_error_(KBBQ_E_ARG, "mmap failed"); }
        std::string err;
        const bool ok = kbbq_inflate_all((const uint8_t*)m, f->size, f->owned, err);
        munmap(m, f->size);
        if (!ok) { f->size = 0; delete f; return kbbq_set_error_(KBBQ_E_ARG, (std::string(path) + ": " + err).c_str()); }
        f->buf = f->owned.data(); f->size = f->owned.size(); f->mapped = false;
    } else {
        if (f->size) {
            void* m = mmap(nullptr, f->size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { close(fd); delete f; return kbbq_set_error_(KBBQ_E_ARG, "mmap failed"); }
            f->buf = (const uint8_t*)m; f->mapped = true;
            madvise(m, f->size, MADV_SEQUENTIAL);
        }
        close(fd);
    }
    const bool whole = range_lo == 0 && range_hi < 0;
    if (!whole && !f->mapped && f->size) { delete f; return kbbq_set_error_(KBBQ_E_ARG, "a byte range of a compressed FASTQ file cannot be opened"); }
    const size_t r0 = (size_t)std::min<int64_t>(std::max<int64_t>(range_lo, 0), (int64_t)f->size);
    const size_t r1 = range_hi < 0 ? f->size : (size_t)std::min<int64_t>(std::max<int64_t>(range_hi, (int64_t)r0), (int64_t)f->size);
    if ((r0 > 0 && f->buf[r0 - 1] != '\n') || (r1 < f->size && r1 > r0 && f->buf[r1 - 1] != '\n')) {
        delete f; return kbbq_set_error_(KBBQ_E_ARG, (std::string(path) + ": the byte range does not start / end at a line end").c_str());
    }
    int bad = kbbq_fastq_index_range_(f, r0, r1);
    if (bad && bad != 3 && whole) {
        // not four lines per record: a WRAPPED file (sequence and quality over several lines each), which pysam's reader -- kseq --
        // accepts (recalibrate.py:56)?  Unwrap it into memory of our own and index that; anything kseq would not take keeps the error.
        kbbq_bytes flat;
        if (unwrap_fastq(f->buf, f->size, flat)) {
            if (f->mapped && f->buf) munmap((void*)f->buf, f->size);
            f->owned.swap(flat);
            f->buf = f->owned.data(); f->size = f->owned.size(); f->mapped = false;
            f->h0.clear(); f->s0.clear(); f->q0.clear(); f->hlen.clear(); f->slen.clear();
            bad = kbbq_fastq_index_range_(f, 0, f->size);
        }
    }
    if (bad) {
        delete f;
        return kbbq_set_error_(KBBQ_E_ARG, bad == 4 ? (std::string(path) + ": not a 4-line-per-record FASTQ file").c_str()
                                          : bad == 1 ? "record header does not start with @"
                                          : bad == 2 ? "sequence and quality lengths differ" : "read longer than 65535 bases");
    }
    *out = f;
    return KBBQ_OK;
}

int kbbq_fastq_open(const char* path, kbbq_fastq** out) { return open_impl(path, 0, -1, out); }
int64_t kbbq_fastq_sync_offset_ex(const char* path, int64_t offset, int skip_second);

// The byte offset of the first record that starts at or after `offset` in an uncompressed 4-line FASTQ file (the file
// size when there is none): how a rank of a multi-GPU run finds ITS part of the file without anybody having indexed all
// of it.  A record starts at a line that begins with '@' and whose second-next line begins with '+': a quality line may
// begin with '@' too, but the line two below it is a sequence line, which cannot begin with '+'.  Deterministic: every
// rank computes the same cut for the same offset.  Returns -1 on error.
int64_t kbbq_fastq_sync_offset(const char* path, int64_t offset)
{
    return kbbq_fastq_sync_offset_ex(path, offset, 0);
}

// skip_second != 0: a record whose name says "second in pair" (compare_reads.py:304-306: the first '_' field ends with
// "/2") is not a cut point -- the cut moves on to the next record, so that mates stay in one rank's range.
int64_t kbbq_fastq_sync_offset_ex(const char* path, int64_t offset, int skip_second)
{
    if (!path || offset < 0) { kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_sync_offset: bad argument"); return -1; }
    int fd = open(path, O_RDONLY);
    if (fd < 0) { kbbq_set_error_(KBBQ_E_ARG, (std::string("cannot open ") + path).c_str()); return -1; }
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); kbbq_set_error_(KBBQ_E_ARG, "fstat failed"); return -1; }
    const int64_t size = (int64_t)st.st_size;
    if (offset == 0 || offset >= size) { close(fd); return offset == 0 ? 0 : size; }
    std::vector<char> buf;
    int64_t result = -2;
    for (size_t window = 1 << 16; result == -2; window <<= 2) {
        const int64_t from = offset - 1;                              // one byte back: is `offset` itself a line start?
        const size_t want = (size_t)std::min<int64_t>((int64_t)window, size - from);
        buf.resize(want);
        size_t got = 0;
        while (got < want) {
            const ssize_t k = pread(fd, buf.data() + got, want - got, from + (int64_t)got);
            if (k <= 0) break;
            got += (size_t)k;
        }
        if (got < want) { close(fd); kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_sync_offset: read failed"); return -1; }
        const bool whole = from + (int64_t)want == size;
        // line starts inside the window (as offsets into buf)
        std::vector<size_t> starts;
        for (size_t i = 0; i + 1 < want; ++i) if (buf[i] == '\n') starts.push_back(i + 1);
        for (size_t k = 0; k < starts.size() && result == -2; ++k) {
            if (buf[starts[k]] != '@') continue;
            if (k + 2 < starts.size()) {
                if (buf[starts[k + 2]] != '+') continue;
                if (skip_second) {
                    size_t e = starts[k] + 1;                          // name: up to the first whitespace; its first '_' field
                    while (e < starts[k + 1] - 1 && buf[e] != ' ' && buf[e] != '\t' && buf[e] != '_' && buf[e] != '\r') ++e;
                    if (e >= starts[k] + 3 && buf[e - 2] == '/' && buf[e - 1] == '2') continue;
                }
                result = from + (int64_t)starts[k];
            }
            else if (!whole) break;                                   // the deciding line is beyond the window: read more
        }
        if (result == -2 && whole) result = size;                     // no further record
        if (window > ((size_t)1 << 30)) { result = size; }
    }
    close(fd);
    return result;
}

// Only the records inside the byte range [byte_lo, byte_hi) of an uncompressed file (both must be record starts, as
// kbbq_fastq_record_offset gives them; byte_hi < 0: to the end): what one rank of a multi-GPU run needs once rank 0
// has indexed the whole file.  Record i of this reader is record first + i of the file, `first` being the caller's.
int kbbq_fastq_open_range(const char* path, int64_t byte_lo, int64_t byte_hi, kbbq_fastq** out)
{
    if (byte_lo < 0) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_open_range: negative offset");
    return open_impl(path, byte_lo, byte_hi, out);
}

// byte offset of record i's '@' (i == number of records: the end of the indexed range)
int64_t kbbq_fastq_record_offset(const kbbq_fastq* f, int64_t i)
{
    if (!f || i < 0 || i > (int64_t)f->h0.size()) { kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_record_offset: bad index"); return -1; }
    if (i < (int64_t)f->h0.size()) return (int64_t)f->h0[i] - 1;
    return f->h0.empty() ? 0 : (int64_t)f->range_end;
}

int kbbq_fastq_is_plain(const kbbq_fastq* f) { return f && (f->mapped || f->size == 0) ? 1 : 0; }

// install the read-group names (first-appearance order of the WHOLE file, from the rank that scanned it) that
// kbbq_fastq_fill_range maps names to: `names` holds `count` NUL-terminated strings back to back
int kbbq_fastq_set_rg_names(kbbq_fastq* f, const char* names, int count)
{
    if (!f || count < 0 || (count > 0 && !names)) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_set_rg_names: bad argument");
    f->rg_names.clear();
    for (int i = 0; i < count; ++i) { f->rg_names.emplace_back(names); names += f->rg_names.back().size() + 1; }
    return KBBQ_OK;
}

int kbbq_fastq_close(kbbq_fastq* f) { delete f; return KBBQ_OK; }
int64_t kbbq_fastq_count(const kbbq_fastq* f) { return f ? (int64_t)f->h0.size() : 0; }

int kbbq_fastq_name(const kbbq_fastq* f, int64_t i, const char** name, int* len)
{
    if (!f || i < 0 || i >= (int64_t)f->h0.size()) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_name: bad index");
    *name = (const char*)f->buf + f->h0[i]; *len = (int)f->hlen[i];
    return KBBQ_OK;
}

int kbbq_fastq_rg_count(const kbbq_fastq* f) { return f ? (int)f->rg_names.size() : 0; }
const char* kbbq_fastq_rg_name(const kbbq_fastq* f, int i)
{
    return (f && i >= 0 && i < (int)f->rg_names.size()) ? f->rg_names[(size_t)i].c_str() : "";
}

} // extern "C"

// 16 characters -> 8 bytes of code nibbles in the layout of KBBQ_ROWS_NIBBLES (word w of a chunk holds bases 8w..8w+3 in
// the low nibbles of its bytes and 8w+4..8w+7 in the high nibbles; A0 T1 G2 C3, N 4: compare_reads.py:199).  Returns
// non-zero when a character is none of ACGTN (such a batch keeps character planes: the reference's TypeError rule).
static inline unsigned pack16_scalar(const uint8_t* c, uint8_t* out)
{
    static const uint8_t code_of[8] = {0, 3, 1, 2, 4, 4, 4, 4};              // index (ch >> 1) & 7: A C T G . . . N
    static const uint8_t expect[8] = {'A', 'C', 'T', 'G', 0, 0, 0, 'N'};
    uint8_t k[16]; unsigned bad = 0;
    for (int i = 0; i < 16; ++i) { const unsigned h = (c[i] >> 1) & 7u; k[i] = code_of[h]; bad |= (unsigned)(expect[h] ^ c[i]); }
    for (int w = 0; w < 2; ++w)
        for (int b = 0; b < 4; ++b) out[4 * w + b] = (uint8_t)(k[8 * w + b] | (k[8 * w + 4 + b] << 4));
    return bad;
}

#if defined(__x86_64__)
__attribute__((target("ssse3"))) static inline unsigned pack16_ssse3(const uint8_t* c, uint8_t* out)
{
    const __m128i v = _mm_loadu_si128(reinterpret_cast<const __m128i*>(c));
    const __m128i h = _mm_and_si128(_mm_srli_epi16(v, 1), _mm_set1_epi8(7));
    const __m128i codes = _mm_setr_epi8(0, 3, 1, 2, 4, 4, 4, 4, 0, 3, 1, 2, 4, 4, 4, 4);
    const __m128i expect = _mm_setr_epi8('A', 'C', 'T', 'G', 0, 0, 0, 'N', 'A', 'C', 'T', 'G', 0, 0, 0, 'N');
    const __m128i code = _mm_shuffle_epi8(codes, h);
    const unsigned ok = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_shuffle_epi8(expect, h), v));
    // dword d of `code` = the codes of bases 4d .. 4d+3: out word 0 = d0 | d1 << 4, out word 1 = d2 | d3 << 4
    const __m128i o = _mm_or_si128(code, _mm_slli_epi16(_mm_srli_si128(code, 4), 4));      // codes <= 4: no carry between bytes
    const uint32_t w0 = (uint32_t)_mm_cvtsi128_si32(o), w1 = (uint32_t)_mm_cvtsi128_si32(_mm_srli_si128(o, 8));
    memcpy(out, &w0, 4); memcpy(out + 4, &w1, 4);
    return ok ^ 0xFFFFu;
}
static const bool g_ssse3 = __builtin_cpu_supports("ssse3");
#else
static const bool g_ssse3 = false;
#endif

// a row of `pitch` characters (pitch a multiple of 16) -> pitch / 2 bytes of nibbles
static inline unsigned pack_row(const uint8_t* chars, int pitch, uint8_t* out)
{
    unsigned bad = 0;
#if defined(__x86_64__)
    static const bool simd = g_ssse3 && !getenv("KBBQ_NO_SIMD");
    if (simd) { for (int j = 0; j < pitch; j += 16) bad |= pack16_ssse3(chars + j, out + (j >> 1)); return bad; }
#endif
    for (int j = 0; j < pitch; j += 16) bad |= pack16_scalar(chars + j, out + (j >> 1));
    return bad;
}

// compare_reads.py:304-306: the first '_' field ends with "/2"
static inline bool name_second(const char* s, int n)
{
    int e = 0; while (e < n && s[e] != '_') ++e;
    return e >= 2 && s[e - 2] == '/' && s[e - 1] == '2';
}

// compare_reads.py:308-318: field 1 of the '_' split must start with "RG"; the id is the text after
// its last ':'.  Returns 0 ok, 1 IndexError (no second field), 2 AssertionError.
static inline int name_rg(const char* s, int n, const char** rg, int* rglen)
{
    int a = 0; while (a < n && s[a] != '_') ++a;
    if (a >= n) return 1;                              // split('_')[1] does not exist
    const int fs = a + 1; int fe = fs; while (fe < n && s[fe] != '_') ++fe;
    if (fe - fs < 2 || s[fs] != 'R' || s[fs + 1] != 'G') return 2;
    int c = fe; while (c > fs && s[c - 1] != ':') --c;
    *rg = s + c; *rglen = fe - c;
    return 0;
}

extern "C" {

// Scan reads [0, min(na, nb)) of the pair.  info: [0] n usable (cut at the first error), [1] S,
// [2] R, [3] error kind (0 none, 1 RG IndexError, 2 RG AssertionError, 3 name prefix, 4 length
// mismatch, 5 shorter than the running maximum), [4] error index.  b may be NULL (single file:
// kinds 3-5 are not checked -- pass 2 of the reference accepts any lengths).
static int scan_impl(kbbq_fastq* a, const kbbq_fastq* b, int infer_rg, bool keep_prior, uint32_t prior_longest, int64_t* info);
int kbbq_fastq_scan(kbbq_fastq* a, const kbbq_fastq* b, int infer_rg, int64_t* info) { return scan_impl(a, b, infer_rg, false, 0, info); }

// The scan of ONE SEGMENT of a pair that is being read piece by piece (fastq_stream.cpp): what the reference's single walk
// over the reads carries from read to read is carried from segment to segment by the caller -- the read groups met so far
// (installed with kbbq_fastq_set_rg_names before the call: new ones are appended, ids stay first-appearance order over the
// whole file, recalibrate.py:59-64) and the longest read so far (`prior_longest`: a read shorter than it is the IndexError of
// recalibrate.py:89-101 even when it is the longest of its own segment).  info as kbbq_fastq_scan, indices relative to the
// segment; [1] is the longest USABLE read of this segment only, [2] the number of read groups including the prior ones.
int kbbq_fastq_scan_next(kbbq_fastq* a, const kbbq_fastq* b, int infer_rg, int64_t prior_longest, int64_t* info)
{
    if (prior_longest < 0 || prior_longest > 65535) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_scan_next: bad prior_longest");
    return scan_impl(a, b, infer_rg, true, (uint32_t)prior_longest, info);
}

static int scan_impl(kbbq_fastq* a, const kbbq_fastq* b, int infer_rg, bool keep_prior, uint32_t prior_longest, int64_t* info)
{
    if (!a || !info) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_scan: NULL argument");
    int64_t n = (int64_t)a->h0.size();
    if (b) n = std::min<int64_t>(n, (int64_t)b->h0.size());
    // The reference walks the reads once and stops at the first offender (checks per read in the order of the
    // kinds).  Here contiguous chunks of reads are walked in parallel, each recording ITS first offender, its read
    // groups in first-appearance order and its longest read; the chunks are then merged in read order.  "Shorter
    // than the running maximum" needs the maximum over all earlier chunks: a second parallel walk over the lengths.
    struct Chunk {
        int64_t lo = 0, hi = 0;
        int64_t rg_err = -1; int rg_kind = 0;              // first read whose read group cannot be inferred
        std::vector<std::string> rgs;                       // first-appearance order, up to rg_err
        int64_t pair_err = -1; int pair_kind = 0;          // first read failing check 3, 4 or 5
        uint32_t longest = 0;
    };
    int64_t chunk_reads = 1 << 16; bool forced = false;
    if (const char* e = getenv("KBBQ_SCAN_CHUNK")) if (atoll(e) > 0) { chunk_reads = atoll(e); forced = true; }   // tests: tiny chunks, on threads
    const int64_t nchunks = std::max<int64_t>(1, (n + chunk_reads - 1) / chunk_reads);
    std::vector<Chunk> ch((size_t)nchunks);
    for (int64_t c = 0; c < nchunks; ++c) { ch[(size_t)c].lo = std::min(n, c * chunk_reads); ch[(size_t)c].hi = std::min(n, (c + 1) * chunk_reads); }
    const unsigned nt = (unsigned)std::min<int64_t>(nthreads_for(forced ? (size_t)1 << 40 : (size_t)n * 256), nchunks);
    auto over_chunks = [&](auto body) {
        std::atomic<int64_t> next(0);
        auto run = [&]() { for (int64_t c; (c = next.fetch_add(1)) < nchunks;) body(ch[(size_t)c]); };
        if (nt <= 1) { run(); return; }
        kbbq_parallel(nt, [&](unsigned) { run(); });
    };
    auto rgs_of = [&](int64_t lo, int64_t hi, std::vector<std::string>& out, int64_t* err, int* kind) {
        const char* last = nullptr; int lastlen = -1;          // consecutive reads mostly share a read group
        for (int64_t i = lo; i < hi; ++i) {
            const char* rg; int rl;
            const int rc = name_rg((const char*)a->buf + a->h0[i], (int)a->hlen[i], &rg, &rl);
            if (rc) { if (err) { *err = i; *kind = rc; } return; }
            if (rl == lastlen && memcmp(rg, last, (size_t)rl) == 0) continue;
            last = rg; lastlen = rl;
            bool seen = false;
            for (const auto& s : out) if ((int)s.size() == rl && memcmp(s.data(), rg, (size_t)rl) == 0) { seen = true; break; }
            if (!seen) out.emplace_back(rg, (size_t)rl);
        }
    };
    over_chunks([&](Chunk& c) {
        if (infer_rg) rgs_of(c.lo, c.hi, c.rgs, &c.rg_err, &c.rg_kind);
        uint32_t longest = 0;
        for (int64_t i = c.lo; i < c.hi; ++i) longest = std::max(longest, a->slen[i]);
        c.longest = longest;
        if (!b) return;
        for (int64_t i = c.lo; i < c.hi; ++i) {
            const uint32_t la = a->hlen[i], lb = b->hlen[i];
            if (lb < la || memcmp(a->buf + a->h0[i], b->buf + b->h0[i], la) != 0) { c.pair_err = i; c.pair_kind = 3; break; }
            if (a->slen[i] != b->slen[i]) { c.pair_err = i; c.pair_kind = 4; break; }
        }
    });
    if (b) {
        std::vector<uint32_t> before((size_t)nchunks, prior_longest);      // longest read of all earlier chunks (and segments)
        for (int64_t c = 1; c < nchunks; ++c) before[(size_t)c] = std::max(before[(size_t)c - 1], ch[(size_t)c - 1].longest);
        over_chunks([&](Chunk& c) {
            uint32_t runmax = before[(size_t)(&c - ch.data())];
            const int64_t stop = c.pair_err >= 0 ? c.pair_err : c.hi;      // checks 3 and 4 come first at the same read
            for (int64_t i = c.lo; i < stop; ++i) {
                if (a->slen[i] < runmax) { c.pair_err = i; c.pair_kind = 5; break; }
                runmax = std::max(runmax, a->slen[i]);
            }
        });
    }
    // merge in read order
    int64_t err_idx = -1; int err_kind = 0;
    auto consider = [&](int64_t idx, int kind) {         // lowest index wins; kinds are in check order
        if (idx >= 0 && (err_idx < 0 || idx < err_idx || (idx == err_idx && kind < err_kind))) { err_idx = idx; err_kind = kind; }
    };
    const std::vector<std::string> prior = keep_prior ? a->rg_names : std::vector<std::string>();
    a->rg_names = prior;
    auto add_rgs = [](std::vector<std::string>& to, const std::vector<std::string>& from) {
        for (const auto& s : from) if (std::find(to.begin(), to.end(), s) == to.end()) to.push_back(s);
    };
    if (infer_rg) {
        for (const auto& c : ch) {
            add_rgs(a->rg_names, c.rgs);
            if (c.rg_err >= 0) { consider(c.rg_err, c.rg_kind); break; }      // the reference never looks further
        }
    } else if (n > 0 && a->rg_names.empty()) a->rg_names.emplace_back("0");
    for (const auto& c : ch) if (c.pair_err >= 0) { consider(c.pair_err, c.pair_kind); break; }
    int64_t usable = n;
    if (err_idx >= 0) usable = err_idx + (err_kind == 5 ? 1 : 0);
    // longest read and read groups among the usable reads only
    uint32_t S = 0;
    std::vector<std::string> seen = infer_rg ? prior : std::vector<std::string>();
    for (const auto& c : ch) {
        if (c.lo >= usable) break;
        if (c.hi <= usable) {
            S = std::max(S, c.longest);
            if (infer_rg) add_rgs(seen, c.rgs);
        } else {
            for (int64_t i = c.lo; i < usable; ++i) S = std::max(S, a->slen[i]);
            if (infer_rg) { std::vector<std::string> part; rgs_of(c.lo, usable, part, nullptr, nullptr); add_rgs(seen, part); }
        }
    }
    const int R = (usable > 0 || !prior.empty()) ? (infer_rg ? (int)seen.size() : 1) : 0;
    if (R > 32767) return kbbq_set_error_(KBBQ_E_ARG, "more than 32767 read groups: the sidecar word holds 15 bits of read-group id");
    info[0] = usable; info[1] = S; info[2] = R; info[3] = err_kind; info[4] = err_idx;
    return KBBQ_OK;
}

// Open + index both files of a pair (at the same time) and scan them, on threads of the library's own: the caller
// -- Python, which would otherwise hold or wait for its interpreter lock between the calls -- is free to do its
// one-time device set-up meanwhile.  kbbq_fastq_pair_wait joins, hands the two readers and the scan's info[5] over
// and frees the job (call it exactly once); file A's error is reported before file B's, like two opens in a row.
struct kbbq_fastq_job {
    std::thread th;
    std::string path_a, path_b; bool has_b = false; int infer_rg = 0;
    kbbq_fastq* a = nullptr; kbbq_fastq* b = nullptr;
    int rc = KBBQ_OK; std::string err; int64_t info[5] = {0, 0, 0, 0, 0};
};

int kbbq_fastq_pair_begin(const char* path_a, const char* path_b, int infer_rg, kbbq_fastq_job** out)
{
    if (!path_a || !out) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_pair_begin: NULL argument");
    kbbq_fastq_job* j = new kbbq_fastq_job();
    j->path_a = path_a; j->has_b = path_b != nullptr; if (path_b) j->path_b = path_b; j->infer_rg = infer_rg;
    j->th = std::thread([j]() {
        int rc_b = KBBQ_OK; std::string err_b;
        std::thread tb;
        if (j->has_b) tb = std::thread([&]() { rc_b = kbbq_fastq_open(j->path_b.c_str(), &j->b); if (rc_b) err_b = kbbq_last_error(); });
        j->rc = kbbq_fastq_open(j->path_a.c_str(), &j->a);
        if (j->rc) j->err = kbbq_last_error();
        if (tb.joinable()) tb.join();
        if (!j->rc && rc_b) { j->rc = rc_b; j->err = err_b; }
        if (!j->rc) { j->rc = kbbq_fastq_scan(j->a, j->b, j->infer_rg, j->info); if (j->rc) j->err = kbbq_last_error(); }
        if (j->rc) { delete j->a; delete j->b; j->a = j->b = nullptr; }
    });
    *out = j;
    return KBBQ_OK;
}

int kbbq_fastq_pair_wait(kbbq_fastq_job* j, kbbq_fastq** a, kbbq_fastq** b, int64_t* info)
{
    if (!j) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_pair_wait: NULL job");
    if (j->th.joinable()) j->th.join();
    const int rc = j->rc;
    if (rc) kbbq_set_error_(rc, j->err.c_str());
    else if (!a || !info || (j->has_b && !b)) { delete j->a; delete j->b; delete j; return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_pair_wait: NULL argument"); }
    else { *a = j->a; if (b) *b = j->b; memcpy(info, j->info, sizeof j->info); }
    delete j;
    return rc;
}

// Fill the padded planes for reads [0, n): seq / qual from `a`, cseq from `b` (b, cseq may be NULL).
// Padding: 'N' in seq / cseq, 0 in qual (include/kbbq_hip.h).  Call kbbq_fastq_scan first (it
// builds the read-group table used here).
// sequence lengths of reads [first, first + n)
int kbbq_fastq_lengths(const kbbq_fastq* f, int64_t first, int64_t n, uint32_t* out)
{
    if (!f || (n > 0 && !out) || first < 0 || n < 0 || first + n > (int64_t)f->h0.size())
        return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_lengths: bad argument");
    for (int64_t i = 0; i < n; ++i) out[i] = f->slen[first + i];
    return KBBQ_OK;
}

// Length bands of reads [first, first + n) (kbbq/fastx.py length_bands): maximal runs of reads whose lengths fall
// into the same class -- class of a length = index of the first entry of classes[nclasses] (ascending) that is >= it,
// empty reads counted as length 1.  out[run] = {lo, hi, longest, shortest non-empty (0: none)}, lo / hi relative to
// `first`.  Returns the number of runs; when there are more than max_runs, ONE run covering everything.
int kbbq_fastq_length_runs(const kbbq_fastq* f, int64_t first, int64_t n, const uint32_t* classes, int nclasses,
                           int max_runs, int64_t* out)
{
    if (!f || !out || (nclasses > 0 && !classes) || max_runs < 1 || first < 0 || n < 0 || first + n > (int64_t)f->h0.size())
        return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_length_runs: bad argument");
    if (n == 0) return 0;
    auto cls = [&](uint32_t len) { return (int)(std::lower_bound(classes, classes + nclasses, std::max<uint32_t>(len, 1)) - classes); };
    int runs = 0, cur = -1;
    uint32_t all_long = 0, all_short = 0;
    uint32_t above = 1, upto = 0;                              // the current class holds lengths in (above, upto]: empty at first
    for (int64_t i = 0; i < n; ++i) {
        const uint32_t L = f->slen[first + i];
        all_long = std::max(all_long, L);
        if (L && (all_short == 0 || L < all_short)) all_short = L;
        if (runs > max_runs) continue;                         // too many: only the overall figures are still needed
        const uint32_t L1 = std::max<uint32_t>(L, 1);
        int c = cur;
        if (!(L1 > above && L1 <= upto)) {                     // sorted input: almost every read stays in its neighbour's class
            c = cls(L);
            if (c != cur) {
                above = c > 0 ? classes[c - 1] : 0u;
                upto = c < nclasses ? classes[c] : 0xFFFFFFFFu;
            }
        }
        if (c != cur) {
            cur = c;
            if (++runs > max_runs) continue;
            int64_t* o = out + 4 * (runs - 1);
            o[0] = i; o[1] = i + 1; o[2] = L; o[3] = L;
        } else {
            int64_t* o = out + 4 * (runs - 1);
            o[1] = i + 1; o[2] = std::max<int64_t>(o[2], L);
            if (L && (o[3] == 0 || L < o[3])) o[3] = L;
        }
    }
    if (runs > max_runs) { out[0] = 0; out[1] = n; out[2] = all_long; out[3] = all_short; return 1; }
    return runs;
}

int kbbq_fastq_fill(const kbbq_fastq* a, const kbbq_fastq* b, int infer_rg, int64_t n, int pitch,
                    uint8_t* seq, uint8_t* cseq, uint8_t* qual, uint32_t* meta)
{
    return kbbq_fastq_fill_range(a, b, infer_rg, 0, n, pitch, seq, cseq, qual, meta);
}

// reads [first, first + n) into rows [0, n): the shard of one rank (read groups keep the ids of the scan)
int kbbq_fastq_fill_range(const kbbq_fastq* a, const kbbq_fastq* b, int infer_rg, int64_t first, int64_t n, int pitch,
                          uint8_t* seq, uint8_t* cseq, uint8_t* qual, uint32_t* meta)
{
    if (!a || (n > 0 && (!seq || !qual || !meta || (b && !cseq)))) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_fill: NULL argument");
    if (first < 0 || n < 0 || first + n > (int64_t)a->h0.size() || (b && first + n > (int64_t)b->h0.size()))
        return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_fill: range out of bounds");
    std::unordered_map<std::string, int> rgmap;
    for (size_t i = 0; i < a->rg_names.size(); ++i) rgmap.emplace(a->rg_names[i], (int)i);
    std::atomic<int> bad(0);
    parallel_for(n, nthreads_for((size_t)n * (size_t)pitch * 3), [&](int64_t lo, int64_t hi) {
        for (int64_t row = lo; row < hi; ++row) {
            const int64_t i = first + row;
            const uint32_t L = a->slen[i];
            if ((int)L > pitch || (b && b->slen[i] != L)) { bad = 1; continue; }
            uint8_t* s = seq + (size_t)row * pitch; uint8_t* q = qual + (size_t)row * pitch;
            memcpy(s, a->buf + a->s0[i], L); memset(s + L, 'N', (size_t)pitch - L);
            memcpy(q, a->buf + a->q0[i], L); memset(q + L, 0, (size_t)pitch - L);
            if (b) {
                uint8_t* c = cseq + (size_t)row * pitch;
                memcpy(c, b->buf + b->s0[i], L); memset(c + L, 'N', (size_t)pitch - L);
            }
            const char* nm = (const char*)a->buf + a->h0[i]; const int nl = (int)a->hlen[i];
            uint32_t rgid = 0;
            if (infer_rg) {
                const char* rg; int rl;
                if (name_rg(nm, nl, &rg, &rl)) { bad = 2; continue; }
                auto it = rgmap.find(std::string(rg, (size_t)rl));
                if (it == rgmap.end()) { bad = 2; continue; }
                rgid = (uint32_t)it->second;
            }
            meta[row] = L | (rgid << 16) | ((uint32_t)name_second(nm, nl) << 31);
        }
    });
    if (bad.load()) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_fill: input does not match the scan (call kbbq_fastq_scan first)");
    return KBBQ_OK;
}

// ---- the packer writes the device layout itself --------------------------------------------
// The kernels run fastest on mate-pair rows with 4-bit sequence planes, rows gathered by read-group segment
// (include/kbbq_hip.h KBBQ_ROWS_*).  Rounds 1-2 uploaded one character row per read and converted on the device
// (k7_lay_out before K1, an unpack pass after K2); here the host writes the destination rows straight from the FASTQ
// text into the page-locked slabs -- same bytes as kbbq_lay_out_dev produces, 2 B/base over PCIe instead of 3, no pass.

// Sidecar words (len | read group << 16 | second << 31; compare_reads.py:304-318) of reads [first, first + n) and the
// statistics the layout decision needs -- the host twin of kbbq_meta_stats_dev: stats8[0] shortest non-empty read
// (0x7FFFFFFF: none), [1] longest, [2] largest read-group id, [3] violations of "uniform first / second pairs of one
// length and read group", [4] empty reads, [5] violations of the KBBQ_ROWS_TWINS preconditions.  Read-group ids are
// those of the scan (kbbq_fastq_scan / kbbq_fastq_set_rg_names).
int kbbq_fastq_meta(const kbbq_fastq* a, int infer_rg, int64_t first, int64_t n, uint32_t* meta, int32_t* stats8)
{
    if (!a || (n > 0 && !meta) || !stats8) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_meta: NULL argument");
    if (first < 0 || n < 0 || first + n > (int64_t)a->h0.size()) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_meta: range out of bounds");
    std::unordered_map<std::string, int> rgmap;
    for (size_t i = 0; i < a->rg_names.size(); ++i) rgmap.emplace(a->rg_names[i], (int)i);
    std::atomic<int> bad(0);
    parallel_for(n, nthreads_for((size_t)n * 256), [&](int64_t lo, int64_t hi) {
        const char* last = nullptr; int lastlen = -1; uint32_t lastid = 0;       // neighbours mostly share a read group
        for (int64_t row = lo; row < hi; ++row) {
            const int64_t i = first + row;
            const char* nm = (const char*)a->buf + a->h0[i]; const int nl = (int)a->hlen[i];
            uint32_t rgid = 0;
            if (infer_rg) {
                const char* rg; int rl;
                if (name_rg(nm, nl, &rg, &rl)) { bad = 2; meta[row] = 0; continue; }
                if (rl == lastlen && memcmp(rg, last, (size_t)rl) == 0) rgid = lastid;
                else {
                    auto it = rgmap.find(std::string(rg, (size_t)rl));
                    if (it == rgmap.end()) { bad = 2; meta[row] = 0; continue; }
                    rgid = (uint32_t)it->second; last = rg; lastlen = rl; lastid = rgid;
                }
            }
            meta[row] = a->slen[i] | (rgid << 16) | ((uint32_t)name_second(nm, nl) << 31);
        }
    });
    if (bad.load()) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_meta: input does not match the scan (call kbbq_fastq_scan first)");
    // statistics (k7_meta_stats, kbbq_layout_kernels.h), pair by pair, on the same threads
    const uint32_t len0 = n > 0 ? (meta[0] & 0xFFFFu) : 0u;
    const int64_t npairs = (n + 1) >> 1;
    struct Part { int mn = 0x7FFFFFFF, mx = 0, rgmax = 0; int64_t viol = 0, empty = 0, tviol = 0; };
    std::vector<Part> parts;
    std::mutex guard;
    parallel_for(npairs, nthreads_for((size_t)n * 256), [&](int64_t lo, int64_t hi) {
        Part p;
        for (int64_t pr = lo; pr < hi; ++pr) {
            const bool has2 = 2 * pr + 1 < n;
            const uint32_t m[2] = {meta[2 * pr], has2 ? meta[2 * pr + 1] : 0xFFFFFFFFu};
            for (int k = 0; k < (has2 ? 2 : 1); ++k) {
                const int len = (int)(m[k] & 0xFFFFu), rg = (int)((m[k] >> 16) & 0x7FFFu);
                if (len) p.mn = std::min(p.mn, len); else ++p.empty;
                p.mx = std::max(p.mx, len); p.rgmax = std::max(p.rgmax, rg);
                if ((m[k] & 0xFFFFu) != len0) { ++p.viol; ++p.tviol; }
                if ((m[k] >> 31) != 0u) ++p.tviol;
            }
            if ((m[0] >> 31) != 0u) ++p.viol;
            if (has2 && ((m[1] >> 31) == 0u || ((m[0] ^ m[1]) & 0x7FFF0000u) != 0u)) ++p.viol;
            if (!has2) ++p.viol;
            if (has2 && ((m[0] ^ m[1]) & 0x7FFF0000u) != 0u) ++p.tviol;
        }
        std::lock_guard<std::mutex> hold(guard);
        parts.push_back(p);
    });
    int mn = 0x7FFFFFFF, mx = 0, rgmax = 0; int64_t viol = 0, empty = 0, tviol = 0;
    for (const Part& p : parts) {
        mn = std::min(mn, p.mn); mx = std::max(mx, p.mx); rgmax = std::max(rgmax, p.rgmax);
        viol += p.viol; empty += p.empty; tviol += p.tviol;
    }
    auto cap = [](int64_t v) { return (int32_t)std::min<int64_t>(v, 0x7FFFFFFF); };
    stats8[0] = mn; stats8[1] = mx; stats8[2] = rgmax; stats8[3] = cap(viol); stats8[4] = cap(empty); stats8[5] = cap(tviol);
    stats8[6] = stats8[7] = 0;
    return KBBQ_OK;
}

// Stable counting sort of rows by read group on the host -- the twin of kbbq_group_rows_dev: perm[nrows] (row i of
// the grouped order is row perm[i]) and seg[R + 1].  pairs != 0: a row is a PAIR of reads, its group the first
// mate's sidecar's.  A sidecar with a read group >= R -> KBBQ_E_RANGE.
int kbbq_group_rows_host(const uint32_t* meta, int64_t nrows, int pairs, int R, int64_t* perm, int64_t* seg)
{
    if (nrows < 0 || R <= 0 || R > 32767 || !seg || (nrows > 0 && (!meta || !perm))) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_group_rows_host: bad argument");
    const unsigned nt = nthreads_for((size_t)nrows * 64);
    const int64_t per = (nrows + nt - 1) / std::max(nt, 1u);
    std::vector<std::vector<int64_t>> cnt(nt, std::vector<int64_t>((size_t)R, 0));
    std::atomic<int> bad(0);
    auto key = [&](int64_t i) { return (int)((meta[pairs ? 2 * i : i] >> 16) & 0x7FFFu); };
    auto over = [&](auto body) {
        kbbq_parallel(nt, [&](unsigned t) {
            const int64_t lo = std::min<int64_t>(nrows, (int64_t)t * per), hi = std::min<int64_t>(nrows, lo + per);
            body(t, lo, hi);
        });
    };
    over([&](unsigned t, int64_t lo, int64_t hi) {
        auto& c = cnt[t];
        for (int64_t i = lo; i < hi; ++i) { const int k = key(i); if (k >= R) { bad = 1; continue; } ++c[(size_t)k]; }
    });
    if (bad.load()) return kbbq_set_error_(KBBQ_E_RANGE, "kbbq_group_rows_host: a row carries a read group >= R");
    int64_t run = 0;
    for (int g = 0; g < R; ++g) {
        seg[g] = run;
        for (unsigned t = 0; t < nt; ++t) { const int64_t c = cnt[t][(size_t)g]; cnt[t][(size_t)g] = run; run += c; }
    }
    seg[R] = nrows;
    over([&](unsigned t, int64_t lo, int64_t hi) {
        auto& c = cnt[t];
        for (int64_t i = lo; i < hi; ++i) perm[c[(size_t)key(i)]++] = i;
    });
    return KBBQ_OK;
}

// Destination rows [row_lo, row_lo + nrows) of reads [first, first + n) in the layout `flags` describes, written into
// plane rows 0 .. nrows - 1 (seq / cseq: row stride pitch / 2 with KBBQ_ROWS_NIBBLES; cseq and b may be NULL):
//   KBBQ_ROWS_PAIRS  row r = reads 2s, 2s + 1 as [mate 1: S][separator][mate 2: S][padding], S = S2 / 2, pitch =
//                    kbbq_pair_pitch(S2) (the `pitch` argument is ignored), sidecar (2S + 1) | read group << 16; every
//                    read must have length S (the caller checked kbbq_fastq_meta's statistics); an odd n leaves the last
//                    row's second half as padding (KBBQ_ROWS_TWINS input);
//   otherwise        row r = read s at the caller's `pitch`, sidecar = the read's;
//   s = perm[r] (rows gathered by read-group segment: kbbq_group_rows_host) or r when perm is NULL.
// meta: the sidecars of kbbq_fastq_meta for the same [first, first + n).  *foreign is set to 1 (never cleared) when
// nibble packing met a seq / cseq character outside ACGTN: the caller then repeats the band with character planes.
int kbbq_fastq_fill_rows(const kbbq_fastq* a, const kbbq_fastq* b, int64_t first, int64_t n, const uint32_t* meta, int flags,
                         int S2, int pitch, const int64_t* perm, int64_t row_lo, int64_t nrows,
                         uint8_t* seq, uint8_t* cseq, uint8_t* qual, uint32_t* dmeta, int* foreign)
{
    if (!a || (nrows > 0 && (!seq || !qual || !dmeta || !meta || (b && !cseq)))) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_fill_rows: NULL argument");
    if (first < 0 || n < 0 || first + n > (int64_t)a->h0.size() || (b && first + n > (int64_t)b->h0.size()))
        return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_fill_rows: range out of bounds");
    if (flags & ~(KBBQ_ROWS_PAIRS | KBBQ_ROWS_NIBBLES | KBBQ_ROWS_TWINS)) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_fill_rows: unknown layout flags");
    const bool pairs = (flags & KBBQ_ROWS_PAIRS) != 0, nib = (flags & KBBQ_ROWS_NIBBLES) != 0;
    const int S = S2 / 2;
    if (pairs) {
        if (S2 <= 0 || (S2 & 1) || S2 > 65534) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_fill_rows: S2 must be positive and even");
        pitch = (S2 + 1 + 15) & ~15;
    }
    if (pitch <= 0 || (pitch & 15)) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_fill_rows: pitch must be a positive multiple of 16");
    const int64_t total = pairs ? (n + 1) / 2 : n;
    if (row_lo < 0 || nrows < 0 || row_lo + nrows > total) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_fill_rows: rows out of bounds");
    const size_t sp = nib ? (size_t)pitch / 2 : (size_t)pitch;
    std::atomic<int> bad(0); std::atomic<unsigned> odd(0);
    parallel_for(nrows, nthreads_for((size_t)nrows * (size_t)pitch * 3), [&](int64_t lo, int64_t hi) {
        std::vector<uint8_t> tmp(nib ? (size_t)pitch : 0);
        unsigned foreign_here = 0;
        for (int64_t row = lo; row < hi; ++row) {
            const int64_t s = perm ? perm[row_lo + row] : row_lo + row;
            if (s < 0 || s >= total) { bad = 3; continue; }
            uint8_t* q = qual + (size_t)row * pitch;
            uint8_t* planes[2] = {seq + (size_t)row * sp, b ? cseq + (size_t)row * sp : nullptr};
            const kbbq_fastq* src[2] = {a, b};
            if (pairs) {
                const int64_t i1 = first + 2 * s; const bool lone = 2 * s + 1 >= n; const int64_t i2 = lone ? i1 : i1 + 1;
                if ((int)a->slen[i1] != S || (int)a->slen[i2] != S || (b && ((int)b->slen[i1] != S || (int)b->slen[i2] != S))) { bad = 1; continue; }
                memcpy(q, a->buf + a->q0[i1], (size_t)S);
                if (lone) memset(q + S, 0, (size_t)(pitch - S));
                else { q[S] = 0; memcpy(q + S + 1, a->buf + a->q0[i2], (size_t)S); memset(q + 2 * S + 1, 0, (size_t)(pitch - 2 * S - 1)); }
                for (int pl = 0; pl < 2; ++pl) {
                    if (!planes[pl]) continue;
                    uint8_t* c = nib ? tmp.data() : planes[pl];
                    memcpy(c, src[pl]->buf + src[pl]->s0[i1], (size_t)S);
                    if (lone) memset(c + S, 'N', (size_t)(pitch - S));
                    else { c[S] = 'N'; memcpy(c + S + 1, src[pl]->buf + src[pl]->s0[i2], (size_t)S); memset(c + 2 * S + 1, 'N', (size_t)(pitch - 2 * S - 1)); }
                    if (nib) foreign_here |= pack_row(c, pitch, planes[pl]);
                }
                dmeta[row] = (uint32_t)(2 * S + 1) | (meta[2 * s] & 0x7FFF0000u);
            } else {
                const int64_t i = first + s;
                const uint32_t L = a->slen[i];
                if ((int)L > pitch || (b && b->slen[i] != L)) { bad = 1; continue; }
                memcpy(q, a->buf + a->q0[i], L); memset(q + L, 0, (size_t)pitch - L);
                for (int pl = 0; pl < 2; ++pl) {
                    if (!planes[pl]) continue;
                    uint8_t* c = nib ? tmp.data() : planes[pl];
                    memcpy(c, src[pl]->buf + src[pl]->s0[i], L); memset(c + L, 'N', (size_t)pitch - L);
                    if (nib) foreign_here |= pack_row(c, pitch, planes[pl]);
                }
                dmeta[row] = meta[s];
            }
        }
        if (foreign_here) odd = 1;
    });
    if (bad.load()) return kbbq_set_error_(KBBQ_E_ARG, bad.load() == 3 ? "kbbq_fastq_fill_rows: perm holds a row out of range"
                                                      : "kbbq_fastq_fill_rows: a read does not fit the layout (lengths differ from what the statistics said)");
    if (odd.load() && foreign) *foreign = 1;
    return KBBQ_OK;
}

// recalibrate.py:153-156 for reads [first, first + n): "@name\nsequence\n+\nquality\n" with the
// quality characters taken from rows [0, n) of `newqual` (pitch bytes each).  Returns the number of
// bytes written into out (capacity cap) or -needed when cap is too small.
int64_t kbbq_fastq_format(const kbbq_fastq* a, int64_t first, int64_t n, int pitch, const uint8_t* newqual,
                          char* out, int64_t cap)
{
    return kbbq_fastq_format_rows(a, first, n, 0, 0, pitch, newqual, out, cap);
}

// The same with the new qualities still in the layout K2 wrote them in: flags & KBBQ_ROWS_PAIRS -> read first + i sits in
// row i / 2 of `newqual` at byte offset (i & 1) * (S2 / 2 + 1), row stride kbbq_pair_pitch(S2) (`first` is the first
// mate of newqual's row 0; `pitch` is ignored) -- the writer reads mate-pair rows as they are, no unpack pass.
int64_t kbbq_fastq_format_rows(const kbbq_fastq* a, int64_t first, int64_t n, int flags, int S2, int pitch,
                               const uint8_t* newqual, char* out, int64_t cap)
{
    if (!a || first < 0 || n < 0 || first + n > (int64_t)a->h0.size()) { kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_format: bad range"); return 0; }
    const bool pairs = (flags & KBBQ_ROWS_PAIRS) != 0;
    const int S = S2 / 2;
    if (pairs) {
        if (S2 <= 0 || (S2 & 1)) { kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_format_rows: S2 must be positive and even"); return 0; }
        pitch = (S2 + 1 + 15) & ~15;
    }
    std::vector<int64_t> off((size_t)n + 1, 0);
    for (int64_t i = 0; i < n; ++i) off[(size_t)i + 1] = off[(size_t)i] + 1 + a->hlen[first + i] + 1 + a->slen[first + i] + 3 + a->slen[first + i] + 1;
    if (off[(size_t)n] > cap) return -off[(size_t)n];
    std::atomic<int> bad(0);
    parallel_for(n, nthreads_for((size_t)off[(size_t)n]), [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            char* p = out + off[(size_t)i];
            const int64_t r = first + i; const uint32_t L = a->slen[r];
            if (pairs && (int)L != S) { bad = 1; continue; }
            const uint8_t* nq = pairs ? newqual + (size_t)(i >> 1) * pitch + (size_t)(i & 1) * (size_t)(S + 1) : newqual + (size_t)i * pitch;
            *p++ = '@'; memcpy(p, a->buf + a->h0[r], a->hlen[r]); p += a->hlen[r]; *p++ = '\n';
            memcpy(p, a->buf + a->s0[r], L); p += L; *p++ = '\n'; *p++ = '+'; *p++ = '\n';
            memcpy(p, nq, L); p += L; *p++ = '\n';
        }
    });
    if (bad.load()) { kbbq_set_error_(KBBQ_E_ARG, "kbbq_fastq_format_rows: a read of a mate-pair row does not have length S2 / 2"); return 0; }
    return off[(size_t)n];
}

} // extern "C"
