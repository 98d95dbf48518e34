// sam_host.cpp -- host-side SAM text reader for the truth-set benchmark and the BAM-sourced tally (no GPU code).
//
// Replaces, for those paths, what the reference gets from pysam.AlignmentFile / AlignedSegment
// (benchmark.py:57-74,102-143; gatk/bqsr.py:23-123): per alignment the flag, contig, position, CIGAR,
// mate position, template length, sequence, qualities and the RG / OQ tags -- as ARRAYS, because the
// kernels (K4 find_errors, K6 canonical reads) take whole batches.  The file is mapped, lines are indexed and
// parsed in parallel; BAM and gzip-compressed SAM are inflated (BGZF blocks in parallel) and BAM records rendered
// as SAM lines first (bam_host.cpp: zlib only, no htslib).
//
// Field semantics follow the SAM specification and pysam's attribute definitions:
//   reference_start = POS - 1; next_reference_start = PNEXT - 1; reference_end = start + sum of M/D/N/=/X;
//   query_alignment_start / _end = leading / trailing soft clip (hard clips outside them are ignored).
// CIGAR operations are stored as (length << 4 | op) with BAM op codes MIDNSHP=X = 0..8; an unknown
// operation letter is stored as op 15 and rejected by K4 (ValueError, like the reference's walk).
#include "../../include/kbbq_hip.h"
#include "host_threads.h"
#include "bam_host.h"

#include <algorithm>
#include <atomic>
#include <functional>
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

int kbbq_set_error_(int code, const char* msg);      // defined in kbbq_hip.hip

struct kbbq_sam {
    const uint8_t* buf = nullptr; size_t size = 0;
    kbbq_bytes owned;                          // the text when it was inflated / rendered from BAM instead of mapped
    std::vector<uint64_t> hdr0; std::vector<uint32_t> hdrlen;      // header lines ('@...')
    std::vector<uint64_t> line0; std::vector<uint32_t> linelen;    // alignment lines
    // per alignment
    std::vector<uint32_t> name_len, seq_len, qual_len, oq_len, rgtag_len, cig_n, clip;
    std::vector<uint64_t> seq0, qual0, oq0, rgtag0, cig_off;
    std::vector<int32_t> flag, contig, ref_span, rg;
    std::vector<int64_t> pos, pnext, tlen;
    std::vector<uint32_t> cigar;
    std::vector<std::string> contigs;          // first-appearance order of RNAME
    std::vector<std::string> rg_ids;           // @RG ID in header order
    ~kbbq_sam() { if (buf && owned.empty()) munmap((void*)buf, size); }
};

namespace {

unsigned threads_for(size_t work) { return kbbq_threads_for(work); }

template <typename F> void par_for(int64_t n, unsigned nt, F f)
{
    if (nt <= 1 || n < 4096) { f(0, n); return; }
    kbbq_parallel_parts((size_t)n, nt, [&](unsigned, size_t lo, size_t hi) { f((int64_t)lo, (int64_t)hi); });      // parked workers (host_threads.h)
}

bool parse_int(const uint8_t* p, const uint8_t* e, int64_t& out)
{
    if (p >= e) return false;
    bool neg = false;
    if (*p == '-' || *p == '+') { neg = *p == '-'; ++p; if (p >= e) return false; }
    int64_t v = 0;
    for (; p < e; ++p) {
        if (*p < '0' || *p > '9') return false;
        if (v > (INT64_MAX - 9) / 10) return false;
        v = v * 10 + (*p - '0');
    }
    out = neg ? -v : v;
    return true;
}

int cigar_code(uint8_t c)
{
    switch (c) {
        case 'M': return 0; case 'I': return 1; case 'D': return 2; case 'N': return 3; case 'S': return 4;
        case 'H': return 5; case 'P': return 6; case '=': return 7; case 'X': return 8; default: return 15;
    }
}

// one pass over a CIGAR string: count the operations (ops == nullptr) or write them
int64_t walk_cigar(const uint8_t* p, const uint8_t* e, uint32_t* ops, bool& bad)
{
    if (e - p == 1 && *p == '*') return 0;
    int64_t n = 0; uint64_t len = 0; bool have = false;
    for (; p < e; ++p) {
        if (*p >= '0' && *p <= '9') { len = len * 10 + (uint64_t)(*p - '0'); have = true; if (len > 0x0FFFFFFFull) bad = true; }
        else {
            if (!have) bad = true;
            if (ops) ops[n] = (uint32_t)(len << 4) | (uint32_t)cigar_code(*p);
            ++n; len = 0; have = false;
        }
    }
    if (have) bad = true;                       // trailing digits without an operation
    return n;
}

}  // namespace

extern "C" {

int kbbq_sam_open(const char* path, kbbq_sam** out)
{
    if (!path || !out) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_sam_open: NULL argument");
    *out = nullptr;
    int fd = open(path, O_RDONLY);
    if (fd < 0) return kbbq_set_error_(KBBQ_E_ARG, (std::string("cannot open ") + path).c_str());
    struct stat st;
    if (fstat(fd, &st) != 0) { close(fd); return kbbq_set_error_(KBBQ_E_ARG, "fstat failed"); }
    kbbq_sam* f = new kbbq_sam();
    f->size = (size_t)st.st_size;
    if (f->size) {
        void* m = mmap(nullptr, f->size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { close(fd); f->size = 0; delete f; return kbbq_set_error_(KBBQ_E_ARG, "mmap failed"); }
        f->buf = (const uint8_t*)m;
        madvise(m, f->size, MADV_SEQUENTIAL);
    }
    close(fd);
    if (f->size >= 4 && (!memcmp(f->buf, "BAM\1", 4) || (f->buf[0] == 0x1f && f->buf[1] == 0x8b))) {
        // gzip / BGZF: inflate; BAM (inflated or not): render the records as SAM lines; then parse the text as usual
        kbbq_bytes raw, text; std::string err;
        bool ok = true;
        const bool zipped = f->buf[0] == 0x1f;
        if (zipped) ok = kbbq_inflate_all(f->buf, f->size, raw, err);
        const uint8_t* img = zipped ? raw.data() : f->buf; const size_t img_n = zipped ? raw.size() : f->size;
        const bool bam = ok && img_n >= 4 && !memcmp(img, "BAM\1", 4);
        if (bam) ok = kbbq_bam_to_sam(img, img_n, text, err);
        if (!ok) { delete f; return kbbq_set_error_(KBBQ_E_ARG, (std::string(path) + ": " + err).c_str()); }
        munmap((void*)f->buf, f->size);
        f->owned = bam ? std::move(text) : std::move(raw);
        if (f->owned.empty()) f->owned.push_back('\n');          // keeps "owned" distinguishable from "mapped"
        f->buf = f->owned.data(); f->size = f->owned.size();
    }
    // line ends, in parallel; then every thread classifies ITS lines (blank / header / alignment) into lists of its own, which are
    // concatenated in thread order = file order (round 4: this loop and the list merge used to run on one thread: 16 M lines)
    const unsigned nt = threads_for(f->size);
    struct Part { std::vector<uint64_t> nl, hdr0, line0; std::vector<uint32_t> hdrlen, linelen; uint64_t first_start = 0; };
    std::vector<Part> parts(nt);
    const size_t per = (f->size + nt - 1) / std::max(nt, 1u);
    kbbq_parallel(nt, [&](unsigned t) {
        const size_t lo = std::min(f->size, (size_t)t * per), hi = std::min(f->size, lo + per);
        const uint8_t* p = f->buf + lo; const uint8_t* e = f->buf + hi;
        auto& v = parts[t].nl;
        v.reserve((hi - lo) / 200 + 16);
        while (p < e) {
            const uint8_t* nl = (const uint8_t*)memchr(p, '\n', (size_t)(e - p));
            if (!nl) break;
            v.push_back((uint64_t)(nl - f->buf));
            p = nl + 1;
        }
    });
    {
        // the last line may lack its line end; a part's first line starts behind the last line end of the parts before it
        int last = -1;
        for (unsigned t = 0; t < nt; ++t) if (!parts[t].nl.empty()) last = (int)t;
        if (f->size && (last < 0 || parts[(size_t)last].nl.back() != f->size - 1)) parts[nt - 1].nl.push_back(f->size);
        uint64_t start = 0;
        for (unsigned t = 0; t < nt; ++t) { parts[t].first_start = start; if (!parts[t].nl.empty()) start = parts[t].nl.back() + 1; }
    }
    kbbq_parallel(nt, [&](unsigned t) {
        Part& P = parts[t];
        uint64_t start = P.first_start;
        for (uint64_t end : P.nl) {
            uint64_t e = end;
            if (e > start && f->buf[e - 1] == '\r') --e;
            bool blank = true;
            for (uint64_t k = start; k < e; ++k) if (f->buf[k] != ' ' && f->buf[k] != '\t') { blank = false; break; }
            if (!blank) {
                if (f->buf[start] == '@') { P.hdr0.push_back(start); P.hdrlen.push_back((uint32_t)(e - start)); }
                else { P.line0.push_back(start); P.linelen.push_back((uint32_t)(e - start)); }
            }
            start = end + 1;
        }
    });
    {
        size_t nh = 0, nl_ = 0;
        for (auto& P : parts) { nh += P.hdr0.size(); nl_ += P.line0.size(); }
        f->hdr0.reserve(nh); f->hdrlen.reserve(nh); f->line0.resize(nl_); f->linelen.resize(nl_);
        std::vector<size_t> at(nt + 1, 0);
        for (unsigned t = 0; t < nt; ++t) at[t + 1] = at[t] + parts[t].line0.size();
        for (auto& P : parts) { f->hdr0.insert(f->hdr0.end(), P.hdr0.begin(), P.hdr0.end()); f->hdrlen.insert(f->hdrlen.end(), P.hdrlen.begin(), P.hdrlen.end()); }
        kbbq_parallel(nt, [&](unsigned t) {
            if (parts[t].line0.empty()) return;
            memcpy(f->line0.data() + at[t], parts[t].line0.data(), parts[t].line0.size() * sizeof(uint64_t));
            memcpy(f->linelen.data() + at[t], parts[t].linelen.data(), parts[t].linelen.size() * sizeof(uint32_t));
        });
    }
    // @RG IDs in header order
    for (size_t h = 0; h < f->hdr0.size(); ++h) {
        const uint8_t* p = f->buf + f->hdr0[h]; const uint8_t* e = p + f->hdrlen[h];
        if (e - p < 3 || memcmp(p, "@RG", 3) != 0) continue;
        std::string id;
        const uint8_t* q = p;
        while (q < e) {
            const uint8_t* t = (const uint8_t*)memchr(q, '\t', (size_t)(e - q));
            const uint8_t* fe = t ? t : e;
            if (fe - q > 3 && q[0] == 'I' && q[1] == 'D' && q[2] == ':') id.assign((const char*)q + 3, (size_t)(fe - q - 3));
            q = fe + 1;
        }
        f->rg_ids.push_back(id);
    }
    std::unordered_map<std::string, int> rgmap;
    for (size_t i = 0; i < f->rg_ids.size(); ++i) rgmap.emplace(f->rg_ids[i], (int)i);

    const int64_t n = (int64_t)f->line0.size();
    std::vector<uint64_t> rname0, cig0; std::vector<uint32_t> rname_len, cig_len;
    {
        // 23 arrays of n entries (2 GB at 16 M alignments): sized and zeroed side by side instead of one after the other
        // (huge pages asked for behind the large ones: raw_vector.h)
        auto fresh = [](auto& v, size_t count, auto value) {
            using T = typename std::remove_reference_t<decltype(v)>::value_type;
            if (count * sizeof(T) >= ((size_t)8 << 20) && v.capacity() < count) { v.reserve(count); kbbq_advise_huge(v.data(), v.capacity() * sizeof(T)); }
            v.assign(count, (T)value);
        };
        const std::function<void()> fills[] = {
            [&]() { fresh(f->name_len, (size_t)(n), 0); }, [&]() { fresh(f->seq_len, (size_t)(n), 0); }, [&]() { fresh(f->qual_len, (size_t)(n), 0); }, [&]() { fresh(f->oq_len, (size_t)(n), 0); },
            [&]() { fresh(f->rgtag_len, (size_t)(n), 0); }, [&]() { fresh(f->cig_n, (size_t)(n), 0); }, [&]() { fresh(f->clip, (size_t)(n), 0); }, [&]() { fresh(f->seq0, (size_t)(n), 0); },
            [&]() { fresh(f->qual0, (size_t)(n), 0); }, [&]() { fresh(f->oq0, (size_t)(n), 0); }, [&]() { fresh(f->rgtag0, (size_t)(n), 0); }, [&]() { fresh(f->cig_off, (size_t)(n + 1), 0); },
            [&]() { fresh(f->flag, (size_t)(n), 0); }, [&]() { fresh(f->contig, (size_t)(n), -1); }, [&]() { fresh(f->ref_span, (size_t)(n), 0); }, [&]() { fresh(f->rg, (size_t)(n), -1); },
            [&]() { fresh(f->pos, (size_t)(n), 0); }, [&]() { fresh(f->pnext, (size_t)(n), 0); }, [&]() { fresh(f->tlen, (size_t)(n), 0); }, [&]() { fresh(rname0, (size_t)(n), 0); },
            [&]() { fresh(cig0, (size_t)(n), 0); }, [&]() { fresh(rname_len, (size_t)(n), 0); }, [&]() { fresh(cig_len, (size_t)(n), 0); }};
        const unsigned nf = (unsigned)(sizeof fills / sizeof fills[0]);
        std::atomic<unsigned> next(0);
        kbbq_parallel(std::min(nt, nf), [&](unsigned) { for (unsigned k; (k = next.fetch_add(1)) < nf;) fills[k](); });
    }
    std::atomic<int64_t> badline(-1);
    auto flagbad = [&](int64_t i) { int64_t cur = badline.load(); while ((cur < 0 || i < cur) && !badline.compare_exchange_weak(cur, i)) {} };
    // pass 1: split fields, numbers, tags, count CIGAR operations
    par_for(n, nt, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            const uint8_t* p = f->buf + f->line0[i]; const uint8_t* e = p + f->linelen[i];
            const uint8_t* fs[12]; const uint8_t* fe[12]; int nf = 0;
            const uint8_t* q = p;
            while (nf < 11 && q <= e) {
                const uint8_t* t = (const uint8_t*)memchr(q, '\t', (size_t)(e - q));
                fs[nf] = q; fe[nf] = t ? t : e; ++nf;
                if (!t) { q = e + 1; break; }
                q = t + 1;
            }
            if (nf < 11) { flagbad(i); continue; }
            int64_t v;
            f->name_len[i] = (uint32_t)(fe[0] - fs[0]);
            if (!parse_int(fs[1], fe[1], v)) { flagbad(i); continue; }
            f->flag[i] = (int32_t)v;
            rname0[i] = (uint64_t)(fs[2] - f->buf); rname_len[i] = (uint32_t)(fe[2] - fs[2]);
            if (!parse_int(fs[3], fe[3], v)) { flagbad(i); continue; }
            f->pos[i] = v - 1;
            cig0[i] = (uint64_t)(fs[5] - f->buf); cig_len[i] = (uint32_t)(fe[5] - fs[5]);
            if (!parse_int(fs[7], fe[7], v)) { flagbad(i); continue; }
            f->pnext[i] = v - 1;
            if (!parse_int(fs[8], fe[8], v)) { flagbad(i); continue; }
            f->tlen[i] = v;
            f->seq0[i] = (uint64_t)(fs[9] - f->buf); f->seq_len[i] = (uint32_t)(fe[9] - fs[9]);
            f->qual0[i] = (uint64_t)(fs[10] - f->buf); f->qual_len[i] = (uint32_t)(fe[10] - fs[10]);
            if (f->seq_len[i] > 65535) { flagbad(i); continue; }
            bool bad = false;
            f->cig_n[i] = (uint32_t)walk_cigar(fs[5], fe[5], nullptr, bad);
            if (bad) { flagbad(i); continue; }
            // tags
            while (q < e) {
                const uint8_t* t = (const uint8_t*)memchr(q, '\t', (size_t)(e - q));
                const uint8_t* te = t ? t : e;
                if (te - q >= 5 && q[2] == ':' && q[4] == ':') {
                    if (q[0] == 'R' && q[1] == 'G' && q[3] == 'Z') { f->rgtag0[i] = (uint64_t)(q + 5 - f->buf); f->rgtag_len[i] = (uint32_t)(te - q - 5) + 1; }
                    else if (q[0] == 'O' && q[1] == 'Q' && q[3] == 'Z') { f->oq0[i] = (uint64_t)(q + 5 - f->buf); f->oq_len[i] = (uint32_t)(te - q - 5) + 1; }
                }
                q = te + 1;
            }
            if (f->rgtag_len[i]) {
                auto it = rgmap.find(std::string((const char*)f->buf + f->rgtag0[i], f->rgtag_len[i] - 1));
                f->rg[i] = it == rgmap.end() ? -2 : it->second;          // -2: tag names a group the header lacks
            }
        }
    });
    if (badline.load() >= 0) {
        const int64_t b = badline.load();
        const std::string head((const char*)f->buf + f->line0[b], std::min<size_t>(60, f->linelen[b]));
        delete f;
        return kbbq_set_error_(KBBQ_E_RANGE, ("not a SAM alignment line: " + head).c_str());
    }
    // contigs in first-appearance order (serial: the order matters)
    std::unordered_map<std::string, int> cmap;
    const uint8_t* last_name = nullptr; uint32_t last_len = 0; int last_id = -1;        // neighbours mostly share a contig: no string, no hash for them
    for (int64_t i = 0; i < n; ++i) {
        const uint8_t* nm = f->buf + rname0[i];
        if (last_name && rname_len[i] == last_len && memcmp(nm, last_name, last_len) == 0) f->contig[i] = last_id;
        else {
            std::string name((const char*)nm, rname_len[i]);
            auto it = cmap.find(name);
            if (it == cmap.end()) { it = cmap.emplace(name, (int)f->contigs.size()).first; f->contigs.push_back(name); }
            f->contig[i] = last_id = it->second; last_name = nm; last_len = rname_len[i];
        }
        f->cig_off[i + 1] = f->cig_off[i] + f->cig_n[i];
    }
    f->cigar.assign(f->cig_off[n], 0);
    // pass 2: CIGAR operations, reference span, soft clips
    par_for(n, nt, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            bool bad = false;
            uint32_t* ops = f->cigar.data() + f->cig_off[i];
            const int64_t m = walk_cigar(f->buf + cig0[i], f->buf + cig0[i] + cig_len[i], ops, bad);
            int64_t span = 0;
            for (int64_t k = 0; k < m; ++k) { const uint32_t op = ops[k] & 15u; if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) span += ops[k] >> 4; }
            f->ref_span[i] = (int32_t)std::min<int64_t>(span, INT32_MAX);
            int64_t lead = 0, tail = 0;
            for (int64_t k = 0; k < m; ++k) { const uint32_t op = ops[k] & 15u; if (op == 4) lead += ops[k] >> 4; else if (op != 5) break; }
            for (int64_t k = m - 1; k >= 0; --k) { const uint32_t op = ops[k] & 15u; if (op == 4) tail += ops[k] >> 4; else if (op != 5) break; }
            const int64_t L = (f->seq_len[i] == 1 && f->buf[f->seq0[i]] == '*') ? 0 : f->seq_len[i];
            const int64_t qs = std::min<int64_t>(lead, 65535), qe = std::max<int64_t>(0, std::min<int64_t>(L - tail, 65535));
            f->clip[i] = (uint32_t)qs | ((uint32_t)qe << 16);
        }
    });
    *out = f;
    return KBBQ_OK;
}

int kbbq_sam_close(kbbq_sam* f) { delete f; return KBBQ_OK; }

// ---- a whole text file, inflated if it is gzip / bgzip: what the Python-side readers of FASTA / VCF / BED files take their bytes from
// (bgzip blocks side by side, gzip members chunk-wise: bam_host.cpp kbbq_inflate_all) instead of Python's gzip module on one thread
struct kbbq_text { const uint8_t* buf = nullptr; size_t size = 0; bool mapped = false; kbbq_bytes owned;
                   ~kbbq_text() { if (mapped && buf) munmap((void*)buf, size); } };

int kbbq_text_open(const char* path, kbbq_text** out, const uint8_t** data, size_t* n)
{
    if (!path || !out || !data || !n) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_text_open: NULL argument");
    *out = nullptr; *data = nullptr; *n = 0;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return kbbq_set_error_(KBBQ_E_ARG, (std::string("cannot open ") + path).c_str());
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) { close(fd); return kbbq_set_error_(KBBQ_E_ARG, (std::string(path) + ": not a regular file").c_str()); }
    kbbq_text* t = new kbbq_text();
    t->size = (size_t)st.st_size;
    if (t->size) {
        void* m = mmap(nullptr, t->size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) { close(fd); delete t; return kbbq_set_error_(KBBQ_E_ARG, "mmap failed"); }
        t->buf = (const uint8_t*)m; t->mapped = true;
    }
    close(fd);
    if (t->size >= 2 && t->buf[0] == 0x1f && t->buf[1] == 0x8b) {
        std::string err;
        if (!kbbq_inflate_all(t->buf, t->size, t->owned, err)) { delete t; return kbbq_set_error_(KBBQ_E_ARG, (std::string(path) + ": " + err).c_str()); }
        munmap((void*)t->buf, t->size);
        t->mapped = false; t->buf = t->owned.data(); t->size = t->owned.size();
    }
    *out = t; *data = t->buf; *n = t->size;
    return KBBQ_OK;
}

int kbbq_text_close(kbbq_text* t) { delete t; return KBBQ_OK; }

// info = { alignments, CIGAR operations in total, longest SEQ, contigs seen, @RG lines, header lines }
int kbbq_sam_info(const kbbq_sam* f, int64_t* info6)
{
    if (!f || !info6) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_sam_info: NULL argument");
    const int64_t n = (int64_t)f->line0.size();
    int64_t maxlen = 0;
    for (int64_t i = 0; i < n; ++i) maxlen = std::max<int64_t>(maxlen, f->seq_len[i]);
    info6[0] = n; info6[1] = (int64_t)f->cigar.size(); info6[2] = maxlen; info6[3] = (int64_t)f->contigs.size();
    info6[4] = (int64_t)f->rg_ids.size(); info6[5] = (int64_t)f->hdr0.size();
    return KBBQ_OK;
}

// any output pointer may be NULL
int kbbq_sam_fields(const kbbq_sam* f, int32_t* flag, int32_t* contig, int64_t* pos, int64_t* pnext, int64_t* tlen,
                    int32_t* qlen, int32_t* ref_span, uint32_t* clip, uint32_t* cig_off, uint32_t* cig_n, int32_t* rg,
                    int32_t* has_qual_oq)
{
    if (!f) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_sam_fields: NULL handle");
    const int64_t n = (int64_t)f->line0.size();
    if (has_qual_oq)                                             // the two lengths share one int32: 16 bits / 15 bits
        for (int64_t i = 0; i < n; ++i)
            if (f->qual_len[i] > 65535 || f->oq_len[i] > 32768)
                return kbbq_set_error_(KBBQ_E_ARG, "a QUAL field longer than 65535 or an OQ tag longer than 32768 characters: malformed line");
    par_for(n, threads_for((size_t)n * 64), [&](int64_t lo, int64_t hi) {
    for (int64_t i = lo; i < hi; ++i) {
        if (flag) flag[i] = f->flag[i];
        if (contig) contig[i] = f->contig[i];
        if (pos) pos[i] = f->pos[i];
        if (pnext) pnext[i] = f->pnext[i];
        if (tlen) tlen[i] = f->tlen[i];
        if (qlen) qlen[i] = (f->seq_len[i] == 1 && f->buf[f->seq0[i]] == '*') ? 0 : (int32_t)f->seq_len[i];
        if (ref_span) ref_span[i] = f->ref_span[i];
        if (clip) clip[i] = f->clip[i];
        if (cig_off) cig_off[i] = (uint32_t)f->cig_off[i];
        if (cig_n) cig_n[i] = f->cig_n[i];
        if (rg) rg[i] = f->rg[i];
        if (has_qual_oq) {
            const bool noqual = f->qual_len[i] == 1 && f->buf[f->qual0[i]] == '*';
            has_qual_oq[i] = (noqual ? 0 : (int32_t)f->qual_len[i]) | (f->oq_len[i] ? (int32_t)(f->oq_len[i] - 1) << 16 : -65536);
        }
    }
    });
    return KBBQ_OK;
}

int kbbq_sam_cigar(const kbbq_sam* f, uint32_t* ops)
{
    if (!f || (!ops && !f->cigar.empty())) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_sam_cigar: NULL argument");
    if (!f->cigar.empty()) memcpy(ops, f->cigar.data(), f->cigar.size() * sizeof(uint32_t));
    return KBBQ_OK;
}

// Query positions [lo, hi) of every alignment that lie past its adaptor boundary, as lo | hi << 16 (0: nothing to trim) --
// what gatk/bqsr.py bamread_adaptor_boundary + _trim_range compute per read (reference bqsr.py:131-206), for all
// alignments at once: the boundary from FLAG / POS / PNEXT / TLEN, then one walk over the CIGAR of the few reads it falls into.
int kbbq_sam_adaptor_trim(const kbbq_sam* f, uint32_t* trim)
{
    if (!f || (!trim && !f->line0.empty())) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_sam_adaptor_trim: NULL argument");
    const int64_t n = (int64_t)f->line0.size();
    par_for(n, threads_for((size_t)n * 64), [&](int64_t lo_i, int64_t hi_i) {
    for (int64_t i = lo_i; i < hi_i; ++i) {
        trim[i] = 0;
        const int32_t fl = f->flag[i];
        const bool rev = fl & 16, mate_rev = fl & 32;
        if (f->tlen[i] == 0 || !(fl & 1) || (fl & 4) || (fl & 8) || rev == mate_rev) continue;
        const int64_t start = f->pos[i], end = start + f->ref_span[i];
        const bool noqual = f->qual_len[i] == 1 && f->buf[f->qual0[i]] == '*';
        const int64_t nq = noqual ? 0 : (int64_t)f->qual_len[i];
        const uint32_t* ops = f->cigar.data() + f->cig_off[i];
        const int64_t nops = f->cig_n[i];
        auto on_query_and_ref = [](uint32_t op) { return op == 0 || op == 7 || op == 8; };
        int64_t lo = 0, hi = 0;
        if (rev) {
            if (!(end - 1 > f->pnext[i])) continue;
            const int64_t boundary = f->pnext[i] - 1;
            if (boundary < start) continue;
            int64_t q = 0, r = end;
            for (int64_t k = 0; k < nops; ++k) { const uint32_t op = ops[k] & 15; if (on_query_and_ref(op) || op == 1 || op == 4) q += ops[k] >> 4; }
            bool reached = false, found = false;
            for (int64_t k = nops - 1; k >= 0 && !found; --k) {
                const uint32_t op = ops[k] & 15; const int64_t l = ops[k] >> 4;
                if (on_query_and_ref(op)) {
                    if (reached) { hi = q; found = true; }
                    else if (r - l <= boundary) { hi = q - ((r - 1) - std::min(boundary, r - 1)); found = true; }
                    else { q -= l; r -= l; }
                } else if (op == 1 || op == 4) {
                    if (reached) { hi = q; found = true; } else q -= l;
                } else if (op == 2 || op == 3) {
                    if (r - l <= boundary) reached = true;
                    r -= l;
                }
            }
            if (!found) continue;
        } else {
            if (!(start <= f->pnext[i] + f->tlen[i])) continue;
            const int64_t boundary = start + (f->tlen[i] < 0 ? -f->tlen[i] : f->tlen[i]);
            if (boundary > end - 1) continue;
            int64_t q = 0, r = start;
            bool reached = false, found = false;
            hi = nq;
            for (int64_t k = 0; k < nops && !found; ++k) {
                const uint32_t op = ops[k] & 15; const int64_t l = ops[k] >> 4;
                if (on_query_and_ref(op)) {
                    if (reached) { lo = q; found = true; }
                    else if (r + l > boundary) { lo = q + std::max<int64_t>(boundary - r, 0); found = true; }
                    else { q += l; r += l; }
                } else if (op == 1 || op == 4) {
                    if (reached) { lo = q; found = true; } else q += l;
                } else if (op == 2 || op == 3) {
                    if (r + l > boundary) reached = true;
                    r += l;
                }
            }
            if (!found) lo = nq;
        }
        trim[i] = (uint32_t)(lo & 0xFFFF) | (uint32_t)(hi & 0xFFFF) << 16;
    }
    });
    return KBBQ_OK;
}

// which: 0 SEQ, 1 QUAL, 2 OQ tag; rows [0, n) <- alignments [first, first + n), zero padded; a missing / '*' field
// leaves its row zero
int kbbq_sam_fill(const kbbq_sam* f, int64_t first, int64_t n, int pitch, int which, uint8_t* plane)
{
    if (!f || (n > 0 && !plane)) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_sam_fill: NULL argument");
    if (first < 0 || n < 0 || first + n > (int64_t)f->line0.size() || pitch <= 0 || which < 0 || which > 2)
        return kbbq_set_error_(KBBQ_E_ARG, "kbbq_sam_fill: bad range / pitch / plane");
    par_for(n, threads_for((size_t)n * (size_t)pitch), [&](int64_t lo, int64_t hi) {
        for (int64_t r = lo; r < hi; ++r) {
            const int64_t i = first + r;
            uint64_t off; uint32_t len;
            if (which == 0) { off = f->seq0[i]; len = f->seq_len[i]; }
            else if (which == 1) { off = f->qual0[i]; len = f->qual_len[i]; }
            else { off = f->oq0[i]; len = f->oq_len[i] ? f->oq_len[i] - 1 : 0; }
            if (len == 1 && which != 2 && f->buf[off] == '*') len = 0;
            len = std::min<uint32_t>(len, (uint32_t)pitch);
            uint8_t* row = plane + (size_t)r * pitch;
            if (len) memcpy(row, f->buf + off, len);
            memset(row + len, 0, (size_t)pitch - len);
        }
    });
    return KBBQ_OK;
}

// what: 0 QNAME of alignment i, 1 whole alignment line i, 2 header line i, 3 contig name i, 4 @RG ID i
int kbbq_sam_text(const kbbq_sam* f, int what, int64_t i, const char** p, int64_t* len)
{
    if (!f || !p || !len) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_sam_text: NULL argument");
    const int64_t lim = what <= 1 ? (int64_t)f->line0.size() : what == 2 ? (int64_t)f->hdr0.size()
                      : what == 3 ? (int64_t)f->contigs.size() : what == 4 ? (int64_t)f->rg_ids.size() : -1;
    if (i < 0 || i >= lim) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_sam_text: index out of range");
    switch (what) {
        case 0: *p = (const char*)f->buf + f->line0[i]; *len = f->name_len[i]; break;
        case 1: *p = (const char*)f->buf + f->line0[i]; *len = f->linelen[i]; break;
        case 2: *p = (const char*)f->buf + f->hdr0[i]; *len = f->hdrlen[i]; break;
        case 3: *p = f->contigs[i].data(); *len = (int64_t)f->contigs[i].size(); break;
        default: *p = f->rg_ids[i].data(); *len = (int64_t)f->rg_ids[i].size(); break;
    }
    return KBBQ_OK;
}

// benchmark.py:102-124: FASTQ reads are matched to alignments by name -- the FASTQ name up to its first '_'
// against QNAME + "/1" | "/2" (second-in-pair flag); of several alignments with one name the LAST wins (a dict
// in the reference).  idx[i] = alignment of FASTQ read i, or -1 (the reference's KeyError).
int kbbq_sam_match_fastq(const kbbq_sam* f, const kbbq_fastq* fq, int64_t* idx)
{
    if (!f || !fq) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_sam_match_fastq: NULL argument");
    const int64_t n = kbbq_fastq_count(fq);
    if (n > 0 && !idx) return kbbq_set_error_(KBBQ_E_ARG, "kbbq_sam_match_fastq: NULL argument");
    std::unordered_map<std::string, int64_t> row;
    const int64_t na = (int64_t)f->line0.size();
    row.reserve((size_t)na * 2);
    for (int64_t i = 0; i < na; ++i) {
        std::string key((const char*)f->buf + f->line0[i], f->name_len[i]);
        key += (f->flag[i] & 128) ? "/2" : "/1";
        row[key] = i;
    }
    par_for(n, threads_for((size_t)n * 64), [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            const char* p = nullptr; int len = 0;
            if (kbbq_fastq_name(fq, i, &p, &len) != KBBQ_OK) { idx[i] = -1; continue; }
            const void* us = memchr(p, '_', (size_t)len);
            if (us) len = (int)((const char*)us - p);
            auto it = row.find(std::string(p, (size_t)len));
            idx[i] = it == row.end() ? -1 : it->second;
        }
    });
    return KBBQ_OK;
}

}  // extern "C"
