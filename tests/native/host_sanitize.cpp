// AddressSanitizer / UBSan harness for the host C++ of libkbbq_hip (fastq_host.cpp, solve_host.cpp):
// g++ -fsanitize=address,undefined ... (tests/test_host_sanitizers.py).  GPU code cannot run under a
// sanitizer on this pool; the host side can, and it is the part that parses untrusted text.
#include "../../include/kbbq_hip.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static thread_local std::string g_err;
int kbbq_set_error_(int code, const char* msg) { g_err = msg ? msg : ""; return code; }
extern "C" const char* kbbq_last_error(void) { return g_err.c_str(); }

static int run_pair(const char* pa, const char* pb, int infer)
{
    {   // the same through the background job: same outcome, nothing leaked on either path
        kbbq_fastq_job* job = nullptr;
        kbbq_fastq *ja = nullptr, *jb = nullptr; int64_t jinfo[5] = {0, 0, 0, 0, 0};
        int jrc = kbbq_fastq_pair_begin(pa, pb, infer, &job);
        if (!jrc) jrc = kbbq_fastq_pair_wait(job, &ja, pb ? &jb : nullptr, jinfo);
        printf("job rc=%d n=%lld S=%lld R=%lld kind=%lld idx=%lld %s\n", jrc, (long long)jinfo[0], (long long)jinfo[1],
               (long long)jinfo[2], (long long)jinfo[3], (long long)jinfo[4], jrc ? g_err.c_str() : "");
        if (!jrc) {
            int64_t runs[4 * 4]; const uint32_t classes[3] = {48, 96, 160};
            const int k = kbbq_fastq_length_runs(ja, 0, kbbq_fastq_count(ja), classes, 3, 4, runs);
            printf("length runs %d\n", k);
            kbbq_fastq_close(ja); if (jb) kbbq_fastq_close(jb);
        }
    }
    kbbq_fastq *a = nullptr, *b = nullptr;
    int rc = kbbq_fastq_open(pa, &a);
    if (rc) { printf("open A rc=%d (%s)\n", rc, g_err.c_str()); return 0; }
    if (pb) { rc = kbbq_fastq_open(pb, &b); if (rc) { printf("open B rc=%d (%s)\n", rc, g_err.c_str()); kbbq_fastq_close(a); return 0; } }
    int64_t info[5] = {0, 0, 0, 0, 0};
    rc = kbbq_fastq_scan(a, b, infer, info);
    printf("scan rc=%d n=%lld S=%lld R=%lld kind=%lld idx=%lld\n", rc, (long long)info[0], (long long)info[1],
           (long long)info[2], (long long)info[3], (long long)info[4]);
    const int64_t n = info[0];
    const int pitch = (int)((info[1] + 15) / 16 * 16 > 0 ? (info[1] + 15) / 16 * 16 : 16);
    std::vector<uint8_t> seq((size_t)n * pitch + 1), cseq((size_t)n * pitch + 1), qual((size_t)n * pitch + 1);
    std::vector<uint32_t> meta((size_t)n + 1);
    rc = kbbq_fastq_fill(a, b, infer, n, pitch, seq.data(), b ? cseq.data() : nullptr, qual.data(), meta.data());
    printf("fill rc=%d\n", rc);
    for (int64_t first : {(int64_t)0, n / 3}) {          // ranged fill + format of a shard
        const int64_t m = n - first;
        rc = kbbq_fastq_fill_range(a, b, infer, first, m, pitch, seq.data(), b ? cseq.data() : nullptr, qual.data(), meta.data());
        const int64_t need = -kbbq_fastq_format(a, first, m, pitch, qual.data(), nullptr, 0);
        std::vector<char> out((size_t)need + 1);
        const int64_t got = kbbq_fastq_format(a, first, m, pitch, qual.data(), out.data(), need);
        printf("range first=%lld rc=%d bytes=%lld/%lld\n", (long long)first, rc, (long long)got, (long long)need);
    }
    if (n >= 3 && kbbq_fastq_is_plain(a) && (!b || kbbq_fastq_is_plain(b))) {
        // a shard's byte-range readers (multi-GPU ingest): records [n / 3, 2 n / 3) through kbbq_fastq_open_range
        const int64_t lo = n / 3, hi = 2 * n / 3, m = hi - lo;
        kbbq_fastq *ra = nullptr, *rb = nullptr;
        int rrc = kbbq_fastq_open_range(pa, kbbq_fastq_record_offset(a, lo), kbbq_fastq_record_offset(a, hi), &ra);
        if (!rrc && b) rrc = kbbq_fastq_open_range(pb, kbbq_fastq_record_offset(b, lo), kbbq_fastq_record_offset(b, hi), &rb);
        std::string names;
        for (int i = 0; i < kbbq_fastq_rg_count(a); ++i) { names += kbbq_fastq_rg_name(a, i); names.push_back('\0'); }
        if (!rrc) rrc = kbbq_fastq_set_rg_names(ra, names.data(), kbbq_fastq_rg_count(a));
        if (!rrc) rrc = kbbq_fastq_fill_range(ra, rb, infer, 0, m, pitch, seq.data(), rb ? cseq.data() : nullptr, qual.data(), meta.data());
        printf("shard rc=%d records=%lld/%lld\n", rrc, (long long)(ra ? kbbq_fastq_count(ra) : -1), (long long)m);
        kbbq_fastq *bad = nullptr;                                   // a range that does not start at a line start
        printf("misaligned rc=%d\n", kbbq_fastq_open_range(pa, kbbq_fastq_record_offset(a, lo) + 1, -1, &bad));
        if (ra) kbbq_fastq_close(ra);
        if (rb) kbbq_fastq_close(rb);
    }
    if (kbbq_fastq_is_plain(a)) {
        // a rank's cut points (multi-GPU ingest without a whole-file index): every offset of a small file, a stride of a
        // large one, and offsets at / past the end; whatever the file holds, the answer is an offset inside it or -1
        const int64_t size = kbbq_fastq_sync_offset(pa, (int64_t)1 << 60);
        const int64_t stride = size > 4096 ? size / 997 + 1 : 1;
        long long found = 0, bad = 0;
        for (int64_t off = 0; off <= size + 2; off += stride)
            for (int second = 0; second < 2; ++second) {
                const int64_t at = kbbq_fastq_sync_offset_ex(pa, off, second);
                if (at >= 0) ++found;
                if (at < -1 || at > size || (at >= 0 && at < off && off <= size)) ++bad;
            }
        printf("sync size=%lld found=%lld bad=%lld negative=%lld\n", (long long)size, found, bad, (long long)kbbq_fastq_sync_offset(pa, -5));
    }
    for (int i = 0; i < kbbq_fastq_rg_count(a); ++i) printf("rg %d = %s\n", i, kbbq_fastq_rg_name(a, i));
    const char* nm; int nl;
    if (kbbq_fastq_count(a) > 0 && kbbq_fastq_name(a, kbbq_fastq_count(a) - 1, &nm, &nl) == 0) printf("last name %.*s\n", nl, nm);
    kbbq_fastq_close(a);
    if (b) kbbq_fastq_close(b);
    return 0;
}

static int run_sam(const char* path)
{
    kbbq_sam* f = nullptr;
    int rc = kbbq_sam_open(path, &f);
    if (rc) { printf("sam open rc=%d (%s)\n", rc, g_err.c_str()); return 0; }
    int64_t info[6];
    kbbq_sam_info(f, info);
    printf("sam n=%lld cigar=%lld maxlen=%lld contigs=%lld rgs=%lld header=%lld\n", (long long)info[0], (long long)info[1],
           (long long)info[2], (long long)info[3], (long long)info[4], (long long)info[5]);
    const size_t n = (size_t)info[0] + 1;
    std::vector<int32_t> flag(n), contig(n), qlen(n), span(n), rg(n), hq(n);
    std::vector<int64_t> pos(n), pnext(n), tlen(n);
    std::vector<uint32_t> clip(n), co(n), cn(n), ops((size_t)info[1] + 1);
    kbbq_sam_fields(f, flag.data(), contig.data(), pos.data(), pnext.data(), tlen.data(), qlen.data(), span.data(), clip.data(),
                    co.data(), cn.data(), rg.data(), hq.data());
    kbbq_sam_fields(f, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    kbbq_sam_cigar(f, ops.data());
    const int pitch = (int)((info[2] + 15) / 16 * 16 > 0 ? (info[2] + 15) / 16 * 16 : 16);
    std::vector<uint8_t> plane((size_t)info[0] * pitch + 1);
    for (int which = 0; which < 3; ++which) printf("fill %d rc=%d\n", which, kbbq_sam_fill(f, 0, info[0], pitch, which, plane.data()));
    if (info[0] > 2) kbbq_sam_fill(f, 1, info[0] - 2, 16, 0, plane.data());          // a narrower pitch truncates
    const char* p; int64_t len; long long total = 0;
    for (int what = 0; what < 5; ++what)
        for (int64_t i = 0; i < 3; ++i) if (kbbq_sam_text(f, what, i, &p, &len) == 0) total += len + (len ? p[len - 1] : 0);
    printf("text %lld\n", total);
    kbbq_sam_close(f);
    return 0;
}

// the sequential reader (fastq_stream.cpp): file A leads in segments of `seg` bytes, file B follows record for record; every
// segment is scanned with the state of the ones before it, filled, formatted and closed
static int run_stream(const char* pa, const char* pb, int infer, size_t seg)
{
    kbbq_fastq_stream *sa = nullptr, *sb = nullptr;
    int rc = kbbq_fastq_stream_open(pa, &sa);
    if (rc) { printf("stream open A rc=%d (%s)\n", rc, g_err.c_str()); return 0; }
    if (pb) { rc = kbbq_fastq_stream_open(pb, &sb); if (rc) { printf("stream open B rc=%d (%s)\n", rc, g_err.c_str()); kbbq_fastq_stream_close(sa); return 0; } }
    std::string rgs; int nrg = 0; int64_t longest = 0, total = 0, segs = 0, usable_all = 0; bool b_ended = pb == nullptr;
    for (;;) {
        kbbq_fastq *a = nullptr, *b = nullptr; int end = 0;
        if (sb && !b_ended) (void)kbbq_fastq_stream_prefetch(sb, seg);
        rc = kbbq_fastq_stream_next(sa, seg, 0, &a, &end);
        if (rc) { printf("stream next A rc=%d (%s)\n", rc, g_err.c_str()); break; }
        if (!a) break;
        const int64_t n = kbbq_fastq_count(a);
        if (sb && !b_ended) {
            rc = kbbq_fastq_stream_next(sb, seg, n, &b, nullptr);
            if (rc) { printf("stream next B rc=%d (%s)\n", rc, g_err.c_str()); kbbq_fastq_close(a); break; }
            if (!b || kbbq_fastq_count(b) < n) b_ended = true;
        }
        (void)kbbq_fastq_set_rg_names(a, rgs.data(), nrg);
        int64_t info[5] = {0, 0, 0, 0, 0};
        rc = kbbq_fastq_scan_next(a, (pb && b) ? b : nullptr, infer, longest, info);
        if (rc) { printf("stream scan rc=%d (%s)\n", rc, g_err.c_str()); kbbq_fastq_close(a); if (b) kbbq_fastq_close(b); break; }
        rgs.clear(); nrg = kbbq_fastq_rg_count(a);
        for (int i = 0; i < nrg; ++i) { rgs += kbbq_fastq_rg_name(a, i); rgs.push_back('\0'); }
        longest = std::max<int64_t>(longest, info[1]);
        const int64_t m = info[0];
        const int pitch = (int)std::max<int64_t>(16, (info[1] + 15) / 16 * 16);
        std::vector<uint8_t> seq((size_t)m * pitch + 1), cseq((size_t)m * pitch + 1), qual((size_t)m * pitch + 1);
        std::vector<uint32_t> meta((size_t)m + 1);
        rc = kbbq_fastq_fill_range(a, (pb && b) ? b : nullptr, infer, 0, m, pitch, seq.data(), (pb && b) ? cseq.data() : nullptr, qual.data(), meta.data());
        const int64_t need = -kbbq_fastq_format(a, 0, m, pitch, qual.data(), nullptr, 0);
        std::vector<char> out((size_t)need + 1);
        const int64_t got = kbbq_fastq_format(a, 0, m, pitch, qual.data(), out.data(), need);
        if (rc || got != need) printf("stream fill rc=%d bytes=%lld/%lld\n", rc, (long long)got, (long long)need);
        total += n; usable_all += m; ++segs;
        kbbq_fastq_close(a); if (b) kbbq_fastq_close(b);
        if (info[3]) { printf("stream first offender kind=%lld at %lld\n", (long long)info[3], (long long)(total - n + info[4])); break; }
        if (end) break;
    }
    printf("stream records=%lld usable=%lld segments=%lld S=%lld R=%d\n", (long long)total, (long long)usable_all, (long long)segs, (long long)longest, nrg);
    kbbq_fastq_stream_close(sa); if (sb) kbbq_fastq_stream_close(sb);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 2) return 2;
    if (!strcmp(argv[1], "stream")) return run_stream(argv[2], argc > 3 && strcmp(argv[3], "-") ? argv[3] : nullptr, argc > 4 ? atoi(argv[4]) : 0,
                                                      argc > 5 ? (size_t)atoll(argv[5]) : (size_t)1 << 16);
    if (!strcmp(argv[1], "sam")) return run_sam(argv[2]);
    if (!strcmp(argv[1], "pair")) return run_pair(argv[2], argc > 3 && strcmp(argv[3], "-") ? argv[3] : nullptr, argc > 4 ? atoi(argv[4]) : 0);
    if (!strcmp(argv[1], "combiln")) {
        const int64_t n = 50000;
        std::vector<int64_t> e((size_t)n), t((size_t)n);
        for (int64_t i = 0; i < n; ++i) { e[(size_t)i] = (i * 7919) % 1000003 - 5; t[(size_t)i] = e[(size_t)i] + (i * 104729) % 9999991 - 3; }
        std::vector<double> out((size_t)n);
        for (int th : {1, 3, 8, 3}) { int rc = kbbq_combiln_host(e.data(), t.data(), n, out.data(), th); printf("combiln threads=%d rc=%d %.6f\n", th, rc, out[123]); }
        std::vector<double> x = {0.5, 1, 2, 12.5, 13, 999, 1e8, 1e300, -1, 0};
        std::vector<double> g(x.size());
        printf("gammaln rc=%d %.6f\n", kbbq_gammaln_host(x.data(), (int64_t)x.size(), g.data()), g[3]);
        // the restated log / gammaln over the constants read out of the mapped libm: short buffers are refused, and on
        // the counts a solve produces the restatement equals the libm-backed routine bit for bit
        std::vector<double> logtab(263), small(10);
        printf("logtab short rc=%d\n", kbbq_libm_log_data(small.data(), 10));
        const int lrc = kbbq_libm_log_data(logtab.data(), 263);
        std::vector<double> cx, want, got;
        for (double v = 1; v < 3e6; v = v < 5000 ? v + 1 : v * 1.37 + 1) cx.push_back((double)(int64_t)v);
        want.resize(cx.size()); got.resize(cx.size());
        kbbq_gammaln_host(cx.data(), (int64_t)cx.size(), want.data());
        long long differ = -1;
        if (!lrc) {
            kbbq_gammaln_restated_host(cx.data(), (int64_t)cx.size(), logtab.data(), got.data());
            differ = 0;
            for (size_t i = 0; i < cx.size(); ++i) differ += memcmp(&want[i], &got[i], sizeof(double)) != 0;
        }
        printf("logtab rc=%d restated differ=%lld of %zu\n", lrc, differ, cx.size());
        // the fused solve prep over a table buffer of 3 read groups x 43 x 150 cycles (+ 16 contexts), on the pool
        const int R = 3, S2 = 150; const int64_t npos = (int64_t)R * 43 * S2, ndn = (int64_t)R * 43 * 16;
        std::vector<int64_t> tab((size_t)(2 * npos + 2 * ndn));
        for (size_t i = 0; i < tab.size(); ++i) tab[i] = (int64_t)((i * 2654435761u) % 100000);
        std::vector<double> aux((size_t)(R + R * 43 + npos + ndn)); std::vector<int64_t> marg((size_t)(2 * R * 43 + 2 * R));
        for (int th : {1, 6, 16, 2}) { int rc = kbbq_solve_prep_host(tab.data(), R, S2, aux.data(), marg.data(), th); printf("prep threads=%d rc=%d %lld\n", th, rc, (long long)marg[5]); }
        return 0;
    }
    return 2;
}
