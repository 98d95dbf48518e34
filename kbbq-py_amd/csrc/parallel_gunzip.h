// parallel_gunzip.h -- a gzip file inflated on many threads (internal to libkbbq_hip's host C++; see parallel_gunzip.cpp).
#pragma once
#include <cstddef>
#include <cstdint>

#include "raw_vector.h"

// A decoder over the gzip file image src[0, n) (mapped or in memory; it must stay there until kbbq_pgz_close) -- one member or
// several, not BGZF (whose blocks say their sizes: bam_host.cpp and fastq_stream.cpp inflate those side by side already).
struct kbbq_pgz;
kbbq_pgz* kbbq_pgz_open(const uint8_t* src, size_t n, unsigned threads);     // threads: at most that many (0: the host-thread ceiling)
void kbbq_pgz_close(kbbq_pgz* z);

// The next stretch of text, APPENDED to `out`: one window of chunks (tens of MB), so memory stays bounded however large the file.
//   1  text was appended, more may follow
//   0  the input has ended: every member inflated, every CRC-32 / size trailer checked (nothing appended by this call)
//  -1  something this decoder does not take on or does not trust (a chunk boundary that did not meet, a trailer that does not match,
//      damaged data): nothing was appended by this call.  kbbq_pgz_delivered() says how much text the earlier calls appended in
//      all -- the caller inflates the input from its start with zlib, which also decides what a damaged input's error is, and
//      skips that much.
int kbbq_pgz_next(kbbq_pgz* z, kbbq_bytes& out);
// The same in two steps, for a caller that has memory of its own for the text: prepare decodes the next window and says how many
// bytes it holds (1 / 0 / -1 as above; a prepared window stays prepared until it is emitted), emit writes them to dst[0, total).
int kbbq_pgz_prepare(kbbq_pgz* z, size_t* total);
int kbbq_pgz_emit(kbbq_pgz* z, uint8_t* dst);
size_t kbbq_pgz_delivered(const kbbq_pgz* z);

// The smallest input (compressed bytes) worth handing to this decoder: 8 MB, or KBBQ_PGZ_MIN_BYTES.
size_t kbbq_pgz_min_bytes();

// All of it at once: true = `out` holds the whole text; false = `out` is as it was (the caller's zlib path takes the input).
bool kbbq_parallel_gunzip(const uint8_t* src, size_t n, kbbq_bytes& out, unsigned threads);
