// fastq_host.h -- the FASTQ reader's record index, shared by fastq_host.cpp (whole files and byte ranges, mapped) and
// fastq_stream.cpp (segments of a file read sequentially: FIFOs, inputs larger than anybody wants to index at once).
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include <sys/mman.h>

#include "raw_vector.h"

// Segment buffers of the sequential reader are kept for the next segment (fastq_stream.cpp): memory that has been touched once costs
// nothing to fill again, fresh pages cost more than reading the file does.
void kbbq_text_pool_give(raw_vector<uint8_t>&& v);
raw_vector<uint8_t> kbbq_text_pool_take(size_t capacity);

struct kbbq_fastq {
    const uint8_t* buf = nullptr; size_t size = 0; bool mapped = false;
    size_t range_end = 0;                  // end of the indexed byte range (the file size for a whole-file reader)
    raw_vector<uint64_t> h0, s0, q0;       // start offsets of header / sequence / quality lines
    raw_vector<uint32_t> hlen, slen;       // header line length (without '@', up to whitespace = name), sequence length
    std::vector<std::string> rg_names;     // first-appearance order (filled by scan / fill with infer_rg)
    kbbq_bytes owned;                      // the inflated text of a compressed file
    raw_vector<uint8_t> text;              // a segment of a sequentially read file (fastq_stream.cpp): uninitialised until read into
    ~kbbq_fastq() { if (mapped && buf) munmap((void*)buf, size); if (text.capacity()) kbbq_text_pool_give(std::move(text)); }
};


// Index the 4-line records of f->buf[r0, r1) (r0 at a line start; r1 at a line end or the end of the buffer): fills h0 / s0 /
// q0 / hlen / slen and range_end.  Returns 0, or 1 header without '@', 2 sequence and quality lengths differ, 3 read longer
// than 65535 bases, 4 line count not a multiple of 4 (fastq_host.cpp).
int kbbq_fastq_index_range_(kbbq_fastq* f, size_t r0, size_t r1);
