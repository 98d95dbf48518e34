// bam_host.h -- gzip / BGZF inflation and BAM -> SAM text, for the SAM reader (internal to libkbbq_hip's host C++).
//
// The reference reads its alignments through pysam / htslib (benchmark.py:57-74, gatk/bqsr.py:52-123), i.e. from
// BAM files.  htslib is not part of this build; zlib is.  A BAM file is a series of BGZF blocks (independent gzip
// members carrying their compressed size in a 'BC' extra field: inflated in parallel here) holding a binary header
// and binary alignment records (SAM specification, section 4); the records are rendered as SAM text lines -- what
// `samtools view -h` prints -- and parsed by the SAM reader like any other text.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "raw_vector.h"

// Inflate a whole gzip / BGZF file image (any number of members).  false + err on corrupt input.
bool kbbq_inflate_all(const uint8_t* src, size_t n, kbbq_bytes& out, std::string& err);

// Uncompressed BAM image -> SAM text (header lines, then one line per record).  false + err on malformed input.
bool kbbq_bam_to_sam(const uint8_t* bam, size_t n, kbbq_bytes& text, std::string& err);
