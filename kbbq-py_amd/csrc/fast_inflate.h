// fast_inflate.h -- whole-buffer DEFLATE through libdeflate when the machine has it (internal to libkbbq_hip's host C++).
//
// zlib's inflate runs at 0.15-0.4 GB/s of output per thread; libdeflate -- the library htslib uses for BGZF when it is built in -- at
// 2-3x that, and its crc32 at many times zlib's.  It has no streaming interface: it suits what is inflated as a whole -- BGZF blocks
// (bgzip'd FASTQ, BAM) and gzip files the mapped reader inflates in one go -- not the member-by-member streaming of fastq_stream.cpp.
// The library is looked up at run time (dlopen "libdeflate.so.0": no header, no link-time dependency); without it, with
// KBBQ_LIBDEFLATE=0, or on ANY result other than success the callers take their zlib path, which also decides what a damaged
// input's error is.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>

#include <dlfcn.h>

struct kbbq_libdeflate {
    void* (*alloc_decompressor)();
    void (*free_decompressor)(void*);
    // 0 success, 1 bad data, 2 short output, 3 insufficient space
    int (*deflate_decompress_ex)(void*, const void* in, size_t in_n, void* out, size_t out_avail, size_t* in_used, size_t* out_used);
    int (*gzip_decompress_ex)(void*, const void* in, size_t in_n, void* out, size_t out_avail, size_t* in_used, size_t* out_used);
    uint32_t (*crc32)(uint32_t, const void*, size_t);
};

// the library's entry points, or nullptr
inline const kbbq_libdeflate* kbbq_libdeflate_get()
{
    static const kbbq_libdeflate* found = []() -> const kbbq_libdeflate* {
        const char* e = getenv("KBBQ_LIBDEFLATE");
        if (e && e[0] == '0') return nullptr;
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return nullptr;
        static kbbq_libdeflate l;
        l.alloc_decompressor = (void* (*)())dlsym(h, "libdeflate_alloc_decompressor");
        l.free_decompressor = (void (*)(void*))dlsym(h, "libdeflate_free_decompressor");
        l.deflate_decompress_ex = (int (*)(void*, const void*, size_t, void*, size_t, size_t*, size_t*))dlsym(h, "libdeflate_deflate_decompress_ex");
        l.gzip_decompress_ex = (int (*)(void*, const void*, size_t, void*, size_t, size_t*, size_t*))dlsym(h, "libdeflate_gzip_decompress_ex");
        l.crc32 = (uint32_t (*)(uint32_t, const void*, size_t))dlsym(h, "libdeflate_crc32");
        if (!l.alloc_decompressor || !l.free_decompressor || !l.deflate_decompress_ex || !l.gzip_decompress_ex || !l.crc32) return nullptr;
        return &l;
    }();
    return found;
}

// One thread's decompressor for raw DEFLATE blocks of known inflated size (BGZF): block() is true when the block inflated to exactly
// `isize` bytes with the CRC its trailer names; false says nothing about the data -- the caller's zlib path looks at it again.
class kbbq_block_inflater {
public:
    kbbq_block_inflater() : l_(kbbq_libdeflate_get()), d_(l_ ? l_->alloc_decompressor() : nullptr) {}
    ~kbbq_block_inflater() { if (d_) l_->free_decompressor(d_); }
    kbbq_block_inflater(const kbbq_block_inflater&) = delete;
    kbbq_block_inflater& operator=(const kbbq_block_inflater&) = delete;
    bool usable() const { return d_ != nullptr; }
    bool block(const uint8_t* src, size_t csize, uint8_t* dst, size_t isize, uint32_t crc) const
    {
        if (!d_ || isize == 0) return false;
        size_t in_used = 0, out_used = 0;
        if (l_->deflate_decompress_ex(d_, src, csize, dst, isize, &in_used, &out_used) != 0 || out_used != isize) return false;
        return l_->crc32(0, dst, isize) == crc;
    }

private:
    const kbbq_libdeflate* l_;
    void* d_;
};
