// raw_vector.h -- std::vector whose resize() leaves new elements uninitialised (internal to libkbbq_hip's host C++): the index arrays of
// a 50 M-read file and the inflated text of a compressed one are GBs that the threads filling them overwrite anyway (and first-touch
// in parallel instead of in one zero-filling thread).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <vector>

#include <sys/mman.h>

template <typename T> struct raw_alloc : std::allocator<T> {
    template <typename U> struct rebind { using other = raw_alloc<U>; };
    template <typename U, typename... A> void construct(U* p, A&&... a)
    {
        if constexpr (sizeof...(A) == 0) ::new ((void*)p) U; else ::new ((void*)p) U(std::forward<A>(a)...);
    }
};
template <typename T> using raw_vector = std::vector<T, raw_alloc<T>>;
using kbbq_bytes = raw_vector<uint8_t>;

// Ask for huge pages behind a large buffer that is about to be filled for the first time (transparent huge pages in "madvise" mode):
// 512 times fewer page faults for the threads that fill it.  KBBQ_HUGE_PAGES=0 turns it off.
inline void kbbq_advise_huge(void* p, size_t bytes)
{
#ifdef MADV_HUGEPAGE
    static const bool on = []() { const char* e = getenv("KBBQ_HUGE_PAGES"); return !(e && e[0] == '0'); }();
    const uintptr_t two_mb = (uintptr_t)2 << 20;
    const uintptr_t lo = ((uintptr_t)p + two_mb - 1) & ~(two_mb - 1), hi = ((uintptr_t)p + bytes) & ~(two_mb - 1);
    if (on && hi > lo) (void)madvise((void*)lo, hi - lo, MADV_HUGEPAGE);
#else
    (void)p; (void)bytes;
#endif
}


// resize() of a vector that is about to be filled for the first time, huge pages asked for when it is large
template <typename V> inline void kbbq_resize_fresh(V& v, size_t n)
{
    if (v.capacity() < n && n * sizeof(typename V::value_type) >= ((size_t)8 << 20)) {
        v.reserve(n);
        kbbq_advise_huge(v.data(), v.capacity() * sizeof(typename V::value_type));
    }
    v.resize(n);
}
