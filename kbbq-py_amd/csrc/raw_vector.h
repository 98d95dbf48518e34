// raw_vector.h -- std::vector whose resize() leaves new elements uninitialised (internal to libkbbq_hip's host C++): the index arrays of
// a 50 M-read file and the inflated text of a compressed one are GBs that the threads filling them overwrite anyway (and first-touch
// in parallel instead of in one zero-filling thread).
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

template <typename T> struct raw_alloc : std::allocator<T> {
    template <typename U> struct rebind { using other = raw_alloc<U>; };
    template <typename U, typename... A> void construct(U* p, A&&... a)
    {
        if constexpr (sizeof...(A) == 0) ::new ((void*)p) U; else ::new ((void*)p) U(std::forward<A>(a)...);
    }
};
template <typename T> using raw_vector = std::vector<T, raw_alloc<T>>;
using kbbq_bytes = raw_vector<uint8_t>;
