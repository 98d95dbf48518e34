// host_threads.h -- how many threads the host-side readers / writers start (internal to libkbbq_hip's host C++).
#pragma once
#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include <sched.h>
#include <unistd.h>

// CPUs this process may actually use: the smaller of the online count, its affinity mask and its cgroup CPU quota
// (a container with 16 CPUs' worth of quota on a 256-thread host is throttled, not sped up, by 256 runnable threads).
// Cached at the first call -- kbbq_bind_host_to_device() makes that call BEFORE it narrows the affinity mask, so the
// figure stays what the whole job may use.
inline unsigned kbbq_usable_cpus()
{
    static const unsigned cached = []() {
        unsigned hw = std::thread::hardware_concurrency();
        if (hw == 0) hw = 4;
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof set, &set) == 0) { const int k = CPU_COUNT(&set); if (k > 0) hw = std::min<unsigned>(hw, (unsigned)k); }
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {                 // cgroup v2: "<quota|max> <period>"
            char quota[32]; long long period = 0;
            if (fscanf(f, "%31s %lld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
                const long long q = atoll(quota);
                if (q > 0) hw = std::min<unsigned>(hw, (unsigned)std::max<long long>(1, (q + period - 1) / period));
            }
            fclose(f);
        }
        return hw;
    }();
    return cached;
}

// Processes of this job that share the host: one per GPU under torch.distributed.run, which exports LOCAL_WORLD_SIZE
// to every rank (KBBQ_LOCAL_RANKS overrides it for other launchers).  1 outside a launcher.
inline unsigned kbbq_local_ranks()
{
    for (const char* name : {"KBBQ_LOCAL_RANKS", "LOCAL_WORLD_SIZE"}) {
        const char* e = getenv(name);
        if (e && atoi(e) > 0) return (unsigned)std::min(atoi(e), 4096);
    }
    return 1;
}

// The ceiling of host threads of THIS process: its share of the usable CPUs -- eight ranks of one node that each started
// "usable CPUs" threads ran scan / fill / format, 99 % of the file path's wall time, 8x oversubscribed.
// KBBQ_HOST_THREADS overrides the ceiling.
inline unsigned kbbq_host_thread_ceiling()
{
    const char* e = getenv("KBBQ_HOST_THREADS");
    if (e && atoi(e) > 0) return (unsigned)atoi(e);
    return std::max(1u, kbbq_usable_cpus() / kbbq_local_ranks());
}

// threads for `work` bytes: one per MiB up to the ceiling
inline unsigned kbbq_threads_for(size_t work)
{
    const unsigned hw = kbbq_host_thread_ceiling();
    return (unsigned)std::max<size_t>(1, std::min<size_t>(hw, work / (1 << 20) + 1));
}

// Parked worker threads for the host stages.  Starting and joining 16 threads costs 0.2-0.3 ms; the file path runs a parallel
// region per 128 K-read slab and stage (fill, sidecars, format, ...), a hundred and more per 8 M reads, and the streaming path
// one set per device slab.  kbbq_parallel(nt, fn) runs fn(0) ... fn(nt - 1) on the parked workers (the caller takes part) when no
// other region is using them; a second region at the same time -- the output pipeline's format stage beside the packer's
// fill, file B being opened beside file A -- starts threads of its own as every region used to, so nobody waits for anybody.
// KBBQ_THREAD_POOL=0: always fresh threads.
class kbbq_pool {
public:
    static kbbq_pool& get()
    {
        static kbbq_pool* p = new kbbq_pool();           // never destroyed: its workers are parked in it when the process exits
        if (p->pid_ != getpid()) p = new kbbq_pool();    // a forked child inherits the object but not the threads
        return *p;
    }

    void run(unsigned nt, const std::function<void(unsigned)>& fn)
    {
        if (nt <= 1) { fn(0); return; }
        std::unique_lock<std::mutex> mine(region_, std::try_to_lock);
        if (!mine.owns_lock() || !enabled_) { fresh(nt, fn); return; }
        std::unique_lock<std::mutex> lk(m_);
        while (workers_ + 1 < nt) { std::thread([this]() { worker(); }).detach(); ++workers_; }
        fn_ = &fn; ntasks_ = nt; next_ = 0; busy_ = workers_; ++gen_;
        work_.notify_all();
        take(lk);
        done_.wait(lk, [&]() { return busy_ == 0; });
        fn_ = nullptr;
    }

private:
    kbbq_pool() : pid_(getpid()) { const char* e = getenv("KBBQ_THREAD_POOL"); enabled_ = !(e && e[0] == '0'); }

    static void fresh(unsigned nt, const std::function<void(unsigned)>& fn)
    {
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; ++t) th.emplace_back([&fn, t]() { fn(t); });
        fn(0);
        for (auto& t : th) t.join();
    }

    void take(std::unique_lock<std::mutex>& lk)          // run tasks of the current region until none is left (m_ held on entry and exit)
    {
        while (next_ < ntasks_) {
            const unsigned t = next_++;
            const std::function<void(unsigned)>* fn = fn_;
            lk.unlock();
            (*fn)(t);
            lk.lock();
        }
    }

    void worker()
    {
        uint64_t seen = 0;
        std::unique_lock<std::mutex> lk(m_);
        if (gen_ && fn_ == nullptr) seen = gen_;          // started between regions: the next one is the first it serves
        for (;;) {
            work_.wait(lk, [&]() { return gen_ != seen; });
            seen = gen_;
            take(lk);
            if (--busy_ == 0) done_.notify_one();
        }
    }

    std::mutex region_, m_;
    std::condition_variable work_, done_;
    const std::function<void(unsigned)>* fn_ = nullptr;
    unsigned workers_ = 0, ntasks_ = 0, next_ = 0, busy_ = 0;
    uint64_t gen_ = 0;
    pid_t pid_;
    bool enabled_ = true;
};

inline void kbbq_parallel(unsigned nt, const std::function<void(unsigned)>& fn) { kbbq_pool::get().run(nt, fn); }

// f(t, lo, hi) over the t-th of nt equal parts of [0, n) (empty parts are skipped)
template <typename F> inline void kbbq_parallel_parts(size_t n, unsigned nt, F f)
{
    if (nt <= 1) { if (n) f(0u, (size_t)0, n); return; }
    const size_t per = (n + nt - 1) / nt;
    kbbq_parallel(nt, [&](unsigned t) {
        const size_t lo = std::min(n, (size_t)t * per), hi = std::min(n, lo + per);
        if (lo < hi) f(t, lo, hi);
    });
}
