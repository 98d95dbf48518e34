// host_threads.h -- how many threads the host-side readers / writers start (internal to libkbbq_hip's host C++).
#pragma once
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include <sched.h>

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
