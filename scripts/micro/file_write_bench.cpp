// How fast can 2.5 GB get into a regular file?  write(2) from one thread, pwrite(2) from 16, memcpy into a shared
// mapping from 16 (after posix_fallocate).  Build: g++ -O2 -pthread -o file_write_bench file_write_bench.cpp
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    const char* path = argc > 1 ? argv[1] : "/tmp/fwb.bin";
    const size_t n = (size_t)2544 << 20; const int T = 16;
    char* src = (char*)malloc(n); memset(src, 'x', n);
    for (int rep = 0; rep < 2; ++rep) {
        { int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644); double t = now();
          for (size_t at = 0; at < n;) { ssize_t k = write(fd, src + at, std::min<size_t>(n - at, 80u << 20)); if (k <= 0) return 1; at += (size_t)k; }
          printf("write, 1 thread, 80 MB pieces      %.3f s  %.1f GB/s\n", now() - t, n / (now() - t) / 1e9); close(fd); }
        { int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644); double t = now();
          std::vector<std::thread> th; const size_t per = n / T;
          for (int i = 0; i < T; ++i) th.emplace_back([=]() { for (size_t at = i * per; at < (i + 1) * per;) { ssize_t k = pwrite(fd, src + at, std::min<size_t>((i + 1) * per - at, 16u << 20), (off_t)at); if (k <= 0) return; at += (size_t)k; } });
          for (auto& x : th) x.join();
          printf("pwrite, %d threads                  %.3f s  %.1f GB/s\n", T, now() - t, n / (now() - t) / 1e9); close(fd); }
        { int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644); double t = now();
          if (posix_fallocate(fd, 0, (off_t)n) != 0) { printf("fallocate failed\n"); return 1; }
          const double tf = now() - t;
          char* m = (char*)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); if (m == MAP_FAILED) return 1;
          std::vector<std::thread> th; const size_t per = n / T;
          for (int i = 0; i < T; ++i) th.emplace_back([=]() { memcpy(m + i * per, src + i * per, per); });
          for (auto& x : th) x.join();
          const double tc = now() - t; munmap(m, n);
          printf("fallocate %.3f s + mmap memcpy, %d thr %.3f s (unmapped %.3f)  %.1f GB/s\n", tf, T, tc - tf, now() - t, n / (now() - t) / 1e9); close(fd); }
    }
    unlink(path);
    return 0;
}
