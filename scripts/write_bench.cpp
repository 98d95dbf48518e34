// write-bandwidth microbench: write(2) of slabs vs ftruncate+mmap+parallel memcpy (+ optional fallocate)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <linux/falloc.h>
#include <sys/mman.h>
#include <unistd.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv)
{
    const char* path = argv[1]; const int mode = atoi(argv[2]); const int nt = atoi(argv[3]);
    const size_t total = (size_t)atoll(argv[4]) << 20, slab = (size_t)80 << 20;
    std::vector<char> src(slab); for (size_t i = 0; i < slab; ++i) src[i] = (char)(i * 7);
    unlink(path);
    int fd = open(path, O_CREAT | O_WRONLY | O_TRUNC, 0644);
    if (mode == 1 || mode == 2) { close(fd); fd = open(path, O_RDWR); }
    const double t0 = now();
    size_t off = 0;
    while (off < total) {
        const size_t n = std::min(slab, total - off);
        if (mode == 3 && posix_fallocate(fd, off, n)) { perror("fallocate"); return 1; }        // the slab's blocks reserved, then plain write(2)
        if (mode == 4 && fallocate(fd, FALLOC_FL_KEEP_SIZE, (off_t)off, (off_t)n)) { perror("fallocate keep-size"); return 1; }   // ... without moving the file's end
        if (mode == 0 || mode == 3 || mode == 4) { size_t w = 0; while (w < n) { ssize_t k = write(fd, src.data() + w, n - w); if (k <= 0) { perror("write"); return 1; } w += k; } }
        else {
            if (mode == 2) { if (posix_fallocate(fd, off, n)) { perror("fallocate"); return 1; } }
            else if (ftruncate(fd, off + n)) { perror("ftruncate"); return 1; }
            const size_t a = off & ~(size_t)4095;
            char* m = (char*)mmap(nullptr, off + n - a, PROT_READ | PROT_WRITE, MAP_SHARED, fd, a);
            if (m == MAP_FAILED) { perror("mmap"); return 1; }
            char* dst = m + (off - a);
            std::vector<std::thread> th;
            const size_t per = (n + nt - 1) / nt;
            for (int t = 0; t < nt; ++t) th.emplace_back([&, t]() { size_t lo = std::min(n, t * per), hi = std::min(n, lo + per); memcpy(dst + lo, src.data() + lo, hi - lo); });
            for (auto& t : th) t.join();
            munmap(m, off + n - a);
        }
        off += n;
    }
    const double t1 = now();
    close(fd);
    const double t2 = now();
    printf("mode %d threads %d: %.3f s (+close %.3f) = %.2f GB/s\n", mode, nt, t1 - t0, t2 - t1, total / (t1 - t0) / 1e9);
    unlink(path);
    return 0;
}
