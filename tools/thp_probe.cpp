// tools/thp_probe.cpp - does this host give transparent huge pages to big anonymous allocations, and what does first touch cost?
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t n = 3ull << 30;
    for (int mode = 0; mode < 3; ++mode) {
        double t0 = now();
        char *p = (char *)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (mode == 1) madvise(p, n, MADV_HUGEPAGE);
        if (mode == 2) madvise(p, n, MADV_NOHUGEPAGE);
        for (size_t i = 0; i < n; i += 4096) p[i] = 1;
        double t1 = now();
        munmap(p, n);
        double t2 = now();
        std::printf("%s: first touch of 3 GiB %.3f s, munmap %.3f s\n", mode == 0 ? "default      " : mode == 1 ? "MADV_HUGEPAGE" : "MADV_NOHUGE  ", t1 - t0, t2 - t1);
    }
    return 0;
}
