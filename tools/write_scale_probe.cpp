// tools/write_scale_probe.cpp - does the page cache take one file's bytes faster from several threads (pwrite at disjoint offsets)?
//   g++ -O2 -std=c++17 -pthread tools/write_scale_probe.cpp -o /tmp/write_scale_probe && /tmp/write_scale_probe /tmp/x.bin
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <thread>
#include <unistd.h>
#include <vector>
int main(int argc, char **argv) {
    const char *path = argc > 1 ? argv[1] : "/tmp/write_scale_probe.bin";
    const size_t total = 2ull << 30, chunk = 8u << 20;
    std::vector<char> buf(chunk, 'x');
    for (int nt : {1, 2, 4, 8}) {
        ::unlink(path);
        const int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t)
            th.emplace_back([&, t] {
                for (size_t o = (size_t)t * chunk; o < total; o += (size_t)nt * chunk)
                    if (::pwrite(fd, buf.data(), chunk, (off_t)o) != (ssize_t)chunk) std::abort();
            });
        for (auto &x : th) x.join();
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("%d writer thread(s): 2 GiB into the page cache in %.3f s = %.1f GB/s\n", nt, s, total / s * 1e-9);
        ::close(fd);
    }
    ::unlink(path);
    return 0;
}
