// tools/format_bench.cpp - how fast does the host format summary rows? (one thread; rows with plausible random fields)
//   g++ -O2 -std=c++17 -Iinclude tools/format_bench.cpp -o build/format_bench -Lplaac_amd -lplaac_native -Wl,-rpath,$PWD/plaac_amd -Wl,-rpath,/opt/rocm/lib
#include <chrono>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
#include "plaac_host.h"
#include "plaac_native.h"
int main(int argc, char **argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 1000000;
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> U(-60.0, 90.0);
    std::vector<plaac_row> rows(n);
    std::vector<uint8_t> codes(700);
    for (auto &c : codes) c = 1 + rng() % 20;
    for (auto &r : rows) {
        std::memset(&r, 0, sizeof r);
        r.prot_len = 300 + rng() % 300;
        r.mw_score = rng() % 40, r.mw_start = rng() % 100, r.mw_end = r.mw_start + 79;
        r.llr_score = U(rng), r.llr_start = rng() % 100, r.llr_end = r.llr_start + 50;
        r.vit_maxrun = rng() % 50;
        r.core_score = (rng() % 4) ? (double)NAN : U(rng), r.core_start = -1, r.core_end = -2;
        r.prd_score = 0.0, r.prd_start = -1, r.prd_end = -2;
        r.hmm_all = U(rng), r.hmm_vit = U(rng);
        r.fi_numaa = rng() % 300, r.fi_meanhydro = U(rng) / 100, r.fi_meancharge = U(rng) / 100, r.fi_meancombo = U(rng) / 100, r.fi_maxrun = rng() % 100;
        r.papa_combo = U(rng) / 100, r.papa_prop = U(rng) / 100, r.papa_fi = U(rng) / 100, r.papa_llr = U(rng) / 10, r.papa_llr2 = U(rng) / 10;
        r.papa_cen = 50 + rng() % 200;
    }
    std::vector<char> buf(1 << 16);
    // variant: every row into fresh memory (a 300 MB buffer, touched for the first time), codes from a cold 700 MB array
    {
        const size_t cap = (size_t)n * 300 + (1 << 20);
        char *big = (char *)std::aligned_alloc(2u << 20, (cap + (2u << 20) - 1) / (2u << 20) * (2u << 20));
        std::vector<uint8_t> bigcodes((size_t)n * 700);
        for (size_t i = 0; i < bigcodes.size(); i += 64) bigcodes[i] = 1 + i % 20;
        size_t at = 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; ++i) {
            const long k = plaac_format_summary_row_n(&rows[i], "sp|P12345|NAME_HUMAN", 20, bigcodes.data() + (size_t)i * 700, 650, 60, 41, big + at, cap - at);
            at += (size_t)k;
            big[at++] = '\n';
        }
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("fresh output + cold codes: %.1f ns per row\n", s * 1e9 / n);
        std::free(big);
    }
    for (int rep = 0; rep < 3; ++rep) {
        size_t bytes = 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < n; ++i) {
            const long k = plaac_format_summary_row(&rows[i], "sp|P12345|NAME_HUMAN", codes.data(), 650, 60, 41, buf.data(), buf.size());
            bytes += (size_t)k;
        }
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("%d rows, %.1f bytes per row: %.1f ns per row (%.2f M rows/s/thread)\n", n, (double)bytes / n, s * 1e9 / n, n / s * 1e-6);
    }
    return 0;
}
