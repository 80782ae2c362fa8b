// sanitizer driver for the host-only sources (no HIP): FASTA reader, parameter setup, encoders, formatters
#include "plaac_host.h"
#include "plaac_native.h"
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
int main(int argc, char **argv) {
    plaac_params P;
    if (plaac_params_init(&P, nullptr, nullptr, 1.0, 60, 41, 41, 41, 1) != PLAAC_OK) return 1;
    char text[16384];
    plaac_format_param_block(&P, text, sizeof text);
    std::vector<char> dot(32768);
    if (plaac_format_hmm_dot(&P, dot.data(), dot.size()) < 0) return 2;
    for (int a = 1; a < argc; ++a) {
        plaac_fasta *f = nullptr;
        plaac_status st = plaac_fasta_read(argv[a], &f);
        std::printf("%s: status %d", argv[a], (int)st);
        if (st != PLAAC_OK) { std::printf("\n"); continue; }
        std::printf(" nrec %u nres %llu\n", f->nrec, (unsigned long long)f->nres);
        std::vector<char> line(1 << 20);
        plaac_row row;
        std::memset(&row, 0, sizeof row);
        row.core_start = -1; row.core_end = -2; row.prd_start = -1; row.prd_end = -2; row.papa_cen = -1;
        row.llr_start = -1; row.llr_end = -2;
        for (uint32_t i = 0; i < f->nrec; ++i) {
            const uint64_t len = f->offsets[i + 1] - f->offsets[i];
            row.prot_len = (int)len;
            if (len) plaac_format_summary_row(&row, f->names + f->name_off[i], f->codes + f->offsets[i], len, 60, 41, line.data(), line.size());
        }
        // the same file as a stream of batches of several sizes: record and residue totals must agree
        const uint32_t nrec_whole = f->nrec;
        const unsigned long long nres_whole = f->nres;
        plaac_fasta_free(f);
        const uint64_t sizes[][2] = {{1, 1ull << 30}, {7, 1ull << 30}, {1u << 20, 1}, {1u << 20, 4096}, {1000, 100000}};
        for (const auto &sz : sizes) {
            plaac_fasta_stream *fs = nullptr;
            if (plaac_fasta_open(argv[a], &fs) != PLAAC_OK) return 3;
            unsigned long long nrec = 0, nres = 0;
            for (;;) {
                plaac_fasta *b = nullptr;
                if (plaac_fasta_next(fs, (uint32_t)sz[0], sz[1], &b) != PLAAC_OK) return 4;
                if (!b) break;
                if (b->nrec == 0 || b->nrec > sz[0]) return 5;
                nrec += b->nrec;
                nres += b->nres;
                plaac_fasta_free(b);
            }
            plaac_fasta_close(fs);
            if (nrec != nrec_whole || nres != nres_whole) {
                std::printf("stream totals differ for %s\n", argv[a]);
                return 6;
            }
        }
    }
    char b[64];
    const double vals[] = {0.0, -0.0, 1.0005, 2.5, 1e-7, 123456.7895, -1e300, 1.0 / 0.0, -1.0 / 0.0, 0.0 / 0.0};
    for (double v : vals) { plaac_format_fixed(v, 3, b, sizeof b); plaac_format_double_tostring(v, b, sizeof b); }
    return 0;
}
