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
        plaac_fasta_free(f);
    }
    char b[64];
    const double vals[] = {0.0, -0.0, 1.0005, 2.5, 1e-7, 123456.7895, -1e300, 1.0 / 0.0, -1.0 / 0.0, 0.0 / 0.0};
    for (double v : vals) { plaac_format_fixed(v, 3, b, sizeof b); plaac_format_double_tostring(v, b, sizeof b); }
    return 0;
}
