// plaac — command-line host of the MI355X-native PLAAC engine: the flag surface and the two output
// tables of `java -jar plaac.jar` (reference: cli/src/plaac.java, main :302-530, scoreallfastas
// :653-950, plotsomefastas :587-649), on top of the C ABI in include/plaac_native.h.
//
//   plaac -i input.fa [-c core] [-a alpha] [-B bg_freqs.txt | -b background.fa] [-F fg_freqs.txt]
//         [-w ww1] [-W ww2] [-p list.txt|all] [-d] [-s] [-m n] [-h file]
//
// Deliberate divergences from the reference: -F reads the -F file (the reference reads
// the -B file by mistake), histogram bins are 64-bit, records with an empty sequence are skipped with
// a note on stderr instead of crashing, the -d column notes
// and the usage text are worded independently. All scoring runs on the GPU; there is no CPU path.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include <unistd.h>

#include "plaac_host.h"
#include "plaac_native.h"

namespace {

struct Options {
    std::string input, bgfile, bgfreq, fgfreq, plotlist, dotfile;
    int corelength = 60, ww1 = 41, ww2 = 41, ww3 = 41, hmmtype = 1;
    double alpha = 1.0;
    bool headers = false, params = true;
};

void put(const std::string &s) { std::fwrite(s.data(), 1, s.size(), stdout); }

// PLAAC_TIMING=1: wall-clock of the host stages on stderr (never on stdout: the tables stay byte-identical)
struct StageTimer {
    bool on = std::getenv("PLAAC_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    void lap(const char *what, double units = 0, const char *unit = "") {
        if (!on) return;
        const auto t1 = std::chrono::steady_clock::now();
        const double s = std::chrono::duration<double>(t1 - t0).count();
        if (units > 0) std::fprintf(stderr, "plaac-timing: %-28s %9.3f ms  %.3g %s/s\n", what, s * 1e3, units / s, unit);
        else std::fprintf(stderr, "plaac-timing: %-28s %9.3f ms\n", what, s * 1e3);
        t0 = t1;
    }
} g_timer;

void usage() {
    put("------------------------------------------------------------\n"
        "plaac (MI355X-native engine) - offered with NO WARRANTY WHATSOEVER.\n"
        "------------------------------------------------------------\n"
        "USAGE: plaac -i input.fa > output.txt   (tab-delimited table, one protein per line)\n"
        "Options:\n"
        "  -c core_length   minimal contiguous prion-like domain length for the HMM parse (default 60)\n"
        "  -B bg_freqs.txt  background AA frequencies/counts, 22 lines in the order\n"
        "                   X A C D E F G H I K L M N P Q R S T V W Y *  (X and * are zeroed, rest normalised)\n"
        "  -b background.fa FASTA used to count background AA frequencies (ignored with -B; defaults to input.fa).\n"
        "                   With -b and no -i the counts are printed in -B format and the program exits.\n"
        "  -a alpha         mix of S. cerevisiae (alpha) and -B/-b/-i (1-alpha) background, in [0,1] (default 1.0)\n"
        "  -F fg_freqs.txt  prion-like AA frequencies in -B format (default: 28 S. cerevisiae domains)\n"
        "  -w window        FoldIndex window (default 41)      -W window   PAPA window (default 41)\n"
        "  -d               print a description of the output columns\n"
        "  -s               do not print the run-time parameter block\n"
        "  -p list.txt|all  per-residue table for the listed sequence names (or all) instead of the summary\n"
        "  -h file          write the HMM (transitions + emission tables) in GraphViz dot format to file\n");
}

void column_notes() {
    static const char *const notes[][2] = {
        {"SEQid", "header line of the FASTA record"},
        {"MW", "largest count of N+Q in any window of MWlen residues (Michelitsch-Weissman)"},
        {"MWstart", "first residue (1-based) of the first window reaching MW"},
        {"MWend", "last residue of that window"},
        {"MWlen", "window length used: min(80, PROTlen)"},
        {"LLR", "largest sum of per-residue PLAAC log-likelihood ratios (natural log) over a window of c residues; NaN if PROTlen < c"},
        {"LLRstart", "first residue of the first window reaching LLR; 0 if PROTlen < c"},
        {"LLRend", "last residue of that window; -1 if PROTlen < c"},
        {"LLRlen", "c, or 0 if PROTlen < c"},
        {"NLLR", "LLR / LLRlen"},
        {"VITmaxrun", "longest run of the PrD-like state in the Viterbi parse"},
        {"COREscore", "largest LLR window sum lying entirely inside the Viterbi PrD parse; NaN if VITmaxrun < c"},
        {"COREstart", "first residue of that core; 0 if none"},
        {"COREend", "last residue of that core; -1 if none"},
        {"CORElen", "c, or 0 if there is no core"},
        {"PRDscore", "sum of LLRs over the whole Viterbi PrD run that contains the core; 0.000 if none"},
        {"PRDstart", "first residue of that run; 0 if none"},
        {"PRDend", "last residue of that run; -1 if none"},
        {"PRDlen", "length of that run"},
        {"PROTlen", "number of residues, a terminal stop excluded"},
        {"HMMall", "log-likelihood ratio of the sequence: two-state HMM vs background-only model"},
        {"HMMvit", "the same along the Viterbi path"},
        {"COREaa", "residues of the core; - if none"},
        {"STARTaa", "first 15 residues of the PrD run; - if none"},
        {"ENDaa", "last 15 residues of the PrD run; - if none"},
        {"PRDaa", "residues of the PrD run; - if none"},
        {"FInumaa", "residues in FoldIndex-disordered runs of at least 5"},
        {"FImeanhydro", "mean scaled hydropathy <H> of the protein"},
        {"FImeancharge", "mean net charge <R> of the protein"},
        {"FImeancombo", "2.785<H> - |<R>| - 1.151"},
        {"FImaxrun", "longest FoldIndex-disordered run"},
        {"PAPAcombo", "largest doubly smoothed PAPA propensity among positions with negative smoothed FoldIndex; NaN if none"},
        {"PAPAprop", "that propensity"},
        {"PAPAfi", "doubly smoothed FoldIndex at PAPAcen"},
        {"PAPAllr", "smoothed PLAAC LLR at PAPAcen"},
        {"PAPAllr2", "doubly smoothed PLAAC LLR at PAPAcen"},
        {"PAPAcen", "position (1-based) where PAPAprop is attained; 0 if none"},
        {"PAPAaa", "residues of the PAPA window centred at PAPAcen"},
    };
    put("############################ Description of output columns ############################\n");
    for (const auto &n : notes) put(std::string("## ") + n[0] + ": " + n[1] + "\n");
    put("#######################################################################################\n");
}

bool die(plaac_ctx *ctx, const char *what, plaac_status st) {
    std::fprintf(stderr, "plaac: %s failed (status %d): %s\n", what, (int)st, plaac_last_error(ctx));
    return false;
}

// counts of a FASTA on the device (computeaafreq :1655-1666)
template <class GetCtx>
bool count_background(GetCtx &&get_ctx, const std::string &path, double out[PLAAC_NAA], plaac_fasta **keep,
                      plaac_batch **keep_batch) {
    for (int i = 0; i < PLAAC_NAA; ++i) out[i] = 0.0;
    plaac_fasta *f = nullptr;
    plaac_status st = plaac_fasta_read(path.c_str(), &f);
    if (st == PLAAC_ERR_IO) {
        put("# Couldn't open " + path + "\n"); // (:4318-4321) and carry on with zero counts
        return true;
    }
    if (st != PLAAC_OK) return die(nullptr, "reading FASTA", st);
    g_timer.lap("read+encode FASTA", (double)f->nres, "residues");
    plaac_ctx *ctx = get_ctx();
    if (!ctx) {
        plaac_fasta_free(f);
        return false;
    }
    int64_t counts[PLAAC_NAA];
    plaac_batch *b = nullptr; // upload once: the scoring pass reuses the resident residues
    st = plaac_batch_upload(ctx, f->codes, f->offsets, f->nrec, &b);
    g_timer.lap("upload batch (H2D)", (double)f->nres, "residues");
    if (st == PLAAC_OK) st = plaac_batch_histogram(b, counts);
    g_timer.lap("background histogram (GPU)", (double)f->nres, "residues");
    if (st != PLAAC_OK) {
        plaac_batch_free(b);
        plaac_fasta_free(f);
        return die(ctx, "plaac_batch_histogram", st);
    }
    for (int i = 0; i < PLAAC_NAA; ++i) out[i] = (double)counts[i];
    if (keep) {
        *keep = f;
        *keep_batch = b;
    } else {
        plaac_batch_free(b);
        plaac_fasta_free(f);
    }
    return true;
}

bool read_params_file(const std::string &path, double vec[PLAAC_NAA]) {
    int warn[PLAAC_NAA];
    plaac_status st = plaac_read_aa_params(path.c_str(), vec, warn);
    if (st == PLAAC_ERR_IO) {
        put("# Couldn't open " + path + "\n"); // (:2708-2710)
        return true;
    }
    if (st != PLAAC_OK) {
        std::fprintf(stderr, "plaac: %s is not a 22-line parameter file\n", path.c_str());
        return false;
    }
    for (int i = 0; i < PLAAC_NAA; ++i)
        if (warn[i]) put("# warning: " + path + " does not have expected name in line" + std::to_string(i + 1) + "\n");
    return true;
}

const char *rec_name(const plaac_fasta *f, uint32_t i) { return f->names + f->name_off[i]; }

bool score_all(plaac_ctx *ctx, const plaac_fasta *f, plaac_batch *batch, const Options &o) {
    if (o.headers) column_notes();
    put(std::string(plaac_summary_header()) + "\n");
    if (f->nrec == 0) return true;
    // not value-initialised: the library overwrites every row, and its copy threads fault the pages in in parallel
    std::unique_ptr<plaac_row[]> rows(new plaac_row[f->nrec]);
    plaac_status st = batch ? plaac_batch_score(batch, rows.get(), nullptr)
                            : plaac_score(ctx, f->codes, f->offsets, f->nrec, rows.get(), nullptr);
    if (st != PLAAC_OK) return die(ctx, "plaac_score", st);
    g_timer.lap(batch ? "score resident batch (GPU + D2H)" : "score (H2D + GPU + D2H)", (double)f->nres, "residues");
    // format in parallel (contiguous row ranges per thread), print in file order
    const unsigned nt = f->nrec < 2048 ? 1u : plaac_host_threads();
    std::vector<std::string> part(nt);
    std::vector<int> bad(nt, 0);
    auto work = [&](unsigned t) {
        const uint32_t r0 = (uint32_t)((uint64_t)f->nrec * t / nt), r1 = (uint32_t)((uint64_t)f->nrec * (t + 1) / nt);
        std::vector<char> line;
        std::string &out = part[t];
        for (uint32_t i = r0; i < r1; ++i) {
            const uint64_t len = f->offsets[i + 1] - f->offsets[i];
            if (len == 0) {
                std::fprintf(stderr, "plaac: record '%s' has no sequence, skipped\n", rec_name(f, i));
                continue;
            }
            line.resize(len * 3 + std::strlen(rec_name(f, i)) + 2048);
            long k = plaac_format_summary_row(&rows[i], rec_name(f, i), f->codes + f->offsets[i], len, o.corelength,
                                              o.ww2, line.data(), line.size());
            if (k < 0) {
                bad[t] = 1;
                return;
            }
            if (k == 0) continue; // nothing left after the stop trim (:762)
            out.append(line.data(), (size_t)k);
            out.push_back('\n');
        }
    };
    if (nt == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nt; ++t) pool.emplace_back(work, t);
        for (auto &th : pool) th.join();
    }
    g_timer.lap("format rows", (double)f->nrec, "rows");
    for (unsigned t = 0; t < nt; ++t) {
        if (bad[t]) return false;
        put(part[t]);
    }
    std::fflush(stdout);
    g_timer.lap("write table");
    return true;
}

// readhashtable (:1866-1893): first column = name, optional second column = display name, line number = order
bool read_plot_list(const std::string &path, std::map<std::string, std::string> &title,
                    std::map<std::string, std::string> &order) {
    FILE *fp = std::fopen(path.c_str(), "rb");
    if (!fp) {
        put("# Couldn't open " + path + "\n");
        put("# Couldn't open " + path + "\n"); // the reference opens the list twice
        return true;
    }
    std::string file;
    char chunk[1 << 14];
    size_t n;
    while ((n = std::fread(chunk, 1, sizeof chunk, fp)) > 0) file.append(chunk, n);
    std::fclose(fp);
    size_t pos = 0;
    int lineno = 1;
    while (pos < file.size()) {
        size_t e = pos;
        while (e < file.size() && file[e] != '\n' && file[e] != '\r') ++e;
        std::string line = file.substr(pos, e - pos);
        if (e < file.size()) {
            if (file[e] == '\r' && e + 1 < file.size() && file[e + 1] == '\n') ++e;
            ++e;
        }
        pos = e;
        // String.split("\\t") drops trailing empty fields
        std::vector<std::string> col;
        size_t p = 0;
        for (;;) {
            size_t q = line.find('\t', p);
            col.push_back(line.substr(p, q == std::string::npos ? std::string::npos : q - p));
            if (q == std::string::npos) break;
            p = q + 1;
        }
        while (col.size() > 1 && col.back().empty()) col.pop_back();
        order[col[0]] = std::to_string(lineno);
        title[col[0]] = col.size() > 1 ? col[1] : col[0];
        ++lineno;
    }
    return true;
}

bool plot_some(plaac_ctx *ctx, const plaac_fasta *f, const Options &o) {
    std::map<std::string, std::string> title, order;
    const bool all = o.plotlist == "all";
    if (!all && !read_plot_list(o.plotlist, title, order)) return false;
    put(std::string(plaac_tracks_header()) + "\n");
    // select records (:617) and build the sub-batch
    std::vector<uint32_t> pick;
    std::vector<std::string> ids, names;
    int genecount = 1;
    for (uint32_t i = 0; i < f->nrec; ++i) {
        const std::string name = rec_name(f, i);
        if (!(all || title.count(name) || title.count(">" + name))) continue;
        const uint64_t b = f->offsets[i], e = f->offsets[i + 1];
        uint64_t n = e - b;
        if (n == 0) {
            std::fprintf(stderr, "plaac: record '%s' has no sequence, skipped\n", name.c_str());
            continue;
        }
        if (f->codes[e - 1] == 21) --n;
        std::string id = std::to_string(genecount), nm = name;
        if (title.count(name)) nm = title[name];
        if (order.count(name)) id = order[name];
        ++genecount;
        if (n == 0) {
            std::fprintf(stderr, "plaac: record '%s' is only a stop codon, skipped\n", name.c_str());
            continue;
        }
        pick.push_back(i);
        ids.push_back(id);
        names.push_back(nm);
    }
    if (pick.empty()) return true;
    std::vector<uint64_t> offs(pick.size() + 1, 0);
    for (size_t k = 0; k < pick.size(); ++k)
        offs[k + 1] = offs[k] + (f->offsets[pick[k] + 1] - f->offsets[pick[k]]);
    const uint64_t total = offs.back();
    std::vector<uint8_t> codes(total + 64);
    for (size_t k = 0; k < pick.size(); ++k)
        std::memcpy(codes.data() + offs[k], f->codes + f->offsets[pick[k]], offs[k + 1] - offs[k]);
    std::vector<plaac_row> rows(pick.size());
    std::vector<uint8_t> t8(2 * total + 2);
    std::vector<double> t64(10 * total + 10);
    plaac_tracks tr;
    tr.vit = t8.data();
    tr.map = t8.data() + total;
    double *d = t64.data();
    tr.charge = d;
    tr.hydro = d + total;
    tr.fi = d + 2 * total;
    tr.plaacllr = d + 3 * total;
    tr.papa = d + 4 * total;
    tr.fix2 = d + 5 * total;
    tr.plaacllrx2 = d + 6 * total;
    tr.papax2 = d + 7 * total;
    tr.post0 = d + 8 * total;
    tr.post1 = d + 9 * total;
    plaac_status st = plaac_score(ctx, codes.data(), offs.data(), (uint32_t)pick.size(), rows.data(), &tr);
    if (st != PLAAC_OK) return die(ctx, "plaac_score", st);
    std::vector<char> buf;
    for (size_t k = 0; k < pick.size(); ++k) {
        const uint32_t n = (uint32_t)rows[k].prot_len;
        buf.resize(plaac_track_rows_bound(n, ids[k].size(), names[k].size()));
        long len = plaac_format_track_rows(&tr, offs[k], codes.data() + offs[k], n, ids[k].c_str(), names[k].c_str(),
                                           buf.data(), buf.size());
        if (len < 0) return false;
        std::fwrite(buf.data(), 1, (size_t)len, stdout);
    }
    return true;
}

} // namespace

int main(int argc, char **argv) {
    g_timer.lap("process start-up (dynamic loading)");
    Options o;
    std::vector<std::string> a(argv + 1, argv + argc);
    // the reference's hand-rolled loop (:337-353): value flags consume the next token; a value flag in
    // the last position is silently ignored; unknown flags are reported on stdout
    size_t i = 0;
    while (i + 1 < a.size() || (i < a.size() && (a[i] == "-d" || a[i] == "-s"))) {
        const std::string &f = a[i];
        if (f == "-i") o.input = a[++i];
        else if (f == "-b") o.bgfile = a[++i];
        else if (f == "-B") o.bgfreq = a[++i];
        else if (f == "-F") o.fgfreq = a[++i];
        else if (f == "-c") o.corelength = std::atoi(a[++i].c_str());
        else if (f == "-w") o.ww1 = std::atoi(a[++i].c_str());
        else if (f == "-W") o.ww2 = std::atoi(a[++i].c_str());
        else if (f == "-a") o.alpha = std::atof(a[++i].c_str());
        else if (f == "-m") o.hmmtype = std::atoi(a[++i].c_str());
        else if (f == "-p") o.plotlist = a[++i];
        else if (f == "-h") o.dotfile = a[++i];
        else if (f == "-d") o.headers = true;
        else if (f == "-s") o.params = false;
        else put("# skipping unknown option " + f + "\n");
        ++i;
    }
    o.ww3 = o.ww2; // (:355)

    if (o.input.empty() && o.bgfile.empty() && o.bgfreq.empty()) {
        usage();
        return 0;
    }
    if (o.corelength < 1 || o.ww1 < 1 || o.ww2 < 1) {
        std::fprintf(stderr, "plaac: -c, -w and -W must be positive\n");
        return 2;
    }

    plaac_params P;
    plaac_params_init(&P, nullptr, nullptr, 1.0, o.corelength, o.ww1, o.ww2, o.ww3, 1);
    // The GPU context (HIP start-up, a few hundred ms) is created on a second thread while this one reads and
    // encodes the FASTA; need_ctx() joins it the first time a device call is due.
    plaac_ctx *ctx = nullptr;
    std::string ctx_err;
    plaac_status ctx_st = PLAAC_OK;
    std::thread ctx_thread;
    bool ctx_started = false, ctx_joined = false;
    if (!o.input.empty() || !o.bgfile.empty()) {
        ctx_started = true;
        ctx_thread = std::thread([&, P0 = P] { // its own copy: main re-initialises P once the background is known
            ctx_st = plaac_ctx_create(&P0, 0, &ctx);
            if (ctx_st != PLAAC_OK) ctx_err = plaac_last_error(nullptr); // thread-local message: copy it here
        });
    }
    auto need_ctx = [&]() -> bool {
        if (!ctx_started) {
            ctx_started = ctx_joined = true;
            ctx_st = plaac_ctx_create(&P, 0, &ctx);
            if (ctx_st != PLAAC_OK) ctx_err = plaac_last_error(nullptr);
        } else if (!ctx_joined) {
            ctx_thread.join();
            ctx_joined = true;
            g_timer.lap("wait for GPU context");
        }
        if (ctx_st != PLAAC_OK) {
            std::fprintf(stderr, "plaac: no usable MI355X (gfx950) device: %s\n", ctx_err.c_str());
            return false;
        }
        return true;
    };
    struct Joiner { // never leave main with the thread still joinable
        std::thread &t;
        bool &joined;
        ~Joiner() {
            if (t.joinable() && !joined) t.join();
        }
    } joiner{ctx_thread, ctx_joined};

    auto get_ctx = [&]() -> plaac_ctx * { return need_ctx() ? ctx : nullptr; };
    // background counts (:377-384)
    double bgf[PLAAC_NAA] = {0}, fgf[PLAAC_NAA];
    plaac_fasta *input = nullptr;
    plaac_batch *input_batch = nullptr;
    bool ok = true;
    if (!o.bgfreq.empty()) {
        ok = read_params_file(o.bgfreq, bgf);
    } else if (!o.bgfile.empty()) {
        ok = count_background(get_ctx, o.bgfile, bgf, o.bgfile == o.input ? &input : nullptr, &input_batch);
    } else if (!o.input.empty()) {
        ok = count_background(get_ctx, o.input, bgf, &input, &input_batch);
    }
    if (!ok) return 1;
    const bool have_fg = !o.fgfreq.empty();
    if (have_fg && !read_params_file(o.fgfreq, fgf)) return 1;

    char text[8192];
    if ((!o.bgfile.empty() || !o.bgfreq.empty()) && o.input.empty()) { // (:394-403): dump and exit
        plaac_format_aa_params(bgf, text, sizeof text);
        put(text);
        if (ctx) plaac_ctx_destroy(ctx);
        return 0;
    }
    if (o.alpha > 1 || o.alpha < 0) { // (:444-447)
        put("# warning: invalid alpha; using alpha = 1.0\n");
        o.alpha = 1.0;
    }
    if (plaac_params_init(&P, have_fg ? fgf : nullptr, bgf, o.alpha, o.corelength, o.ww1, o.ww2, o.ww3, 1) != PLAAC_OK) {
        std::fprintf(stderr, "plaac: bad parameters\n");
        return 2;
    }
    if (o.params) {
        plaac_format_param_block(&P, text, sizeof text);
        put(text);
    }
    if (!o.dotfile.empty()) { // hmm1.dottify(hmmdotfile, true) (:520-522)
        std::vector<char> dot(16384);
        const long k = plaac_format_hmm_dot(&P, dot.data(), dot.size());
        FILE *fp = k >= 0 ? std::fopen(o.dotfile.c_str(), "wb") : nullptr;
        if (fp) {
            std::fwrite(dot.data(), 1, (size_t)k, fp);
            std::fclose(fp);
        } else {
            put("## problem writing to dotfile\n");
        }
    }

    if (!need_ctx()) return 1;
    if (plaac_ctx_set_params(ctx, &P) != PLAAC_OK) {
        die(ctx, "plaac_ctx_set_params", PLAAC_ERR_ARG);
        return 1;
    }
    if (!input) {
        plaac_status st = plaac_fasta_read(o.input.c_str(), &input);
        if (st == PLAAC_ERR_IO) {
            put("# Couldn't open " + o.input + "\n");
            plaac_fasta empty{};
            uint64_t zero = 0;
            empty.offsets = &zero;
            ok = o.plotlist.empty() ? score_all(ctx, &empty, nullptr, o) : plot_some(ctx, &empty, o);
            plaac_ctx_destroy(ctx);
            return ok ? 0 : 1;
        }
        if (st != PLAAC_OK) {
            die(nullptr, "reading FASTA", st);
            return 1;
        }
    }
    ok = o.plotlist.empty() ? score_all(ctx, input, input_batch, o) : plot_some(ctx, input, o);
    // The output is complete and flushed. Leaving through _exit skips unmapping hundreds of MB of host and device
    // buffers and the HIP runtime's own shutdown, which the operating system does faster (PLAAC_TEARDOWN=1 keeps
    // the orderly path, e.g. under leak checkers).
    std::fflush(stdout);
    std::fflush(stderr);
    if (!std::getenv("PLAAC_TEARDOWN")) ::_exit(ok ? 0 : 1);
    plaac_batch_free(input_batch);
    plaac_fasta_free(input);
    plaac_ctx_destroy(ctx);
    g_timer.lap("teardown");
    return ok ? 0 : 1;
}
