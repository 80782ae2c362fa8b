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
//
// The input is STREAMED (the reference reads record by record, fastareader :4302-4375): a reader thread cuts the
// FASTA into batches, one worker thread per scoring context (two contexts per visible GPU, every GPU of the node)
// scores them, and the main thread formats and prints them in file order while later batches are still being read
// and scored. Memory is bounded by the batches in flight, whatever the size of the file.
//   PLAAC_BATCH_RECORDS / PLAAC_BATCH_BYTES   batch size (default 262144 records / 96 MiB of FASTA text)
//   PLAAC_DEVICES=0,1,...                     devices to use, a device may be repeated (default: one context on each of the
//                                             first ceil(input bytes / PLAAC_BYTES_PER_DEVICE) GPUs, 4 GiB per device: a context
//                                             costs 0.15 s to bring up, and the kernels of 3 GB of FASTA take 0.14 s of ONE GPU)
//   PLAAC_KEEP_BYTES                          the background pass keeps the parsed batches for the scoring pass up to this many
//                                             bytes (default: 4 GiB, at most a quarter of the memory the host / cgroup has
//                                             free), beyond it the kept batches are released and the scoring pass reads the
//                                             file again. Resident set of the default two-pass run: that budget + the batches
//                                             in flight (4 x PLAAC_BATCH_BYTES of text, parsed) + 256 MiB of formatted output
//   PLAAC_SINGLE_PASS=0                       always the reference's two passes. Default (round 5): when the scoring tables do
//                                             not depend on the input's residue counts - alpha = 1, background counted from the
//                                             scored input itself (plaac.java:377-384, :458) - and the input fits
//                                             PLAAC_KEEP_BYTES (any size when stdout is a plain file: the table is then written
//                                             in place and nothing waits in memory), the counting pass runs INSIDE the scoring pass
//                                             (plaac_score_begin_counting): the table is formatted while the file is still being
//                                             read, the parameter block (whose "## bg_input" line needs the final counts) and
//                                             everything behind it are held back until the last batch has been counted, then
//                                             written in the reference's order. The bytes on stdout are the same (tested).
//   PLAAC_PLACED_WRITE=0                      single pass with stdout a plain file (not a pipe, not opened for appending): the
//                                             table is written where it belongs while the block still waits - a placeholder of
//                                             the block's length (made-up counts) first, the real block over it at the end; if
//                                             the length differs after all (no valid residue: NaN) the table is moved. 0: hold
//                                             everything back, as for a pipe. 10 M sequences: 1.40 - 1.50 s -> 1.25 - 1.35 s
//   PLAAC_DEVICE_PARSE=0                      single pass: the HOST splits lines and encodes (plaac_fasta_next). Default (round
//                                             5, K1): the host only finds the records and keeps their names, the device parses
//                                             (plaac_score_begin_text); the few residues a row prints are read from the text
//   PLAAC_DEVICE_FORMAT=0                     the HOST formats the rows of a text batch. Default: the device writes the
//                                             table's text (plaac_score_end_text_table: the rows never cross PCIe; a batch with
//                                             a value it will not vouch for comes back the old way). 1.25 - 1.44 s -> 0.60 - 0.78 s
//   PLAAC_TIMING=1                            stage clock on stderr; with PLAAC_TIMING_T0=<the launcher's CLOCK_MONOTONIC, ns>
//                                             also since the launch, with PLAAC_TIMING_MAPS=1 the large resident mappings
//   PLAAC_HUGE_PAGES=0                        plain allocations for the big host buffers (encoded residues, rows, formatted
//                                             text) instead of transparent huge pages on request. 10 M sequences, same box:
//                                             two passes 2.18 - 2.47 s, + huge pages 2.09 - 2.16, + single pass 1.94 - 2.11
//                                             (profiles/r05_e2e_single_pass.txt; first touch of 3 GiB 0.44 -> 0.12 s)
#include <atomic>
#include <fcntl.h>
#include <chrono>
#include <ctime>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unordered_set>
#include <vector>

#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
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

// single-pass runs hold back what the reference prints before the table until the input's residue counts are final:
// while g_defer is set, put() collects instead of writing
std::string *g_defer = nullptr;
void put(const std::string &s) {
    if (g_defer) g_defer->append(s);
    else std::fwrite(s.data(), 1, s.size(), stdout);
}

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
    // PLAAC_TIMING_T0 = the launcher's CLOCK_MONOTONIC in ns when it started this process: what exec + dynamic loading took
    // (this object is constructed after the libraries are in), and `mark` = the same clock at a point of interest
    static double mono_ms() {
        timespec ts;
        clock_gettime(CLOCK_MONOTONIC, &ts);
        return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
    }
    double loaded_ms = mono_ms();
    void mark(const char *what) const {
        const char *e = std::getenv("PLAAC_TIMING_T0");
        if (!on || !e) return;
        const double t0ms = std::strtod(e, nullptr) * 1e-6;
        std::fprintf(stderr, "plaac-timing: %-28s %9.3f ms after the launch (libraries loaded at %.3f ms)\n", what, mono_ms() - t0ms,
                     loaded_ms - t0ms);
        if (FILE *fp = std::fopen("/proc/self/status", "r")) { // what the operating system will have to take back at exit
            char line[256];
            while (std::fgets(line, sizeof line, fp))
                if (!std::strncmp(line, "VmRSS", 5) || !std::strncmp(line, "Rss", 3) || !std::strncmp(line, "VmPTE", 5) ||
                    !std::strncmp(line, "VmSize", 6) || !std::strncmp(line, "VmLck", 5) || !std::strncmp(line, "VmPin", 5))
                    std::fprintf(stderr, "plaac-timing:     %s", line);
            std::fclose(fp);
        }
        {
            rusage ru;
            getrusage(RUSAGE_SELF, &ru);
            std::fprintf(stderr, "plaac-timing:     cpu used so far: user %.3f s, system %.3f s\n", ru.ru_utime.tv_sec + ru.ru_utime.tv_usec * 1e-6,
                         ru.ru_stime.tv_sec + ru.ru_stime.tv_usec * 1e-6);
            if (FILE *fp = std::fopen("/sys/fs/cgroup/cpu.stat", "r")) { // (the container's CPU quota: was the process held back?)
                char line[256];
                while (std::fgets(line, sizeof line, fp))
                    if (std::strstr(line, "throttled")) std::fprintf(stderr, "plaac-timing:     cgroup %s", line);
                std::fclose(fp);
            }
        }
        if (std::getenv("PLAAC_TIMING_MAPS"))
            if (FILE *fp = std::fopen("/proc/self/smaps", "r")) { // the mappings with more than 32 MB resident
                char line[512], head[512] = "";
                while (std::fgets(line, sizeof line, fp)) {
                    unsigned long a, b, kb;
                    if (std::sscanf(line, "%lx-%lx ", &a, &b) == 2 && std::strchr(line, '-') && !std::strstr(line, "kB")) std::strncpy(head, line, sizeof head - 1);
                    else if (std::sscanf(line, "Rss: %lu kB", &kb) == 1 && kb > 32768) std::fprintf(stderr, "plaac-timing:     %8lu kB resident in %s", kb, head);
                }
                std::fclose(fp);
            }
    }
} g_timer;

// PLAAC_TIMING: how long each stage of the pipeline was BUSY (summed over its calls; the stages run side by side)
struct Busy {
    std::atomic<long long> ns[10] = {};
    std::atomic<long long> cpu_fmt{0};
    const char *name[10] = {"reader: next batch", "begin: upload + parse kernels (+ enqueue without an uploader thread)", "worker: end (wait, download)",
                           "sink: format a batch (all threads, wall)", "sink: hand parts to the writer (waits for room)",
                           "writer: fwrite", "writer: waiting for text", "sink: waiting for a scored batch",
                           "sink: format threads, summed over threads", "worker: begin of an uploaded batch (enqueue)"};
    struct Scope {
        Busy &b;
        int k;
        std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        ~Scope() { b.ns[k] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
    };
    Scope in(int k) { return Scope{*this, k}; }
    void report() {
        if (!g_timer.on) return;
        for (int k = 0; k < 10; ++k)
            if (ns[k]) std::fprintf(stderr, "plaac-timing:   busy %-52s %9.3f ms\n", name[k], ns[k] * 1e-6);
        if (cpu_fmt) std::fprintf(stderr, "plaac-timing:   cpu  %-52s %9.3f ms\n", "sink: format threads, CPU time summed", cpu_fmt * 1e-6);
    }
} g_busy;

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


// ------------------------------------------------------------------------------------------------
// pipeline plumbing
// ------------------------------------------------------------------------------------------------
uint64_t env_u64(const char *name, uint64_t dflt) {
    const char *e = std::getenv(name);
    if (!e || !*e) return dflt;
    const long long v = std::atoll(e);
    return v > 0 ? (uint64_t)v : dflt;
}

// on / off switches: unset = dflt, "0" = off, anything else = on (env_u64 takes 0 for "use the default")
bool env_flag(const char *name, bool dflt) {
    const char *e = std::getenv(name);
    if (!e || !*e) return dflt;
    return !(e[0] == '0' && e[1] == 0);
}

// Batches the background pass keeps for the scoring pass. Once the input turns out to be larger than the budget the
// kept ones are of no use: those the pass is done with are released at once, the others as soon as their Batch goes.
struct KeptState {
    std::mutex m;
    std::unordered_set<plaac_fasta *> done, freed; // kept batches whose Batch object is gone / that have been released
    bool overflow = false, closed = false;
    void batch_gone(plaac_fasta *f) {
        std::lock_guard<std::mutex> l(m);
        if (closed) return;
        if (overflow) {
            if (freed.insert(f).second) plaac_fasta_free(f);
        } else {
            done.insert(f);
        }
    }
    void on_overflow() {
        std::lock_guard<std::mutex> l(m);
        overflow = true;
        for (plaac_fasta *f : done)
            if (freed.insert(f).second) plaac_fasta_free(f);
        done.clear();
    }
    void release_rest(std::vector<plaac_fasta *> &keep) { // end of the pass: nothing of `keep` survives
        std::lock_guard<std::mutex> l(m);
        closed = true;
        for (plaac_fasta *f : keep)
            if (freed.insert(f).second) plaac_fasta_free(f);
        keep.clear();
    }
};

// default budget of the kept batches (PLAAC_KEEP_BYTES overrides): 4 GiB, but never more than a quarter of what the
// host / the cgroup has free, so that a small-memory box re-reads the file instead of being OOM-killed
uint64_t default_keep_bytes() {
    uint64_t avail = UINT64_MAX;
    if (FILE *fp = std::fopen("/proc/meminfo", "r")) {
        char line[256];
        while (std::fgets(line, sizeof line, fp)) {
            unsigned long long kb = 0;
            if (std::sscanf(line, "MemAvailable: %llu kB", &kb) == 1) avail = (uint64_t)kb << 10;
        }
        std::fclose(fp);
    }
    auto read_u64 = [](const char *path, uint64_t &v) {
        FILE *fp = std::fopen(path, "r");
        if (!fp) return false;
        char buf[64] = {0};
        const bool ok = std::fgets(buf, sizeof buf, fp) && buf[0] >= '0' && buf[0] <= '9';
        std::fclose(fp);
        if (ok) v = std::strtoull(buf, nullptr, 10);
        return ok;
    };
    uint64_t lim = 0, cur = 0;
    if (read_u64("/sys/fs/cgroup/memory.max", lim) && read_u64("/sys/fs/cgroup/memory.current", cur))
        avail = std::min(avail, lim > cur ? lim - cur : 0);
    else if (read_u64("/sys/fs/cgroup/memory/memory.limit_in_bytes", lim) &&
             read_u64("/sys/fs/cgroup/memory/memory.usage_in_bytes", cur))
        avail = std::min(avail, lim > cur ? lim - cur : 0);
    return std::min<uint64_t>(4ull << 30, avail / 4);
}

// Host buffers of a batch's rows / formatted text on transparent huge pages where the host grants them on request (THP mode
// "madvise" on the MI355X boxes; tools/thp_probe.cpp: first touch of 3 GiB 0.44 -> 0.12 s, unmapping 0.27 -> 0.12 s): a
// 10 M-sequence run touches 1.6 GB of rows and 2.1 GB of text once each.
constexpr size_t HUGE_PAGE = 2u << 20;
inline bool huge_pages_on() { // PLAAC_HUGE_PAGES=0: plain allocations (A/B switch)
    static const bool on = !(std::getenv("PLAAC_HUGE_PAGES") && std::getenv("PLAAC_HUGE_PAGES")[0] == '0');
    return on;
}
inline void advise_huge(void *p, size_t bytes) {
    if (!huge_pages_on()) return;
    const uintptr_t a = ((uintptr_t)p + HUGE_PAGE - 1) / HUGE_PAGE * HUGE_PAGE, e = ((uintptr_t)p + bytes) / HUGE_PAGE * HUGE_PAGE;
    if (e > a) (void)madvise((void *)a, e - a, MADV_HUGEPAGE);
}
// the rows (codes) of a batch: uninitialised (the device copy writes every element), huge pages
template <class T>
struct HugeBuf {
    T *p = nullptr;
    size_t n = 0;
    void resize(size_t k) {
        std::free(p);
        n = k;
        const size_t bytes = k * sizeof(T);
        if (bytes >= 4 * HUGE_PAGE && huge_pages_on()) {
            const size_t rounded = (bytes + HUGE_PAGE - 1) / HUGE_PAGE * HUGE_PAGE;
            p = (T *)std::aligned_alloc(HUGE_PAGE, rounded);
            if (p) (void)madvise(p, rounded, MADV_HUGEPAGE);
        } else {
            p = (T *)std::malloc(bytes ? bytes : 1);
        }
        if (!p) throw std::bad_alloc();
    }
    T *data() { return p; }
    T &operator[](size_t i) { return p[i]; }
    ~HugeBuf() { std::free(p); }
    HugeBuf() = default;
    HugeBuf(const HugeBuf &) = delete;
    HugeBuf &operator=(const HugeBuf &) = delete;
};
using RowBuf = HugeBuf<plaac_row>;

struct TextBuf {
    char *p = nullptr;
    size_t n = 0, cap = 0;
    bool foreign = false; // the library's buffer (plaac_score_tracks_table): released with plaac_table_free, never reused or grown
    void release() {
        if (foreign) plaac_table_free(p);
        else std::free(p);
        p = nullptr, n = cap = 0, foreign = false;
    }
    void append(const char *s, size_t k) {
        if (n + k > cap) grow(n + k);
        std::memcpy(p + n, s, k);
        n += k;
    }
    void push_back(char c) { append(&c, 1); }
    static char *alloc(size_t &cap) {
        cap = (cap + HUGE_PAGE - 1) / HUGE_PAGE * HUGE_PAGE;
        char *q = (char *)std::aligned_alloc(HUGE_PAGE, cap);
        if (!q) throw std::bad_alloc();
        if (huge_pages_on()) (void)madvise(q, cap, MADV_HUGEPAGE);
        return q;
    }
    void grow(size_t need) {
        size_t c2 = std::max(need, cap * 2);
        char *q = alloc(c2);
        if (n) std::memcpy(q, p, n);
        std::free(p);
        p = q;
        cap = c2;
    }
};
struct Batch {
    uint64_t seq = 0;
    plaac_fasta *f = nullptr;
    bool owned = true; // false: kept by the background pass (KeptState decides when it is released)
    KeptState *kept = nullptr;
    RowBuf rows;
    // track mode: the selected records as their own batch
    std::vector<uint32_t> pick;
    std::vector<std::string> ids, names;
    std::vector<uint8_t> pcodes;
    std::vector<uint64_t> poffs;
    std::vector<uint8_t> t8;
    std::vector<double> t64;
    plaac_tracks tr{};
    plaac_status st = PLAAC_OK;
    bool begun = false; // pipelined scoring: plaac_score_begin has taken the batch (plaac_score_end is owed)
    std::string err;
    // device-side parse (K1): the batch as located text; `view` is what the device made of it (f points at it once the
    // batch has been collected; the names are the text batch's, trimmed by the sink, which sees the batches in file order)
    plaac_fasta_text *ft = nullptr;
    plaac_fasta view{};
    std::vector<uint64_t> toffs;
    std::vector<uint8_t> tblank;
    std::vector<uint32_t> text_ext; // per record: where its header line and its sequence end (plaac_score_end_text)
    plaac_text_batch *tb = nullptr; // the batch uploaded and parsed ahead of its scoring call (plaac_text_upload), until that call
    TextBuf table{};                // the batch's rows as text, made on the device (plaac_score_end_text_table)
    bool have_table = false;
    int last_blank = 0;
    uint64_t table_residues = 0;
    ~Batch() {
        if (table.p) table.release(); // (a batch that never reached the sink: its table is still here)
        if (tb) plaac_text_batch_free(tb);
        if (ft) plaac_fasta_text_free(ft);
        else if (f && owned) plaac_fasta_free(f);
        else if (f && kept) kept->batch_gone(f);
    }
};
using BatchPtr = std::unique_ptr<Batch>;

// bounded FIFO between the reader and the workers
class Queue {
    std::mutex m;
    std::condition_variable cv_put, cv_get;
    std::vector<BatchPtr> q;
    size_t cap;
    bool closed = false;

  public:
    explicit Queue(size_t c) : cap(c) {}
    void set_cap(size_t c) {
        std::lock_guard<std::mutex> l(m);
        cap = c;
        cv_put.notify_all();
    }
    void put(BatchPtr b) {
        std::unique_lock<std::mutex> l(m);
        cv_put.wait(l, [&] { return q.size() < cap || closed; });
        if (closed) return;
        q.push_back(std::move(b));
        cv_get.notify_one();
    }
    BatchPtr get() { // nullptr: closed and drained
        std::unique_lock<std::mutex> l(m);
        cv_get.wait(l, [&] { return !q.empty() || closed; });
        if (q.empty()) return nullptr;
        BatchPtr b = std::move(q.front());
        q.erase(q.begin());
        cv_put.notify_one();
        return b;
    }
    void close() {
        std::lock_guard<std::mutex> l(m);
        closed = true;
        cv_put.notify_all();
        cv_get.notify_all();
    }
};

// scored batches, handed to the main thread in file order; at most `window` batches may be ahead of the printer
class Reorder {
    std::mutex m;
    std::condition_variable cv, cv_room;
    std::map<uint64_t, BatchPtr> done;
    uint64_t next = 0, total = UINT64_MAX, window;

  public:
    explicit Reorder(uint64_t w) : window(w) {}
    void set_window(uint64_t w) {
        std::lock_guard<std::mutex> l(m);
        window = w;
        cv_room.notify_all();
    }
    void wait_room(uint64_t seq) { // called by a worker BEFORE it scores batch `seq`
        std::unique_lock<std::mutex> l(m);
        cv_room.wait(l, [&] { return seq < next + window; });
    }
    bool has_room(uint64_t seq) {
        std::lock_guard<std::mutex> l(m);
        return seq < next + window;
    }
    void put(BatchPtr b) {
        std::lock_guard<std::mutex> l(m);
        const uint64_t s = b->seq;
        done[s] = std::move(b);
        cv.notify_all();
    }
    void set_total(uint64_t n) {
        std::lock_guard<std::mutex> l(m);
        total = n;
        cv.notify_all();
    }
    BatchPtr take() { // next batch in file order; nullptr after the last one
        std::unique_lock<std::mutex> l(m);
        cv.wait(l, [&] { return done.count(next) || next >= total; });
        if (next >= total) return nullptr;
        BatchPtr b = std::move(done[next]);
        done.erase(next);
        ++next;
        cv_room.notify_all();
        return b;
    }
};

// stdout writer on its own thread: the formatter hands over finished text and goes on with the next batch
// Formatted text on its way to stdout: buffers on huge pages that go back to a pool when they have been written, instead of
// a std::string per formatter thread and batch. (Those came from malloc arenas - a fresh set of threads per batch, up to 8 x
// cores arenas - that never shrink: 2.1 GB resident at the end of a 10 M-sequence run, first touched 4 KB at a time and
// handed back 4 KB at a time by the exit, 0.5 s after the output was complete; profiles/r05_e2e_device_parse.txt.)
class TextPool {
    std::mutex m;
    std::vector<TextBuf> idle;
    static constexpr size_t KEEP = 96; // buffers kept for reuse (a part of a full batch is ~5 MB)

  public:
    TextBuf get(size_t cap) {
        {
            std::lock_guard<std::mutex> l(m);
            for (size_t k = idle.size(); k-- > 0;)
                if (idle[k].cap >= cap) {
                    TextBuf b = idle[k];
                    idle[k] = idle.back();
                    idle.pop_back();
                    b.n = 0;
                    return b;
                }
        }
        TextBuf b;
        b.cap = cap;
        b.p = TextBuf::alloc(b.cap);
        return b;
    }
    void put(TextBuf b) {
        if (!b.p) return;
        if (b.foreign) return b.release(); // (not the pool's to recycle)
        {
            std::lock_guard<std::mutex> l(m);
            if (idle.size() < KEEP) {
                idle.push_back(b);
                return;
            }
        }
        std::free(b.p);
    }
    ~TextPool() {
        for (TextBuf &b : idle) std::free(b.p);
    }
};

// Threads that stay: a batch's rows are formatted by the same threads as the batch before. (A fresh std::thread per batch and
// range lands on one of the host's 256 CPUs that has been idle - clocked down, caches cold - and is gone after 10 ms: the
// formatter ran at 480 ns per row inside the run against 205 ns in a loop of its own, profiles/r05_e2e_device_parse.txt.)
class ThreadTeam {
    std::vector<std::thread> th;
    std::mutex m;
    std::condition_variable cv_work, cv_done;
    const std::function<void(unsigned)> *job = nullptr;
    unsigned long gen = 0;
    unsigned remaining = 0;
    bool stop = false;

  public:
    explicit ThreadTeam(unsigned n) {
        for (unsigned t = 0; t < n; ++t)
            th.emplace_back([this, t] {
                unsigned long seen = 0;
                for (;;) {
                    const std::function<void(unsigned)> *f;
                    {
                        std::unique_lock<std::mutex> l(m);
                        cv_work.wait(l, [&] { return stop || gen != seen; });
                        if (stop) return;
                        seen = gen;
                        f = job;
                    }
                    (*f)(t);
                    std::lock_guard<std::mutex> l(m);
                    if (--remaining == 0) cv_done.notify_all();
                }
            });
    }
    unsigned size() const { return (unsigned)th.size(); }
    void run(const std::function<void(unsigned)> &f) { // f(t) for every t < size(), returns when all are through
        std::unique_lock<std::mutex> l(m);
        job = &f;
        remaining = (unsigned)th.size();
        ++gen;
        cv_work.notify_all();
        cv_done.wait(l, [&] { return remaining == 0; });
    }
    ~ThreadTeam() {
        {
            std::lock_guard<std::mutex> l(m);
            stop = true;
            cv_work.notify_all();
        }
        for (auto &t : th) t.join();
    }
};

class Writer {
    std::mutex m;
    std::condition_variable cv_put, cv_get;
    std::vector<TextBuf> q;
    TextPool pool;
    size_t bytes = 0;
    bool closed = false, held = false;
    std::atomic<bool> bad{false}; // a short write or a stream error (ENOSPC, EIO, closed pipe): the table is incomplete
    std::thread th;
    static constexpr size_t CAP = 256u << 20; // text waiting to be written

  public:
    // hold: nothing is written (and nothing waits for room) until release(), which puts `prefix` in front of what has come in
    explicit Writer(bool hold = false) : held(hold) {
        th = std::thread([this] {
            for (;;) {
                std::vector<TextBuf> take;
                {
                    auto busy = g_busy.in(6);
                    std::unique_lock<std::mutex> l(m);
                    cv_get.wait(l, [&] { return (!held && !q.empty()) || closed; });
                    if (q.empty()) return;
                    take.swap(q);
                    bytes = 0;
                    cv_put.notify_all();
                }
                auto busy = g_busy.in(5);
                for (TextBuf &s : take) {
                    if (!bad && std::fwrite(s.p, 1, s.n, stdout) != s.n) bad = true;
                    pool.put(s);
                }
            }
        });
    }
    bool failed() const { return bad; }
    TextBuf buffer(size_t cap) { return pool.get(cap); } // for write(): goes back to the pool when it has been written
    void write(TextBuf s) {
        if (s.n == 0) {
            pool.put(s);
            return;
        }
        std::unique_lock<std::mutex> l(m);
        cv_put.wait(l, [&] { return held || bytes < CAP; });
        bytes += s.n;
        q.push_back(s);
        cv_get.notify_one();
    }
    void write(const std::string &s) {
        TextBuf b = pool.get(s.size() + 1);
        b.append(s.data(), s.size());
        write(b);
    }
    void release(const std::string &prefix) {
        TextBuf b{};
        if (!prefix.empty()) {
            b = pool.get(prefix.size() + 1);
            b.append(prefix.data(), prefix.size());
        }
        std::lock_guard<std::mutex> l(m);
        if (!held) {
            pool.put(b);
            return;
        }
        held = false;
        if (b.n) q.insert(q.begin(), b);
        cv_get.notify_all();
    }
    bool is_held() {
        std::lock_guard<std::mutex> l(m);
        return held;
    }
    bool finish() { // everything handed over so far has been flushed when this returns; false: a write failed
        {
            std::lock_guard<std::mutex> l(m);
            closed = true;
            held = false;
            cv_get.notify_all();
        }
        if (th.joinable()) th.join();
        if (std::fflush(stdout) != 0 || std::ferror(stdout)) bad = true;
        return !bad;
    }
    ~Writer() { finish(); }
};

const char *rec_name(const plaac_fasta *f, uint32_t i) { return f->names + f->name_off[i]; }

struct Engine {
    plaac_node *node = nullptr;
    std::thread starter;
    plaac_status st = PLAAC_OK;
    std::string err;
    bool started = false, joined = false;
    // max_devices: how many of the visible GPUs a run of this size is worth (a context costs ~0.15 s to bring up and ~40 ms to
    // take down, and the summary table of a 3 GB input keeps ONE GPU busy for a tenth of the run: the host formats)
    void start(const plaac_params &P, int max_devices = 1 << 30) { // HIP start-up takes a few hundred ms: it runs beside option handling / parsing
        started = true;
        starter = std::thread([this, P, max_devices] {
            std::vector<int> devs;
            if (const char *e = std::getenv("PLAAC_DEVICES")) {
                for (const char *p = e; *p;) {
                    char *end = nullptr;
                    const long d = std::strtol(p, &end, 10);
                    if (end == p) break;
                    devs.push_back((int)d);
                    p = *end ? end + 1 : end;
                }
            }
            if (devs.empty()) {
                const int n = std::min(plaac_device_count(), std::max(1, max_devices));
                // pipelined scoring (default): ONE context per GPU with two batches in flight; PLAAC_PIPELINE=0: the
                // synchronous entry point on two contexts per GPU (round 3)
                const int per = (int)env_u64("PLAAC_CTX_PER_DEVICE", env_flag("PLAAC_PIPELINE", true) ? 1 : 2);
                for (int k = 0; k < per; ++k)
                    for (int d = 0; d < n; ++d) devs.push_back(d);
            }
            if (devs.empty()) {
                st = PLAAC_ERR_DEVICE;
                err = "no HIP device available";
                return;
            }
            st = plaac_node_create(&P, devs.data(), (int)devs.size(), &node);
            if (st != PLAAC_OK) err = plaac_node_last_error(nullptr);
            // consecutive batches of a context overlap on the device too (the head of a batch beside the tail of the one
            // before it; the long chains of consecutive batches side by side)
            else if (env_flag("PLAAC_PIPELINE", true) && env_flag("PLAAC_OVERLAP_CALLS", true)) (void)plaac_node_set_overlap(node, 1);
        });
    }
    bool ready(const plaac_params &P) {
        if (!started) start(P);
        if (!joined) {
            starter.join();
            joined = true;
            g_timer.lap("wait for GPU contexts");
        }
        if (st != PLAAC_OK) {
            std::fprintf(stderr, "plaac: no usable MI355X (gfx950) device: %s\n", err.c_str());
            return false;
        }
        return true;
    }
    ~Engine() {
        if (started && !joined) starter.join();
    }
};

struct Stream {
    uint32_t batch_records;
    uint64_t batch_bytes;
};

// Reads `path` batch by batch on a reader thread, runs `work(ctx, batch)` on one worker thread per context and hands
// the batches to `sink` in file order on the calling thread. `replay` (nullable): batches kept by an earlier pass, used
// instead of reading the file. `keep` (nullable): keep the parsed batches (up to keep_bytes) for a later pass.
// opens `path` as a batch stream; a file that cannot be opened is reported the reference's way (:4318-4321) and
// treated as empty (*fs stays null, the caller carries on with no records)
bool open_stream(const std::string &path, plaac_fasta_stream **fs) {
    *fs = nullptr;
    const plaac_status st = plaac_fasta_open(path.c_str(), fs);
    if (st == PLAAC_ERR_IO) {
        put("# Couldn't open " + path + "\n");
        return true;
    }
    if (st != PLAAC_OK) {
        std::fprintf(stderr, "plaac: cannot read %s (status %d)\n", path.c_str(), (int)st);
        return false;
    }
    return true;
}

// `finish` (nullable functor): pipelined scoring - `work` only BEGINS a batch (plaac_score_begin: upload + kernels
// enqueued), `finish` collects it (plaac_score_end); a worker keeps two batches in flight on its context, so that the
// upload of a batch, the download of the one before it and the host's own copies run beside the kernels.
struct NoFinish {
    plaac_status operator()(plaac_ctx *, struct Batch &) const { return PLAAC_OK; }
    static constexpr bool enabled = false;
};
template <class F>
struct Finish {
    F f;
    plaac_status operator()(plaac_ctx *c, struct Batch &b) const { return f(c, b); }
    static constexpr bool enabled = true;
};
template <class F>
Finish<F> make_finish(F f) {
    return Finish<F>{f};
}

// `drained` (nullable): called once, by the last worker to leave, when every batch of the input has been worked (and
// collected) - before the sink has seen them all. With it the pass runs AHEAD of its sink like a keeping pass does (the
// reader and the workers are not bound to the printer's window): the single-pass run needs the counts of the whole input
// while the formatter is still busy.
template <class Prep, class Work, class Sink, class Fin = NoFinish>
bool run_pipeline(Engine &eng, const plaac_params &P0, const std::string &path, plaac_fasta_stream *fs, const Stream &sp,
                  std::vector<plaac_fasta *> *replay, std::vector<plaac_fasta *> *keep, uint64_t keep_bytes, Prep &&prep,
                  Work &&work, Sink &&sink, Fin finish = Fin(), const std::function<void()> *drained = nullptr,
                  bool as_text = false, bool run_ahead = true,
                  const std::function<plaac_status(plaac_ctx *, Batch &)> *upload = nullptr) {
    if (!fs && !replay) {
        if (drained) (*drained)();
        return true; // nothing to read
    }
    // A pass that keeps its parsed batches anyway (the background pass, up to keep_bytes) lets the reader run as far
    // ahead as it likes: it parses through the few hundred ms in which the GPU contexts come up instead of stopping
    // four batches in. A pass that does not keep them is bounded to four batches in flight.
    KeptState ks; // (declared before the queues: batches still queued at a failure are destroyed before it)
    // (run_ahead = false: a pass with `drained` whose sink does not hold anything back - the table is being written in place -
    //  stays within the printer's window like any other: memory bounded whatever the size of the input)
    Queue q((keep || (drained && run_ahead)) ? (size_t)1 << 30 : 4);
    Reorder ro(4); // widened once the number of contexts is known
    std::atomic<bool> failed{false};
    bool keeping = keep != nullptr, keep_overflow = false;
    uint64_t kept_bytes = 0;
    std::thread reader([&] {
        uint64_t seq = 0;
        for (;;) {
            if (failed) break;
            BatchPtr b(new Batch());
            if (replay) {
                if (seq >= replay->size()) break;
                b->f = (*replay)[seq];
                b->owned = false;
            } else if (as_text) { // K1: the records are only located; the device parses them
                plaac_fasta_text *t = nullptr;
                auto busy = g_busy.in(0);
                const plaac_status st = plaac_fasta_next_text(fs, sp.batch_records, sp.batch_bytes, &t);
                if (st != PLAAC_OK) {
                    std::fprintf(stderr, "plaac: reading %s failed (status %d)\n", path.c_str(), (int)st);
                    failed = true;
                    break;
                }
                if (!t) break;
                b->ft = t;
            } else {
                plaac_fasta *f = nullptr;
                const plaac_status st = plaac_fasta_next(fs, sp.batch_records, sp.batch_bytes, &f);
                if (st != PLAAC_OK) {
                    std::fprintf(stderr, "plaac: reading %s failed (status %d)\n", path.c_str(), (int)st);
                    failed = true;
                    break;
                }
                if (!f) break;
                b->f = f;
                if (keeping) {
                    kept_bytes += f->nres + 16ull * f->nrec + f->name_off[f->nrec];
                    if (kept_bytes <= keep_bytes) {
                        keep->push_back(f);
                        b->owned = false;
                        b->kept = &ks;
                    } else { // too large to keep: the later pass reads the file again; what this pass is done with is
                        keeping = false; // released now, the kept batches still in flight when their Batch goes
                        keep_overflow = true;
                        ks.on_overflow();
                        q.set_cap(4);
                    }
                }
            }
            b->seq = seq++;
            prep(*b);
            q.put(std::move(b));
        }
        q.close();
        ro.set_total(seq);
    });
    // the reader is already parsing while the GPU contexts come up (HIP start-up takes a few hundred ms)
    if (!eng.ready(P0)) {
        failed = true;
        q.close();
        reader.join();
        if (fs) plaac_fasta_close(fs);
        if (drained) (*drained)();
        return false;
    }
    const int nctx = plaac_node_size(eng.node);
    ro.set_window((drained && run_ahead) ? (uint64_t)1 << 40 : (uint64_t)2 * nctx + 2);
    std::atomic<int> live_workers{nctx};
    std::vector<std::thread> workers, uploaders;
    // `upload` (text batches): an uploader thread per context takes the batches off the reader's queue, uploads and parses them
    // (plaac_text_upload) and hands them to the context's worker - the upload of batch k + 1 beside the download of batch k
    std::vector<std::unique_ptr<Queue>> uq;
    for (int k = 0; upload && k < nctx; ++k) uq.emplace_back(new Queue(1));
    for (int k = 0; upload && k < nctx; ++k)
        uploaders.emplace_back([&, k] {
            plaac_ctx *ctx = plaac_node_ctx(eng.node, k);
            for (;;) {
                BatchPtr b = q.get();
                if (!b) break;
                if (!failed) {
                    auto busy = g_busy.in(1);
                    b->st = (*upload)(ctx, *b);
                    if (b->st != PLAAC_OK) {
                        b->err = plaac_text_upload_error(ctx); // (the uploader's own slot: plaac_last_error is the scoring thread's)
                        failed = true;
                    }
                }
                uq[k]->put(std::move(b));
            }
            uq[k]->close();
        });
    for (int k = 0; k < nctx; ++k)
        workers.emplace_back([&, k] {
            plaac_ctx *ctx = plaac_node_ctx(eng.node, k);
            BatchPtr pend; // pipelined scoring: the batch begun in the previous round
            auto collect_pending = [&] { // (an end is owed for every begin, whatever has failed since)
                if (pend->begun) {
                    auto busy = g_busy.in(2);
                    const plaac_status st = finish(ctx, *pend);
                    if (st != PLAAC_OK && pend->st == PLAAC_OK) {
                        pend->st = st;
                        pend->err = plaac_last_error(ctx);
                        failed = true;
                    }
                }
                ro.put(std::move(pend));
            };
            for (;;) {
                BatchPtr b = upload ? uq[k]->get() : q.get();
                if (b) {
                    // A worker never waits for room while it holds a begun batch: that batch may be the very one the printer
                    // is waiting for (a slow worker - the first call of a context measures its streams - sits on the oldest
                    // batch while the others run the window full of finished ones).
                    if (pend && !ro.has_room(b->seq)) collect_pending();
                    ro.wait_room(b->seq);
                    if (!failed && b->st == PLAAC_OK) {
                        auto busy = g_busy.in(upload ? 9 : 1);
                        b->st = work(ctx, *b);
                        if (b->st != PLAAC_OK) {
                            b->err = plaac_last_error(ctx);
                            failed = true;
                        } else {
                            b->begun = Fin::enabled;
                        }
                    }
                }
                if (pend) collect_pending();
                if (!b) break;
                if (Fin::enabled) pend = std::move(b);
                else ro.put(std::move(b));
            }
            if (--live_workers == 0 && drained) (*drained)();
        });
    bool ok = true;
    for (;;) {
        BatchPtr b;
        {
            auto busy = g_busy.in(7);
            b = ro.take();
        }
        if (!b) break;
        if (b->st != PLAAC_OK && ok) {
            std::fprintf(stderr, "plaac: scoring failed (status %d): %s\n", (int)b->st, b->err.c_str());
            ok = false;
        }
        if (ok && !failed && !sink(*b)) {
            ok = false;
            failed = true;
        }
    }
    reader.join();
    for (auto &u : uploaders) u.join();
    for (auto &w : workers) w.join();
    if (fs) plaac_fasta_close(fs);
    if (keep && (keep_overflow || !ok || failed)) ks.release_rest(*keep);
    else if (keep) { // the batches stay for the scoring pass: from now on nobody but `keep` refers to them
        std::lock_guard<std::mutex> l(ks.m);
        ks.closed = true;
    }
    return ok && !failed;
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


// ---- pass 1: residue counts of a FASTA on the GPUs (computeaafreq :1655-1666) ----
bool count_background(Engine &eng, const plaac_params &P, const std::string &path, const Stream &sp,
                      double out[PLAAC_NAA], std::vector<plaac_fasta *> *keep, uint64_t keep_bytes) {
    for (int i = 0; i < PLAAC_NAA; ++i) out[i] = 0.0;
    std::mutex m;
    int64_t total[PLAAC_NAA] = {0};
    uint64_t nres = 0;
    plaac_fasta_stream *fs = nullptr;
    if (!open_stream(path, &fs)) return false; // a missing file: reported, zero counts (:4318-4321)
    // round 5, late: the counting pass fed with TEXT too (parsed on the device, nothing kept: the scoring pass reads the file
    // again - it is in the page cache - and parses it on the device as well). PLAAC_DEVICE_PARSE=0 / PLAAC_PIPELINE=0: the
    // host parses and keeps its batches for the scoring pass, as before.
    if (env_flag("PLAAC_DEVICE_PARSE", true) && env_flag("PLAAC_PIPELINE", true)) {
        const bool ok = run_pipeline(
            eng, P, path, fs, sp, nullptr, nullptr, 0, [](Batch &) {},
            [&](plaac_ctx *ctx, Batch &b) { return plaac_histogram_begin_text(ctx, b.ft->text, b.ft->len, b.ft->starts, b.ft->nrec); },
            [](Batch &) { return true; },
            make_finish([&](plaac_ctx *ctx, Batch &b) {
                (void)b;
                int64_t c[PLAAC_NAA];
                uint64_t r = 0;
                const plaac_status st = plaac_histogram_end_text(ctx, c, &r);
                if (st == PLAAC_OK) {
                    std::lock_guard<std::mutex> l(m);
                    for (int i = 0; i < PLAAC_NAA; ++i) total[i] += c[i];
                    nres += r;
                }
                return st;
            }),
            nullptr, true);
        for (int i = 0; i < PLAAC_NAA; ++i) out[i] = (double)total[i];
        g_timer.lap("background pass (read + H2D + parse + histogram on the device)", (double)nres, "residues");
        return ok;
    }
    const bool ok = run_pipeline(
        eng, P, path, fs, sp, nullptr, keep, keep_bytes, [](Batch &) {},
        [&](plaac_ctx *ctx, Batch &b) {
            int64_t c[PLAAC_NAA];
            const plaac_status st = plaac_histogram(ctx, b.f->codes, b.f->offsets, b.f->nrec, c);
            if (st == PLAAC_OK) {
                std::lock_guard<std::mutex> l(m);
                for (int i = 0; i < PLAAC_NAA; ++i) total[i] += c[i];
                nres += b.f->nres;
            }
            return st;
        },
        [](Batch &) { return true; });
    for (int i = 0; i < PLAAC_NAA; ++i) out[i] = (double)total[i];
    g_timer.lap("background pass (read + H2D + histogram)", (double)nres, "residues");
    return ok;
}

// The single pass (round 5; alpha = 1, background counted from the scored input): what the reference prints before the first
// table row is held back in `held` (put() collects while g_defer points at it); `block_at` is where the parameter block goes
// once the counts are final, `make_block(counts, text)` formats it (and checks that the tables the batches were scored
// with are the tables of those counts).
struct SinglePass {
    std::string held;
    size_t block_at = 0;
    bool want_block = false;
    std::function<bool(const double counts[PLAAC_NAA], std::string &text)> make_block;
};

// Is stdout a plain file this process may write at offsets of its choosing (not a pipe / terminal, not opened for appending)?
// *at = where the next byte goes. The single pass then writes the table where it belongs while the parameter block in
// front of it still waits for the counts (PLAAC_PLACED_WRITE=0: hold everything back, as for a pipe).
bool stdout_is_plain_file(off_t *at) {
    struct stat sb;
    if (::fstat(STDOUT_FILENO, &sb) != 0 || !S_ISREG(sb.st_mode)) return false;
    const int fl = ::fcntl(STDOUT_FILENO, F_GETFL);
    if (fl < 0 || (fl & O_APPEND)) return false;
    const off_t o = ::lseek(STDOUT_FILENO, 0, SEEK_CUR);
    if (o < 0) return false;
    *at = o;
    return true;
}
bool pwrite_all(int fd, const char *p, size_t n, off_t at) {
    while (n) {
        const ssize_t k = ::pwrite(fd, p, n, at);
        if (k <= 0) return false;
        p += k;
        n -= (size_t)k;
        at += k;
    }
    return true;
}
// moves the bytes of stdout's file from `from` to its end by `delta` (the parameter block came out longer or shorter than its
// placeholder: an input without a single valid residue prints NaN frequencies). stdout is write-only as a rule, so the file
// is opened a second time through /proc.
bool shift_tail(off_t from, off_t end, long delta) {
    if (delta == 0 || end <= from) return true;
    const int fd = ::open("/proc/self/fd/1", O_RDWR);
    if (fd < 0) return false;
    std::vector<char> buf(8u << 20);
    bool ok = true;
    if (delta > 0) { // towards the end: last chunk first
        for (off_t hi = end; ok && hi > from;) {
            const off_t lo = std::max<off_t>(from, hi - (off_t)buf.size());
            ok = ::pread(fd, buf.data(), (size_t)(hi - lo), lo) == (ssize_t)(hi - lo) && pwrite_all(fd, buf.data(), (size_t)(hi - lo), lo + delta);
            hi = lo;
        }
    } else {
        for (off_t lo = from; ok && lo < end;) {
            const off_t hi = std::min<off_t>(end, lo + (off_t)buf.size());
            ok = ::pread(fd, buf.data(), (size_t)(hi - lo), lo) == (ssize_t)(hi - lo) && pwrite_all(fd, buf.data(), (size_t)(hi - lo), lo + delta);
            lo = hi;
        }
        if (ok) ok = ::ftruncate(fd, end + delta) == 0;
    }
    ::close(fd);
    return ok;
}

// ---- pass 2, summary mode (scoreallfastas :653-950) ----
bool score_all(Engine &eng, const plaac_params &P, const Options &o, const Stream &sp, std::vector<plaac_fasta *> *replay,
               SinglePass *single = nullptr) {
    if (o.headers) column_notes();
    put(std::string(plaac_summary_header()) + "\n");
    plaac_fasta_stream *fs = nullptr; // the reference opens the file after it has printed the header (:715-750)
    if (!replay && !open_stream(o.input, &fs)) return false;
    g_defer = nullptr; // (single pass: everything in front of the table is in single->held now)
    const unsigned nt_max = plaac_host_threads();
    std::unique_ptr<ThreadTeam> team; // (made when the first large batch arrives; PLAAC_FORMAT_TEAM=0: a thread per range and batch)
    const bool use_team = env_flag("PLAAC_FORMAT_TEAM", true);
    uint64_t nres = 0, nrec = 0;
    std::fflush(stdout);
    // single pass into a plain file: the text in front of the table goes out now, with a block of the right length computed
    // from made-up counts in the block's place; the real block is written over it once the counts are final
    bool placed = false;
    off_t placed_at = 0;
    size_t placeholder_len = 0;
    std::string final_block;
    if (single && env_flag("PLAAC_PLACED_WRITE", true) && stdout_is_plain_file(&placed_at)) {
        placed = true;
        std::string blk0;
        if (single->want_block) {
            double ones[PLAAC_NAA];
            for (double &v : ones) v = 1.0;
            (void)single->make_block(ones, blk0);
            // Only the block's LENGTH is wanted: what stands in its place until the counts are final must not pass for output
            // if the run dies first (a block of made-up frequencies would) - comment lines of '#' with the reason up front.
            const char note[] = "## INCOMPLETE RUN: the parameter block is written here when the run ends ";
            size_t at = 0;
            for (char &ch : blk0) {
                if (ch == '\n') continue;
                ch = at < sizeof note - 1 ? note[at] : '#';
                ++at;
            }
        }
        placeholder_len = blk0.size();
        placed_at += (off_t)single->block_at;
        single->held.insert(single->block_at, blk0);
    }
    Writer writer(single != nullptr && !placed); // from here on the table goes through the writer thread
    if (placed) writer.write(single->held);
    const bool pipelined = env_flag("PLAAC_PIPELINE", true) || single;
    std::mutex cm;
    int64_t total_counts[PLAAC_NAA] = {0};
    std::atomic<bool> block_failed{false};
    // K1: the single pass hands the device the file's text (PLAAC_DEVICE_PARSE=0: the host parses, as in the two passes)
    // (round 5, late: also every other pipelined scoring pass that reads the file itself - background from -B / -b another file,
    //  or an input too large to keep: the web application's own invocation is "-B bg_freqs_X.txt -a 0.5", web/lib/server.rb:152-155)
    const bool counting = single != nullptr;
    const bool as_text = pipelined && !replay && env_flag("PLAAC_DEVICE_PARSE", true);
    int prev_blank = 1; // (the sink's: how the record before the batch it is looking at ended)
    // ... and the rows as TEXT from the device as well (plaac_score_end_text_table; the name of a batch's first record is trimmed
    // or not by how the batch before it ended, which the reader finds out itself: any context, any order). PLAAC_DEVICE_FORMAT=0: host.
    const bool device_format = as_text && env_flag("PLAAC_DEVICE_FORMAT", true);
    auto collect = make_finish([&](plaac_ctx *ctx, Batch &b) {
        if (!b.ft && !counting) return plaac_score_end(ctx, b.rows.data());
        int64_t c[PLAAC_NAA] = {0};
        int64_t *const cp = counting ? c : nullptr;
        plaac_status st;
        if (b.ft && device_format) {
            // the rows as text, made on the device; a batch it will not vouch for (a value of 1e9 or more, an infinity, a record
            // without a sequence) comes back the old way and is formatted below
            uint64_t bytes = 0;
            int needs_host = 0, lastb = 0;
            st = plaac_score_end_text_table_size(ctx, o.corelength, o.ww2, b.ft->prev_blank, &bytes, &needs_host, &lastb, &b.table_residues);
            if (st != PLAAC_OK) return st;
            b.last_blank = lastb; // (== b.ft->last_blank: the reader's own look at the batch's last record)
            if (!needs_host) {
                b.table = writer.buffer((size_t)bytes + 1);
                st = plaac_score_end_text_table(ctx, b.table.p, b.table.cap, cp);
                b.table.n = (size_t)bytes;
                b.have_table = st == PLAAC_OK;
                if (st == PLAAC_OK) {
                    std::lock_guard<std::mutex> l(cm);
                    for (int i = 0; i < PLAAC_NAA; ++i) total_counts[i] += c[i];
                }
                return st;
            }
            b.rows.resize(b.ft->nrec);
        }
        if (b.ft) {
            const uint32_t n = b.ft->nrec;
            b.toffs.resize((size_t)n + 1);
            b.tblank.resize((size_t)n + 1);
            b.text_ext.resize(2 * (size_t)n + 2);
            // (no copy of the codes: a summary row prints a few residues, read from the text - plaac_fasta_text_codes)
            st = plaac_score_end_text(ctx, b.rows.data(), nullptr, 0, b.toffs.data(), b.tblank.data(), b.text_ext.data(), cp);
            b.view.nrec = n;
            b.view.nres = b.toffs[n];
            b.view.codes = nullptr;
            b.view.offsets = b.toffs.data();
            b.view.names = nullptr; // (a text batch's names lie in its text: name_of below)
            b.view.name_off = nullptr;
            b.f = &b.view;
            b.owned = false;
        } else {
            st = plaac_score_end_counts(ctx, b.rows.data(), c);
        }
        if (st == PLAAC_OK) {
            std::lock_guard<std::mutex> l(cm);
            for (int i = 0; i < PLAAC_NAA; ++i) total_counts[i] += c[i];
        }
        return st;
    });
    // every batch of the input has been counted (the formatter is still at work): the parameter block can be written, the
    // table behind it released
    const std::function<void()> drained = [&] {
        if (!single) return;
        std::string block;
        double cd[PLAAC_NAA];
        {
            std::lock_guard<std::mutex> l(cm);
            for (int i = 0; i < PLAAC_NAA; ++i) cd[i] = (double)total_counts[i];
        }
        if (single->want_block && !single->make_block(cd, block)) block_failed = true;
        g_timer.lap("single pass: input read, counted, scored");
        if (placed) {
            final_block = std::move(block); // (written over its placeholder when the table is out)
            return;
        }
        single->held.insert(single->block_at, block);
        writer.release(single->held);
    };
    const std::function<plaac_status(plaac_ctx *, Batch &)> upload = [&](plaac_ctx *ctx, Batch &b) {
        return plaac_text_upload(ctx, b.ft->text, b.ft->len, b.ft->starts, b.ft->nrec, &b.tb);
    };
    auto run = [&](auto &&...a) {
        return pipelined ? run_pipeline(std::forward<decltype(a)>(a)..., collect, single ? &drained : nullptr, as_text, !placed,
                                        (as_text && env_flag("PLAAC_UPLOAD_THREAD", true)) ? &upload : nullptr)
                         : run_pipeline(std::forward<decltype(a)>(a)...);
    };
    const bool ok = run(
        eng, P, o.input, fs, sp, replay, (std::vector<plaac_fasta *> *)nullptr, (uint64_t)0, [](Batch &) {},
        [&](plaac_ctx *ctx, Batch &b) {
            if (b.ft) {
                if (!device_format) b.rows.resize(b.ft->nrec);
                if (b.tb) { // (uploaded and parsed by the context's uploader thread already)
                    plaac_text_batch *t = b.tb;
                    b.tb = nullptr;
                    return plaac_score_begin_uploaded(ctx, t, counting ? 1 : 0);
                }
                return plaac_score_begin_text(ctx, b.ft->text, b.ft->len, b.ft->starts, b.ft->nrec, counting ? 1 : 0);
            }
            b.rows.resize(b.f->nrec);
            if (single) return plaac_score_begin_counting(ctx, b.f->codes, b.f->offsets, b.f->nrec);
            if (pipelined) return plaac_score_begin(ctx, b.f->codes, b.f->offsets, b.f->nrec);
            return plaac_score(ctx, b.f->codes, b.f->offsets, b.f->nrec, b.rows.data(), nullptr);
        },
        [&](Batch &b) {
            if (b.have_table) { // (formatted on the device)
                if (b.last_blank != b.ft->last_blank) {
                    std::fprintf(stderr, "plaac: the reader and the device disagree about how a record ends - rerun with PLAAC_DEVICE_FORMAT=0\n");
                    return false;
                }
                prev_blank = b.last_blank;
                nres += b.table_residues;
                nrec += b.ft->nrec;
                auto busy = g_busy.in(4);
                writer.write(b.table);
                b.table = TextBuf{};
                return true;
            }
            if (b.ft) prev_blank = plaac_fasta_text_trim_names(b.ft, b.tblank.data(), b.ft->prev_blank);
            const plaac_fasta *f = b.f;
            // format in parallel (contiguous row ranges per thread), print in file order
            const unsigned nt = f->nrec < 2048 ? 1u : nt_max;
            std::vector<TextBuf> part(nt);
            std::vector<int> bad(nt, 0);
            auto fmt = [&](unsigned t) {
                auto busy_thread = g_busy.in(8);
                struct CpuClock { // (thread CPU time beside the wall time: the difference is time the thread was not running)
                    timespec t0;
                    CpuClock() { clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t0); }
                    ~CpuClock() {
                        timespec t1;
                        clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t1);
                        g_busy.cpu_fmt += (t1.tv_sec - t0.tv_sec) * 1000000000ll + (t1.tv_nsec - t0.tv_nsec);
                    }
                } cpu_clock;
                const uint32_t r0 = (uint32_t)((uint64_t)f->nrec * t / nt), r1 = (uint32_t)((uint64_t)f->nrec * (t + 1) / nt);
                std::vector<uint8_t> some; // device-parsed batch: the record's codes where the row prints them
                // (one buffer per thread and batch, from the writer's pool: ~230 bytes of numbers per row + its name, sequences
                //  are printed for the few records with a PrD; it grows if that was too little)
                auto name_of = [&](uint32_t i, size_t &nl) -> const char * {
                    if (b.ft) {
                        nl = b.ft->name_len[i];
                        return b.ft->text + b.ft->starts[i] + 1;
                    }
                    const char *nm = rec_name(f, i);
                    nl = std::strlen(nm);
                    return nm;
                };
                const size_t names_bytes = b.ft ? (size_t)(r1 - r0) * 24 : (size_t)(f->name_off[r1] - f->name_off[r0]);
                part[t] = writer.buffer((size_t)(r1 - r0) * 260 + names_bytes + (size_t)(f->offsets[r1] - f->offsets[r0]) / 8 + 16384);
                TextBuf &out = part[t];
                for (uint32_t i = r0; i < r1; ++i) {
                    const uint64_t len = f->offsets[i + 1] - f->offsets[i];
                    size_t nl = 0;
                    const char *nm = name_of(i, nl);
                    if (len == 0) {
                        std::fprintf(stderr, "plaac: record '%.*s' has no sequence, skipped\n", (int)nl, nm);
                        continue;
                    }
                    // (the row goes straight into the part's buffer: room for the longest a row can be)
                    const size_t need = nl + len * 3 + (size_t)o.ww2 + 8800;
                    if (out.n + need > out.cap) out.grow(out.n + need);
                    const uint8_t *rec_codes;
                    if (b.ft) {
                        // The residues a row prints (plaac_format_summary_row: the PAPA window; core, PrD and its two ends when
                        // the PrD is long enough; every range clamped into the protein): one stretch of the record, decoded
                        // from the text into its place
                        const plaac_row &r = b.rows[i];
                        const long n = r.prot_len;
                        if (some.size() < len) some.resize(len);
                        if (n > 0) {
                            long lo = n - 1, hi = 0;
                            auto want = [&](long v) {
                                v = std::min(std::max(v, 0l), n - 1);
                                lo = std::min(lo, v);
                                hi = std::max(hi, v);
                            };
                            want((long)r.papa_cen - o.ww2 / 2);
                            want((long)r.papa_cen + o.ww2 / 2);
                            if (r.prd_end - r.prd_start + 1 >= o.corelength)
                                for (long v : {(long)r.core_start, (long)r.core_end, (long)r.prd_start, (long)r.prd_start + 14,
                                               (long)r.prd_end - 14, (long)r.prd_end})
                                    want(v);
                            (void)plaac_fasta_text_codes(b.ft->text, b.ft->starts, b.text_ext.data(), i, (uint64_t)lo,
                                                         (uint64_t)(hi - lo + 1), some.data() + lo);
                        }
                        rec_codes = some.data();
                    } else {
                        rec_codes = f->codes + f->offsets[i];
                    }
                    const long k = plaac_format_summary_row_n(&b.rows[i], nm, nl, rec_codes, len, o.corelength, o.ww2, out.p + out.n,
                                                              out.cap - out.n - 1);
                    if (k < 0) {
                        bad[t] = 1;
                        return;
                    }
                    if (k == 0) continue; // nothing left after the stop trim (:762)
                    out.n += (size_t)k;
                    out.p[out.n++] = '\n';
                }
            };
            {
                auto busy = g_busy.in(3);
                if (nt == 1) {
                    fmt(0);
                } else if (use_team) {
                    if (!team) team.reset(new ThreadTeam(nt));
                    const std::function<void(unsigned)> job = fmt;
                    team->run(job);
                } else {
                    std::vector<std::thread> pool;
                    for (unsigned t = 0; t < nt; ++t) pool.emplace_back(fmt, t);
                    for (auto &th : pool) th.join();
                }
            }
            for (unsigned t = 0; t < nt; ++t)
                if (bad[t]) return false;
            {
                auto busy = g_busy.in(4);
                for (unsigned t = 0; t < nt; ++t) writer.write(part[t]);
            }
            nres += f->nres;
            nrec += f->nrec;
            return true;
        });
    if (single && writer.is_held()) drained(); // (a failed pipeline never drained: what was held still goes out)
    if (!writer.finish()) {
        std::fprintf(stderr, "plaac: writing the table to stdout failed (disk full / closed pipe?): output is incomplete\n");
        return false;
    }
    if (placed && single->want_block && ok) {
        const long delta = (long)final_block.size() - (long)placeholder_len;
        const off_t end = ::lseek(STDOUT_FILENO, 0, SEEK_CUR);
        bool done = end >= 0 && shift_tail(placed_at + (off_t)placeholder_len, end, delta) &&
                    pwrite_all(STDOUT_FILENO, final_block.data(), final_block.size(), placed_at);
        if (done && delta != 0) done = ::lseek(STDOUT_FILENO, end + delta, SEEK_SET) >= 0;
        if (!done) {
            std::fprintf(stderr, "plaac: could not put the parameter block in its place in the output file - rerun with PLAAC_PLACED_WRITE=0\n");
            return false;
        }
    }
    if (block_failed) {
        std::fprintf(stderr, "plaac: the scoring tables depend on the input's residue counts after all - rerun with PLAAC_SINGLE_PASS=0\n");
        return false;
    }
    g_timer.lap("scoring pass (read + H2D + GPU + D2H + format + write)", (double)nres, "residues");
    g_busy.report();
    return ok;
}

// ---- pass 2, track mode (plotsomefastas :587-649) ----
bool plot_some(Engine &eng, const plaac_params &P, const Options &o, const Stream &sp, std::vector<plaac_fasta *> *replay) {
    std::map<std::string, std::string> title, order;
    const bool all = o.plotlist == "all";
    if (!all && !read_plot_list(o.plotlist, title, order)) return false;
    plaac_fasta_stream *fs = nullptr;
    if (!replay && !open_stream(o.input, &fs)) return false;
    put(std::string(plaac_tracks_header()) + "\n");
    std::fflush(stdout);
    const unsigned nt_max = plaac_host_threads();
    std::unique_ptr<ThreadTeam> team;
    Writer writer; // (from here on the table goes through the writer thread)
    const bool device_table = env_flag("PLAAC_DEVICE_FORMAT", true);
    int genecount = 1; // advanced by the reader thread only (records are selected in file order, :617)
    const bool ok = run_pipeline(
        eng, P, o.input, fs, sp, replay, nullptr, 0,
        [&](Batch &b) { // select records and build the sub-batch that is scored
            const plaac_fasta *f = b.f;
            for (uint32_t i = 0; i < f->nrec; ++i) {
                const std::string name = rec_name(f, i);
                if (!(all || title.count(name) || title.count(">" + name))) continue;
                const uint64_t bb = f->offsets[i], e = f->offsets[i + 1];
                uint64_t n = e - bb;
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
                b.pick.push_back(i);
                b.ids.push_back(id);
                b.names.push_back(nm);
            }
            b.poffs.assign(b.pick.size() + 1, 0);
            for (size_t k = 0; k < b.pick.size(); ++k)
                b.poffs[k + 1] = b.poffs[k] + (f->offsets[b.pick[k] + 1] - f->offsets[b.pick[k]]);
            b.pcodes.resize(b.poffs.back() + 64);
            for (size_t k = 0; k < b.pick.size(); ++k)
                std::memcpy(b.pcodes.data() + b.poffs[k], f->codes + f->offsets[b.pick[k]], b.poffs[k + 1] - b.poffs[k]);
        },
        [&](plaac_ctx *ctx, Batch &b) {
            if (b.pick.empty()) return PLAAC_OK;
            const uint64_t total = b.poffs.back();
            b.rows.resize(b.pick.size());
            if (device_table && total < 0xfffffff0ull) {
                // the batch's lines from the device (plaac_score_tracks_table): the twelve track arrays never cross PCIe, the
                // text does; a batch with a value the device will not vouch for takes the old way below
                std::string labels;
                std::vector<uint64_t> loff(b.pick.size() + 1, 0);
                for (size_t k = 0; k < b.pick.size(); ++k) {
                    labels += b.ids[k];
                    labels += '\t';
                    labels += b.names[k];
                    loff[k + 1] = labels.size();
                }
                char *table = nullptr;
                uint64_t len = 0;
                int needs_host = 0;
                const plaac_status st = plaac_score_tracks_table(ctx, b.pcodes.data(), b.poffs.data(), (uint32_t)b.pick.size(), labels.data(),
                                                                 loff.data(), b.rows.data(), &table, &len, &needs_host);
                if (st != PLAAC_OK) return st;
                if (!needs_host) {
                    b.table = TextBuf{table, (size_t)len, (size_t)len, true};
                    b.have_table = true;
                    return PLAAC_OK;
                }
            }
            b.t8.resize(2 * total + 2);
            b.t64.resize(10 * total + 10);
            double *d = b.t64.data();
            b.tr = plaac_tracks{b.t8.data(), b.t8.data() + total, d,           d + total,     d + 2 * total, d + 3 * total,
                                d + 4 * total, d + 5 * total,   d + 6 * total, d + 7 * total, d + 8 * total, d + 9 * total};
            return plaac_score(ctx, b.pcodes.data(), b.poffs.data(), (uint32_t)b.pick.size(), b.rows.data(), &b.tr);
        },
        [&](Batch &b) {
            // The proteins of a batch formatted side by side (round 5, late: one thread formatted 112 bytes per residue at 6 M
            // residues/s - 4.7 s of a 5.1 s run over 100,000 sequences): contiguous ranges of proteins with about the same
            // number of residues per thread, each into a buffer of its own; the writer thread prints them in file order.
            const size_t K = b.pick.size();
            if (K == 0) return true;
            if (b.have_table) { // (formatted on the device)
                writer.write(b.table);
                b.table = TextBuf{};
                return true;
            }
            const uint64_t total = b.poffs[K];
            const unsigned nt = total < 65536 ? 1u : nt_max;
            std::vector<size_t> cut(nt + 1, K);
            cut[0] = 0;
            for (unsigned t = 1; t < nt; ++t) // first protein whose offset reaches the t-th share of the residues
                cut[t] = (size_t)(std::lower_bound(b.poffs.begin(), b.poffs.begin() + K, total * t / nt) - b.poffs.begin());
            std::vector<TextBuf> part(nt);
            std::vector<int> bad(nt, 0);
            auto fmt = [&](unsigned t) {
                // (a line is the two labels + ~110 bytes of numbers; plaac_track_rows_bound - 400 per residue - is what a line can
                //  be at most: the buffer starts at the usual size and grows to the bound only if a protein asks for it)
                size_t need = 4096;
                for (size_t k = cut[t]; k < cut[t + 1]; ++k)
                    need += (size_t)b.rows[k].prot_len * (b.ids[k].size() + b.names[k].size() + 130) + 64;
                part[t] = writer.buffer(need);
                TextBuf &out = part[t];
                for (size_t k = cut[t]; k < cut[t + 1]; ++k) {
                    const uint32_t n = (uint32_t)b.rows[k].prot_len;
                    long len = plaac_format_track_rows(&b.tr, b.poffs[k], b.pcodes.data() + b.poffs[k], n, b.ids[k].c_str(),
                                                       b.names[k].c_str(), out.p + out.n, out.cap - out.n);
                    if (len < 0) {
                        out.grow(out.n + plaac_track_rows_bound(n, b.ids[k].size(), b.names[k].size()));
                        len = plaac_format_track_rows(&b.tr, b.poffs[k], b.pcodes.data() + b.poffs[k], n, b.ids[k].c_str(),
                                                      b.names[k].c_str(), out.p + out.n, out.cap - out.n);
                    }
                    if (len < 0) {
                        bad[t] = 1;
                        return;
                    }
                    out.n += (size_t)len;
                }
            };
            if (nt == 1) {
                fmt(0);
            } else {
                if (!team) team.reset(new ThreadTeam(nt));
                const std::function<void(unsigned)> job = fmt;
                team->run(job);
            }
            for (unsigned t = 0; t < nt; ++t)
                if (bad[t]) return false;
            for (unsigned t = 0; t < nt; ++t) writer.write(part[t]);
            return true;
        });
    if (!writer.finish()) {
        std::fprintf(stderr, "plaac: writing the tracks to stdout failed (disk full / closed pipe?): output is incomplete\n");
        return false;
    }
    if (std::fflush(stdout) != 0 || std::ferror(stdout)) {
        std::fprintf(stderr, "plaac: writing the tracks to stdout failed (disk full / closed pipe?): output is incomplete\n");
        return false;
    }
    g_timer.lap("track pass");
    return ok;
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
    Engine eng;
    const bool need_gpu = !o.input.empty() || (!o.bgfile.empty() && o.bgfreq.empty());
    // One GPU per PLAAC_BYTES_PER_DEVICE of input (default 4 GiB; PLAAC_DEVICES names the devices itself): the kernels of a
    // 3 GB FASTA take 0.14 s of one MI355X, the host's share of that run 1.2 s - seven more contexts only add their start-up
    // and teardown (profiles/r05_e2e_device_parse.txt: 1.9 - 2.1 s with eight contexts against 1.25 - 1.35 s with one).
    int worth = 1;
    {
        uint64_t in_bytes = 0;
        struct stat sb;
        for (const std::string *f : {&o.input, &o.bgfile})
            if (!f->empty() && ::stat(f->c_str(), &sb) == 0 && S_ISREG(sb.st_mode)) in_bytes = std::max<uint64_t>(in_bytes, (uint64_t)sb.st_size);
        const uint64_t per = std::max<uint64_t>(1, env_u64("PLAAC_BYTES_PER_DEVICE", 4ull << 30));
        worth = (int)std::min<uint64_t>(1u << 20, std::max<uint64_t>(1, (in_bytes + per - 1) / per));
    }
    if (need_gpu) eng.start(P, worth); // HIP start-up runs beside the rest of the set-up
    // track mode moves 82 bytes per residue through the host: smaller batches
    Stream sp{(uint32_t)env_u64("PLAAC_BATCH_RECORDS", 262144),
              env_u64("PLAAC_BATCH_BYTES", o.plotlist.empty() ? (96ull << 20) : (8ull << 20))};
    const uint64_t keep_bytes = env_u64("PLAAC_KEEP_BYTES", default_keep_bytes());

    // ONE pass instead of the reference's two (counting pass :377-384, scoring pass :755)? Only when the tables the batches
    // are scored with cannot depend on the counts: alpha = 1 after its clamp (:444-447, :458 - the input's background then
    // only shows in the "## bg_input" line), background = the scored input itself, summary mode, an input that can be opened
    // and whose formatted table may wait in memory (the budget of the kept batches of the two-pass run)
    SinglePass single;
    bool single_pass = false;
    {
        const double a_eff = (o.alpha > 1 || o.alpha < 0) ? 1.0 : o.alpha;
        struct stat sb;
        single_pass = env_flag("PLAAC_SINGLE_PASS", true) && env_flag("PLAAC_PIPELINE", true) && !o.input.empty() &&
                      o.plotlist.empty() && o.bgfreq.empty() && (o.bgfile.empty() || o.bgfile == o.input) && a_eff == 1.0 &&
                      ::stat(o.input.c_str(), &sb) == 0 && S_ISREG(sb.st_mode) && ::access(o.input.c_str(), R_OK) == 0;
        // An input of any size when the table can be written in place (nothing waits in memory then); otherwise one whose
        // table may wait: the budget of the kept batches of the two-pass run
        off_t at = 0;
        if (single_pass && (uint64_t)sb.st_size > keep_bytes)
            single_pass = env_flag("PLAAC_PLACED_WRITE", true) && stdout_is_plain_file(&at);
    }
    if (single_pass) g_defer = &single.held;

    // background counts (:377-384)
    double bgf[PLAAC_NAA] = {0}, fgf[PLAAC_NAA];
    std::vector<plaac_fasta *> kept; // parsed batches of the input, when the background pass read it
    bool kept_valid = false, ok = true;
    if (single_pass) {
        // (counted inside the scoring pass)
    } else if (!o.bgfreq.empty()) {
        ok = read_params_file(o.bgfreq, bgf);
    } else if (!o.bgfile.empty()) {
        const bool same = o.bgfile == o.input;
        ok = count_background(eng, P, o.bgfile, sp, bgf, same ? &kept : nullptr, keep_bytes);
        kept_valid = same && !kept.empty();
    } else if (!o.input.empty()) {
        ok = count_background(eng, P, o.input, sp, bgf, &kept, keep_bytes);
        kept_valid = !kept.empty();
    }
    if (!ok) return 1;
    const bool have_fg = !o.fgfreq.empty();
    if (have_fg && !read_params_file(o.fgfreq, fgf)) return 1;

    char text[8192];
    if ((!o.bgfile.empty() || !o.bgfreq.empty()) && o.input.empty()) { // (:394-403): dump and exit
        plaac_format_aa_params(bgf, text, sizeof text);
        put(text);
        std::fflush(stdout);
        if (eng.started && eng.ready(P)) plaac_node_destroy(eng.node);
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
    if (o.params && single_pass) { // the block needs the final counts: its place is kept, its text comes with them
        single.want_block = true;
        single.block_at = single.held.size();
        const plaac_params Pscored = P;
        const Options oc = o;
        const bool fg_given = have_fg;
        std::vector<double> fgv(fgf, fgf + PLAAC_NAA);
        single.make_block = [Pscored, oc, fg_given, fgv](const double counts[PLAAC_NAA], std::string &out) {
            plaac_params P2;
            if (plaac_params_init(&P2, fg_given ? fgv.data() : nullptr, counts, oc.alpha, oc.corelength, oc.ww1, oc.ww2, oc.ww3, 1) != PLAAC_OK)
                return false;
            char t2[8192];
            plaac_format_param_block(&P2, t2, sizeof t2);
            out = t2;
            plaac_params Pc = Pscored; // identical but for the input's own frequencies?
            std::memcpy(Pc.bgthis, P2.bgthis, sizeof Pc.bgthis);
            return std::memcmp(&Pc, &P2, sizeof P2) == 0;
        };
    } else if (o.params) {
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

    if (!eng.ready(P)) return 1;
    if (plaac_node_set_params(eng.node, &P) != PLAAC_OK) {
        std::fprintf(stderr, "plaac: plaac_node_set_params failed: %s\n", plaac_node_last_error(eng.node));
        return 1;
    }
    ok = o.plotlist.empty() ? score_all(eng, P, o, sp, kept_valid ? &kept : nullptr, single_pass ? &single : nullptr)
                            : plot_some(eng, P, o, sp, kept_valid ? &kept : nullptr);
    // The output is complete and flushed; contexts, batches and the HIP runtime are torn down in order (≈55 ms with
    // two contexts on an MI355X). PLAAC_FAST_EXIT=1 leaves through _exit instead and lets the operating system reclaim
    // everything.
    std::fflush(stdout);
    g_timer.mark("output complete");
    std::fflush(stderr);
    if (std::getenv("PLAAC_FAST_EXIT")) ::_exit(ok ? 0 : 1);
    plaac_node_destroy(eng.node);
    g_timer.lap("teardown: GPU contexts");
    g_timer.mark("contexts destroyed");
    // The parsed input kept for the scoring pass (up to PLAAC_KEEP_BYTES) is handed back by process exit: unmapping
    // gigabytes piecewise costs 0.3 s that the exit path does not have to pay (PLAAC_TEARDOWN=1: free it, for leak checkers)
    if (std::getenv("PLAAC_TEARDOWN")) {
        for (plaac_fasta *k : kept) plaac_fasta_free(k);
        g_timer.lap("teardown: parsed batches");
    }
    return ok ? 0 : 1;
}
