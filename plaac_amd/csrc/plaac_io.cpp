// plaac_io.cpp — host text I/O of libplaac_native.so (see include/plaac_host.h). No device code.
//
// Reference behaviour reproduced here (cli/src/plaac.java):
//   fastareader :4302-4375 — header = line starting with '>'; sequence lines are concatenated until a
//       blank line or the next header; after a blank line the rest of the record is skipped; the name of the
//       first record (and of any record found after a blank line) is trimmed, names met while reading a
//       sequence are not; line content is never trimmed (blanks become X).
//   java.util.Formatter %.Nf — FormattedFloatingDecimal: take the SHORTEST decimal digit string that
//       identifies the double (what Double.toString prints), then round HALF_UP at the requested place
//       (so 0.0045 -> "0.005" although the double is 0.004499999...; C printf gives "0.004").
#include "plaac_host.h"

#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <array>
#include <string>
#include <thread>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

namespace {

const char kAlphabet[] = "XACDEFGHIKLMNPQRSTVWY*"; // aanames (:26)

// shortest round-trip digits of |v| (finite, non-zero): value = 0.d1d2...dn x 10^decexp
struct Digits {
    std::string d;
    int decexp;
};

Digits shortest_digits(double av) {
    char tmp[64];
    auto r = std::to_chars(tmp, tmp + sizeof tmp, av, std::chars_format::scientific);
    std::string s(tmp, r.ptr);
    const size_t epos = s.find('e');
    std::string mant = s.substr(0, epos);
    const int e10 = std::atoi(s.c_str() + epos + 1);
    Digits out;
    for (char c : mant)
        if (c != '.') out.d.push_back(c);
    while (out.d.size() > 1 && out.d.back() == '0') out.d.pop_back();
    out.decexp = e10 + 1;
    return out;
}

// FormattedFloatingDecimal.applyPrecision: keep `keep` leading digits, HALF_UP on the next one
void round_half_up(Digits &g, int keep) {
    const int n = (int)g.d.size();
    if (keep >= n || keep < 0) return;
    if (keep == 0) {
        if (g.d[0] >= '5') {
            g.d = "1";
            g.decexp += 1;
        } else {
            g.d = "0";
        }
        return;
    }
    const bool up = g.d[keep] >= '5';
    g.d.resize(keep);
    if (!up) return;
    int i = keep - 1;
    while (i >= 0 && g.d[i] == '9') g.d[i--] = '0';
    if (i >= 0) {
        g.d[i]++;
    } else {
        g.d.insert(g.d.begin(), '1');
        g.d.resize(keep);
        g.decexp += 1;
    }
}

std::string fixed_string(double v, int decimals) {
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v > 0 ? "Infinity" : "-Infinity";
    std::string out;
    if (std::signbit(v)) out.push_back('-');
    const double av = std::fabs(v);
    Digits g{"0", 1};
    if (av != 0.0) {
        g = shortest_digits(av);
        round_half_up(g, g.decexp + decimals);
    }
    // integer part
    if (g.decexp <= 0 || g.d == "0") {
        out.push_back('0');
    } else {
        for (int i = 0; i < g.decexp; ++i) out.push_back(i < (int)g.d.size() ? g.d[i] : '0');
    }
    if (decimals > 0) {
        out.push_back('.');
        for (int k = 0; k < decimals; ++k) {
            const int idx = g.decexp + k; // digit index of the k-th decimal
            char c = '0';
            if (g.d != "0" && idx >= 0 && idx < (int)g.d.size()) c = g.d[idx];
            out.push_back(c);
        }
    }
    return out;
}

// %.Nf of java.util.Formatter appended to `s` without the digit strings of fixed_string. Scaled by 10^decimals, a value
// whose fraction is not within `err` of one half rounds the same way whether one looks at the double's exact expansion,
// at its shortest round-trip digits (what the Formatter does) or at the rounded product y: the three differ by less
// than y * 2^-51. Everything else (ties and near-ties, huge values, NaN / infinities) takes fixed_string.
void append_fixed(std::string &s, double v, int decimals) {
    static const double p10[10] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9};
    const double av = std::fabs(v);
    if (decimals >= 0 && decimals <= 9 && av < 1e9) { // (false for NaN)
        const double y = av * p10[decimals];
        const double fl = std::floor(y), f = y - fl, err = y * 0x1p-51 + 0x1p-60;
        const bool clear = std::fabs(f - 0.5) > err;
        if (err < 0.25 && (clear || y < 0x1p51)) {
            unsigned long long r = (unsigned long long)fl;
            // Within the noise of the tie t = (k + 1/2) / 10^decimals (means of small integers sit there all the time): the
            // Formatter rounds the SHORTEST digits D that read back as v. c = the double nearest t (a correctly rounded
            // quotient of two integers). v == c: D is t itself (no decimal with fewer places lies within half an ulp of c, and
            // t reads back as c) - HALF_UP goes up. v != c: D lies in v's own rounding interval, which ends before (begins
            // after) c's, and t lies in c's - D is on v's side of t. Either way: up iff v >= c. (The device's formatter,
            // format_device.hip.inc, takes the same step; the differential test against the digit-string path covers it.)
            r += clear ? (f > 0.5 ? 1ull : 0ull) : (av >= (double)(2ull * r + 1ull) / (2.0 * p10[decimals]) ? 1ull : 0ull);
            char tmp[40];
            int k = 40;
            for (int i = 0; i < decimals; ++i) {
                tmp[--k] = (char)('0' + r % 10);
                r /= 10;
            }
            if (decimals > 0) tmp[--k] = '.';
            do {
                tmp[--k] = (char)('0' + r % 10);
                r /= 10;
            } while (r);
            if (std::signbit(v)) tmp[--k] = '-';
            s.append(tmp + k, (size_t)(40 - k));
            return;
        }
    }
    s += fixed_string(v, decimals);
}

void append_int(std::string &s, long v) {
    char tmp[24];
    auto r = std::to_chars(tmp, tmp + sizeof tmp, v);
    s.append(tmp, (size_t)(r.ptr - tmp));
}

// java.lang.Double.toString
std::string java_double_tostring(double v) {
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v > 0 ? "Infinity" : "-Infinity";
    std::string out;
    if (std::signbit(v)) out.push_back('-');
    const double av = std::fabs(v);
    if (av == 0.0) return out + "0.0";
    Digits g = shortest_digits(av);
    if (av >= 1e-3 && av < 1e7) {
        if (g.decexp <= 0) {
            out += "0.";
            out.append((size_t)(-g.decexp), '0');
            out += g.d;
        } else {
            for (int i = 0; i < g.decexp; ++i) out.push_back(i < (int)g.d.size() ? g.d[i] : '0');
            out.push_back('.');
            if ((int)g.d.size() > g.decexp) out += g.d.substr((size_t)g.decexp);
            else out.push_back('0');
        }
    } else {
        out.push_back(g.d[0]);
        out.push_back('.');
        if (g.d.size() > 1) out += g.d.substr(1);
        else out.push_back('0');
        out += "E" + std::to_string(g.decexp - 1);
    }
    return out;
}

// BufferedReader.readLine over an in-memory file: terminators \n, \r, \r\n
struct LineReader {
    const std::string &buf;
    size_t pos = 0;
    explicit LineReader(const std::string &b) : buf(b) {}
    bool next(std::string &line) {
        if (pos >= buf.size()) return false;
        size_t e = pos;
        while (e < buf.size() && buf[e] != '\n' && buf[e] != '\r') ++e;
        line.assign(buf, pos, e - pos);
        if (e < buf.size()) {
            if (buf[e] == '\r' && e + 1 < buf.size() && buf[e + 1] == '\n') ++e;
            ++e;
        }
        pos = e;
        return true;
    }
};

bool slurp(const char *path, std::string &out) {
    FILE *f = std::fopen(path, "rb");
    if (!f) return false;
    if (std::fseek(f, 0, SEEK_END) == 0) { // regular file: one sized read
        const long sz = std::ftell(f);
        std::rewind(f);
        if (sz > 0) {
            out.resize((size_t)sz);
            const size_t got = std::fread(&out[0], 1, (size_t)sz, f);
            out.resize(got);
        }
    }
    char chunk[1 << 16]; // pipes / growing files: whatever is left
    size_t n;
    while ((n = std::fread(chunk, 1, sizeof chunk, f)) > 0) out.append(chunk, n);
    std::fclose(f);
    return true;
}

// read-only view of a whole file: mmap for regular files (no copy; the parser's threads fault the pages in in
// parallel), a buffered read for pipes and the like
struct FileView {
    const char *data = nullptr;
    size_t size = 0;
    void *map = nullptr;
    std::string buf;
    bool open(const char *path) {
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (::fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
            void *m = ::mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m != MAP_FAILED) {
                ::madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
                map = m;
                data = (const char *)m;
                size = (size_t)st.st_size;
                ::close(fd);
                return true;
            }
        }
        ::close(fd);
        if (!slurp(path, buf)) return false;
        data = buf.data();
        size = buf.size();
        return true;
    }
    ~FileView() {
        if (map) ::munmap(map, size);
    }
};

// submatrix(int[], r1, r2) clamps (:1445-1456)
void clamp_range(int m, int &r1, int &r2) {
    if (r1 < 0) r1 = 0;
    if (r2 < r1) r2 = r1;
    if (r1 >= m) r1 = m - 1;
    if (r2 >= m) r2 = m - 1;
}

long emit(const std::string &s, char *buf, size_t cap) {
    if (s.size() + 1 > cap) return -1;
    std::memcpy(buf, s.data(), s.size());
    buf[s.size()] = '\0';
    return (long)s.size();
}

double inf2nan(double x) { return std::isinf(x) ? std::nan("") : x; } // (:1008)

// worker threads for parsing / formatting: hardware threads capped by the cgroup CPU quota and PLAAC_THREADS
// Large host buffers (the encoded residues of a batch: up to gigabytes for an input that is kept for a second pass) on
// transparent huge pages where the host offers them on request (THP mode "madvise", as on the MI355X boxes): first touch of
// 3 GiB 0.44 s -> 0.12 s, unmapping 0.27 s -> 0.12 s (tools/thp_probe.cpp). free() releases them like any malloc'ed block.
void *big_alloc(size_t bytes) {
    constexpr size_t HUGE = 2u << 20;
    if (bytes < 4 * HUGE) return std::malloc(bytes);
    const size_t rounded = (bytes + HUGE - 1) / HUGE * HUGE;
    static const bool off = std::getenv("PLAAC_HUGE_PAGES") && std::getenv("PLAAC_HUGE_PAGES")[0] == '0'; // (A/B switch)
    if (off) return std::malloc(bytes);
    void *p = std::aligned_alloc(HUGE, rounded);
    if (p) (void)madvise(p, rounded, MADV_HUGEPAGE);
    return p;
}

unsigned host_threads() {
    unsigned n = std::thread::hardware_concurrency();
    if (n == 0) n = 1;
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        long period = 0;
        if (std::fscanf(f, "%63s %ld", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0) {
            const long cap = std::atol(q) / period;
            if (cap >= 1 && (unsigned)cap < n) n = (unsigned)cap;
        }
        std::fclose(f);
    }
    if (const char *e = std::getenv("PLAAC_THREADS")) {
        const int v = std::atoi(e);
        if (v >= 1) n = (unsigned)v;
    }
    return n > 64 ? 64 : n;
}

} // namespace

extern "C" {

// Parses the records that start inside [rb0, re0) of the file image d[0, nbytes): re0 is a record start or the end
// of the file, so every record of the range is whole. `trim_first`: the range's first record is the first of the
// file or follows a record that ended in a blank line (its name is then trimmed, see below). `last_blank` (nullable)
// receives how the last record ended, for the next range.
static plaac_status parse_records(const char *d, size_t nbytes, size_t rb0, size_t re0, bool trim_first,
                                  plaac_fasta **out, bool *last_blank) {
    // Record starts = lines beginning with '>' (fastareader :4325-4372: a header ends the previous record
    // whether that record was still being read or was being skipped after a blank line). Everything between two
    // starts is one record, so records can be parsed independently; only the trimming of a name depends on how
    // the PREVIOUS record ended (first record / after a blank line: found by hasmorefastas, trimmed).
    const unsigned nthreads = host_threads();
    const size_t span = re0 - rb0;
    // a '>' starts a line iff it is the first byte or follows a line terminator (\n, \r or \r\n): found per byte
    // range in parallel, concatenated in order
    std::vector<size_t> starts;
    {
        const unsigned nt = span < (8u << 20) ? 1u : nthreads;
        std::vector<std::vector<size_t>> part(nt);
        auto scan = [&](unsigned t) {
            size_t p = rb0 + span / nt * t;
            const size_t e = t + 1 == nt ? re0 : rb0 + span / nt * (t + 1);
            std::vector<size_t> &out = part[t];
            while (p < e) {
                const char *q = (const char *)memchr(d + p, '>', e - p);
                if (!q) break;
                p = (size_t)(q - d);
                if (p == 0 || d[p - 1] == '\n' || d[p - 1] == '\r') out.push_back(p);
                ++p;
            }
        };
        if (nt == 1) {
            scan(0);
        } else {
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < nt; ++t) pool.emplace_back(scan, t);
            for (auto &th : pool) th.join();
        }
        size_t total = 0;
        for (auto &v : part) total += v.size();
        starts.reserve(total);
        for (auto &v : part) starts.insert(starts.end(), v.begin(), v.end());
    }
    const size_t nrec = starts.size();
    if (nrec > 0xffffffffull) return PLAAC_ERR_ARG; // a batch is indexed with 32 bits: read such a file as a stream
    struct Rec {
        size_t name_b = 0, name_e = 0; // header text after '>' (untrimmed)
        size_t seq_len = 0;
        bool blank_end = false;
    };
    std::vector<Rec> recs(nrec);
    std::vector<std::pair<size_t, size_t>> nm(nrec); // name extents after trimming
    char *names_out = nullptr;
    const uint64_t *name_off = nullptr;
    // pass 1 (parallel): header extent, sequence length, how the record ended
    auto parse_range = [&](size_t r0, size_t r1, uint8_t *codes_out, const uint64_t *offs) {
        for (size_t i = r0; i < r1; ++i) {
            const size_t rb = starts[i], re = i + 1 < nrec ? starts[i + 1] : re0;
            size_t p = rb;
            auto next_line = [&](size_t &lb, size_t &le) -> bool { // BufferedReader.readLine on [p, re)
                if (p >= re) return false;
                // the line ends at the first \n or \r: two library scans (vectorised) instead of a byte loop - the \n
                // first, then a \r inside what it found (absent from Unix files: one more pass over a short line)
                const char *nl = (const char *)memchr(d + p, '\n', re - p);
                size_t e = nl ? (size_t)(nl - d) : re;
                if (const char *cr = (const char *)memchr(d + p, '\r', e - p)) e = (size_t)(cr - d);
                lb = p;
                le = e;
                if (e < re) {
                    if (d[e] == '\r' && e + 1 < nbytes && d[e + 1] == '\n') ++e;
                    ++e;
                }
                p = e;
                return true;
            };
            size_t lb, le;
            next_line(lb, le); // the header line
            Rec &R = recs[i];
            if (!codes_out) {
                R.name_b = lb + 1;
                R.name_e = le;
            } else { // pass 2: the (possibly trimmed) name goes to its final place
                std::memcpy(names_out + name_off[i], d + nm[i].first, nm[i].second - nm[i].first);
                names_out[name_off[i + 1] - 1] = '\0';
            }
            size_t len = 0;
            bool blank = false;
            while (next_line(lb, le)) {
                if (le == lb) { // blank line: the rest of the record is skipped
                    blank = true;
                    break;
                }
                if (codes_out) plaac_encode(d + lb, le - lb, codes_out + offs[i] + len);
                len += le - lb;
            }
            if (!codes_out) {
                R.seq_len = len;
                R.blank_end = blank;
            }
        }
    };
    auto run_parallel = [&](uint8_t *codes_out, const uint64_t *offs) {
        if (nrec < 4096 || nthreads <= 1) {
            parse_range(0, nrec, codes_out, offs);
            return;
        }
        std::vector<std::thread> pool;
        // split by bytes, not by record count, so threads get equal work
        size_t r0 = 0;
        for (unsigned t = 0; t < nthreads; ++t) {
            const size_t target = rb0 + span / nthreads * (t + 1);
            size_t r1 = t + 1 == nthreads ? nrec
                                          : (size_t)(std::lower_bound(starts.begin(), starts.end(), target) - starts.begin());
            if (r1 < r0) r1 = r0;
            pool.emplace_back(parse_range, r0, r1, codes_out, offs);
            r0 = r1;
        }
        for (auto &th : pool) th.join();
    };
    run_parallel(nullptr, nullptr);

    plaac_fasta *f = (plaac_fasta *)std::calloc(1, sizeof(plaac_fasta));
    if (!f) return PLAAC_ERR_NOMEM;
    f->nrec = (uint32_t)nrec;
    f->offsets = (uint64_t *)std::malloc((nrec + 1) * sizeof(uint64_t));
    f->name_off = (uint64_t *)std::malloc((nrec + 1) * sizeof(uint64_t));
    if (!f->offsets || !f->name_off) {
        plaac_fasta_free(f);
        return PLAAC_ERR_NOMEM;
    }
    // names: first record and records that follow a blank-line-terminated one are trimmed (:4362), others not
    uint64_t off = 0, noff = 0;
    for (size_t i = 0; i < nrec; ++i) {
        size_t b = recs[i].name_b, e = recs[i].name_e;
        if (i == 0 ? trim_first : recs[i - 1].blank_end) { // hasmorefastas: line.trim().substring(1)
            while (e > b && (unsigned char)d[e - 1] <= ' ') --e;
            // leading blanks cannot precede '>' (the line starts with it); blanks after '>' stay
        }
        nm[i] = {b, e};
        f->offsets[i] = off;
        f->name_off[i] = noff;
        off += recs[i].seq_len;
        noff += (e - b) + 1;
    }
    f->offsets[nrec] = off;
    f->name_off[nrec] = noff;
    f->nres = off;
    f->codes = (uint8_t *)big_alloc(off + 64);
    f->names = (char *)std::malloc(noff + 1);
    if (!f->codes || !f->names) {
        plaac_fasta_free(f);
        return PLAAC_ERR_NOMEM;
    }
    f->names[noff] = '\0';
    names_out = f->names;
    name_off = f->name_off;
    run_parallel(f->codes, f->offsets); // pass 2 (parallel): encode and copy names straight into the final buffers
    if (last_blank && nrec) *last_blank = recs[nrec - 1].blank_end;
    *out = f;
    return PLAAC_OK;
}

plaac_status plaac_fasta_read(const char *path, plaac_fasta **out) {
    if (!path || !out) return PLAAC_ERR_ARG;
    *out = nullptr;
    FileView file;
    if (!file.open(path)) return PLAAC_ERR_IO;
    return parse_records(file.data, file.size, 0, file.size, true, out, nullptr);
}

// ---- the same reader as a stream of batches: bounded memory whatever the size of the file (the reference reads
//      record by record, fastareader.hasmorefastas / nextfasta :4302-4375) ----
struct plaac_fasta_stream {
    FileView file;
    size_t cursor = 0;     // a record start, or 0 before the first batch
    size_t released = 0;   // bytes of the mapping already handed back to the kernel
    bool trim_next = true; // the next record is the file's first or follows a blank-line-terminated one
    int prev_blank = 1;    // the same for the text batches (plaac_fasta_next_text)
    // Text batches point into the file image and let its pages go when they are freed: the stream outlives them. live counts
    // the batches handed out and not yet freed (+ 1 for the handle itself until plaac_fasta_close); whoever brings it to zero
    // deletes the stream - a close with batches still queued somewhere (bin/plaac's no-GPU exit) defers the unmapping.
    std::atomic<long> live{1};
};
static void fasta_stream_release(plaac_fasta_stream *s) {
    if (s && s->live.fetch_sub(1, std::memory_order_acq_rel) == 1) delete s;
}

plaac_status plaac_fasta_open(const char *path, plaac_fasta_stream **out) {
    if (!path || !out) return PLAAC_ERR_ARG;
    *out = nullptr;
    plaac_fasta_stream *s = new (std::nothrow) plaac_fasta_stream();
    if (!s) return PLAAC_ERR_NOMEM;
    if (!s->file.open(path)) {
        delete s;
        return PLAAC_ERR_IO;
    }
    *out = s;
    return PLAAC_OK;
}

plaac_status plaac_fasta_next(plaac_fasta_stream *s, uint32_t max_records, uint64_t max_bytes, plaac_fasta **out) {
    if (!s || !out) return PLAAC_ERR_ARG;
    *out = nullptr;
    const char *d = s->file.data;
    const size_t nbytes = s->file.size;
    if (max_records == 0) max_records = 1;
    if (max_bytes == 0) max_bytes = 1;
    auto is_start = [&](size_t p) { return d[p] == '>' && (p == 0 || d[p - 1] == '\n' || d[p - 1] == '\r'); };
    while (s->cursor < nbytes) {
        // the batch ends at the first record start at or after cursor + max_bytes (a record is never split) ...
        size_t e = nbytes;
        if (nbytes - s->cursor > max_bytes) {
            size_t p = s->cursor + (size_t)max_bytes;
            while (p < nbytes) {
                const char *q = (const char *)memchr(d + p, '>', nbytes - p);
                if (!q) break;
                p = (size_t)(q - d);
                if (is_start(p)) {
                    e = p;
                    break;
                }
                ++p;
            }
        }
        // ... or after max_records records
        {
            size_t p = s->cursor, n = 0;
            while (p < e) {
                const char *q = (const char *)memchr(d + p, '>', e - p);
                if (!q) break;
                p = (size_t)(q - d);
                if (is_start(p) && n++ == max_records) {
                    e = p;
                    break;
                }
                ++p;
            }
        }
        plaac_fasta *f = nullptr;
        bool last_blank = false;
        const plaac_status st = parse_records(d, nbytes, s->cursor, e, s->trim_next, &f, &last_blank);
        if (st != PLAAC_OK) return st;
        s->cursor = e;
        if (s->file.map) { // pages behind the cursor are not needed again: keep the resident set bounded
            const size_t page = 4096, upto = e & ~(page - 1);
            if (upto > s->released) {
                ::madvise((char *)s->file.map + s->released, upto - s->released, MADV_DONTNEED);
                s->released = upto;
            }
        }
        if (f->nrec == 0) { // bytes before the first header (hasmorefastas skips them)
            plaac_fasta_free(f);
            continue;
        }
        s->trim_next = last_blank;
        *out = f;
        return PLAAC_OK;
    }
    return PLAAC_OK; // end of file: *out stays NULL
}

void plaac_fasta_close(plaac_fasta_stream *s) { fasta_stream_release(s); } // (deferred while text batches are alive)

// ---- the stream as batches of text for the device-side parser (K1, round 5) ----
// The reader is the one serial stage of bin/plaac's single pass, so it does as little as it can: it finds the lines that
// begin with '>' (memchr, the batch's bytes split over the host's threads) and, in the same sweep, where each header line
// ends; names are NOT copied (a name is text[starts[i] + 1 .. + name_len[i]), the text stays mapped while the batch lives).
plaac_status plaac_fasta_next_text(plaac_fasta_stream *s, uint32_t max_records, uint64_t max_bytes, plaac_fasta_text **out) {
    if (!s || !out) return PLAAC_ERR_ARG;
    *out = nullptr;
    const char *d = s->file.data;
    const size_t nbytes = s->file.size;
    if (max_records == 0) max_records = 1;
    if (max_bytes == 0) max_bytes = 1;
    auto is_start = [&](size_t p) { return d[p] == '>' && (p == 0 || d[p - 1] == '\n' || d[p - 1] == '\r'); };
    // bytes before the first header of the rest of the file belong to no record (hasmorefastas skips them)
    while (s->cursor < nbytes && !is_start(s->cursor)) {
        const char *q = (const char *)memchr(d + s->cursor + 1, '>', nbytes - s->cursor - 1);
        s->cursor = q ? (size_t)(q - d) : nbytes;
    }
    if (s->cursor >= nbytes) return PLAAC_OK;
    const size_t b0 = s->cursor;
    // the batch ends at the first record start at or after b0 + max_bytes (a record is never split), or at max_records
    size_t hard = nbytes;
    if (nbytes - b0 > max_bytes) {
        size_t p = b0 + (size_t)max_bytes;
        while (p < nbytes) {
            const char *q = (const char *)memchr(d + p, '>', nbytes - p);
            if (!q) break;
            p = (size_t)(q - d);
            if (is_start(p)) {
                hard = p;
                break;
            }
            ++p;
        }
    }
    const size_t span = hard - b0;
    const unsigned nt = span < (4u << 20) ? 1u : host_threads();
    struct Found {
        uint64_t start;
        uint32_t name_len;
    };
    std::vector<std::vector<Found>> part(nt);
    auto scan = [&](unsigned t) {
        size_t p = b0 + span / nt * t;
        const size_t e = t + 1 == nt ? hard : b0 + span / nt * (t + 1);
        std::vector<Found> &f = part[t];
        f.reserve((e - p) / 200 + 16);
        while (p < e) {
            const char *q = (const char *)memchr(d + p, '>', e - p);
            if (!q) break;
            p = (size_t)(q - d);
            if (is_start(p)) {
                // the header line ends at the first \n or \r (or with the file): the next record cannot begin before that
                size_t le = p + 1;
                while (le < nbytes && d[le] != '\n' && d[le] != '\r') ++le;
                f.push_back(Found{(uint64_t)p, (uint32_t)std::min<size_t>(le - p - 1, 0xffffffffu)});
                p = le;
            } else {
                ++p;
            }
        }
    };
    if (nt == 1) {
        scan(0);
    } else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nt; ++t) pool.emplace_back(scan, t);
        for (auto &th : pool) th.join();
    }
    size_t nrec = 0;
    for (auto &v : part) nrec += v.size();
    size_t e = hard;
    plaac_fasta_text *t = (plaac_fasta_text *)std::calloc(1, sizeof(plaac_fasta_text));
    if (!t) return PLAAC_ERR_NOMEM;
    const size_t keep = std::min<size_t>(nrec, max_records);
    t->starts = (uint64_t *)std::malloc((keep + 1) * sizeof(uint64_t));
    t->name_len = (uint32_t *)std::malloc((keep + 1) * sizeof(uint32_t));
    if (!t->starts || !t->name_len) {
        plaac_fasta_text_free(t);
        return PLAAC_ERR_NOMEM;
    }
    size_t k = 0;
    for (auto &v : part)
        for (const Found &f : v) {
            if (k == keep) {
                if (e == hard) e = (size_t)f.start; // the first record that no longer fits: the batch ends where it begins
                break;
            }
            t->starts[k] = f.start - b0;
            t->name_len[k] = f.name_len;
            ++k;
        }
    t->starts[keep] = e - b0;
    // how the batch's last record ends, for the next batch's first name (the device reports it for every record; the reader
    // needs it for this ONE record so that batches can be collected by any context in any order): the rule of
    // fasta_device.hip.inc - an empty line behind the header = a terminator preceded by \n, or a \r preceded by \r
    t->prev_blank = s->prev_blank;
    t->last_blank = t->prev_blank;
    if (keep) {
        const size_t rb = b0 + (size_t)t->starts[keep - 1];
        size_t h = rb;
        while (h < e && d[h] != '\n' && d[h] != '\r') ++h;
        int blank = 0;
        for (size_t q = h + 1; q < e && !blank; ++q) {
            const char c = d[q], pc = d[q - 1];
            blank = (c == '\n' || c == '\r') && (pc == '\n' || (pc == '\r' && c == '\r'));
        }
        t->last_blank = blank;
    }
    s->prev_blank = t->last_blank;
    t->text = d + b0;
    t->len = e - b0;
    t->nrec = (uint32_t)keep;
    t->owner_ = s;
    s->live.fetch_add(1, std::memory_order_relaxed);
    t->file_off_ = b0;
    s->cursor = e;
    *out = t;
    return PLAAC_OK;
}

void plaac_fasta_text_free(plaac_fasta_text *t) {
    if (!t) return;
    // (the pages of the file image behind this batch are not needed again: the stream lets them go in file order)
    plaac_fasta_stream *s = (plaac_fasta_stream *)t->owner_;
    if (s && s->file.map && t->starts) {
        const size_t page = 4096, a = (size_t)t->file_off_ & ~(page - 1), b = ((size_t)t->file_off_ + (size_t)t->len) & ~(page - 1);
        if (b > a) ::madvise((char *)s->file.map + a, b - a, MADV_DONTNEED);
    }
    std::free(t->starts);
    std::free(t->name_len);
    std::free(t);
    fasta_stream_release(s);
}

int plaac_fasta_text_trim_names(plaac_fasta_text *t, const uint8_t *blank_end, int prev_blank) {
    if (!t || !blank_end) return prev_blank;
    for (uint32_t i = 0; i < t->nrec; ++i) {
        if (i == 0 ? prev_blank != 0 : blank_end[i - 1] != 0) { // hasmorefastas: line.trim().substring(1) (:4362)
            const unsigned char *b = (const unsigned char *)t->text + t->starts[i] + 1;
            uint32_t e = t->name_len[i];
            while (e > 0 && b[e - 1] <= ' ') --e;
            t->name_len[i] = e;
        }
    }
    return t->nrec ? (blank_end[t->nrec - 1] != 0) : prev_blank;
}

uint64_t plaac_fasta_text_codes(const char *text, const uint64_t *starts, const uint32_t *extents, uint32_t i, uint64_t first,
                                uint64_t count, uint8_t *out) {
    if (!text || !starts || !extents || !out || count == 0) return 0;
    static const std::array<uint8_t, 256> lut = [] {
        std::array<uint8_t, 256> l{};
        char all[256];
        for (int k = 0; k < 256; ++k) all[k] = (char)k;
        plaac_encode(all, 256, l.data());
        return l;
    }();
    const unsigned char *p = (const unsigned char *)text + starts[i] + extents[2 * (size_t)i] + 1;
    const unsigned char *e = (const unsigned char *)text + starts[i] + extents[2 * (size_t)i + 1];
    uint64_t idx = 0, k = 0;
    for (; p < e && idx < first; ++p) idx += !(*p == '\n' || *p == '\r'); // (the residues in front of the range)
    for (; p < e && k < count; ++p)
        if (!(*p == '\n' || *p == '\r')) out[k++] = lut[*p];
    return k;
}

void plaac_fasta_free(plaac_fasta *f) {
    if (!f) return;
    std::free(f->codes);
    std::free(f->offsets);
    std::free(f->names);
    std::free(f->name_off);
    std::free(f);
}

plaac_status plaac_read_aa_params(const char *path, double vec[PLAAC_NAA], int *warn_line) {
    if (!path || !vec) return PLAAC_ERR_ARG;
    for (int i = 0; i < PLAAC_NAA; ++i) {
        vec[i] = 0.0;
        if (warn_line) warn_line[i] = 0;
    }
    std::string file;
    if (!slurp(path, file)) return PLAAC_ERR_IO;
    LineReader rd(file);
    std::string line;
    for (int i = 0; i < PLAAC_NAA; ++i) {
        if (!rd.next(line)) return PLAAC_ERR_ARG; // the reference throws on a short file
        std::vector<std::string> tok; // StringTokenizer: split on blanks, tabs, newlines, form feeds
        size_t p = 0;
        while (p < line.size()) {
            while (p < line.size() && std::strchr(" \t\n\r\f", line[p])) ++p;
            size_t q = p;
            while (q < line.size() && !std::strchr(" \t\n\r\f", line[q])) ++q;
            if (q > p) tok.push_back(line.substr(p, q - p));
            p = q;
        }
        if (tok.empty()) return PLAAC_ERR_ARG;
        char *endp = nullptr;
        vec[i] = std::strtod(tok[0].c_str(), &endp);
        if (endp == tok[0].c_str()) return PLAAC_ERR_ARG;
        if (tok.size() > 2 && warn_line && tok[2][0] != kAlphabet[i]) warn_line[i] = 1;
    }
    return PLAAC_OK;
}

unsigned plaac_host_threads(void) { return host_threads(); }

int plaac_format_fixed(double v, int decimals, char *buf, size_t cap) {
    std::string s;
    append_fixed(s, v, decimals);
    return (int)emit(s, buf, cap);
}

// the digit-string path alone (what append_fixed falls back to): the differential test compares the two
int plaac_format_fixed_reference(double v, int decimals, char *buf, size_t cap) {
    return (int)emit(fixed_string(v, decimals), buf, cap);
}

int plaac_format_double_tostring(double v, char *buf, size_t cap) {
    return (int)emit(java_double_tostring(v), buf, cap);
}

const char *plaac_summary_header(void) {
    return "SEQid\tMW\tMWstart\tMWend\tMWlen\tLLR\tLLRstart\tLLRend\tLLRlen\tNLLR\tVITmaxrun\tCOREscore\tCOREstart\t"
           "COREend\tCORElen\tPRDscore\tPRDstart\tPRDend\tPRDlen\tPROTlen\tHMMall\tHMMvit\tCOREaa\tSTARTaa\tENDaa\t"
           "PRDaa\tFInumaa\tFImeanhydro\tFImeancharge\tFImeancombo\tFImaxrun\tPAPAcombo\tPAPAprop\tPAPAfi\tPAPAllr\t"
           "PAPAllr2\tPAPAcen\tPAPAaa";
}

const char *plaac_tracks_header(void) {
    return "ORDER\tSEQid\tAANUM\tAA\tVIT\tMAP\tCHARGE\tHYDRO\tFI\tPLAAC\tPAPA\tFIx2\tPLAACx2\tPAPAx2\tHMM.background\t"
           "HMM.PrD-like";
}

// The row straight into the caller's buffer (round 5: 10 M of these are the longest stage of bin/plaac's run once the parse
// is the device's): no string object, the integer and the three decimals of a %.3f from one rounded product.
namespace {
struct RowOut {
    char *p;
    inline void ch(char c) { *p++ = c; }
    inline void lit(const char *s, size_t n) {
        std::memcpy(p, s, n);
        p += n;
    }
    inline void u64(unsigned long long v) {
        char tmp[24];
        int k = 24;
        do {
            tmp[--k] = (char)('0' + v % 10);
            v /= 10;
        } while (v);
        lit(tmp + k, (size_t)(24 - k));
    }
    inline void I(long v) { // "\t" + decimal
        ch('\t');
        if (v < 0) {
            ch('-');
            u64(0ull - (unsigned long long)v);
        } else {
            u64((unsigned long long)v);
        }
    }
    inline void F(double v) { // "\t" + %.3f of java.util.Formatter (append_fixed's fast path; everything else through it)
        ch('\t');
        const double av = std::fabs(v);
        if (av < 1e9) { // (false for NaN)
            const double y = av * 1e3;
            const double fl = std::floor(y), f = y - fl, err = y * 0x1p-51 + 0x1p-60;
            if (err < 0.25) {
                unsigned long long r = (unsigned long long)fl;
                r += std::fabs(f - 0.5) > err ? (f > 0.5 ? 1ull : 0ull) : (av >= (double)(2ull * r + 1ull) / 2000.0 ? 1ull : 0ull); // (append_fixed)
                if (std::signbit(v)) ch('-');
                u64(r / 1000ull);
                const unsigned fr = (unsigned)(r % 1000ull);
                ch('.');
                ch((char)('0' + fr / 100u));
                ch((char)('0' + fr / 10u % 10u));
                ch((char)('0' + fr % 10u));
                return;
            }
        }
        thread_local std::string s;
        s.clear();
        append_fixed(s, v, 3);
        lit(s.data(), s.size());
    }
    inline void aa(const uint8_t *codes, int m, int r1, int r2) {
        clamp_range(m, r1, r2);
        for (int i = r1; i <= r2; ++i) *p++ = kAlphabet[codes[i] <= 21 ? codes[i] : 0];
    }
};
} // namespace

long plaac_format_summary_row_n(const plaac_row *r, const char *name, size_t name_len, const uint8_t *codes, uint64_t reclen,
                                int corelength, int ww2, char *buf, size_t cap) {
    if (!r || !name || !buf) return -1;
    if (r->prot_len <= 0) return 0; // skipped record (:762)
    const int n = r->prot_len;
    (void)reclen;
    // the longest a row can be: the name, 37 fields of at most 26 characters (a %.3f below 1e9; larger values and the
    // non-finite ones are at most 330), four sequences inside the protein and the PAPA window
    if (cap < name_len + 3 * (size_t)n + (size_t)(ww2 > 0 ? ww2 : 0) + 37 * 26 + 23 * 330 + 64) return -1;
    RowOut o{buf};
    o.lit(name, name_len);
    // one-based indices; the -1/-2 sentinels are shifted too, as the reference does (:902-913)
    o.I(r->mw_score);
    o.I(r->mw_start + 1);
    o.I(r->mw_end + 1);
    o.I(r->mw_end - r->mw_start + 1);
    o.F(inf2nan(r->llr_score));
    o.I(r->llr_start + 1);
    o.I(r->llr_end + 1);
    o.I(r->llr_end - r->llr_start + 1);
    o.F(inf2nan(r->llr_score) / (double)(r->llr_end - r->llr_start + 1)); // NLLR (:907)
    o.I(r->vit_maxrun);
    o.F(inf2nan(r->core_score));
    o.I(r->core_start + 1);
    o.I(r->core_end + 1);
    o.I(r->core_end - r->core_start + 1);
    o.F(r->prd_score);
    o.I(r->prd_start + 1);
    o.I(r->prd_end + 1);
    o.I(r->prd_end - r->prd_start + 1);
    o.I(n);
    o.F(r->hmm_all);
    o.F(r->hmm_vit);
    o.ch('\t');
    if (r->prd_end - r->prd_start + 1 >= corelength) { // (:915-922)
        o.aa(codes, n, r->core_start, r->core_end);
        o.ch('\t');
        o.aa(codes, n, r->prd_start, r->prd_start + 14);
        o.ch('\t');
        o.aa(codes, n, r->prd_end - 14, r->prd_end);
        o.ch('\t');
        o.aa(codes, n, r->prd_start, r->prd_end);
    } else {
        o.lit("-\t-\t-\t-", 7);
    }
    o.I(r->fi_numaa);
    o.F(r->fi_meanhydro);
    o.F(r->fi_meancharge);
    o.F(r->fi_meancombo);
    o.I(r->fi_maxrun);
    o.F(inf2nan(r->papa_combo));
    o.F(r->papa_prop);
    o.F(r->papa_fi);
    o.F(r->papa_llr);
    o.F(r->papa_llr2);
    o.I(r->papa_cen + 1);
    o.ch('\t');
    o.aa(codes, n, r->papa_cen - ww2 / 2, r->papa_cen + ww2 / 2); // (:944)
    return (long)(o.p - buf);
}

long plaac_format_summary_row(const plaac_row *r, const char *name, const uint8_t *codes, uint64_t reclen,
                              int corelength, int ww2, char *buf, size_t cap) {
    if (!name) return -1;
    // (the bound of the direct writer is generous; a caller with a tight buffer gets the row through a string)
    const size_t nl = std::strlen(name);
    const long k = plaac_format_summary_row_n(r, name, nl, codes, reclen, corelength, ww2, buf, cap);
    if (k >= 0 && buf && cap) buf[(size_t)k < cap ? (size_t)k : cap - 1] = '\0';
    if (k >= 0 || !r || !buf || r->prot_len <= 0) return k;
    std::vector<char> big(nl + 3 * (size_t)r->prot_len + (size_t)(ww2 > 0 ? ww2 : 0) + 37 * 26 + 23 * 330 + 64);
    const long k2 = plaac_format_summary_row_n(r, name, nl, codes, reclen, corelength, ww2, big.data(), big.size());
    if (k2 < 0 || (size_t)k2 + 1 > cap) return -1;
    std::memcpy(buf, big.data(), (size_t)k2);
    buf[k2] = '\0';
    return k2;
}

size_t plaac_track_rows_bound(uint32_t n, size_t id_len, size_t name_len) {
    return (size_t)n * (id_len + name_len + 400) + 128;
}

long plaac_format_track_rows(const plaac_tracks *t, uint64_t first, const uint8_t *codes, uint32_t n,
                             const char *order_id, const char *name, char *buf, size_t cap) {
    if (!t || !codes || !order_id || !name || !buf) return -1;
    std::string s;
    s.reserve((size_t)n * 160 + 64);
    static const int prec[8] = {4, 4, 8, 4, 8, 8, 4, 8}; // (:638)
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t k = first + i;
        s += order_id;
        s += '\t';
        s += name;
        s += '\t';
        append_int(s, (long)i + 1);
        s += '\t';
        s += kAlphabet[codes[i] <= 21 ? codes[i] : 0];
        s += '\t';
        append_int(s, (long)t->vit[k]);
        s += '\t';
        append_int(s, (long)t->map[k]);
        const double v[8] = {t->charge[k], t->hydro[k], t->fi[k], t->plaacllr[k], t->papa[k], t->fix2[k],
                             t->plaacllrx2[k], t->papax2[k]};
        for (int j = 0; j < 8; ++j) {
            s += '\t';
            append_fixed(s, v[j], prec[j]);
        }
        s += '\t';
        append_fixed(s, t->post0[k], 4);
        s += '\t';
        append_fixed(s, t->post1[k], 4);
        s += '\n';
    }
    s += "########################################################\n";
    return emit(s, buf, cap);
}

long plaac_format_param_block(const plaac_params *p, char *buf, size_t cap) {
    if (!p || !buf) return -1;
    auto aa = [](const double *v) {
        std::string s;
        for (int i = 0; i < PLAAC_NAA; ++i) {
            s.push_back(kAlphabet[i]);
            s.push_back('=');
            s += fixed_string(v[i], 5);
            s.push_back(';');
        }
        return s;
    };
    std::string s = "############################ parameters at run-time ####################################\n";
    s += "## alpha=" + java_double_tostring(p->alpha) + "; corelength=" + std::to_string(p->corelength) +
         "; ww1=" + std::to_string(p->ww1) + "; ww2=" + std::to_string(p->ww2) + "; ww3=" + std::to_string(p->ww3) +
         "; adjustprolines=" + (p->adjustprolines ? "true" : "false") + ";\n";
    s += "## fg_used: {" + aa(p->fg) + "}\n";
    s += "## bg_scer: {" + aa(p->bgscer) + "}\n";
    s += "## bg_input: {" + aa(p->bgthis) + "}\n";
    s += "## bg_used: {" + aa(p->bg) + "}\n";
    s += "## plaac_llr: {" + aa(p->llr) + "}\n";
    s += "## papa_lods: {" + aa(p->lodpapa) + "}\n";
    s += "#######################################################################################\n";
    return emit(s, buf, cap);
}

// hmm.dottify(filename, true) (:4209-4287) for the two-state model of prionhmm1 (:968-981): the same nodes,
// edges, labels and attributes, so `dot -Tpng` draws the same picture.
long plaac_format_hmm_dot(const plaac_params *p, char *buf, size_t cap) {
    if (!p || !buf) return -1;
    const double trans[2][2] = {{99.9 / 100, 0.1 / 100}, {2.0 / 100, 98.0 / 100}};
    const double init[2] = {0.9524, 0.0476};
    const char *names[2] = {"background", "PrD-like"};
    double eprob[2][PLAAC_NAA]; // emat = normalize(bg), normalize(fg) (:974-975)
    for (int st = 0; st < 2; ++st) {
        const double *src = st == 0 ? p->bg : p->fg;
        double sum = 0;
        for (int k = 0; k < PLAAC_NAA; ++k) sum = sum + src[k];
        sum = 1.0 * sum;
        if (sum < 0.000000000001) sum = 1;
        for (int k = 0; k < PLAAC_NAA; ++k) eprob[st][k] = src[k] / sum;
    }
    std::string s = "Digraph G {\n"
                    "edge [fontname=Courier, fontsize=8, labelfontname=Courier,labelfontsize=8];\n"
                    " node [fontname=Courier, fontsize=10]\n"
                    " start [label=start, shape=circle, height=0.25, style=filled, color=grey, rank=source];\n";
    auto node = [](int i) { return "n" + std::to_string(i); };
    for (int i = 0; i < 2; ++i) {
        s += "  " + node(i) + " [label=\"" + names[i] + "\", shape=circle, height=1.2];\n";
        s += "  start -> " + node(i) + " [label=\"" + fixed_string(init[i], 3) + "\", color=gray];\n";
    }
    for (int i = 0; i < 2; ++i) {
        for (int j = 0; j < 2; ++j) {
            if (!(trans[i][j] > 0)) continue;
            const std::string lab = fixed_string(trans[i][j], 3);
            if (i == j) { // an invisible inner self-edge makes the visible outer one a little bigger
                const std::string port = (i % 2 == 0) ? ":w" : ":e";
                const std::string ends = "  " + node(i) + port + " -> " + node(j) + port;
                s += ends + " [label=\"\", color=gray, style=invis];\n";
                s += ends + " [label=\"" + lab + "\", color=gray];\n";
            } else {
                const std::string ends = "  " + node(i) + " -> " + node(j);
                if (i < j) s += ends + " [label=\"spacerlabel\", color=gray, constraint=false, style=invis];\n";
                s += ends + " [label=\"" + lab + "\", color=gray, constraint=false];\n";
            }
        }
    }
    for (int i = 0; i < 2; ++i) { // emission records, X and * left out
        s += "rec" + std::to_string(i) + " [shape=record, label=\"{ <fs> AA";
        for (int k = 1; k < PLAAC_NAA - 1; ++k) {
            s += '|';
            s += kAlphabet[k];
        }
        s += "}|{ <f" + std::to_string(i) + "> prob";
        for (int k = 1; k < PLAAC_NAA - 1; ++k) s += "|" + fixed_string(eprob[i][k], 4);
        s += "}\"];\n";
    }
    for (int i = 0; i < 2; ++i) s += "  " + node(i) + " -> rec" + std::to_string(i) + " [style=dashed];\n";
    s += "}\n";
    return emit(s, buf, cap);
}

long plaac_format_aa_params(const double vec[PLAAC_NAA], char *buf, size_t cap) {
    if (!vec || !buf) return -1;
    std::string s;
    for (int i = 0; i < PLAAC_NAA; ++i) {
        s += fixed_string(vec[i], 6);
        s += " # ";
        s.push_back(kAlphabet[i]);
        s.push_back('\n');
    }
    return emit(s, buf, cap);
}

} // extern "C"
