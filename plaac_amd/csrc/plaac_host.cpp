// plaac_host.cpp — host-side (no device) half of libplaac_native.so:
// parameter/table setup and residue encoding. Product code; never touches oracle/.
//
// Reference semantics implemented here (cli/src/plaac.java of whitehead/plaac):
//   table setup in main           :444-500   (normalise x3, alpha mixing, eps for X and *, llr = ln(fg/bg))
//   plaac() constructor           :279-291   (loglut, lodpapa1)
//   aahydro2                      :90        ((1/9)*aahydro + 0.5)
//   prionhmm1 / prionhmm0         :968-1001  (T, I, E = normalize(bg|fg))
//   hmm.initialize                :2893-2935 (logs; all fprob <= 1e-4 -> free end -> lfprob = 0)
//   aatoint                       :1508-1534
// Every sum is a plain left-to-right fp64 loop and nothing is fused (build with -ffp-contract=off).
#include "plaac_native.h"

#include <array>
#include <cmath>
#include <cstring>

namespace {

using Vec = std::array<double, PLAAC_NAA>;

// Data tables of the reference (plaac.java:37-60, 64-87, 206-229, 261-270).
constexpr Vec kCharge = {0, 0, 0, 1, 1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 0, -1, 0, 0, 0, 0, 0, 0};
constexpr Vec kHydro = {0.0, 1.8, 2.5, -3.5, -3.5, 2.8, -0.4, -3.2, 4.5, -3.9, 3.8,
                        1.9, -3.5, -1.6, -3.5, -4.5, -0.8, -0.7, 4.2, -0.9, -1.3, 0.0};
constexpr Vec kOddsPapa = {0.0, 0.67267686, 1.5146198, 0.27887323, 0.5460614, 2.313433, 0.96153843, 0.75686276,
                           2.2562358, 0.20664589, 0.9607843, 1.9615384, 1.0836071, 0.30196398, 1.0716166, 0.6664044,
                           1.1432927, 0.8917492, 2.2562358, 1.9478673, 2.1785367, 0.0};
constexpr Vec kBgScer = {0, 0.0550, 0.0126, 0.0586, 0.0655, 0.0441, 0.0498, 0.0217, 0.0655, 0.0735, 0.0950,
                         0.0207, 0.0615, 0.0438, 0.0396, 0.0444, 0.0899, 0.0592, 0.0556, 0.0104, 0.0337, 0};
constexpr Vec kFg04 = {0, 0.0488, 0.0032, 0.0202, 0.0234, 0.0276, 0.1157, 0.0149, 0.0191, 0.0329, 0.0456,
                       0.0149, 0.1444, 0.0308, 0.2208, 0.0202, 0.1008, 0.0297, 0.0234, 0.0064, 0.0573, 0};
constexpr Vec kFg28 = {0, 0.04865, 0.00219, 0.01638, 0.00783, 0.02537, 0.07603, 0.0181, 0.02018, 0.01641, 0.02639,
                       0.02975, 0.25885, 0.05126, 0.15178, 0.025, 0.10988, 0.03841, 0.01972, 0.00157, 0.05624, 0};

// normalize (:1933-1941): divide by the left-to-right sum unless it is < 1e-12.
Vec unit_sum(const Vec &v) {
    double total = 0;
    for (double e : v) total = total + e;
    total = 1.0 * total;
    if (total < 0.000000000001) total = 1;
    Vec r;
    for (int i = 0; i < PLAAC_NAA; ++i) r[i] = v[i] / total;
    return r;
}

void fill_hmm(plaac_hmm &h, const double (&trans)[2][2], const double (&init)[2], const Vec &emit0,
              const Vec &emit1) {
    double endp[2];
    bool free_end = true;
    for (int s = 0; s < 2; ++s) {
        double rowsum = 0;
        for (int d = 0; d < 2; ++d) {
            rowsum = rowsum + trans[s][d];
            h.lt[s][d] = std::log(trans[s][d]);
        }
        endp[s] = std::fmax(0.0, 1.0 - rowsum);
        if (endp[s] > 0.0001) free_end = false;
        h.li[s] = std::log(init[s]);
    }
    for (int s = 0; s < 2; ++s) h.lf[s] = std::log(free_end ? 1.0 : endp[s]);
    for (int k = 0; k < PLAAC_NAA; ++k) {
        h.le[0][k] = std::log(emit0[k]);
        h.le[1][k] = std::log(emit1[k]);
    }
}

struct CodeLut {
    uint8_t t[256];
    CodeLut() {
        std::memset(t, 0, sizeof t);
        const char *aas = "ACDEFGHIKLMNPQRSTVWY";
        for (int i = 0; i < 20; ++i) {
            t[(unsigned char)aas[i]] = (uint8_t)(i + 1);
            t[(unsigned char)(aas[i] + ('a' - 'A'))] = (uint8_t)(i + 1);
        }
        t[(unsigned char)'*'] = 21;
    }
};
const CodeLut kCodeLut;

} // namespace

extern "C" {

int plaac_abi_version(void) { return PLAAC_ABI_VERSION; }
size_t plaac_sizeof_params(void) { return sizeof(plaac_params); }
size_t plaac_sizeof_row(void) { return sizeof(plaac_row); }

void plaac_builtin_tables(double bg_scer[PLAAC_NAA], double fg28[PLAAC_NAA], double fg04[PLAAC_NAA]) {
    if (bg_scer) std::memcpy(bg_scer, kBgScer.data(), sizeof(Vec));
    if (fg28) std::memcpy(fg28, kFg28.data(), sizeof(Vec));
    if (fg04) std::memcpy(fg04, kFg04.data(), sizeof(Vec));
}

plaac_status plaac_params_init(plaac_params *out, const double fgfreq[PLAAC_NAA], const double bgcounts[PLAAC_NAA],
                               double alpha, int corelength, int ww1, int ww2, int ww3, int adjustprolines) {
    if (!out || corelength < 1 || ww1 < 1 || ww2 < 1 || ww3 < 1) return PLAAC_ERR_ARG;
    std::memset(out, 0, sizeof *out);
    if (alpha > 1 || alpha < 0) alpha = 1.0; // :444-447
    out->corelength = corelength;
    out->ww1 = ww1;
    out->ww2 = ww2;
    out->ww3 = ww3;
    out->adjustprolines = adjustprolines ? 1 : 0;
    out->alpha = alpha;
    out->cc[0] = 2.785;
    out->cc[1] = -1;
    out->cc[2] = -1.151;
    out->big_neg = -1000000.0;

    Vec fgraw = kFg28, bgraw{};
    if (fgfreq) std::memcpy(fgraw.data(), fgfreq, sizeof(Vec));
    if (bgcounts) std::memcpy(bgraw.data(), bgcounts, sizeof(Vec));

    const Vec scer = unit_sum(kBgScer); // :312
    fgraw[0] = fgraw[21] = 0;           // :449
    Vec fgn = unit_sum(fgraw);          // :452
    bgraw[0] = bgraw[21] = 0;           // :454
    const Vec own = unit_sum(bgraw);    // :456
    Vec blend;
    for (int i = 0; i < PLAAC_NAA; ++i) blend[i] = alpha * scer[i] + (1 - alpha) * own[i]; // axpby :1981
    Vec mix = unit_sum(blend);                                                           // :458
    const double tiny = 0.00001;                                                         // :490
    fgn[0] = fgn[21] = tiny;
    mix[0] = mix[21] = tiny;
    const Vec fg = unit_sum(fgn); // :496
    const Vec bg = unit_sum(mix); // :497

    for (int i = 0; i < PLAAC_NAA; ++i) {
        out->fg[i] = fg[i];
        out->bg[i] = bg[i];
        out->bgscer[i] = scer[i];
        out->bgthis[i] = own[i];
        out->hydro2[i] = (1.0 / 9.0) * kHydro[i] + 0.5;
        out->charge[i] = kCharge[i];
        out->llr[i] = (i >= 1 && i <= 20) ? std::log(fg[i] / bg[i]) : 0.0;       // :500
        out->lodpapa[i] = (i >= 1 && i <= 20) ? std::log(kOddsPapa[i]) : 0.0;    // :289
    }
    for (int i = 0; i < PLAAC_LUTLEN; ++i) out->loglut[i] = std::log(1.0 + std::exp(-i / 100.0)); // :283

    const Vec ebg = unit_sum(bg), efg = unit_sum(fg); // :974-975, :994-995
    const double t1[2][2] = {{99.9 / 100, 0.1 / 100}, {2.0 / 100, 98.0 / 100}};
    const double i1[2] = {0.9524, 0.0476};
    fill_hmm(out->hmm1, t1, i1, ebg, efg);
    const double t0[2][2] = {{1, 0}, {0, 1}};
    const double i0[2] = {1, 0};
    fill_hmm(out->hmm0, t0, i0, ebg, ebg);
    return PLAAC_OK;
}

void plaac_encode(const char *text, size_t n, uint8_t *codes) {
    for (size_t i = 0; i < n; ++i) codes[i] = kCodeLut.t[(unsigned char)text[i]];
}

// ---- wire rows: what a summary row needs to cross a link (include/plaac_native.h) ----
// Word layout of plaac_row as 40 x 32-bit words: 0..25 the thirteen doubles, 26..39 the fourteen ints. The wire row keeps
// words 0..17 (llr_score .. papa_combo), 20..25 (papa_fi, papa_llr, papa_llr2), then mw_score (| stop flag << 30),
// mw_start, llr_start, vit_maxrun, core_start, prd_start, prd_end, fi_numaa, fi_maxrun, papa_cen.
static const int kWireWords[PLAAC_WIRE_ROW_BYTES / 4] = {0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 16,
                                                         17, 20, 21, 22, 23, 24, 25, 26, 27, 29, 31, 32, 34, 35, 37, 38, 39};

plaac_status plaac_rows_to_wire(const plaac_row *rows, const uint64_t *offsets, uint32_t n, uint8_t *wire) {
    if (n && (!rows || !offsets || !wire)) return PLAAC_ERR_ARG;
    static_assert(sizeof(plaac_row) == 160, "plaac_row layout");
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t w[40], o[PLAAC_WIRE_ROW_BYTES / 4];
        std::memcpy(w, &rows[i], sizeof w);
        const uint64_t raw = offsets[i + 1] - offsets[i];
        const uint64_t flag = raw - (uint64_t)(uint32_t)rows[i].prot_len; // one trailing stop trimmed (:758)?
        if (offsets[i + 1] < offsets[i] || flag > 1u || rows[i].mw_score < 0 || rows[i].mw_score > 80) return PLAAC_ERR_ARG;
        for (int k = 0; k < PLAAC_WIRE_ROW_BYTES / 4; ++k) o[k] = w[kWireWords[k]];
        o[24] |= (uint32_t)flag << 30;
        std::memcpy(wire + (size_t)i * PLAAC_WIRE_ROW_BYTES, o, sizeof o);
    }
    return PLAAC_OK;
}

plaac_status plaac_rows_from_wire(const uint8_t *wire, const uint64_t *offsets, uint32_t n, int32_t corelength, plaac_row *rows) {
    if ((n && (!rows || !offsets || !wire)) || corelength < 1) return PLAAC_ERR_ARG;
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t w[40] = {0}, o[PLAAC_WIRE_ROW_BYTES / 4];
        std::memcpy(o, wire + (size_t)i * PLAAC_WIRE_ROW_BYTES, sizeof o);
        const uint32_t flag = (o[24] >> 30) & 1u;
        o[24] &= ~(1u << 30);
        for (int k = 0; k < PLAAC_WIRE_ROW_BYTES / 4; ++k) w[kWireWords[k]] = o[k];
        plaac_row &r = rows[i];
        std::memcpy(&r, w, sizeof w);
        if (offsets[i + 1] < offsets[i] || offsets[i + 1] - offsets[i] < flag) return PLAAC_ERR_ARG;
        const uint64_t len = offsets[i + 1] - offsets[i] - flag;
        if (len > 0x7fffffffull) return PLAAC_ERR_ARG;
        r.prot_len = (int32_t)len;
        if (len == 0) { // a skipped record (:762): its row is all zero
            std::memset(&r, 0, sizeof r);
            continue;
        }
        // hss2 with min == max == L returns end = start + L - 1, or (-1, -2) when L > n (:1210-1215, :1253-1256)
        r.mw_end = r.mw_start + (int32_t)(len < 80 ? len : 80) - 1;          // :769-771
        r.llr_end = r.llr_start >= 0 ? r.llr_start + corelength - 1 : -2;     // :782-783
        r.core_end = r.core_start >= 0 ? r.core_start + corelength - 1 : -2;  // :818-833, :873-880
        // papamaxprop = papax2[papamaxcenter] = the maximum itself when there is a centre, else NaN (:4944-4946, :4993)
        if (r.papa_cen >= 0) {
            r.papa_prop = r.papa_combo;
        } else {
            const uint64_t qnan = 0x7ff8000000000000ull;
            std::memcpy(&r.papa_prop, &qnan, 8);
        }
    }
    return PLAAC_OK;
}

} // extern "C"
