/*
 * oracle/plaac_oracle.c — TEST INFRASTRUCTURE ONLY (see plaac_oracle.h).
 *
 * CPU restatement of the PLAAC per-protein scoring path. Written from the
 * operation-order contract in SURVEY.md §8(a)/§9; every function cites the
 * reference lines (cli/src/plaac.java) whose arithmetic it restates.
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (Java never fuses mul+add).
 *
 * Parity status: Viterbi parse pinned by tests/golden/kat28.tsv (28/28);
 * all float outputs PARITY UNPINNED (reference has no golden values, no JVM here).
 *
 * Dead work of the reference that cannot change any printed column is omitted:
 * logprobsubtrellis/margcollapse/etst (:3264-3282), hssr/hssr2 (:5002-5007),
 * numdisordered(strict) and per-run local stats (:4908-4929, :5038-5067), and in
 * summary mode the backward/posterior/MAP pass and hmm0's Viterbi/posterior arrays.
 */
#include "plaac_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NAA ORACLE_NAA

/* ---- constant tables: data of plaac.java:37-60, 64-87, 206-229, 261-270 ---- */
static const double k_charge[NAA] = {0, 0, 0, 1, 1, 0, 0, 0, 0, -1, 0, 0, 0, 0, 0, -1, 0, 0, 0, 0, 0, 0};
static const double k_hydro[NAA] = {0.0, 1.8, 2.5, -3.5, -3.5, 2.8, -0.4, -3.2, 4.5, -3.9, 3.8,
                                    1.9, -3.5, -1.6, -3.5, -4.5, -0.8, -0.7, 4.2, -0.9, -1.3, 0.0};
static const double k_odpapa1[NAA] = {0.0, 0.67267686, 1.5146198, 0.27887323, 0.5460614, 2.313433, 0.96153843,
                                      0.75686276, 2.2562358, 0.20664589, 0.9607843, 1.9615384, 1.0836071,
                                      0.30196398, 1.0716166, 0.6664044, 1.1432927, 0.8917492, 2.2562358,
                                      1.9478673, 2.1785367, 0.0};
static const double k_bg_scer[NAA] = {0, 0.0550, 0.0126, 0.0586, 0.0655, 0.0441, 0.0498, 0.0217, 0.0655, 0.0735, 0.0950,
                                      0.0207, 0.0615, 0.0438, 0.0396, 0.0444, 0.0899, 0.0592, 0.0556, 0.0104, 0.0337, 0};
static const double k_fg04[NAA] = {0, 0.0488, 0.0032, 0.0202, 0.0234, 0.0276, 0.1157, 0.0149, 0.0191, 0.0329, 0.0456,
                                   0.0149, 0.1444, 0.0308, 0.2208, 0.0202, 0.1008, 0.0297, 0.0234, 0.0064, 0.0573, 0};
static const double k_fg28[NAA] = {0, 0.04865, 0.00219, 0.01638, 0.00783, 0.02537, 0.07603, 0.0181, 0.02018, 0.01641,
                                   0.02639, 0.02975, 0.25885, 0.05126, 0.15178, 0.025, 0.10988, 0.03841, 0.01972,
                                   0.00157, 0.05624, 0};

void oracle_const_tables(double bg_scer[NAA], double fg28[NAA], double fg04[NAA]) {
    memcpy(bg_scer, k_bg_scer, sizeof k_bg_scer);
    memcpy(fg28, k_fg28, sizeof k_fg28);
    memcpy(fg04, k_fg04, sizeof k_fg04);
}

size_t oracle_sizeof_params(void) { return sizeof(oracle_params); }
size_t oracle_sizeof_row(void) { return sizeof(oracle_row); }

/* normalize, plaac.java:1933-1941 (sum left->right :1570; "if sm<eps sm=1") */
static void normalize22(const double *a, double *out) {
    double sm = 0;
    for (int i = 0; i < NAA; i++) sm = sm + a[i];
    sm = 1.0 * sm;
    if (sm < 0.000000000001) sm = 1;
    for (int i = 0; i < NAA; i++) out[i] = a[i] / sm;
}

/* hmm.initialize, plaac.java:2893-2935: logs of T/E/I; all fprob<=1e-4 -> freeend -> lf=log(1)=0 */
static void hmm_setup(oracle_hmm *h, const double t[2][2], const double ini[2], const double *e0, const double *e1) {
    int freeend = 1;
    double f[2];
    for (int i = 0; i < 2; i++) {
        double rs = 0;
        for (int j = 0; j < 2; j++) {
            h->lt[i][j] = log(t[i][j]);
            rs = rs + t[i][j];
        }
        h->li[i] = log(ini[i]);
        f[i] = fmax(0.0, 1.0 - rs);
        if (f[i] > 0.0001) freeend = 0;
    }
    for (int k = 0; k < NAA; k++) {
        h->le[0][k] = log(e0[k]);
        h->le[1][k] = log(e1[k]);
    }
    for (int i = 0; i < 2; i++) h->lf[i] = log(freeend ? 1.0 : f[i]);
}

/* T1-T3. main :444-500 (with the -F defect fixed: fg comes from the caller),
 * plaac() :279-291, aahydro2 :90, prionhmm1 :968-981, prionhmm0 :988-1001. */
void oracle_build_params(const double fgfreq_in[NAA], const double bgcounts_in[NAA], double alpha, int corelength,
                         int ww1, int ww2, int ww3, int adjustprolines, oracle_params *P) {
    double fgfreq[NAA], bgf[NAA], mix[NAA], bgcombo[NAA];
    memset(P, 0, sizeof *P);
    if (alpha > 1 || alpha < 0) alpha = 1.0; /* :444-447 (the warning line is the host's business) */
    P->alpha = alpha;
    P->corelength = corelength;
    P->ww1 = ww1;
    P->ww2 = ww2;
    P->ww3 = ww3;
    P->adjustprolines = adjustprolines;
    P->cc[0] = 2.785;
    P->cc[1] = -1;
    P->cc[2] = -1.151; /* :800 */

    normalize22(k_bg_scer, P->bgscer); /* :312 */

    memcpy(fgfreq, fgfreq_in, sizeof fgfreq);
    fgfreq[0] = 0;
    fgfreq[21] = 0; /* :449 */
    normalize22(fgfreq, fgfreq); /* :452 (in-place is safe: sum is taken first) */

    memcpy(bgf, bgcounts_in, sizeof bgf);
    bgf[0] = 0;
    bgf[21] = 0;                 /* :454 */
    normalize22(bgf, P->bgthis); /* :456 */
    for (int i = 0; i < NAA; i++) mix[i] = alpha * P->bgscer[i] + (1 - alpha) * P->bgthis[i]; /* axpby :1981 */
    normalize22(mix, bgcombo);                                                               /* :458 */

    const double epsx = 0.00001; /* :490-495 */
    fgfreq[0] = epsx;
    fgfreq[21] = epsx;
    bgcombo[0] = epsx;
    bgcombo[21] = epsx;
    normalize22(fgfreq, P->fg);  /* :496 */
    normalize22(bgcombo, P->bg); /* :497 */
    for (int j = 1; j < 21; j++) P->llr[j] = log(P->fg[j] / P->bg[j]); /* :500 */

    for (int i = 0; i <= 4000; i++) P->loglut[i] = log(1.0 + exp(-i / 100.0)); /* :283 */
    for (int k = 1; k <= 20; k++) P->lodpapa[k] = log(k_odpapa1[k]);            /* :289 */
    for (int k = 0; k < NAA; k++) {
        P->hydro2[k] = (1.0 / 9.0) * k_hydro[k] + 0.5; /* axpb :2060 via :90 */
        P->charge[k] = k_charge[k];
    }

    double e_bg[NAA], e_fg[NAA];
    normalize22(P->bg, e_bg); /* :974 */
    normalize22(P->fg, e_fg); /* :975 */
    const double t1[2][2] = {{99.9 / 100, 0.1 / 100}, {2.0 / 100, 98.0 / 100}};
    const double i1[2] = {0.9524, 0.0476};
    hmm_setup(&P->hmm1, t1, i1, e_bg, e_fg);
    const double t0[2][2] = {{1, 0}, {0, 1}};
    const double i0[2] = {1, 0};
    hmm_setup(&P->hmm0, t0, i0, e_bg, e_bg); /* :994-995 */
}

/* E1: aatoint, plaac.java:1508-1534 — 20 AAs case-insensitive, '*'->21, everything else->0 */
uint8_t oracle_aatoint(char c) {
    switch (c) {
    case 'A': case 'a': return 1;
    case 'C': case 'c': return 2;
    case 'D': case 'd': return 3;
    case 'E': case 'e': return 4;
    case 'F': case 'f': return 5;
    case 'G': case 'g': return 6;
    case 'H': case 'h': return 7;
    case 'I': case 'i': return 8;
    case 'K': case 'k': return 9;
    case 'L': case 'l': return 10;
    case 'M': case 'm': return 11;
    case 'N': case 'n': return 12;
    case 'P': case 'p': return 13;
    case 'Q': case 'q': return 14;
    case 'R': case 'r': return 15;
    case 'S': case 's': return 16;
    case 'T': case 't': return 17;
    case 'V': case 'v': return 18;
    case 'W': case 'w': return 19;
    case 'Y': case 'y': return 20;
    case '*': return 21;
    default: return 0;
    }
}

void oracle_encode(const char *s, size_t n, uint8_t *out) {
    for (size_t i = 0; i < n; i++) out[i] = oracle_aatoint(s[i]);
}

/* E2: countaas :1698-1706 + isvalidprotein :1732-1739 over untrimmed records.
 * (int64 bins: the reference's 32-bit int overflow is a documented fix) */
void oracle_histogram(const uint8_t *codes, const uint64_t *offsets, uint32_t nprot, int64_t counts[NAA]) {
    for (int k = 0; k < NAA; k++) counts[k] = 0;
    for (uint32_t p = 0; p < nprot; p++) {
        const uint8_t *aa = codes + offsets[p];
        int64_t m = (int64_t)(offsets[p + 1] - offsets[p]);
        if (m < 1) continue; /* the reference throws on an empty record; skipped here */
        int valid = 1;
        for (int64_t i = 1; i < m - 1; i++)
            if (aa[i] == 0 || aa[i] == 21) { valid = 0; break; }
        if (aa[m - 1] == 0) valid = 0;
        if (!valid) continue;
        for (int64_t i = 0; i < m; i++) counts[aa[i]]++;
    }
}

/* LSE with lookup table: logeapeb, plaac.java:1024-1047 */
double oracle_logeapeb(const double *lut, double a, double b) {
    if (a > b) {
        double c = a - b;
        if (!(c < 40)) return a;
        int dex = (int)floor(100 * c);
        return a + ((100 * c - dex) * lut[dex + 1] + (dex + 1 - 100 * c) * lut[dex]);
    } else if (b > a) {
        double c = b - a;
        if (!(c < 40)) return b;
        int dex = (int)floor(100 * c);
        return b + ((100 * c - dex) * lut[dex + 1] + (dex + 1 - 100 * c) * lut[dex]);
    }
    return a + log(2); /* plaac.log2 :30 */
}

/* hss2(seq,min,max), plaac.java:1206-1257, general form (the drivers only call it with min==max) */
static void hss2_buf(const double *seq, int n, int minlength, int maxlength, double out[3], double *psum) {
    if (minlength > n || minlength > maxlength) {
        out[0] = -1.0;
        out[1] = -2.0;
        out[2] = -INFINITY;
        return;
    }
    if (maxlength > n) maxlength = n;
    psum[0] = 0;
    for (int i = 0; i < n; i++) psum[i + 1] = psum[i] + seq[i];
    int beststart = 0, beststop = minlength - 1, curstart = 0, newstart = 0;
    double d = psum[minlength], best = d;
    for (int i = minlength; i < n; i++) {
        if ((i - curstart) >= maxlength) curstart++;
        d = psum[i + 1] - psum[curstart];
        newstart = curstart;
        for (int j = curstart + 1; j < i - minlength; j++) {
            if (psum[i + 1] - psum[j] >= d) {
                d = psum[i + 1] - psum[j];
                newstart = j;
            }
            curstart = newstart;
        }
        if (d > best) {
            best = d;
            beststop = i;
            beststart = curstart;
        }
    }
    out[0] = beststart;
    out[1] = beststop;
    out[2] = best;
}

void oracle_hss2(const double *seq, int n, int minlength, int maxlength, double out[3]) {
    double *psum = (double *)malloc(sizeof(double) * ((size_t)(n > 0 ? n : 0) + 1));
    hss2_buf(seq, n, minlength, maxlength, out, psum);
    free(psum);
}

/* Brute-force fixed-width window over the same rounded prefix sums (the disabled
 * debug cross-check of plaac.java:772-791 uses hss :1073-1114; here used as a property test). */
void oracle_hss_brute(const double *seq, int n, int len, double out[3]) {
    if (len > n) {
        out[0] = -1.0;
        out[1] = -2.0;
        out[2] = -INFINITY;
        return;
    }
    double *psum = (double *)malloc(sizeof(double) * ((size_t)n + 1));
    psum[0] = 0;
    for (int i = 0; i < n; i++) psum[i + 1] = psum[i] + seq[i];
    int bs = 0;
    double best = psum[len];
    for (int s = 1; s + len <= n; s++) {
        double d = psum[s + len] - psum[s];
        if (d > best) {
            best = d;
            bs = s;
        }
    }
    free(psum);
    out[0] = bs;
    out[1] = bs + len - 1;
    out[2] = best;
}

/* slidingaverage(arr,ww,shrink=true,weight=false[,mergeme=13,seq]), plaac.java:2585-2622 / :2626-2662 */
static void window_mean_shrink(const double *arr, int n, int ww, const uint8_t *seq_or_null, double *sa) {
    int w = ww / 2;
    if (w >= n) w = n - 1;
    for (int i = 0; i < n; i++) {
        double score = 0.0, denom = 0.0;
        for (int j = -w; j <= w; j++) {
            int p = i + j;
            if (p >= 0 && p < n) {
                denom = denom + 1.0;
                if (seq_or_null) {
                    const uint8_t *s = seq_or_null;
                    int dup1 = (s[p] == 13) && (p >= 1) && (s[p - 1] == 13);
                    int dup2 = (s[p] == 13) && (p >= 2) && (s[p - 2] == 13);
                    if (!dup1 && !dup2) score = score + 1.0 * arr[p];
                } else {
                    score = score + 1.0 * arr[p];
                }
            }
        }
        sa[i] = score / denom;
    }
}

/* slidingaverage(arr,ww,shrink=false,weight=true), plaac.java:2585-2622 */
static void window_mean_weighted(const double *arr, int n, int ww, double *sa) {
    int w = ww / 2;
    if (w >= n) w = n - 1;
    int mini = w, maxi = n - w - 1;
    for (int i = 0; i < mini && i < n; i++) sa[i] = NAN;
    for (int i = (maxi + 1 > 0 ? maxi + 1 : 0); i < n; i++) sa[i] = NAN;
    for (int i = mini; i <= maxi; i++) {
        double score = 0.0, denom = 0.0;
        for (int j = -w; j <= w; j++) {
            int p = i + j;
            if (p >= 0 && p < n) {
                int m1 = p < w ? p : w;
                int m2 = (n - p - 1) < w ? (n - p - 1) : w;
                double wt = 1.0 + m1 + m2;
                denom = denom + wt;
                score = score + wt * arr[p];
            }
        }
        sa[i] = score / denom;
    }
}

typedef struct scratch {
    double *d[16];
    int32_t *tb;
    uint8_t *b[2];
    size_t cap;
} scratch;

static void scratch_reserve(scratch *S, size_t n) {
    if (n <= S->cap) return;
    for (int k = 0; k < 16; k++) {
        free(S->d[k]);
        S->d[k] = (double *)malloc(sizeof(double) * (n + 1));
    }
    free(S->tb);
    S->tb = (int32_t *)malloc(sizeof(int32_t) * 2 * n);
    for (int k = 0; k < 2; k++) {
        free(S->b[k]);
        S->b[k] = (uint8_t *)malloc(n + 1);
    }
    S->cap = n;
}

static void scratch_free(scratch *S) {
    for (int k = 0; k < 16; k++) free(S->d[k]);
    free(S->tb);
    free(S->b[0]);
    free(S->b[1]);
    memset(S, 0, sizeof *S);
}

/* viterbidecodel, plaac.java:3077-3121 (2 states; strict '>' keeps state 0 on ties) */
static double viterbi2(const oracle_hmm *h, const uint8_t *x, int n, double *s0, double *s1, int32_t *tb,
                       uint8_t *path) {
    s0[0] = h->li[0] + h->le[0][x[0]];
    s1[0] = h->li[1] + h->le[1][x[0]];
    for (int t = 1; t < n; t++) {
        for (int i = 0; i < 2; i++) {
            int arg = 0;
            double best = h->lt[0][i] + s0[t - 1];
            if (h->lt[1][i] + s1[t - 1] > best) {
                best = h->lt[1][i] + s1[t - 1];
                arg = 1;
            }
            (i ? s1 : s0)[t] = best + h->le[i][x[t]];
            tb[2 * (size_t)t + i] = arg;
        }
    }
    int arg = 0;
    double best = s0[n - 1] + h->lf[0];
    if (s1[n - 1] + h->lf[1] > best) {
        best = s1[n - 1] + h->lf[1];
        arg = 1;
    }
    path[n - 1] = (uint8_t)arg;
    for (int t = n - 2; t >= 0; t--) path[t] = (uint8_t)tb[2 * (size_t)(t + 1) + path[t + 1]];
    return best;
}

/* posteriorl forward part, plaac.java:3354-3375 */
static double forward2(const oracle_params *P, const oracle_hmm *h, const uint8_t *x, int n, double *a0, double *a1) {
    const double *lut = P->loglut;
    a0[0] = h->li[0] + h->le[0][x[0]];
    a1[0] = h->li[1] + h->le[1][x[0]];
    for (int t = 1; t < n; t++) {
        for (int i = 0; i < 2; i++) {
            double score = -INFINITY;
            score = oracle_logeapeb(lut, score, h->lt[0][i] + a0[t - 1]);
            score = oracle_logeapeb(lut, score, h->lt[1][i] + a1[t - 1]);
            (i ? a1 : a0)[t] = score + h->le[i][x[t]];
        }
    }
    double tot = -INFINITY;
    tot = oracle_logeapeb(lut, tot, a0[n - 1] + h->lf[0]);
    tot = oracle_logeapeb(lut, tot, a1[n - 1] + h->lf[1]);
    return tot;
}

/* posteriorl backward + posterior :3377-3405, mapdecodel :4032-4045 */
static void backward_posterior2(const oracle_params *P, const oracle_hmm *h, const uint8_t *x, int n, const double *a0,
                                const double *a1, double *b0, double *b1, double *pp0, double *pp1, uint8_t *map) {
    const double *lut = P->loglut;
    b0[n - 1] = h->lf[0];
    b1[n - 1] = h->lf[1];
    for (int t = n - 2; t >= 0; t--) {
        for (int i = 0; i < 2; i++) {
            double score = -INFINITY;
            score = oracle_logeapeb(lut, score, h->lt[i][0] + b0[t + 1] + h->le[0][x[t + 1]]);
            score = oracle_logeapeb(lut, score, h->lt[i][1] + b1[t + 1] + h->le[1][x[t + 1]]);
            (i ? b1 : b0)[t] = score;
        }
    }
    double lpseq = -INFINITY;
    lpseq = oracle_logeapeb(lut, lpseq, a0[0] + b0[0]);
    lpseq = oracle_logeapeb(lut, lpseq, a1[0] + b1[0]);
    for (int t = 0; t < n; t++) {
        pp0[t] = exp((a0[t] + b0[t]) - lpseq);
        pp1[t] = exp((a1[t] + b1[t]) - lpseq);
        map[t] = (uint8_t)(pp1[t] > pp0[t] ? 1 : 0);
    }
}

static void score_protein_s(const oracle_params *P, const uint8_t *aa, int n, oracle_row *row, oracle_tracks *tr,
                            scratch *S) {
    if ((size_t)n > S->cap) scratch_reserve(S, (size_t)n + (size_t)n / 4 + 64);
    double *m = S->d[0], *s0 = S->d[1], *s1 = S->d[2], *a0 = S->d[3], *a1 = S->d[4];
    double *hydro = S->d[5], *charge = S->d[6], *fi = S->d[7], *pllr = S->d[8], *papa = S->d[9];
    double *fix2 = S->d[10], *pllrx2 = S->d[11], *papax2 = S->d[12];
    double hs[3];
    const int c = P->corelength;
    memset(row, 0, sizeof *row);
    row->prot_len = n;

    /* W1: MW, plaac.java:767-771 (qnmask: 1.0 for N=12 and Q=14) */
    for (int i = 0; i < n; i++) m[i] = (aa[i] == 12 || aa[i] == 14) ? 1.0 : 0.0;
    int mw = n < 80 ? n : 80;
    hss2_buf(m, n, mw, mw, hs, S->d[15]);
    row->mw_score = (int32_t)hs[2]; /* (int) inf2nan(.) :901; never infinite since mw<=n */
    row->mw_start = (int32_t)hs[0];
    row->mw_end = (int32_t)hs[1];

    /* W2: LLR window, :782-783 */
    for (int i = 0; i < n; i++) m[i] = P->llr[aa[i]];
    hss2_buf(m, n, c, c, hs, S->d[15]);
    row->llr_score = hs[2];
    row->llr_start = (int32_t)hs[0];
    row->llr_end = (int32_t)hs[1];

    /* H1, H2, H4, H5: :794-798 */
    uint8_t *vit = S->b[0];
    double lvit1 = viterbi2(&P->hmm1, aa, n, s0, s1, S->tb, vit);
    double lmarg1 = forward2(P, &P->hmm1, aa, n, a0, a1);
    if (tr) {
        double *b0 = S->d[13], *b1 = S->d[14];
        backward_posterior2(P, &P->hmm1, aa, n, a0, a1, b0, b1, tr->post0, tr->post1, tr->map);
        memcpy(tr->vit, vit, (size_t)n);
    }
    /* hmm0 through the same general code (it degenerates to a running sum) */
    uint8_t *vit0 = S->b[1];
    double lvit0 = viterbi2(&P->hmm0, aa, n, s0, s1, S->tb, vit0);
    double lmarg0 = forward2(P, &P->hmm0, aa, n, a0, a1);
    row->hmm_all = lmarg1 - lmarg0;
    row->hmm_vit = lvit1 - lvit0;

    /* P1: longestrun :1787-1804, masked core :818-833, PRD :861-880 */
    int maxrun = 0;
    for (int i = 0; i < n;) {
        if (vit[i] > 0) {
            int st = i++;
            while (i < n && vit[i] > 0) i++;
            if (i - st >= maxrun) maxrun = i - st;
        } else i++;
    }
    row->vit_maxrun = maxrun;
    const double big_neg = -1000000.0;
    for (int i = 0; i < n; i++) m[i] = vit[i] == 0 ? big_neg : P->llr[aa[i]];
    hss2_buf(m, n, c, c, hs, S->d[15]);
    int corestart = (int)hs[0], corestop = (int)hs[1];
    int aastart = corestart, aastop = corestop;
    double prdscore = 0;
    if (hs[2] > big_neg / 2) {
        while (aastart >= 0 && vit[aastart] == 1) aastart--;
        aastart++;
        while (aastop < n && vit[aastop] == 1) aastop++;
        aastop--;
        for (int k = aastart; k <= aastop; k++) prdscore = prdscore + P->llr[aa[k]];
    } else {
        hs[2] = NAN;
        aastart = -1;
        aastop = -2;
        corestart = -1;
        corestop = -2;
    }
    row->core_score = hs[2];
    row->core_start = corestart;
    row->core_end = corestop;
    row->prd_score = prdscore;
    row->prd_start = aastart;
    row->prd_end = aastop;

    /* D1/D2: disorderreport :4877-4887; mean :1584-1588 */
    double sm = 0;
    for (int i = 0; i < n; i++) { m[i] = P->hydro2[aa[i]]; sm = sm + m[i]; }
    double meanhydro = (1.0 * sm) / n;
    window_mean_shrink(m, n, P->ww1, NULL, hydro);
    sm = 0;
    for (int i = 0; i < n; i++) { m[i] = P->charge[aa[i]]; sm = sm + m[i]; }
    double meancharge = (1.0 * sm) / n;
    window_mean_shrink(m, n, P->ww1, NULL, charge);
    double meanfi = P->cc[2] + P->cc[1] * fabs(meancharge) + P->cc[0] * meanhydro;
    for (int i = 0; i < n; i++) fi[i] = P->cc[0] * hydro[i] + P->cc[1] * fabs(charge[i]) + P->cc[2]; /* axpbypc :2050 */
    row->fi_meanhydro = meanhydro;
    row->fi_meancharge = meancharge;
    row->fi_meancombo = meanfi;

    /* D1 (llr track), D3 (PAPA with proline adjustment) :4889-4897 */
    for (int i = 0; i < n; i++) m[i] = P->llr[aa[i]];
    window_mean_shrink(m, n, P->ww3, NULL, pllr);
    for (int i = 0; i < n; i++) m[i] = P->lodpapa[aa[i]];
    window_mean_shrink(m, n, P->ww2, P->adjustprolines ? aa : NULL, papa);

    /* D4: :4903-4905 */
    window_mean_weighted(papa, n, P->ww2, papax2);
    window_mean_weighted(pllr, n, P->ww3, pllrx2);
    window_mean_weighted(fi, n, P->ww1, fix2);

    /* D5: papamode 1, :4932-4948, :4986-4997 */
    double pmax = -INFINITY;
    int pcen = -1;
    for (int k = (P->ww2 - 1) / 2; k < n - (P->ww2 - 1) / 2; k++) {
        double ps = papax2[k];
        if ((ps > pmax) & (fix2[k] < 0)) {
            pcen = k;
            pmax = ps;
        }
    }
    row->papa_combo = pmax;
    row->papa_cen = pcen;
    row->papa_prop = row->papa_fi = row->papa_llr = row->papa_llr2 = NAN;
    if (pcen >= 0) {
        row->papa_prop = papax2[pcen];
        row->papa_fi = fix2[pcen];
        row->papa_llr2 = pllrx2[pcen];
        row->papa_llr = pllr[pcen];
    }

    /* D6: strict2 run scan :5010-5059 (minlen = 5); maxint over the zero-initialised lenaa :5060 */
    int halfw = (P->ww1 - 1) / 2;
    if (halfw > n / 2) halfw = n / 2;
    int numstrict2 = 0, maxlen = 0;
    for (int i = halfw; i < n - halfw;) {
        if (fi[i] < 0) {
            int startdex = i++;
            while (i < n - halfw && fi[i] < 0) i++;
            int stopdex = i - 1;
            if (startdex == halfw) startdex = 0;
            if (stopdex == n - halfw - 1) stopdex = n - 1;
            int len = stopdex - startdex + 1;
            if (len >= 5) {
                numstrict2 += len;
                if (len > maxlen) maxlen = len;
            }
        } else i++;
    }
    row->fi_numaa = numstrict2;
    row->fi_maxrun = maxlen;

    if (tr) {
        memcpy(tr->charge, charge, sizeof(double) * n);
        memcpy(tr->hydro, hydro, sizeof(double) * n);
        memcpy(tr->fi, fi, sizeof(double) * n);
        memcpy(tr->plaacllr, pllr, sizeof(double) * n);
        memcpy(tr->papa, papa, sizeof(double) * n);
        memcpy(tr->fix2, fix2, sizeof(double) * n);
        memcpy(tr->plaacllrx2, pllrx2, sizeof(double) * n);
        memcpy(tr->papax2, papax2, sizeof(double) * n);
    }
}

void oracle_score_protein(const oracle_params *P, const uint8_t *aa, int n, oracle_row *row, oracle_tracks *tr) {
    scratch S;
    memset(&S, 0, sizeof S);
    score_protein_s(P, aa, n, row, tr, &S);
    scratch_free(&S);
}

static void offset_tracks(const oracle_tracks *in, uint64_t off, oracle_tracks *out) {
    out->vit = in->vit + off;
    out->map = in->map + off;
    out->charge = in->charge + off;
    out->hydro = in->hydro + off;
    out->fi = in->fi + off;
    out->plaacllr = in->plaacllr + off;
    out->papa = in->papa + off;
    out->fix2 = in->fix2 + off;
    out->plaacllrx2 = in->plaacllrx2 + off;
    out->papax2 = in->papax2 + off;
    out->post0 = in->post0 + off;
    out->post1 = in->post1 + off;
}

void oracle_score_batch(const oracle_params *P, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot,
                        oracle_row *rows, oracle_tracks *tracks, int nthreads) {
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        scratch S;
        memset(&S, 0, sizeof S);
#pragma omp for schedule(dynamic, 16)
        for (int64_t p = 0; p < (int64_t)nprot; p++) {
            const uint8_t *aa = codes + offsets[p];
            int64_t n = (int64_t)(offsets[p + 1] - offsets[p]);
            if (n > 0 && aa[n - 1] == 21) n--; /* kill one terminal stop, :758 */
            if (n < 1) {                        /* :762 (and empty records, which the reference cannot survive) */
                memset(&rows[p], 0, sizeof rows[p]);
                continue;
            }
            oracle_tracks t, *tp = NULL;
            if (tracks) {
                offset_tracks(tracks, offsets[p], &t);
                tp = &t;
            }
            score_protein_s(P, aa, (int)n, &rows[p], tp, &S);
        }
        scratch_free(&S);
    }
}
