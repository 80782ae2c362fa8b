/*
 * oracle/plaac_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * Scalar CPU restatement of the PLAAC scoring hot path (reference:
 * /root/reference/cli/src/plaac.java, cited per function in plaac_oracle.c).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported baseline. The product
 * (plaac_amd/csrc, libplaac_native.so) never links, loads or calls it.
 *
 * Pinning status: the reference has no test-suite and cannot run here (no JVM).
 *   - Viterbi parse boundaries (tables T1/T2, encoding E1, Viterbi H1): PINNED by
 *     the 28 [start-end] annotations of cli/src/scer_fg_28.fasta
 *     (tests/golden/kat28.tsv, 28/28 reproduced).
 *   - every float output, forward-backward, FoldIndex, PAPA: PARITY UNPINNED
 *     (no reference golden values exist for them).
 */
#ifndef PLAAC_ORACLE_H
#define PLAAC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_NAA 22
#define ORACLE_LUTLEN 4001

typedef struct oracle_hmm {
    double lt[2][2];          /* log transition  [from][to]            */
    double li[2];             /* log initial                           */
    double le[2][ORACLE_NAA]; /* log emission    [state][code]         */
    double lf[2];             /* log end-transition (0.0: free end)    */
} oracle_hmm;

typedef struct oracle_params {
    int32_t corelength, ww1, ww2, ww3, adjustprolines, pad_;
    double alpha;
    double cc[3];
    double fg[ORACLE_NAA];      /* "## fg_used"   */
    double bgscer[ORACLE_NAA];  /* "## bg_scer"   */
    double bgthis[ORACLE_NAA];  /* "## bg_input"  */
    double bg[ORACLE_NAA];      /* "## bg_used"   */
    double llr[ORACLE_NAA];     /* "## plaac_llr" */
    double lodpapa[ORACLE_NAA]; /* "## papa_lods" */
    double hydro2[ORACLE_NAA];
    double charge[ORACLE_NAA];
    oracle_hmm hmm1, hmm0;
    double loglut[ORACLE_LUTLEN];
} oracle_params;

/* Raw per-protein results, zero-based indices, reference sentinels
 * (-1/-2 for "none", -inf / NaN scores) exactly as the Java locals hold them
 * before formatting (plaac.java:899-945). */
typedef struct oracle_row {
    double llr_score, core_score, prd_score, hmm_all, hmm_vit;
    double fi_meanhydro, fi_meancharge, fi_meancombo;
    double papa_combo, papa_prop, papa_fi, papa_llr, papa_llr2;
    int32_t mw_score, mw_start, mw_end;
    int32_t llr_start, llr_end;
    int32_t vit_maxrun;
    int32_t core_start, core_end;
    int32_t prd_start, prd_end;
    int32_t prot_len;
    int32_t fi_numaa, fi_maxrun;
    int32_t papa_cen;
} oracle_row;

/* Per-residue tracks (plaac.java:635-643); each array has room for n entries. */
typedef struct oracle_tracks {
    uint8_t *vit, *map;
    double *charge, *hydro, *fi, *plaacllr, *papa, *fix2, *plaacllrx2, *papax2;
    double *post0, *post1;
} oracle_tracks;

/* constant tables of the reference (data, plaac.java:261-270) */
void oracle_const_tables(double bg_scer[ORACLE_NAA], double fg28[ORACLE_NAA], double fg04[ORACLE_NAA]);

/* T1-T3: plaac.java:279-291, 444-500, 968-1001, 2893-2935 */
void oracle_build_params(const double fgfreq_in[ORACLE_NAA], const double bgcounts_in[ORACLE_NAA],
                         double alpha, int corelength, int ww1, int ww2, int ww3, int adjustprolines,
                         oracle_params *out);

/* E1: plaac.java:1508-1534 */
uint8_t oracle_aatoint(char c);
void oracle_encode(const char *s, size_t n, uint8_t *out);

/* E2: plaac.java:1655-1666, 1698-1706, 1732-1739. codes are UNTRIMMED records. */
void oracle_histogram(const uint8_t *codes, const uint64_t *offsets, uint32_t nprot, int64_t counts[ORACLE_NAA]);

/* LSE: plaac.java:1024-1047 */
double oracle_logeapeb(const double *loglut, double a, double b);

/* H-hss: plaac.java:1206-1257 (out = start, end, score) */
void oracle_hss2(const double *seq, int n, int minlength, int maxlength, double out[3]);
/* brute-force cross-check, plaac.java:1073-1114 idea (fixed width) */
void oracle_hss_brute(const double *seq, int n, int len, double out[3]);

/* Whole per-protein path: plaac.java:759-880 + disorderreport :4866-5068.
 * aa = codes AFTER the one-trailing-stop trim, n >= 1. tracks may be NULL. */
void oracle_score_protein(const oracle_params *P, const uint8_t *aa, int n, oracle_row *row, oracle_tracks *tr);

/* Batch over UNTRIMMED records (trims one trailing code 21 like :758). Rows of
 * records with n<1 after the trim get prot_len=0 and are otherwise zeroed.
 * tracks (nullable) are SoA arrays over the untrimmed residue index space
 * (entries of a trimmed stop are left untouched). nthreads<=1: serial. */
void oracle_score_batch(const oracle_params *P, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot,
                        oracle_row *rows, oracle_tracks *tracks, int nthreads);

size_t oracle_sizeof_params(void);
size_t oracle_sizeof_row(void);

#ifdef __cplusplus
}
#endif
#endif
