/*
 * plaac_native.h — C ABI of the MI355X-native PLAAC scoring engine (libplaac_native.so).
 *
 * This is the drop-in boundary for the reference's per-protein scoring loops
 * (reference = cli/src/plaac.java of whitehead/plaac). The reference has no FFI of its own:
 * the seam is the body of `scoreallfastas` (plaac.java:759-880) and `plotsomefastas`
 * (plaac.java:625-633). A JNI / ctypes / C++ host encodes FASTA records to residue codes,
 * calls plaac_score once per batch and formats rows (see INTEGRATION.md for the JNI stub).
 *
 * Conventions
 *   - plain C, no exceptions across the boundary; every call returns a plaac_status and
 *     leaves a message retrievable with plaac_last_error().
 *   - caller owns every buffer; the ctx owns device memory and streams. A ctx is
 *     single-caller; distinct ctxs are independent; there are no process globals.
 *   - residue codes 0..21 = X A C D E F G H I K L M N P Q R S T V W Y *  (plaac.java:26).
 *   - a batch is `codes` (all records concatenated, UNTRIMMED) + `offsets[nprot+1]`.
 *     Scoring drops exactly one trailing stop code (21) per record (plaac.java:758);
 *     the background histogram does not (plaac.java:1698-1706).
 *   - results do not depend on batch split, protein order or device count.
 *   - there is NO CPU fallback: every compute entry point runs the HIP kernels or fails.
 */
#ifndef PLAAC_NATIVE_H
#define PLAAC_NATIVE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 5): plaac_node_batch_last_error; node batches are detached (not dangling) when their node is destroyed first;
 * plaac_node_batch_upload rejects offsets[0] != 0; wire rows (plaac_rows_to_wire / _from_wire) for the cross-process gather;
 * plaac_score_begin_counting / plaac_score_end_counts (scoring and background counts in one pass). A binding
 * compares plaac_abi_version() with the PLAAC_ABI_VERSION it was compiled against before its first call. */
#define PLAAC_ABI_VERSION 2
#define PLAAC_NAA 22
#define PLAAC_LUTLEN 4001

typedef enum plaac_status {
    PLAAC_OK = 0,
    PLAAC_ERR_ARG = 1,    /* bad argument (null pointer, non-monotone offsets, code > 21 ...) */
    PLAAC_ERR_DEVICE = 2, /* HIP runtime / kernel failure, or no gfx950 device                */
    PLAAC_ERR_NOMEM = 3,  /* host or device allocation failed                                 */
    PLAAC_ERR_IO = 4,     /* file could not be read (host helpers only)                       */
    PLAAC_ERR_UNSUPPORTED = 5 /* a diagnostic-build switch asked of the release library (plaac_debug_set_knob) */
} plaac_status;

/* Log-space 2-state HMM as hmm.initialize leaves it (plaac.java:2893-2935). */
typedef struct plaac_hmm {
    double lt[2][2];         /* ltprob[from][to] */
    double li[2];            /* liprob           */
    double le[2][PLAAC_NAA]; /* leprob[state][code] */
    double lf[2];            /* lfprob (0.0: free end) */
} plaac_hmm;

/* Everything the kernels need, so the device never calls log():
 * replaces the locals of main (plaac.java:444-500), the statics loglut/lodpapa1/aahydro2/aacharge
 * (:33, :201, :90, :37) and the two hmm objects (:517-518). */
typedef struct plaac_params {
    int32_t corelength;      /* -c, default 60 (:319) */
    int32_t ww1, ww2, ww3;   /* -w, -W, ww3=ww2 (:320-322, :355) */
    int32_t adjustprolines;  /* always true in the reference (:329) */
    int32_t reserved_;
    double alpha;            /* -a after the [0,1] clamp (:444-447) */
    double cc[3];            /* FoldIndex coefficients {2.785,-1,-1.151} (:800) */
    double big_neg;          /* -1000000.0 (:819) */
    double fg[PLAAC_NAA];      /* "## fg_used"   */
    double bgscer[PLAAC_NAA];  /* "## bg_scer"   */
    double bgthis[PLAAC_NAA];  /* "## bg_input"  */
    double bg[PLAAC_NAA];      /* "## bg_used"   */
    double llr[PLAAC_NAA];     /* "## plaac_llr" */
    double lodpapa[PLAAC_NAA]; /* "## papa_lods" */
    double hydro2[PLAAC_NAA];  /* aahydro2 */
    double charge[PLAAC_NAA];  /* aacharge */
    plaac_hmm hmm1;            /* prionhmm1 (:968-981) */
    plaac_hmm hmm0;            /* prionhmm0 (:988-1001) */
    double loglut[PLAAC_LUTLEN]; /* loglut (:283) */
} plaac_params;

/* One summary row per record: the raw values behind the 38 printed columns of
 * scoreallfastas (plaac.java:899-945). Indices are ZERO-based with the reference's
 * sentinels (start -1 / end -2 when there is none); the host adds 1 when printing.
 * prot_len == 0 marks a record that is skipped (empty after the stop trim, :762). */
typedef struct plaac_row {
    double llr_score;     /* LLR      (-inf when n < corelength)            */
    double core_score;    /* COREscore (NaN when no core)                   */
    double prd_score;     /* PRDscore (0.0 when no core)                    */
    double hmm_all;       /* HMMall                                         */
    double hmm_vit;       /* HMMvit                                         */
    double fi_meanhydro;  /* FImeanhydro                                    */
    double fi_meancharge; /* FImeancharge                                   */
    double fi_meancombo;  /* FImeancombo                                    */
    double papa_combo;    /* PAPAcombo (-inf when no centre; printed NaN)   */
    double papa_prop;     /* PAPAprop                                       */
    double papa_fi;       /* PAPAfi                                         */
    double papa_llr;      /* PAPAllr                                        */
    double papa_llr2;     /* PAPAllr2                                       */
    int32_t mw_score, mw_start, mw_end;
    int32_t llr_start, llr_end;
    int32_t vit_maxrun;
    int32_t core_start, core_end;
    int32_t prd_start, prd_end;
    int32_t prot_len;
    int32_t fi_numaa, fi_maxrun;
    int32_t papa_cen;
} plaac_row; /* 13 x f64 + 14 x i32 = 160 bytes */

/* WIRE ROWS (round 5): what a summary row needs to cross a link (xGMI between ranks, a socket between hosts). Of the 160
 * bytes the receiver can rebuild 24 from what it holds anyway - the batch's offsets and the core length:
 *   prot_len  = offsets[i+1] - offsets[i] - (one trailing stop trimmed, plaac.java:758: a flag in bit 30 of mw_score)
 *   mw_end    = mw_start + min(80, prot_len) - 1            (hss2 with min == max, :769-771, :1253-1256)
 *   llr_end   = llr_start + corelength - 1, core_end likewise; -2 where the start is -1 (:1210-1215, :873-880)
 *   papa_prop = papa_combo when there is a centre, NaN otherwise (:4944-4946, :4993)
 * so a wire row is 12 doubles + 10 ints = 136 bytes (the field order of plaac_row without those five). A skipped record
 * (prot_len 0) comes back as the all-zero row the scorer writes. plaac_rows_from_wire(plaac_rows_to_wire(rows)) == rows
 * byte for byte (tests/test_dist.py; plaac_amd/dist.py does the same on the device for the RCCL gather). Host only. */
#define PLAAC_WIRE_ROW_BYTES 136
plaac_status plaac_rows_to_wire(const plaac_row *rows, const uint64_t *offsets, uint32_t n, uint8_t *wire);
plaac_status plaac_rows_from_wire(const uint8_t *wire, const uint64_t *offsets, uint32_t n, int32_t corelength,
                                  plaac_row *rows);

/* Per-residue tracks (plotsomefastas, plaac.java:635-643): SoA arrays indexed by the
 * position of the residue in `codes` (the entry of a trimmed stop is unspecified).
 * All twelve pointers must be non-null when a tracks struct is passed. */
typedef struct plaac_tracks {
    uint8_t *vit, *map;
    double *charge, *hydro, *fi, *plaacllr, *papa, *fix2, *plaacllrx2, *papax2;
    double *post0, *post1; /* HMM.background, HMM.PrD-like */
} plaac_tracks;

typedef struct plaac_ctx plaac_ctx;

int plaac_abi_version(void);
size_t plaac_sizeof_params(void);
size_t plaac_sizeof_row(void);

/* ---- host-side parameter setup (no device needed) ------------------------------------ */

/* Built-in tables of the reference: bg_freq_scer (:261), prd_freq_scer_28 (:269), prd_freq_scer_04 (:265). */
void plaac_builtin_tables(double bg_scer[PLAAC_NAA], double fg28[PLAAC_NAA], double fg04[PLAAC_NAA]);

/* Table setup of main (plaac.java:444-500) + plaac() (:279-291) + prionhmm1/0 (:968-1001).
 * fgfreq: foreground frequencies or counts (NULL = prd_freq_scer_28); bgcounts: background counts or
 * frequencies of the scored organism, i.e. what -B / -b / -i produce (NULL = all zero).
 * alpha outside [0,1] is replaced by 1.0 (the caller prints the reference's warning line). */
plaac_status plaac_params_init(plaac_params *out, const double fgfreq[PLAAC_NAA], const double bgcounts[PLAAC_NAA],
                               double alpha, int corelength, int ww1, int ww2, int ww3, int adjustprolines);

/* aatoint / string2aa (plaac.java:1508-1534, :1764-1769): text -> codes, one byte per residue. */
void plaac_encode(const char *text, size_t n, uint8_t *codes);

/* ---- device context -------------------------------------------------------------------- */

/* Binds a gfx950 device, uploads the tables, creates streams. Fails (PLAAC_ERR_DEVICE) when no GPU is usable. */
plaac_status plaac_ctx_create(const plaac_params *params, int device_id, plaac_ctx **out);
/* Re-upload tables (parameter sweeps reuse one ctx and one resident batch). */
plaac_status plaac_ctx_set_params(plaac_ctx *ctx, const plaac_params *params);
void plaac_ctx_destroy(plaac_ctx *ctx);
/* Message of the last failing call on this ctx (ctx == NULL: last failing ctx_create on this thread). */
const char *plaac_last_error(const plaac_ctx *ctx);

/* ---- the hot path ---------------------------------------------------------------------- */

/* computeaafreq / countaas / isvalidprotein (plaac.java:1655-1666, :1698-1706, :1732-1739):
 * 22-bin residue histogram over the valid records of the batch (raw counts, bins 0 and 21 included). */
plaac_status plaac_histogram(plaac_ctx *ctx, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot,
                             int64_t counts[PLAAC_NAA]);

/* Body of the per-protein loops (plaac.java:759-880 and :625-633) for a whole batch.
 * Host buffers in, host buffers out; rows in input order. tracks may be NULL (summary mode). */
plaac_status plaac_score(plaac_ctx *ctx, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot,
                         plaac_row *rows, const plaac_tracks *tracks);

/* Pipelined form of plaac_score (summary mode) for hosts that stream batches through one context - the reference's
 * loop over records (plaac.java:755) as a loop over batches: plaac_score_begin copies the batch to the device on a copy
 * stream of the context's own (beside the kernels of the batch before it), enqueues its scoring kernels and returns;
 * plaac_score_end waits for the OLDEST batch begun and copies its rows out (input order). At most two batches may be
 * pending. With plaac_ctx_set_overlap on, the device work of consecutive batches overlaps as well (bin/plaac does both:
 * one context per GPU, two batches in flight). The buffers handed to _begin may be reused as soon as it returns. */
plaac_status plaac_score_begin(plaac_ctx *ctx, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot);
plaac_status plaac_score_end(plaac_ctx *ctx, plaac_row *rows);
/* The same pair with the reference's FIRST pass folded in (round 5): the batch's 22 background counts (computeaafreq,
 * plaac.java:1655-1666: the histogram over its valid records) are taken on the device copy the scoring kernels read and come
 * back with the rows. For hosts whose scoring tables do not depend on the counts - alpha = 1 (plaac.java:458: the input's
 * background then only appears in the "## bg_input" line of the parameter block) - the reference's two passes over the input
 * become one: bin/plaac runs this way by default. A batch begun with plaac_score_begin_counting must be collected with
 * plaac_score_end_counts. */
plaac_status plaac_score_begin_counting(plaac_ctx *ctx, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot);
plaac_status plaac_score_end_counts(plaac_ctx *ctx, plaac_row *rows, int64_t counts[PLAAC_NAA]);
/* The same pipeline fed with FASTA TEXT (round 5; SURVEY.md section 2, K1 "encode / validate"): the host only finds where the
 * records begin - text[starts[i]] is the '>' of record i, starts[nrec] = text_len: whole records, as plaac_fasta_next_text
 * (plaac_host.h) cuts them - and keeps their names; the DEVICE splits the lines, drops the line terminators, stops a record
 * at an empty line and encodes the residues (fastareader plaac.java:4302-4375, string2aa / aatoint :1764-1769, :1508-1534,
 * exactly as plaac_fasta_next / plaac_encode do on the host - tested against them). counting != 0: the background counts as
 * in plaac_score_begin_counting. plaac_score_end_text returns the rows and, for the host's formatter, what the device parsed:
 * codes (NULLABLE; capacity codes_cap bytes, text_len always suffices), offsets (nrec + 1), blank_end (nrec: 1 = the record's
 * sequence ended at an empty line, which makes the reference trim the NEXT record's name, :4362), extents (nullable, 2 * nrec:
 * per record the positions, relative to its start, of the header's line end and of the empty line that ended it - or the
 * record's length; its residues are the bytes between them that are not line ends), counts (nullable). A host that prints a
 * few residues per record (the summary table: the PAPA window, the sequences of the records with a PrD) leaves `codes` NULL
 * and reads them from the text it still holds with plaac_fasta_text_codes (plaac_host.h) - bin/plaac does: a copy of every
 * code back to the host costs the collecting thread more than the parse it saved. */
plaac_status plaac_score_begin_text(plaac_ctx *ctx, const char *text, uint64_t text_len, const uint64_t *starts, uint32_t nrec,
                                    int counting);
plaac_status plaac_score_end_text(plaac_ctx *ctx, plaac_row *rows, uint8_t *codes, uint64_t codes_cap, uint64_t *offsets,
                                  uint8_t *blank_end, uint32_t *extents, int64_t *counts);
/* The oldest pending TEXT batch's summary rows as the TABLE'S TEXT, made on the device (round 5): scoreallfastas' output line
 * (plaac.java:899-945) for every record of the batch, exactly the bytes plaac_format_summary_row_n writes on the host (tested
 * byte for byte), names taken from the text (trimmed where the reference trims them, :4362; prev_blank: how the record before
 * this batch ended, 1 for the first batch of a file). plaac_score_end_text_table_size waits for the batch and reports the
 * table's size, how the batch's last record ended, and whether the host must format this batch itself (*needs_host != 0: a
 * value of 1e9 or more, an infinity the reference prints as such, a record without a sequence, for which the host prints
 * a note - then collect the batch with plaac_score_end_text and format as before; the device never guesses).
 * plaac_score_end_text_table copies the text into `table` (capacity >= the reported size) and gives the slot up. */
plaac_status plaac_score_end_text_table_size(plaac_ctx *ctx, int corelength, int ww2, int prev_blank, uint64_t *table_bytes,
                                             int *needs_host, int *last_blank, uint64_t *residues /* nullable: the batch's */);
plaac_status plaac_score_end_text_table(plaac_ctx *ctx, char *table, uint64_t table_cap, int64_t *counts);
/* plotsomefastas' PER-RESIDUE table (plaac.java:587-649; the lines of :635-645) for a batch of selected records, made on the
 * device (round 5, late): scored in track mode like plaac_score with tracks, but the twelve arrays stay on the device and the
 * table's text comes back - for record k the residues 1 .. prot_len, each "label_k \t AANUM \t AA \t VIT \t MAP \t" + the eight
 * window tracks with the decimals of :638 + the two posteriors, then the closing line of 56 '#'. labels: for record k the
 * bytes of "ORDER \t SEQid" at label_off[k] .. label_off[k+1]. rows (nullable): the batch's summary rows as well. *table is
 * malloc'ed (plaac_table_free) - or NULL with *needs_host != 0 when a value needs the host's formatter (>= 1e9, an infinity):
 * then plaac_score with tracks and plaac_format_track_rows as before. No batch may be pending on the context. */
plaac_status plaac_score_tracks_table(plaac_ctx *ctx, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot, const char *labels,
                                      const uint64_t *label_off, plaac_row *rows, char **table, uint64_t *table_len, int *needs_host);
void plaac_table_free(char *table);
/* A text batch uploaded and parsed AHEAD of its scoring call by ANOTHER host thread (round 5, late): plaac_text_upload may run
 * beside the scoring calls of the same context (one upload at a time per context; own stream, own pinned and device buffers),
 * plaac_score_begin_uploaded takes the batch over (what plaac_score_begin_text does minus the upload and the parse; the batch
 * object is consumed) and is collected like any text batch (plaac_score_end_text / _end_text_table*). bin/plaac runs an
 * uploader thread per context: the upload of batch k + 1 beside the download of batch k. */
typedef struct plaac_text_batch plaac_text_batch;
plaac_status plaac_text_upload(plaac_ctx *ctx, const char *text, uint64_t text_len, const uint64_t *starts, uint32_t nrec,
                               plaac_text_batch **out);
plaac_status plaac_score_begin_uploaded(plaac_ctx *ctx, plaac_text_batch *batch, int counting);
void plaac_text_batch_free(plaac_text_batch *batch); /* an uploaded batch that will not be scored */
/* message of the last failing plaac_text_upload on this context: the uploader thread has a slot of its own (plaac_last_error
 * belongs to the thread that scores) */
const char *plaac_text_upload_error(const plaac_ctx *ctx);
/* The COUNTING pass of a two-pass run fed with text (computeaafreq plaac.java:1655-1666 over nextfasta's records): the batch is
 * parsed on the device like a scored one and counted (countaas / isvalidprotein :1698-1739), not scored. Shares a context's
 * two pending slots with the scoring calls; residues (nullable) = the batch's residue count. */
plaac_status plaac_histogram_begin_text(plaac_ctx *ctx, const char *text, uint64_t text_len, const uint64_t *starts, uint32_t nrec);
plaac_status plaac_histogram_end_text(plaac_ctx *ctx, int64_t counts[PLAAC_NAA], uint64_t *residues);

/* ---- resident batches: upload once, use many times -------------------------------------------------------
 * The reference makes one full pass over the input for the background counts and a second one for scoring
 * (plaac.java:377-384 then :755), and a parameter sweep (BASELINE config 5) re-scores the same proteome under
 * several (alpha, core length) settings. A resident batch keeps the encoded residues in HBM across those calls. */
typedef struct plaac_batch plaac_batch;
plaac_status plaac_batch_upload(plaac_ctx *ctx, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot,
                                plaac_batch **out);
plaac_status plaac_batch_histogram(plaac_batch *b, int64_t counts[PLAAC_NAA]);
/* scores with the ctx's CURRENT parameters (plaac_ctx_set_params between calls = one sweep point each) */
plaac_status plaac_batch_score(plaac_batch *b, plaac_row *rows, const plaac_tracks *tracks);
void plaac_batch_free(plaac_batch *b);

/* Same, on buffers already resident in device memory (all pointers are device pointers, including
 * the ones inside *tracks; the tracks struct itself lives on the host). `stream` is a hipStream_t
 * (NULL = the ctx's own stream). All work is ordered after earlier work on `stream` and before later work on
 * it (internal side streams are forked and joined with events). The call waits once for its own planning
 * kernels (it needs one number back to size the work buffers) and then returns after enqueueing the scoring
 * kernels: synchronise the stream (or call plaac_ctx_sync when stream == NULL) before reading rows.
 * d_codes must be 16-byte aligned and readable up to the next 16-byte boundary after total_residues
 * (true of any hipMalloc / torch allocation). Work buffers are grown on demand and reused: consecutive calls on one
 * ctx are ordered after each other on the device whatever stream each is given (a ctx is still single-caller on
 * the host side), and plaac_ctx_set_params waits for the last scored batch before it replaces the tables. */
plaac_status plaac_score_device(plaac_ctx *ctx, const uint8_t *d_codes, const uint64_t *d_offsets, uint32_t nprot,
                                uint64_t total_residues, plaac_row *d_rows, const plaac_tracks *d_tracks,
                                void *stream);
/* Parameter sweep over one resident batch (BASELINE config 5: alpha x core length): npoints parameter sets,
 * one device row array per point. Points that differ only in the core length share everything that does not
 * depend on it (sort, packed copy, forward pass, FoldIndex/PAPA tracks, Viterbi + traceback); only the two
 * prefix-sum window searches run per core length. Each d_rows[i] is bit-identical to what plaac_score_device
 * produces for points[i] alone. The ctx's own parameters are not changed. */
plaac_status plaac_score_sweep_device(plaac_ctx *ctx, const uint8_t *d_codes, const uint64_t *d_offsets,
                                      uint32_t nprot, uint64_t total_residues, const plaac_params *points,
                                      uint32_t npoints, plaac_row *const *d_rows, void *stream);
/* host-buffer form on a resident batch: rows[i] = host array of nprot rows for points[i] */
plaac_status plaac_batch_sweep(plaac_batch *b, const plaac_params *points, uint32_t npoints, plaac_row *const *rows);
plaac_status plaac_histogram_device(plaac_ctx *ctx, const uint8_t *d_codes, const uint64_t *d_offsets,
                                    uint32_t nprot, int64_t *d_counts, void *stream);
plaac_status plaac_ctx_sync(plaac_ctx *ctx);

/* Device time in ms (HIP events recorded on the stream each kernel is launched on) of the most recent
 * scored batch: [0] whole call (first launch -> all kernels done), [1] plan (length sort),
 * [2] k_vit (Viterbi / traceback / core), [3] k_fwd (forward recurrence),
 * [4] k_win (MW / LLR windows, means), [5] k_tracks (FoldIndex / PAPA window tracks),
 * [6] group-interleaved packing of the residues for [2..4] (incl. the host round trip for its size),
 * [7] k_bwd (backward recurrence) in track mode; in summary mode the long run of a chain-bound call (k_long: the latency
 *     forms of the long wave-groups + their core search; [2..4] are the throughput-form runs then), else 0.
 * The four scoring kernels run concurrently on separate streams (PLAAC_SERIAL_STREAMS=1 in the
 * environment at ctx creation serialises them for profiling), so [2..5] overlap and do not add up to [0].
 * Blocks until the batch has completed. */
plaac_status plaac_last_timings(plaac_ctx *ctx, float ms[8]);
/* Same figures averaged over the most recent `ncalls` scored batches (the ctx keeps the events of
 * the last 32). This is how bench.py times the kernels over its whole timed region without a sync per step. */
plaac_status plaac_timings_mean(plaac_ctx *ctx, uint32_t ncalls, float ms[8]);

/* ---- all GPUs of one node ---------------------------------------------------------
 * The reference's serial per-protein loop (plaac.java:755, :610) sharded BY SEQUENCE: one plaac_ctx per listed
 * device (a device may be listed more than once: two contexts on one GPU overlap the copies of one batch with the
 * kernels of another), one host thread per context, the records dealt to the contexts by plaac_shard_plan (below:
 * sorted by length, dealt in turn - equal residue counts and an equal share of the long proteins), rows written in
 * input order. No data-path collective; the 22 x int64 histogram is summed on the host. Results are
 * identical to a single context's whatever the device list. device_ids == NULL or ndev <= 0: every visible device once.
 * Same calling rules as a ctx: single-caller, no process globals, status codes + plaac_node_last_error. */
typedef struct plaac_node plaac_node;
int plaac_device_count(void); /* visible HIP devices (0 when there is none or the runtime fails) */
plaac_status plaac_node_create(const plaac_params *params, const int *device_ids, int ndev, plaac_node **out);
void plaac_node_destroy(plaac_node *node);
int plaac_node_size(const plaac_node *node);      /* number of contexts */
plaac_ctx *plaac_node_ctx(plaac_node *node, int k); /* borrowed: context k, for pipelines that feed the devices themselves */
plaac_status plaac_node_set_params(plaac_node *node, const plaac_params *params);
plaac_status plaac_node_histogram(plaac_node *node, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot,
                                  int64_t counts[PLAAC_NAA]);
plaac_status plaac_node_score(plaac_node *node, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot,
                              plaac_row *rows, const plaac_tracks *tracks);
/* message of the last failing call on this node (node == NULL: last failing plaac_node_create on this thread) */
const char *plaac_node_last_error(const plaac_node *node);
/* plaac_ctx_set_overlap on every context of the node (for hosts that feed the contexts' device entry points themselves) */
plaac_status plaac_node_set_overlap(plaac_node *node, int on);

/* ---- FASTA text through a node (round 6): the text entry points of a context, dealt over the node's contexts ---------------
 * A host that reads a FASTA file (fastareader, plaac.java:4302-4375) and prints scoreallfastas' table (:755-948) hands the
 * node BATCHES OF TEXT in file order (whole records, text[starts[i]] = the '>' of record i, starts[nrec] = text_len, as
 * plaac_fasta_next_text cuts them) and takes the table's text back in the same order; which context scores a batch is the
 * node's business (round robin; every context holds two pending batches, so 2 x plaac_node_size batches may be in flight).
 * Every "end" call below serves the OLDEST pending batch of the node, whatever context it went to. The node itself keeps
 * how the record before a batch ended (the reference trims the NEXT name behind an empty line, :4362); plaac_node_text_reset
 * starts a new file. Results are byte for byte those of a single context fed the same batches.
 *   plaac_node_text_begin          score (counting != 0: and count, as plaac_score_begin_counting) the batch
 *   plaac_node_text_table_size     wait for the oldest batch; its table's size, its residues, and *needs_host != 0 when the
 *                                  device will not vouch for a value of it (>= 1e9, an infinity, a record without a sequence)
 *   plaac_node_text_table          the table's bytes (capacity >= the size reported); gives the batch up
 *   plaac_node_text_rows           the oldest batch as rows + what the device parsed (plaac_score_end_text), for a host that
 *                                  formats it itself - the way out when needs_host != 0; gives the batch up
 *   plaac_node_text_discard        gives the oldest batch up without collecting anything
 *   plaac_node_text_pending        batches in flight
 * The COUNTING pass of a two-pass run (computeaafreq, :1655-1666): plaac_node_histogram_text_begin per batch, then
 * plaac_node_histogram_text_end once per batch in the same order (counts and *residues are ADDED to what the caller passes
 * in, so a loop over a file ends with the file's totals; zero them first).
 * An uploader thread of the host may run plaac_node_text_upload beside the scoring calls (plaac_text_upload: one upload at
 * a time per context - the node serialises them per context); plaac_node_text_begin_uploaded then takes the batch over. */
typedef struct plaac_node_text_batch plaac_node_text_batch;
plaac_status plaac_node_text_begin(plaac_node *node, const char *text, uint64_t text_len, const uint64_t *starts, uint32_t nrec,
                                   int counting);
plaac_status plaac_node_text_upload(plaac_node *node, const char *text, uint64_t text_len, const uint64_t *starts, uint32_t nrec,
                                    plaac_node_text_batch **out);
plaac_status plaac_node_text_begin_uploaded(plaac_node *node, plaac_node_text_batch *batch, int counting);
void plaac_node_text_batch_free(plaac_node_text_batch *batch); /* an uploaded batch that will not be scored */
plaac_status plaac_node_text_table_size(plaac_node *node, int corelength, int ww2, uint64_t *table_bytes, int *needs_host,
                                        uint64_t *residues /* nullable */);
plaac_status plaac_node_text_table(plaac_node *node, char *table, uint64_t table_cap, int64_t *counts /* nullable */);
plaac_status plaac_node_text_rows(plaac_node *node, plaac_row *rows, uint8_t *codes /* nullable */, uint64_t codes_cap,
                                  uint64_t *offsets, uint8_t *blank_end, uint32_t *extents /* nullable */,
                                  int64_t *counts /* nullable */);
plaac_status plaac_node_text_discard(plaac_node *node);
int plaac_node_text_pending(const plaac_node *node);
uint32_t plaac_node_text_oldest_records(const plaac_node *node); /* records of the oldest pending batch (0: none pending) */
void plaac_node_text_reset(plaac_node *node); /* a new file begins: the next batch's first name is trimmed (:4362) */
plaac_status plaac_node_histogram_text_begin(plaac_node *node, const char *text, uint64_t text_len, const uint64_t *starts,
                                             uint32_t nrec);
plaac_status plaac_node_histogram_text_end(plaac_node *node, int64_t counts[PLAAC_NAA], uint64_t *residues /* nullable */);
/* plotsomefastas' per-residue table (plaac.java:587-649) for a batch of selected records on one of the node's contexts
 * (plaac_score_tracks_table; round robin). No text batch may be pending on the node. */
plaac_status plaac_node_score_tracks_table(plaac_node *node, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot,
                                           const char *labels, const uint64_t *label_off, plaac_row *rows, char **table,
                                           uint64_t *table_len, int *needs_host);

/* THE partitioner of every multi-GPU layer (plaac_node_*, plaac_amd/dist.py, bench.py): SURVEY.md 8(e) G1's "sort by
 * length, deal". The records are sorted by length (descending, stable) and dealt to `parts` shards boustrophedon
 * (0 .. parts-1, parts-1 .. 0, ...): equal residue counts, and every shard gets the same share of the long proteins whose
 * serial chains bound a step (plaac.java:3360-3368). index_out[part_start[k] .. part_start[k+1]) = the records of shard k
 * in ascending input order; part_start has parts + 1 entries. Host-only: needs no device. */
plaac_status plaac_shard_plan(const uint64_t *offsets, uint32_t nprot, uint32_t parts, uint32_t *index_out,
                              uint32_t *part_start);

/* Resident batches on all devices of a node: the batch is cut with plaac_shard_plan, every shard is uploaded ONCE to its
 * device and stays there for the background pass (plaac.java:377-384), the scoring pass (:755) - the reference's two
 * passes over one input - and for parameter sweeps, which the reference runs as one `main` invocation per point
 * (plaac.java:337-353, web/lib/server.rb:152-155: nine uploads of the same proteome). Rows (and tracks) come back in
 * INPUT order. plaac_node_score / plaac_node_histogram are upload + use + free of such a batch. offsets[0] must be 0 (rows
 * and tracks are indexed from the first residue of the batch, as in the single-device entry points).
 * LIFETIME: a batch belongs to its node. Free every batch before plaac_node_destroy; a batch that is still alive when its
 * node is destroyed is DETACHED by the node (its device memory is released with the contexts): every later call on it
 * returns PLAAC_ERR_ARG with the message of plaac_node_batch_last_error, and plaac_node_batch_free of it is still required
 * (it releases the host side) and safe. */
typedef struct plaac_node_batch plaac_node_batch;
plaac_status plaac_node_batch_upload(plaac_node *node, const uint8_t *codes, const uint64_t *offsets, uint32_t nprot,
                                     plaac_node_batch **out);
plaac_status plaac_node_batch_histogram(plaac_node_batch *b, int64_t counts[PLAAC_NAA]);
/* scores with the node's CURRENT parameters (plaac_node_set_params between calls: the two-pass run on one upload) */
plaac_status plaac_node_batch_score(plaac_node_batch *b, plaac_row *rows, const plaac_tracks *tracks);
/* BASELINE config 5 on all devices: rows[i] = host array of nprot rows for points[i]; every device runs the sweep-aware
 * schedule of plaac_score_sweep_device over its resident shard. The node's own parameters are not changed. */
plaac_status plaac_node_batch_sweep(plaac_node_batch *b, const plaac_params *points, uint32_t npoints,
                                    plaac_row *const *rows);
void plaac_node_batch_free(plaac_node_batch *b);
/* message of the last failing call on the batch's node, or that the node is gone (bindings that keep only the batch handle) */
const char *plaac_node_batch_last_error(const plaac_node_batch *b);
/* records / residues of the uploaded batch: the sizes of the row and track arrays the calls above fill */
uint32_t plaac_node_batch_records(const plaac_node_batch *b);
uint64_t plaac_node_batch_residues(const plaac_node_batch *b);

/* Summary mode scores the FoldIndex / PAPA window tracks in two tiers: a filter that decides from error-bounded
 * prefix sums, and the exact fixed-order kernel for every protein the bounds cannot decide (results are identical
 * either way; PLAAC_KB_FILTER=0 at ctx creation sends everything to the exact kernel). This returns how many
 * proteins of the most recent scored batch took the exact tier. Blocks until that batch has completed. */
plaac_status plaac_last_exact_fallbacks(plaac_ctx *ctx, uint32_t *count);

/* DIAGNOSTIC (tools/pmc.sh): three kernels that stream once over the resident residue buffer - 16 bytes per lane,
 * aligned dwords, unaligned dwords - so that a rocprofv3 PMC pass over the same process can calibrate FETCH_SIZE on a
 * known byte count in the access shapes of the scoring kernels (MI355X_MICROARCH.md: the counter is uncalibrated for
 * anything but wide streaming reads). Asynchronous on `stream` (NULL = the ctx's own). No result. */
plaac_status plaac_calibration_reads(plaac_ctx *ctx, const uint8_t *d_codes, uint64_t total_residues, void *stream);

/* Pipelines of batches: let consecutive scoring calls of this context OVERLAP (default off = every call is ordered behind
 * everything enqueued on its stream before it). With overlap on, the planning and packing kernels of a call - which read
 * `d_codes` / `d_offsets` and write context scratch only (the plan and the packed copy exist twice, used by alternate
 * calls) - run on a stream of their own as soon as the call BEFORE the previous one has let go of that scratch, i.e. beside
 * the previous call's scoring kernels; the lane-per-protein kernels of the call follow as soon as those of the previous
 * call are through; its window kernels stay on the caller's stream. The caller guarantees: (1) `d_codes` / `d_offsets` of
 * a call are complete when the call is made (not produced by work still pending on a stream) and stay unchanged until it
 * has completed; (2) `d_rows` (and the tracks) of a call are not in use by anything ELSE that is still pending when the
 * call is made - a consumer of an earlier call's rows on another stream has to be waited for on the host (an event
 * synchronisation, plaac_ctx_sync), not by a stream-side wait on the caller's stream; consecutive calls may write the
 * same buffer only when they score the same batch with the same parameters (every row field is written once per call,
 * with its final value; the long runs of consecutive chain-bound calls run side by side, so the LATER call is not
 * always the last writer). Results are unchanged; a lone call is unchanged. (bench.py switches it on: its steps are back to back on one
 * resident batch. Measured: EXPERIMENTS.md 4.9.) */
plaac_status plaac_ctx_set_overlap(plaac_ctx *ctx, int on);
/* OFF by default. on != 0: summary mode forms the five floats reported AT the PAPA centre (PAPAcombo = PAPAprop, PAPAfi,
 * PAPAllr, PAPAllr2; disorderreport :4986-4997) from first-level window sums that SLIDE over six neighbouring positions instead
 * of 41 fixed-order taps per position: they then differ from the reference-order values by less than 1e-12 (the north star's
 * bar for floats is 1e-6; tested at 1e-9), and the summary step is about 8 % shorter. Every index (PAPAcen included), every
 * decision and every other float of the row stay exactly the reference-order results. Track mode is not affected. */
plaac_status plaac_ctx_set_value_tolerance(plaac_ctx *ctx, int on);

/* DIAGNOSTIC (bench.py --clock-probe): the shader clock the chip actually holds while the scoring kernels run. One wave
 * on a stream of its own sleeps in s_sleep 127 steps (64 x 127 shader cycles each) for `micros` microseconds of the
 * constant 100 MHz counter; *mhz = slept cycles / elapsed time. The instruction-issue roof of the path is priced in
 * cycles, so its fraction depends on this clock (nominal 2.4 GHz; under the fp64 + LDS load of this path the chip runs
 * lower). Blocks the calling thread (not the scoring streams) until the wave has finished. */
plaac_status plaac_clock_probe(plaac_ctx *ctx, uint32_t micros, double *mhz);

/* The filter tier needs the SIGN of FoldIndex (plaac.java:4885, :5020-5058) and of its second smoothing (:4903, :4944)
 * at every position. When hydro2[] and cc[] are rationals with small denominators - the reference's own tables are:
 * aahydro / 9 + 0.5 with one-decimal aahydro (:90), cc = {2.785, -1, -1.151} - m * FoldIndex is an integer over the
 * fixed denominator SH * SC, so the sign comes from exact 32-bit integer arithmetic (no error bound involved); other
 * tables keep the error-bounded fp64 form. Returns 1 when `params` qualify for the integer form (and the library will
 * use it; PLAAC_FI_INT=0 at ctx creation forces the fp64 form), else 0. info (nullable) receives
 * {A2, B2, C2, SH, SC, Hmin}: I2 = A2 * sum(H - Hmin) + B2 * count + C2 * |charge sum| - 1 = 2 * SH * SC * m * fi - 1.
 * Host-only: needs no device. */
int plaac_fi_integer_form(const plaac_params *params, int32_t info[6]);

/* DIAGNOSTIC / tests (host only, needs no device): the schedule of a scoring call as text. A call is some forty kernel
 * launches on up to ten streams; which kernels, in which forms, on which streams is decided by a pure function of the
 * call's kind, the words the planning kernels send back, the context's history and the environment switches
 * (plaac_amd/csrc/schedule.hip.inc). This entry point runs that function for a DESCRIBED call - no batch, no device - and
 * writes one line per decision ("F name=value") and per operation ("L stream kernel ... accesses", "R stream event",
 * "W stream event"; "ALIAS a b": two stream slots that are one stream). tests/test_schedule.py walks the decision table with
 * it and checks that no wait can deadlock and that conflicting accesses to a buffer are ordered, within a call and across
 * overlapping calls. The environment switches are read at the call. Returns the length written, or -1 (buffer too small). */
typedef struct plaac_sched_query {
    uint32_t nprot;
    uint32_t npoints;            /* 1, or the points of a sweep */
    uint64_t residues;
    uint32_t total_rows, rows_first, long_groups, long_rows; /* what the planning kernels would report */
    uint32_t run_mark[7];
    uint32_t ngroups_sweep;      /* sweep groups (<= 11) */
    uint32_t group_members[11];  /* points per group (sum = npoints) */
    int32_t kb_base[11];         /* per group: -1, or the earlier group whose window tracks it shares */
    int32_t lane_possible, fast20, wmax, core_par_tables; /* properties of the tables (all groups alike) */
    int32_t tracks, overlap;
    uint64_t ncalls;             /* calls the context has scored before this one */
    int32_t last_chain_bound, last_mixed, last_single_summary, old_tail;
} plaac_sched_query;
long plaac_debug_schedule(const plaac_sched_query *q, char *buf, size_t cap);

/* ---- environment and test hooks ---------------------------------------------------------------------------------------
 * The release library reads NINE environment variables of its own - eight when a context is created, one in the host I/O
 * helpers (and nothing else of the host's environment but the HIP runtime's GPU_MAX_HW_QUEUES):
 *   PLAAC_LATENCY_MODE=0|1   force the throughput / the latency forms of the chain kernels (default: chosen per batch)
 *   PLAAC_KB_FILTER=0        summary mode scores every window track with the exact kernel (no filter tier)
 *   PLAAC_OVERLAP=1          contexts start with plaac_ctx_set_overlap on
 *   PLAAC_SERIAL_STREAMS=1   every kernel of a call on one stream (profilers that serialise dispatches anyway)
 *   PLAAC_SWEEP_SPREAD=0|1   a sweep's groups on streams of their own / spread over the call's streams (default: by
 *                            GPU_MAX_HW_QUEUES)
 *   PLAAC_STREAM_DEBUG=1     print the role streams' hardware queues at context creation (stderr)
 *   PLAAC_CTX_TIMING=1       print what bringing a context up is made of (stderr)
 *   PLAAC_THREADS=n          host threads of plaac_score's staging copies
 *   PLAAC_HUGE_PAGES=0       plain allocations for the large host buffers of the FASTA / table helpers (plaac_host.h) instead
 *                            of transparent huge pages on request
 * None changes a result. Every other switch of earlier rounds is a TEST HOOK: plaac_debug_set_knob(key, value) - key without
 * the PLAAC_ prefix, value NULL removes it - sets a process-wide table the contexts created afterwards read. The hooks force
 * a form the library selects by itself for other batch shapes or tables (KB_LANE_MIN_GROUPS, MIXED_GROUPS, GENERIC_TRACKS,
 * FI_INT, LSE_CLAMP ...: plaac_amd/csrc/schedule.hip.inc lists them), so that parity tests reach it with small batches;
 * plaac_amd/native.py turns PLAAC_<KEY> of its own process environment into these calls. Keys of forms that were measured
 * slower and are no longer selected at all (TRACK_ONE_PASS, FWD_DIRECT, TRACK_CKPT ...; EXPERIMENTS.md) and the result-breaking
 * ablation switches exist in the diagnostic build only (`make DIAG=1`): the release library answers PLAAC_ERR_UNSUPPORTED and
 * does not contain their kernels. plaac_diag_build() != 0: this library is the diagnostic build. */
plaac_status plaac_debug_set_knob(const char *key, const char *value);
int plaac_diag_build(void);

#ifdef __cplusplus
}
#endif
#endif /* PLAAC_NATIVE_H */
