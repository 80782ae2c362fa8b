/*
 * plaac_host.h — host-side text I/O of the PLAAC engine (part of libplaac_native.so, no device code).
 *
 * These helpers reproduce the reference's input and output contracts around the scoring kernels
 * (reference = cli/src/plaac.java of whitehead/plaac):
 *   fastareader                :4302-4375  record splitting quirks
 *   read_aa_params / print_aa_params  :2684-2713 / :2665-2669   the -B / -F / -b file format
 *   System.out.format rows     :899-945 (summary, 38 columns), :635-645 (per-residue tracks)
 *   "## parameters at run-time" block :503-514, aaparams2string :2717-2723
 * Number formatting follows java.util.Formatter (HALF_UP on the exact decimal expansion, "NaN",
 * "Infinity"), not C printf.
 * C linkage so that the C++ CLI, a JNI shim and ctypes tests share one implementation.
 */
#ifndef PLAAC_HOST_H
#define PLAAC_HOST_H

#include <stddef.h>
#include <stdint.h>

#include "plaac_native.h"

#ifdef __cplusplus
extern "C" {
#endif

/* A FASTA file split into records exactly as fastareader does, residues already encoded (plaac_encode).
 * codes/offsets are a ready-made UNTRIMMED batch for plaac_histogram / plaac_score. */
typedef struct plaac_fasta {
    uint32_t nrec;
    uint64_t nres;
    uint8_t *codes;     /* nres bytes                                  */
    uint64_t *offsets;  /* nrec + 1                                    */
    char *names;        /* all headers, NUL-separated                  */
    uint64_t *name_off; /* nrec + 1 offsets into names                 */
} plaac_fasta;

/* Reads `path`. Returns PLAAC_ERR_IO when the file cannot be opened (the reference prints
 * "# Couldn't open <file>" to stdout and carries on with no records; the CLI mirrors that). */
plaac_status plaac_fasta_read(const char *path, plaac_fasta **out);
void plaac_fasta_free(plaac_fasta *f);

/* The same reader as a stream of batches, for inputs of any size in bounded memory (the reference reads record by
 * record: fastareader.hasmorefastas / nextfasta, plaac.java:4302-4375). plaac_fasta_next returns the next batch of
 * at most max_records records and about max_bytes bytes of file text (a record is never split; every batch holds
 * at least one record), *out = NULL at the end of the file. Batches are ordinary plaac_fasta objects (free each with
 * plaac_fasta_free); concatenated they equal what plaac_fasta_read returns for the file. */
typedef struct plaac_fasta_stream plaac_fasta_stream;
plaac_status plaac_fasta_open(const char *path, plaac_fasta_stream **out);
plaac_status plaac_fasta_next(plaac_fasta_stream *s, uint32_t max_records, uint64_t max_bytes, plaac_fasta **out);
void plaac_fasta_close(plaac_fasta_stream *s); /* text batches still alive keep the file image mapped until the last is freed */

/* The stream cut into batches of TEXT for the device-side parser (plaac_score_begin_text, round 5): the same batch boundaries
 * as plaac_fasta_next, but the records are only LOCATED - starts[i] = offset of record i's '>' inside `text`, starts[nrec] = len;
 * record i's name is the name_len[i] bytes behind its '>' (NOT copied, not NUL-terminated; untrimmed: whether a name is
 * trimmed depends on how the record before it ended, :4362, which the device reports as blank_end;
 * plaac_fasta_text_trim_names applies it to name_len). `text` points into the stream's file image and stays valid until
 * plaac_fasta_text_free (which also lets the stream release the pages behind it). *out = NULL at the end. */
typedef struct plaac_fasta_text {
    const char *text;
    uint64_t len;
    uint32_t nrec;
    uint64_t *starts;   /* nrec + 1 */
    uint32_t *name_len; /* nrec */
    int prev_blank;     /* how the record BEFORE this batch ended: 1 = in an empty line (or this is the file's first batch): the
                           reference trims the first name of this batch then (:4362); the reader looks at that one record itself */
    int last_blank;     /* the same for this batch's last record (= the next batch's prev_blank) */
    void *owner_;       /* the stream (page release bookkeeping) */
    uint64_t file_off_; /* offset of `text` in the file */
} plaac_fasta_text;
plaac_status plaac_fasta_next_text(plaac_fasta_stream *s, uint32_t max_records, uint64_t max_bytes, plaac_fasta_text **out);
void plaac_fasta_text_free(plaac_fasta_text *t);
/* trims (shortens name_len of) the names the reference trims: the first record of the file and every record whose
 * predecessor ended in an empty line. prev_blank: how the record before this batch ended (1 for the first batch); returns the flag for the next batch. */
int plaac_fasta_text_trim_names(plaac_fasta_text *t, const uint8_t *blank_end, int prev_blank);
/* Residues [first, first + count) of record i of a text batch, encoded as plaac_encode does, read from the TEXT with the
 * extents the device reported for the batch (plaac_score_end_text: extents[2i], extents[2i+1]): the host's way to the few
 * residues a summary row prints without a copy of all codes. Returns how many were written (fewer when the record ends). */
uint64_t plaac_fasta_text_codes(const char *text, const uint64_t *starts, const uint32_t *extents, uint32_t i, uint64_t first,
                                uint64_t count, uint8_t *out);

/* Worker threads the host helpers use for parsing / formatting: hardware threads, capped by the cgroup CPU
 * quota, overridable with PLAAC_THREADS. plaac_fasta_read parses records in parallel (they are independent once
 * the header lines are located). */
unsigned plaac_host_threads(void);

/* -B / -F parameter files: 22 lines, first token = number (:2684-2713).
 * warn_line (nullable, >= 22 ints): set to 1 where the optional "# name" column disagrees with the alphabet. */
plaac_status plaac_read_aa_params(const char *path, double vec[PLAAC_NAA], int *warn_line);

/* java.util.Formatter "%.<decimals>f" of v into buf (cap >= 400). Returns the length written. */
int plaac_format_fixed(double v, int decimals, char *buf, size_t cap);
/* The same text by the digit-string path alone (shortest round-trip digits, HALF_UP on them): plaac_format_fixed and
 * the row formatters take an arithmetic short cut for values that are not within rounding noise of a tie and fall
 * back to this path otherwise; the two must agree on every double (differential test). */
int plaac_format_fixed_reference(double v, int decimals, char *buf, size_t cap);
/* java.lang.Double.toString(v) (used for alpha in the parameter block). */
int plaac_format_double_tostring(double v, char *buf, size_t cap);

/* One line of the per-protein summary table, without the trailing newline (:899-945).
 * codes = the record's UNTRIMMED codes, reclen their number. Returns the length, or -1 if cap is too small.
 * Records with row->prot_len == 0 produce no line (returns 0). */
long plaac_format_summary_row(const plaac_row *row, const char *name, const uint8_t *codes, uint64_t reclen,
                              int corelength, int ww2, char *buf, size_t cap);
/* The same with the name given by length (not NUL-terminated: the names of a text batch lie in the file's text) and written
 * without a terminating NUL. cap must cover name_len + 3 * prot_len + ww2 + 8700 (the longest a row can be); -1 otherwise. */
long plaac_format_summary_row_n(const plaac_row *row, const char *name, size_t name_len, const uint8_t *codes, uint64_t reclen,
                                int corelength, int ww2, char *buf, size_t cap);
/* header line of the summary table (:715-719) and of the per-residue table (:603-605), no newline */
const char *plaac_summary_header(void);
const char *plaac_tracks_header(void);

/* All per-residue lines of one protein plus the closing line of 56 '#' (:635-645), newline-terminated.
 * `first` = index of the record's first residue in the track arrays. Returns the length or -1. */
long plaac_format_track_rows(const plaac_tracks *tracks, uint64_t first, const uint8_t *codes, uint32_t n,
                             const char *order_id, const char *name, char *buf, size_t cap);
/* worst-case bytes plaac_format_track_rows needs for a protein of n residues */
size_t plaac_track_rows_bound(uint32_t n, size_t id_len, size_t name_len);

/* "## parameters at run-time" block (:503-514), newline-terminated. Returns the length or -1. */
long plaac_format_param_block(const plaac_params *p, char *buf, size_t cap);
/* GraphViz description of the two-state HMM with its emission tables: hmm.dottify(file, true), the -h flag
 * (:4209-4287, :520-522). Returns the length or -1. */
long plaac_format_hmm_dot(const plaac_params *p, char *buf, size_t cap);
/* the 22 lines "%.6f # %s" of print_aa_params (:2665-2669). Returns the length or -1. */
long plaac_format_aa_params(const double vec[PLAAC_NAA], char *buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* PLAAC_HOST_H */
