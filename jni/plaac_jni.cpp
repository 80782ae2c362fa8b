// plaac_jni.cpp — JNI shim between PlaacNative.java and the C ABI of libplaac_native.so (include/plaac_native.h).
// One C call per native, no logic of its own: direct-buffer addresses in, plaac_status out; a non-zero status is thrown
// as java.lang.IllegalStateException(plaac_last_error) — the reference's own error convention at this seam is an
// uncaught exception. Build: `make jni` (needs $(JAVA_HOME)/include/jni.h; this image has no JDK,
// so the default build skips it; tests/test_abi.py compile-checks it against a declarations-only stand-in).
#include <jni.h>

#include <cstdint>
#include <new>
#include <string>

#include "plaac_native.h"

namespace {

void raise(JNIEnv *env, const char *msg) {
    jclass cls = env->FindClass("java/lang/IllegalStateException");
    if (cls) env->ThrowNew(cls, msg ? msg : "plaac_native failed");
}

// address of a direct buffer that must hold at least `need` bytes; throws and returns null otherwise
void *direct(JNIEnv *env, jobject buf, uint64_t need, const char *what) {
    void *p = buf ? env->GetDirectBufferAddress(buf) : nullptr;
    if (!p || (uint64_t)env->GetDirectBufferCapacity(buf) < need) {
        raise(env, (std::string(what) + ": a direct ByteBuffer of sufficient capacity is required").c_str());
        return nullptr;
    }
    return p;
}

plaac_node *node_of(jlong h) { return reinterpret_cast<plaac_node *>(static_cast<intptr_t>(h)); }

// message of a failed node call; a null handle has no message of its own (plaac_node_last_error(nullptr) would hand back
// the text of this thread's last failed plaac_node_create - another call's error)
const char *node_error(jlong h) { return node_of(h) ? plaac_node_last_error(node_of(h)) : "null node handle"; }

} // namespace

extern "C" {

// System.loadLibrary: refuse a libplaac_native.so of another ABI generation before any native is bound
JNIEXPORT jint JNICALL JNI_OnLoad(JavaVM *, void *) {
    return plaac_abi_version() == PLAAC_ABI_VERSION ? JNI_VERSION_1_6 : JNI_ERR;
}

JNIEXPORT jint JNICALL Java_PlaacNative_abiVersion(JNIEnv *, jclass) { return plaac_abi_version(); }

JNIEXPORT jint JNICALL Java_PlaacNative_deviceCount(JNIEnv *, jclass) { return plaac_device_count(); }

JNIEXPORT jint JNICALL Java_PlaacNative_paramsBytes(JNIEnv *, jclass) { return (jint)plaac_sizeof_params(); }

JNIEXPORT void JNICALL Java_PlaacNative_paramsInit(JNIEnv *env, jclass, jobject paramsOut, jdoubleArray fg,
                                                   jdoubleArray bgCounts, jdouble alpha, jint corelength, jint ww1,
                                                   jint ww2, jint ww3, jboolean adjustProlines) {
    plaac_params *P = (plaac_params *)direct(env, paramsOut, plaac_sizeof_params(), "paramsOut");
    if (!P) return;
    double f[PLAAC_NAA], b[PLAAC_NAA];
    if (fg) {
        if (env->GetArrayLength(fg) != PLAAC_NAA) return raise(env, "fg must have 22 entries");
        env->GetDoubleArrayRegion(fg, 0, PLAAC_NAA, f);
    }
    if (bgCounts) {
        if (env->GetArrayLength(bgCounts) != PLAAC_NAA) return raise(env, "bgCounts must have 22 entries");
        env->GetDoubleArrayRegion(bgCounts, 0, PLAAC_NAA, b);
    }
    if (plaac_params_init(P, fg ? f : nullptr, bgCounts ? b : nullptr, alpha, corelength, ww1, ww2, ww3,
                          adjustProlines ? 1 : 0) != PLAAC_OK)
        raise(env, "plaac_params_init rejected its arguments");
}

JNIEXPORT void JNICALL Java_PlaacNative_encode(JNIEnv *env, jclass, jobject text, jint n, jobject codesOut) {
    if (n < 0) return raise(env, "negative length");
    const char *t = (const char *)direct(env, text, (uint64_t)n, "text");
    uint8_t *c = t ? (uint8_t *)direct(env, codesOut, (uint64_t)n, "codesOut") : nullptr;
    if (c) plaac_encode(t, (size_t)n, c);
}

JNIEXPORT jlong JNICALL Java_PlaacNative_nodeCreate(JNIEnv *env, jclass, jobject params, jintArray devices) {
    const plaac_params *P = (const plaac_params *)direct(env, params, plaac_sizeof_params(), "params");
    if (!P) return 0;
    jint ids[64];
    jsize nd = devices ? env->GetArrayLength(devices) : 0;
    if (nd > 64) {
        raise(env, "at most 64 contexts");
        return 0;
    }
    if (nd > 0) env->GetIntArrayRegion(devices, 0, nd, ids);
    int dev[64];
    for (jsize i = 0; i < nd; ++i) dev[i] = (int)ids[i];
    plaac_node *node = nullptr;
    if (plaac_node_create(P, nd > 0 ? dev : nullptr, (int)nd, &node) != PLAAC_OK) {
        raise(env, plaac_node_last_error(nullptr));
        return 0;
    }
    return (jlong) reinterpret_cast<intptr_t>(node);
}

JNIEXPORT void JNICALL Java_PlaacNative_nodeSetParams(JNIEnv *env, jclass, jlong node, jobject params) {
    const plaac_params *P = (const plaac_params *)direct(env, params, plaac_sizeof_params(), "params");
    if (P && plaac_node_set_params(node_of(node), P) != PLAAC_OK) raise(env, node_error(node));
}

JNIEXPORT void JNICALL Java_PlaacNative_nodeDestroy(JNIEnv *, jclass, jlong node) { plaac_node_destroy(node_of(node)); }

JNIEXPORT void JNICALL Java_PlaacNative_histogram(JNIEnv *env, jclass, jlong node, jobject codes, jobject offsets,
                                                  jint nprot, jlongArray counts22) {
    if (nprot < 0 || !counts22 || env->GetArrayLength(counts22) != PLAAC_NAA)
        return raise(env, "histogram: nprot >= 0 and a long[22] are required");
    const uint64_t *off = (const uint64_t *)direct(env, offsets, 8ull * ((uint64_t)nprot + 1), "offsets");
    if (!off) return;
    const uint8_t *c = (const uint8_t *)direct(env, codes, off[nprot], "codes");
    if (!c) return;
    int64_t counts[PLAAC_NAA];
    if (plaac_node_histogram(node_of(node), c, off, (uint32_t)nprot, counts) != PLAAC_OK)
        return raise(env, node_error(node));
    jlong out[PLAAC_NAA];
    for (int i = 0; i < PLAAC_NAA; ++i) out[i] = (jlong)counts[i];
    env->SetLongArrayRegion(counts22, 0, PLAAC_NAA, out);
}

JNIEXPORT void JNICALL Java_PlaacNative_score(JNIEnv *env, jclass, jlong node, jobject codes, jobject offsets,
                                              jint nprot, jobject rowsOut, jobjectArray tracks) {
    if (nprot < 0) return raise(env, "score: negative nprot");
    const uint64_t *off = (const uint64_t *)direct(env, offsets, 8ull * ((uint64_t)nprot + 1), "offsets");
    if (!off) return;
    const uint64_t total = off[nprot];
    const uint8_t *c = (const uint8_t *)direct(env, codes, total, "codes");
    plaac_row *rows = c ? (plaac_row *)direct(env, rowsOut, (uint64_t)nprot * sizeof(plaac_row), "rowsOut") : nullptr;
    if (!rows) return;
    plaac_tracks t, *tp = nullptr;
    if (tracks) {
        if (env->GetArrayLength(tracks) != 12) return raise(env, "tracks must be 12 direct buffers");
        void *a[12];
        for (int i = 0; i < 12; ++i) {
            a[i] = direct(env, env->GetObjectArrayElement(tracks, i), total * (i < 2 ? 1u : 8u), "tracks[i]");
            if (!a[i]) return;
        }
        t = plaac_tracks{(uint8_t *)a[0], (uint8_t *)a[1], (double *)a[2], (double *)a[3], (double *)a[4], (double *)a[5],
                         (double *)a[6],  (double *)a[7],  (double *)a[8], (double *)a[9], (double *)a[10], (double *)a[11]};
        tp = &t;
    }
    if (plaac_node_score(node_of(node), c, off, (uint32_t)nprot, rows, tp) != PLAAC_OK)
        raise(env, node_error(node));
}

// ---- resident batches (plaac_node_batch_*). The Java side passes the batch handle only; error text comes from
// plaac_node_batch_last_error, which also answers for a batch whose node was destroyed first (the library detaches such a
// batch: its calls fail with a message, batchFree stays safe - no raw node pointer is kept here).
struct BatchRef {
    plaac_node_batch *b;
};
static BatchRef *ref_of(jlong h) { return reinterpret_cast<BatchRef *>(static_cast<intptr_t>(h)); }

JNIEXPORT jlong JNICALL Java_PlaacNative_batchUpload(JNIEnv *env, jclass, jlong node, jobject codes, jobject offsets,
                                                     jint nprot) {
    if (nprot < 0) {
        raise(env, "batchUpload: negative nprot");
        return 0;
    }
    if (!node_of(node)) {
        raise(env, "null node handle");
        return 0;
    }
    const uint64_t *off = (const uint64_t *)direct(env, offsets, 8ull * ((uint64_t)nprot + 1), "offsets");
    if (!off) return 0;
    const uint8_t *c = (const uint8_t *)direct(env, codes, off[nprot], "codes");
    if (!c) return 0;
    plaac_node_batch *b = nullptr;
    if (plaac_node_batch_upload(node_of(node), c, off, (uint32_t)nprot, &b) != PLAAC_OK) {
        raise(env, node_error(node));
        return 0;
    }
    BatchRef *r = new (std::nothrow) BatchRef{b};
    if (!r) {
        plaac_node_batch_free(b);
        raise(env, "out of host memory");
        return 0;
    }
    return (jlong) reinterpret_cast<intptr_t>(r);
}

JNIEXPORT void JNICALL Java_PlaacNative_batchFree(JNIEnv *, jclass, jlong batch) {
    BatchRef *r = ref_of(batch);
    if (!r) return;
    plaac_node_batch_free(r->b);
    delete r;
}

JNIEXPORT void JNICALL Java_PlaacNative_batchHistogram(JNIEnv *env, jclass, jlong batch, jlongArray counts22) {
    BatchRef *r = ref_of(batch);
    if (!r) return raise(env, "null batch handle");
    if (!counts22 || env->GetArrayLength(counts22) != PLAAC_NAA) return raise(env, "batchHistogram: a long[22] is required");
    int64_t counts[PLAAC_NAA];
    if (plaac_node_batch_histogram(r->b, counts) != PLAAC_OK) return raise(env, plaac_node_batch_last_error(r->b));
    jlong out[PLAAC_NAA];
    for (int i = 0; i < PLAAC_NAA; ++i) out[i] = (jlong)counts[i];
    env->SetLongArrayRegion(counts22, 0, PLAAC_NAA, out);
}

JNIEXPORT void JNICALL Java_PlaacNative_batchScore(JNIEnv *env, jclass, jlong batch, jobject rowsOut, jobjectArray tracks) {
    BatchRef *r = ref_of(batch);
    if (!r) return raise(env, "null batch handle");
    const uint64_t nprot = plaac_node_batch_records(r->b), totalResidues = plaac_node_batch_residues(r->b);
    plaac_row *rows = (plaac_row *)direct(env, rowsOut, (uint64_t)nprot * sizeof(plaac_row), "rowsOut");
    if (!rows) return;
    plaac_tracks t, *tp = nullptr;
    if (tracks) {
        if (env->GetArrayLength(tracks) != 12) return raise(env, "tracks must be 12 direct buffers");
        void *a[12];
        for (int i = 0; i < 12; ++i) {
            a[i] = direct(env, env->GetObjectArrayElement(tracks, i), (uint64_t)totalResidues * (i < 2 ? 1u : 8u), "tracks[i]");
            if (!a[i]) return;
        }
        t = plaac_tracks{(uint8_t *)a[0], (uint8_t *)a[1], (double *)a[2], (double *)a[3], (double *)a[4], (double *)a[5],
                         (double *)a[6],  (double *)a[7],  (double *)a[8], (double *)a[9], (double *)a[10], (double *)a[11]};
        tp = &t;
    }
    if (plaac_node_batch_score(r->b, rows, tp) != PLAAC_OK) raise(env, plaac_node_batch_last_error(r->b));
}

JNIEXPORT void JNICALL Java_PlaacNative_batchSweep(JNIEnv *env, jclass, jlong batch, jobject params, jint npoints,
                                                   jobjectArray rowsOut) {
    BatchRef *r = ref_of(batch);
    if (!r) return raise(env, "null batch handle");
    if (npoints < 0 || npoints > 4096) return raise(env, "batchSweep: between 0 and 4096 points");
    const uint64_t nprot = plaac_node_batch_records(r->b);
    if (!rowsOut || env->GetArrayLength(rowsOut) != npoints) return raise(env, "batchSweep: one row buffer per point");
    const plaac_params *P = (const plaac_params *)direct(env, params, (uint64_t)npoints * plaac_sizeof_params(), "params");
    if (!P) return;
    plaac_row *rows[4096];
    for (jint i = 0; i < npoints; ++i) {
        rows[i] = (plaac_row *)direct(env, env->GetObjectArrayElement(rowsOut, i), (uint64_t)nprot * sizeof(plaac_row), "rowsOut[i]");
        if (!rows[i]) return;
    }
    if (plaac_node_batch_sweep(r->b, P, (uint32_t)npoints, rows) != PLAAC_OK) raise(env, plaac_node_batch_last_error(r->b));
}

JNIEXPORT void JNICALL Java_PlaacNative_nodeSetOverlap(JNIEnv *env, jclass, jlong node, jboolean on) {
    if (plaac_node_set_overlap(node_of(node), on ? 1 : 0) != PLAAC_OK) raise(env, node_error(node));
}

// ---- FASTA text in, table text out (plaac_node_text_*): fastareader.nextfasta (:4325-4357) in front of the scoring loop and
//      the output line of scoreallfastas (:899-945) behind it run on the device. A Java host hands over the bytes of whole
//      records and where each begins, in file order, and collects the batches in the same order; which device scores a batch
//      is the node's business. Every collecting native serves the OLDEST pending batch.
JNIEXPORT void JNICALL Java_PlaacNative_textBegin(JNIEnv *env, jclass, jlong node, jobject text, jlong textLen, jobject starts,
                                                  jint nrec, jboolean counting) {
    if (nrec < 0 || textLen < 0) return raise(env, "textBegin: negative size");
    const uint64_t *st = (const uint64_t *)direct(env, starts, 8ull * ((uint64_t)nrec + 1), "starts");
    if (!st) return;
    const char *t = (const char *)direct(env, text, (uint64_t)textLen, "text");
    if (!t) return;
    if (plaac_node_text_begin(node_of(node), t, (uint64_t)textLen, st, (uint32_t)nrec, counting ? 1 : 0) != PLAAC_OK)
        raise(env, node_error(node));
}

// returns the table's size in bytes; out2 = {needsHost, residues}. needsHost != 0: the device will not vouch for a value of
// this batch (>= 1e9, an infinity, a record without a sequence) - collect it with textRows and format it on the host, or
// give it up with textDiscard; textTable refuses it.
JNIEXPORT jlong JNICALL Java_PlaacNative_textTableSize(JNIEnv *env, jclass, jlong node, jint corelength, jint ww2, jlongArray out2) {
    if (!out2 || env->GetArrayLength(out2) != 2) {
        raise(env, "textTableSize: a long[2] is required");
        return -1;
    }
    uint64_t bytes = 0, residues = 0;
    int needs = 0;
    if (plaac_node_text_table_size(node_of(node), corelength, ww2, &bytes, &needs, &residues) != PLAAC_OK) {
        raise(env, node_error(node));
        return -1;
    }
    const jlong out[2] = {(jlong)needs, (jlong)residues};
    env->SetLongArrayRegion(out2, 0, 2, out);
    return (jlong)bytes;
}

static void put_counts(JNIEnv *env, jlongArray counts22, const int64_t *counts) {
    jlong out[PLAAC_NAA];
    for (int i = 0; i < PLAAC_NAA; ++i) out[i] = (jlong)counts[i];
    env->SetLongArrayRegion(counts22, 0, PLAAC_NAA, out);
}

JNIEXPORT void JNICALL Java_PlaacNative_textTable(JNIEnv *env, jclass, jlong node, jobject tableOut, jlong tableCap,
                                                  jlongArray counts22) {
    if (tableCap < 0 || (counts22 && env->GetArrayLength(counts22) != PLAAC_NAA)) return raise(env, "textTable: capacity >= 0, counts null or long[22]");
    char *t = (char *)direct(env, tableOut, (uint64_t)tableCap, "tableOut");
    if (!t) return;
    int64_t counts[PLAAC_NAA];
    if (plaac_node_text_table(node_of(node), t, (uint64_t)tableCap, counts22 ? counts : nullptr) != PLAAC_OK) return raise(env, node_error(node));
    if (counts22) put_counts(env, counts22, counts);
}

// the oldest batch as rows and what the device parsed (plaac_score_end_text): rowsOut nrec * ROW_BYTES, offsetsOut nrec + 1
// longs, blankEndOut nrec bytes, extentsOut null or 2 * nrec ints, codesOut null or textLen bytes; counts22 null or long[22]
JNIEXPORT void JNICALL Java_PlaacNative_textRows(JNIEnv *env, jclass, jlong node, jobject rowsOut, jobject codesOut, jobject offsetsOut,
                                                 jobject blankEndOut, jobject extentsOut, jlongArray counts22) {
    if (counts22 && env->GetArrayLength(counts22) != PLAAC_NAA) return raise(env, "textRows: counts null or long[22]");
    const uint64_t nrec = plaac_node_text_oldest_records(node_of(node));
    plaac_row *rows = (plaac_row *)direct(env, rowsOut, nrec * sizeof(plaac_row), "rowsOut");
    if (!rows) return;
    uint64_t *offs = (uint64_t *)direct(env, offsetsOut, 8ull * (nrec + 1), "offsetsOut");
    if (!offs) return;
    uint8_t *blank = (uint8_t *)direct(env, blankEndOut, nrec, "blankEndOut");
    if (!blank) return;
    uint32_t *ext = nullptr;
    if (extentsOut && !(ext = (uint32_t *)direct(env, extentsOut, 8ull * nrec, "extentsOut"))) return;
    uint8_t *codes = nullptr;
    uint64_t cap = 0;
    if (codesOut) {
        codes = (uint8_t *)direct(env, codesOut, 0, "codesOut");
        if (!codes) return;
        cap = (uint64_t)env->GetDirectBufferCapacity(codesOut);
    }
    int64_t counts[PLAAC_NAA];
    if (plaac_node_text_rows(node_of(node), rows, codes, cap, offs, blank, ext, counts22 ? counts : nullptr) != PLAAC_OK)
        return raise(env, node_error(node));
    if (counts22) put_counts(env, counts22, counts);
}

JNIEXPORT void JNICALL Java_PlaacNative_textDiscard(JNIEnv *env, jclass, jlong node) {
    if (plaac_node_text_discard(node_of(node)) != PLAAC_OK) raise(env, node_error(node));
}
JNIEXPORT jint JNICALL Java_PlaacNative_textPending(JNIEnv *, jclass, jlong node) { return plaac_node_text_pending(node_of(node)); }
JNIEXPORT jint JNICALL Java_PlaacNative_textOldestRecords(JNIEnv *, jclass, jlong node) {
    return (jint)plaac_node_text_oldest_records(node_of(node));
}
JNIEXPORT void JNICALL Java_PlaacNative_textReset(JNIEnv *, jclass, jlong node) { plaac_node_text_reset(node_of(node)); }

// an uploader thread's half of textBegin (plaac_node_text_upload): returns the uploaded batch's handle
JNIEXPORT jlong JNICALL Java_PlaacNative_textUpload(JNIEnv *env, jclass, jlong node, jobject text, jlong textLen, jobject starts, jint nrec) {
    if (nrec < 0 || textLen < 0) {
        raise(env, "textUpload: negative size");
        return 0;
    }
    const uint64_t *st = (const uint64_t *)direct(env, starts, 8ull * ((uint64_t)nrec + 1), "starts");
    if (!st) return 0;
    const char *t = (const char *)direct(env, text, (uint64_t)textLen, "text");
    if (!t) return 0;
    plaac_node_text_batch *b = nullptr;
    if (plaac_node_text_upload(node_of(node), t, (uint64_t)textLen, st, (uint32_t)nrec, &b) != PLAAC_OK) {
        raise(env, "textUpload failed (out of memory on the host or the device, or a malformed batch)");
        return 0;
    }
    return (jlong)reinterpret_cast<intptr_t>(b);
}
JNIEXPORT void JNICALL Java_PlaacNative_textBeginUploaded(JNIEnv *env, jclass, jlong node, jlong batch, jboolean counting) {
    plaac_node_text_batch *b = reinterpret_cast<plaac_node_text_batch *>(static_cast<intptr_t>(batch));
    if (plaac_node_text_begin_uploaded(node_of(node), b, counting ? 1 : 0) != PLAAC_OK) raise(env, node_error(node));
}
JNIEXPORT void JNICALL Java_PlaacNative_textBatchFree(JNIEnv *, jclass, jlong batch) {
    plaac_node_text_batch_free(reinterpret_cast<plaac_node_text_batch *>(static_cast<intptr_t>(batch)));
}

// the counting pass of a two-pass run over text (computeaafreq, :1655-1666): begin per batch, end per batch in the same
// order; counts22 / residues1 are ADDED to (zero them before a file)
JNIEXPORT void JNICALL Java_PlaacNative_histogramTextBegin(JNIEnv *env, jclass, jlong node, jobject text, jlong textLen, jobject starts,
                                                           jint nrec) {
    if (nrec < 0 || textLen < 0) return raise(env, "histogramTextBegin: negative size");
    const uint64_t *st = (const uint64_t *)direct(env, starts, 8ull * ((uint64_t)nrec + 1), "starts");
    if (!st) return;
    const char *t = (const char *)direct(env, text, (uint64_t)textLen, "text");
    if (!t) return;
    if (plaac_node_histogram_text_begin(node_of(node), t, (uint64_t)textLen, st, (uint32_t)nrec) != PLAAC_OK) raise(env, node_error(node));
}
JNIEXPORT void JNICALL Java_PlaacNative_histogramTextEnd(JNIEnv *env, jclass, jlong node, jlongArray counts22, jlongArray residues1) {
    if (!counts22 || env->GetArrayLength(counts22) != PLAAC_NAA || (residues1 && env->GetArrayLength(residues1) != 1))
        return raise(env, "histogramTextEnd: a long[22] (and null or a long[1]) are required");
    jlong cur[PLAAC_NAA], res[1] = {0};
    env->GetLongArrayRegion(counts22, 0, PLAAC_NAA, cur);
    if (residues1) env->GetLongArrayRegion(residues1, 0, 1, res);
    int64_t counts[PLAAC_NAA];
    for (int i = 0; i < PLAAC_NAA; ++i) counts[i] = (int64_t)cur[i];
    uint64_t r = (uint64_t)res[0];
    if (plaac_node_histogram_text_end(node_of(node), counts, &r) != PLAAC_OK) return raise(env, node_error(node));
    put_counts(env, counts22, counts);
    if (residues1) {
        res[0] = (jlong)r;
        env->SetLongArrayRegion(residues1, 0, 1, res);
    }
}

// plotsomefastas' per-residue table (:587-649) made on the device (plaac_node_score_tracks_table): returns the table as a
// byte[] - or null when a value needs the host's formatter (then score(..., tracks) and format on the host as before).
// labels = for record k the bytes "ORDER \t SEQid" at labelOff[k] .. labelOff[k+1]; rowsOut null or nprot * ROW_BYTES.
JNIEXPORT jbyteArray JNICALL Java_PlaacNative_tracksTable(JNIEnv *env, jclass, jlong node, jobject codes, jobject offsets, jint nprot,
                                                          jobject labels, jobject labelOff, jobject rowsOut) {
    if (nprot < 0) {
        raise(env, "tracksTable: negative nprot");
        return nullptr;
    }
    const uint64_t *off = (const uint64_t *)direct(env, offsets, 8ull * ((uint64_t)nprot + 1), "offsets");
    if (!off) return nullptr;
    const uint8_t *c = (const uint8_t *)direct(env, codes, off[nprot], "codes");
    if (!c) return nullptr;
    const uint64_t *lo = (const uint64_t *)direct(env, labelOff, 8ull * ((uint64_t)nprot + 1), "labelOff");
    if (!lo) return nullptr;
    const char *lb = (const char *)direct(env, labels, lo[nprot], "labels");
    if (!lb) return nullptr;
    plaac_row *rows = nullptr;
    if (rowsOut && !(rows = (plaac_row *)direct(env, rowsOut, (uint64_t)nprot * sizeof(plaac_row), "rowsOut"))) return nullptr;
    char *table = nullptr;
    uint64_t len = 0;
    int needs = 0;
    if (plaac_node_score_tracks_table(node_of(node), c, off, (uint32_t)nprot, lb, lo, rows, &table, &len, &needs) != PLAAC_OK) {
        raise(env, node_error(node));
        return nullptr;
    }
    if (needs || !table) {
        plaac_table_free(table);
        return nullptr;
    }
    if (len > 0x7fffffffull) {
        plaac_table_free(table);
        raise(env, "tracksTable: the table exceeds a Java array (select fewer records per call)");
        return nullptr;
    }
    jbyteArray out = env->NewByteArray((jsize)len);
    if (out) env->SetByteArrayRegion(out, 0, (jsize)len, (const jbyte *)table);
    plaac_table_free(table);
    return out;
}

} // extern "C"
