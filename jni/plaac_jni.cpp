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

// ---- FASTA text in, summary-table text out (round 5): fastareader.nextfasta (:4325-4357) in front of the scoring loop and the
//      output line of scoreallfastas (:899-945) behind it run on the device; a Java host hands over the bytes of whole records
//      and where each begins, and gets the table's bytes back. Context k of the node; two batches may be pending per context.
static plaac_ctx *ctx_of(JNIEnv *env, jlong node, jint k) {
    plaac_ctx *c = plaac_node_ctx(node_of(node), k);
    if (!c) raise(env, "no such context on this node");
    return c;
}

JNIEXPORT void JNICALL Java_PlaacNative_textBegin(JNIEnv *env, jclass, jlong node, jint k, jobject text, jlong textLen,
                                                  jobject starts, jint nrec, jboolean counting) {
    plaac_ctx *c = ctx_of(env, node, k);
    if (!c) return;
    if (nrec < 0 || textLen < 0) return raise(env, "textBegin: negative size");
    const uint64_t *st = (const uint64_t *)direct(env, starts, 8ull * ((uint64_t)nrec + 1), "starts");
    if (!st) return;
    const char *t = (const char *)direct(env, text, (uint64_t)textLen, "text");
    if (!t) return;
    if (plaac_score_begin_text(c, t, (uint64_t)textLen, st, (uint32_t)nrec, counting ? 1 : 0) != PLAAC_OK) raise(env, plaac_last_error(c));
}

// returns the table's size in bytes; out3 = {needsHost, lastBlank, residues}. needsHost != 0: the device will not vouch for a
// value of this batch - collect it with the row-level natives (not bound here: score / batchScore) after formatting on the host
JNIEXPORT jlong JNICALL Java_PlaacNative_textTableSize(JNIEnv *env, jclass, jlong node, jint k, jint corelength, jint ww2,
                                                       jint prevBlank, jlongArray out3) {
    plaac_ctx *c = ctx_of(env, node, k);
    if (!c) return -1;
    if (!out3 || env->GetArrayLength(out3) != 3) {
        raise(env, "textTableSize: a long[3] is required");
        return -1;
    }
    uint64_t bytes = 0, residues = 0;
    int needs = 0, lastb = 0;
    if (plaac_score_end_text_table_size(c, corelength, ww2, prevBlank, &bytes, &needs, &lastb, &residues) != PLAAC_OK) {
        raise(env, plaac_last_error(c));
        return -1;
    }
    const jlong out[3] = {(jlong)needs, (jlong)lastb, (jlong)residues};
    env->SetLongArrayRegion(out3, 0, 3, out);
    return (jlong)bytes;
}

JNIEXPORT void JNICALL Java_PlaacNative_textTable(JNIEnv *env, jclass, jlong node, jint k, jobject tableOut, jlong tableCap,
                                                  jlongArray counts22) {
    plaac_ctx *c = ctx_of(env, node, k);
    if (!c) return;
    if (tableCap < 0 || (counts22 && env->GetArrayLength(counts22) != PLAAC_NAA)) return raise(env, "textTable: capacity >= 0, counts null or long[22]");
    char *t = (char *)direct(env, tableOut, (uint64_t)tableCap, "tableOut");
    if (!t) return;
    int64_t counts[PLAAC_NAA];
    if (plaac_score_end_text_table(c, t, (uint64_t)tableCap, counts22 ? counts : nullptr) != PLAAC_OK) return raise(env, plaac_last_error(c));
    if (counts22) {
        jlong out[PLAAC_NAA];
        for (int i = 0; i < PLAAC_NAA; ++i) out[i] = (jlong)counts[i];
        env->SetLongArrayRegion(counts22, 0, PLAAC_NAA, out);
    }
}

} // extern "C"
