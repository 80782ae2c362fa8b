// PlaacNative.java — the Java side of the JNI binding of libplaac_native.so (MI355X-native PLAAC engine).
//
// Goes next to plaac.java (cli/src/ of whitehead/plaac, default package like every class there). The natives replace
// the bodies of the reference's per-protein loops:
//   plaac.java:444-500 (table setup in main)            -> paramsInit
//   plaac.java:1655-1666 (computeaafreq, pass 1)        -> histogram
//   plaac.java:759-880 (scoreallfastas body)            -> score(..., null)
//   plaac.java:625-633 (plotsomefastas body)            -> score(..., tracks)
// All buffers are DIRECT ByteBuffers in native byte order (zero copy: the shim hands their addresses to the C ABI);
// one JNI call per batch. A non-zero plaac_status becomes an IllegalStateException carrying plaac_last_error().
import java.nio.ByteBuffer;

final class PlaacNative {
    static { System.loadLibrary("plaac_jni"); } // thin shim (jni/plaac_jni.cpp) linked against libplaac_native.so

    static final int NAA = 22;        // X A C D E F G H I K L M N P Q R S T V W Y *  (plaac.java:26)
    static final int ROW_BYTES = 160; // sizeof(plaac_row): 13 doubles then 14 ints, see include/plaac_native.h
    static final int TRACKS = 12;     // vit, map (1 byte/residue); charge hydro fi plaacllr papa fix2 plaacllrx2 papax2 post0 post1

    private PlaacNative() {}

    /** plaac_abi_version() of the loaded library (the shim's JNI_OnLoad has already refused another generation) */
    static native int abiVersion();
    /** visible gfx950 devices */
    static native int deviceCount();
    /** sizeof(plaac_params): capacity of the params buffers below */
    static native int paramsBytes();
    /** table setup of main (fg / bgCounts may be null = built-in prd_freq_scer_28 / all zero) */
    static native void paramsInit(ByteBuffer paramsOut, double[] fg, double[] bgCounts, double alpha, int corelength,
                                  int ww1, int ww2, int ww3, boolean adjustProlines);
    /** string2aa: n text bytes -> n residue codes */
    static native void encode(ByteBuffer text, int n, ByteBuffer codesOut);
    /** one scoring context per listed device (null: every visible device); returns the plaac_node handle */
    static native long nodeCreate(ByteBuffer params, int[] devices);
    static native void nodeSetParams(long node, ByteBuffer params);
    static native void nodeDestroy(long node);
    /** 22 residue counts over the valid records of the batch (codes untrimmed, offsets = nprot + 1 longs) */
    static native void histogram(long node, ByteBuffer codes, ByteBuffer offsets, int nprot, long[] counts22);
    /** rowsOut: nprot * ROW_BYTES; tracks: null (summary mode) or TRACKS direct buffers of total-residue elements */
    static native void score(long node, ByteBuffer codes, ByteBuffer offsets, int nprot, ByteBuffer rowsOut,
                             ByteBuffer[] tracks);

    // ---- resident batches: ONE upload for the background pass (plaac.java:377-384), the scoring pass (:755) and every
    // point of a parameter sweep - the reference runs a sweep as one main() per point (plaac.java:337-353,
    // web/lib/server.rb:152-155), i.e. it reads and encodes the same proteome once per point. The batch is cut over the
    // node's devices by length-sorted dealing (plaac_shard_plan); rows come back in input order.
    // A direct ByteBuffer holds at most 2^31 - 1 bytes: upload a proteome above 2.1 G residues as several batches.
    /** uploads the batch to the node's devices; returns the plaac_node_batch handle */
    static native long batchUpload(long node, ByteBuffer codes, ByteBuffer offsets, int nprot);
    /** required for every batch, also one whose node was destroyed first (such a batch is detached: its calls throw) */
    static native void batchFree(long batch);
    static native void batchHistogram(long batch, long[] counts22);
    /** scores with the node's CURRENT parameters (nodeSetParams between calls: the two-pass run on one upload) */
    static native void batchScore(long batch, ByteBuffer rowsOut, ByteBuffer[] tracks);
    /** npoints parameter sets (params: npoints * paramsBytes() bytes) over the resident batch; rowsOut[i]: nprot * ROW_BYTES */
    static native void batchSweep(long batch, ByteBuffer params, int npoints, ByteBuffer[] rowsOut);
    /** pipelines of batches: consecutive scoring calls of a context may overlap on the device (plaac_ctx_set_overlap) */
    static native void nodeSetOverlap(long node, boolean on);

    // ---- FASTA text in, table text out (plaac_node_text_*): the bytes of whole records ('>' of record i at starts[i],
    // starts[nrec] = textLen) in FILE ORDER; the device parses (fastareader, plaac.java:4302-4375), scores and writes
    // scoreallfastas' lines (:899-945). The node deals the batches over its devices (2 x devices may be in flight) and every
    // collecting native serves the OLDEST pending batch, so the table comes back in file order. textReset() before a new file.
    static native void textBegin(long node, ByteBuffer text, long textLen, ByteBuffer starts, int nrec, boolean counting);
    /** waits for the oldest batch; returns its table's size; out2 = {needsHost, residues} */
    static native long textTableSize(long node, int corelength, int ww2, long[] out2);
    /** the oldest batch's table text (after textTableSize said needsHost == 0); gives the batch up */
    static native void textTable(long node, ByteBuffer tableOut, long tableCap, long[] counts22);
    /** needsHost != 0 (a value >= 1e9, an infinity, a record without a sequence): the oldest batch as rows + what the device
     *  parsed, for the host's own formatter (plaac_score_end_text); gives the batch up. textOldestRecords() sizes the buffers. */
    static native void textRows(long node, ByteBuffer rowsOut, ByteBuffer codesOut, ByteBuffer offsetsOut, ByteBuffer blankEndOut,
                                ByteBuffer extentsOut, long[] counts22);
    /** gives the oldest batch up without collecting it */
    static native void textDiscard(long node);
    static native int textPending(long node);
    static native int textOldestRecords(long node);
    static native void textReset(long node);
    /** an uploader thread's half of textBegin: upload + parse now, score later (returns the uploaded batch's handle) */
    static native long textUpload(long node, ByteBuffer text, long textLen, ByteBuffer starts, int nrec);
    static native void textBeginUploaded(long node, long uploadedBatch, boolean counting);
    static native void textBatchFree(long uploadedBatch);
    // the counting pass of a two-pass run over text (computeaafreq, plaac.java:1655-1666): one begin and, in the same order,
    // one end per batch; counts22 / residues1 are added to
    static native void histogramTextBegin(long node, ByteBuffer text, long textLen, ByteBuffer starts, int nrec);
    static native void histogramTextEnd(long node, long[] counts22, long[] residues1);
    /** plotsomefastas (plaac.java:587-649): the per-residue table of the selected records, made on the device; labels = for
     *  record k the bytes "ORDER\tSEQid" at labelOff[k] .. labelOff[k+1]; null when a value needs the host's formatter */
    static native byte[] tracksTable(long node, ByteBuffer codes, ByteBuffer offsets, int nprot, ByteBuffer labels,
                                     ByteBuffer labelOff, ByteBuffer rowsOut);
}
