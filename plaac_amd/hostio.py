"""ctypes binding of include/plaac_host.h: the FASTA reader and the Java-compatible text formatting
that live in libplaac_native.so (C++). Python adds nothing but marshalling."""
import ctypes as C

import numpy as np

from . import native


class _Fasta(C.Structure):
    _fields_ = [("nrec", C.c_uint32), ("nres", C.c_uint64), ("codes", C.POINTER(C.c_uint8)),
                ("offsets", C.POINTER(C.c_uint64)), ("names", C.POINTER(C.c_char)),
                ("name_off", C.POINTER(C.c_uint64))]


HOST_EXPORTS = (
    "plaac_fasta_read", "plaac_fasta_free", "plaac_fasta_next_text", "plaac_fasta_text_free", "plaac_fasta_text_trim_names", "plaac_fasta_text_codes", "plaac_format_summary_row_n", "plaac_read_aa_params", "plaac_format_fixed",
    "plaac_format_fixed_reference",
    "plaac_format_double_tostring", "plaac_format_summary_row", "plaac_summary_header", "plaac_tracks_header",
    "plaac_format_track_rows", "plaac_track_rows_bound", "plaac_format_param_block", "plaac_format_aa_params",
    "plaac_host_threads", "plaac_format_hmm_dot", "plaac_fasta_open", "plaac_fasta_next", "plaac_fasta_close",
)

_ready = False


def _lib():
    global _ready
    L = native.load()
    if not _ready:
        L.plaac_fasta_read.argtypes = [C.c_char_p, C.POINTER(C.POINTER(_Fasta))]
        L.plaac_fasta_free.argtypes = [C.POINTER(_Fasta)]
        L.plaac_fasta_free.restype = None
        L.plaac_fasta_open.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        L.plaac_fasta_next.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(C.POINTER(_Fasta))]
        L.plaac_fasta_close.argtypes = [C.c_void_p]
        L.plaac_fasta_close.restype = None
        L.plaac_read_aa_params.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p]
        L.plaac_format_fixed.argtypes = [C.c_double, C.c_int, C.c_char_p, C.c_size_t]
        L.plaac_format_fixed_reference.argtypes = [C.c_double, C.c_int, C.c_char_p, C.c_size_t]
        L.plaac_format_double_tostring.argtypes = [C.c_double, C.c_char_p, C.c_size_t]
        L.plaac_format_summary_row.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int,
                                               C.c_char_p, C.c_size_t]
        L.plaac_format_summary_row.restype = C.c_long
        L.plaac_summary_header.restype = C.c_char_p
        L.plaac_tracks_header.restype = C.c_char_p
        L.plaac_format_track_rows.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_char_p, C.c_char_p,
                                              C.c_char_p, C.c_size_t]
        L.plaac_format_track_rows.restype = C.c_long
        L.plaac_track_rows_bound.argtypes = [C.c_uint32, C.c_size_t, C.c_size_t]
        L.plaac_track_rows_bound.restype = C.c_size_t
        L.plaac_format_param_block.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.plaac_format_param_block.restype = C.c_long
        L.plaac_format_aa_params.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.plaac_format_aa_params.restype = C.c_long
        L.plaac_format_hmm_dot.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.plaac_format_hmm_dot.restype = C.c_long
        _ready = True
    return L


def _take(pf):
    """(names, codes, offsets) out of a plaac_fasta*, which is freed"""
    f = pf.contents
    n = f.nrec
    codes = np.ctypeslib.as_array(f.codes, shape=(max(int(f.nres), 1),))[:int(f.nres)].copy()
    offs = np.ctypeslib.as_array(f.offsets, shape=(n + 1,)).copy()
    noff = np.ctypeslib.as_array(f.name_off, shape=(n + 1,)).copy()
    blob = C.string_at(f.names, int(noff[-1]))
    names = [blob[int(noff[i]):int(noff[i + 1]) - 1] for i in range(n)]
    _lib().plaac_fasta_free(pf)
    return names, codes, offs


def read_fasta(path):
    """-> (names list[bytes], codes u8, offsets u64) exactly as the reference's fastareader splits the file"""
    pf = C.POINTER(_Fasta)()
    st = _lib().plaac_fasta_read(str(path).encode(), C.byref(pf))
    if st != native.PLAAC_OK:
        raise native.PlaacError(st, "cannot read " + str(path))
    return _take(pf)


def stream_fasta(path, max_records=262144, max_bytes=128 << 20):
    """the same records as read_fasta, as a generator of (names, codes, offsets) batches (plaac_fasta_open / _next)"""
    L = _lib()
    h = C.c_void_p()
    st = L.plaac_fasta_open(str(path).encode(), C.byref(h))
    if st != native.PLAAC_OK:
        raise native.PlaacError(st, "cannot read " + str(path))
    try:
        while True:
            pf = C.POINTER(_Fasta)()
            st = L.plaac_fasta_next(h, int(max_records), int(max_bytes), C.byref(pf))
            if st != native.PLAAC_OK:
                raise native.PlaacError(st, "plaac_fasta_next")
            if not pf:
                return
            yield _take(pf)
    finally:
        L.plaac_fasta_close(h)


class _FastaText(C.Structure):
    """plaac_fasta_text"""
    _fields_ = [("text", C.c_void_p), ("len", C.c_uint64), ("nrec", C.c_uint32), ("starts", C.POINTER(C.c_uint64)),
                ("name_len", C.POINTER(C.c_uint32)), ("prev_blank", C.c_int), ("last_blank", C.c_int), ("owner_", C.c_void_p),
                ("file_off_", C.c_uint64)]


def stream_fasta_text(path, max_records=262144, max_bytes=128 << 20):
    """the stream as batches of TEXT for the device-side parser (plaac_fasta_next_text): a generator of
    (text bytes, starts u64[nrec + 1], trim) where trim(blank_end, prev_blank) -> (names list[bytes], next prev_blank)
    applies the reference's name trimming once the device has reported how the records ended"""
    L = _lib()
    L.plaac_fasta_next_text.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(C.POINTER(_FastaText))]
    L.plaac_fasta_text_free.argtypes = [C.POINTER(_FastaText)]
    L.plaac_fasta_text_free.restype = None
    L.plaac_fasta_text_trim_names.argtypes = [C.POINTER(_FastaText), C.c_void_p, C.c_int]
    h = C.c_void_p()
    st = L.plaac_fasta_open(str(path).encode(), C.byref(h))
    if st != native.PLAAC_OK:
        raise native.PlaacError(st, "cannot read " + str(path))
    try:
        while True:
            pt = C.POINTER(_FastaText)()
            st = L.plaac_fasta_next_text(h, int(max_records), int(max_bytes), C.byref(pt))
            if st != native.PLAAC_OK:
                raise native.PlaacError(st, "plaac_fasta_next_text")
            if not pt:
                return
            t = pt.contents
            n = t.nrec
            text = C.string_at(t.text, int(t.len))
            starts = np.ctypeslib.as_array(t.starts, shape=(n + 1,)).copy()

            trim_flags = (int(t.prev_blank), int(t.last_blank))  # (what the reader itself says about the batch's two ends)

            def trim(blank_end, prev_blank, pt=pt, n=n, text=text, starts=starts):
                be = np.ascontiguousarray(blank_end, dtype=np.uint8)
                nxt = L.plaac_fasta_text_trim_names(pt, be.ctypes.data, int(prev_blank))
                nlen = np.ctypeslib.as_array(pt.contents.name_len, shape=(max(n, 1),))
                names = [text[int(starts[i]) + 1:int(starts[i]) + 1 + int(nlen[i])] for i in range(n)]
                return names, int(nxt)

            trim.flags = trim_flags
            yield text, starts, trim
            L.plaac_fasta_text_free(pt)
    finally:
        L.plaac_fasta_close(h)


def text_codes(text, starts, extents, i, first, count):
    """plaac_fasta_text_codes: residues [first, first + count) of record i of a text batch, read from the text with the
    extents the device reported (Context.score_text(..., want_codes=False))"""
    L = _lib()
    L.plaac_fasta_text_codes.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p]
    L.plaac_fasta_text_codes.restype = C.c_uint64
    starts = np.ascontiguousarray(starts, dtype=np.uint64)
    extents = np.ascontiguousarray(extents, dtype=np.uint32)
    out = np.zeros(max(int(count), 1), dtype=np.uint8)
    k = L.plaac_fasta_text_codes(text, starts.ctypes.data, extents.ctypes.data, int(i), int(first), int(count), out.ctypes.data)
    return out[:int(k)]


def read_aa_params(path):
    vec = np.zeros(22)
    warn = np.zeros(22, dtype=np.int32)
    st = _lib().plaac_read_aa_params(str(path).encode(), vec.ctypes.data, warn.ctypes.data)
    if st != native.PLAAC_OK:
        raise native.PlaacError(st, "cannot read " + str(path))
    return vec, warn


def format_fixed(v, decimals):
    buf = C.create_string_buffer(512)
    _lib().plaac_format_fixed(float(v), int(decimals), buf, 512)
    return buf.value.decode()


def format_fixed_reference(v, decimals):
    buf = C.create_string_buffer(512)
    _lib().plaac_format_fixed_reference(float(v), int(decimals), buf, 512)
    return buf.value.decode()


def double_tostring(v):
    buf = C.create_string_buffer(64)
    _lib().plaac_format_double_tostring(float(v), buf, 64)
    return buf.value.decode()


def summary_header():
    return _lib().plaac_summary_header().decode()


def tracks_header():
    return _lib().plaac_tracks_header().decode()


def format_summary_row(row, name, codes, corelength=60, ww2=41):
    """row: one element of a ROW_DTYPE array; codes: the record's untrimmed codes"""
    r = np.array([row], dtype=native.ROW_DTYPE)
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    cap = 3 * len(codes) + len(name) + 4096
    buf = C.create_string_buffer(cap)
    k = _lib().plaac_format_summary_row(r.ctypes.data, name if isinstance(name, bytes) else name.encode(),
                                        codes.ctypes.data, len(codes), corelength, ww2, buf, cap)
    if k < 0:
        raise RuntimeError("summary row buffer too small")
    return buf.raw[:k].decode()


def format_summary_rows(rows, names, codes, offsets, corelength=60, ww2=41):
    """The table lines (newline-terminated) of a whole slice as one bytes object: rows = ROW_DTYPE array, names = list of
    bytes, codes / offsets = the slice's untrimmed residue codes and its n + 1 offsets (relative to codes). One
    plaac_format_summary_row_n call per record into one buffer; records without a line (prot_len 0) add nothing."""
    L = _lib()
    rows = np.ascontiguousarray(rows, dtype=native.ROW_DTYPE)
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    offs = np.asarray(offsets, dtype=np.int64)
    n = len(rows)
    f = L.plaac_format_summary_row_n
    f.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    f.restype = C.c_long
    lens = np.diff(offs)
    cap = int(3 * lens.sum() + sum(len(x) for x in names) + n * (ww2 + 8701) + 1)
    buf = np.empty(cap, dtype=np.uint8)
    base, at = buf.ctypes.data, 0
    rbase, cbase, rb = rows.ctypes.data, codes.ctypes.data, rows.dtype.itemsize
    for i in range(n):
        nm = names[i]
        k = f(rbase + i * rb, nm, len(nm), cbase + int(offs[i]), int(lens[i]), corelength, ww2, base + at, cap - at)
        if k < 0:
            raise RuntimeError("summary row buffer too small")
        if k:
            buf[at + k] = 10
            at += k + 1
    return buf[:at].tobytes()


def format_track_rows(tracks, first, codes, n, order_id, name):
    T = native.Tracks(**{k: tracks[k].ctypes.data for k in native.TRACK_U8 + native.TRACK_F64})
    codes = np.ascontiguousarray(codes, dtype=np.uint8)
    cap = _lib().plaac_track_rows_bound(n, len(order_id), len(name))
    buf = C.create_string_buffer(cap)
    k = _lib().plaac_format_track_rows(C.addressof(T), first, codes.ctypes.data, n, order_id.encode(), name.encode(),
                                       buf, cap)
    if k < 0:
        raise RuntimeError("track rows buffer too small")
    return buf.raw[:k].decode()


def format_param_block(P):
    buf = C.create_string_buffer(16384)
    k = _lib().plaac_format_param_block(C.addressof(P), buf, 16384)
    return buf.raw[:k].decode()


def format_aa_params(vec):
    vec = np.ascontiguousarray(vec, dtype=np.float64)
    buf = C.create_string_buffer(2048)
    k = _lib().plaac_format_aa_params(vec.ctypes.data, buf, 2048)
    return buf.raw[:k].decode()


def format_hmm_dot(P):
    buf = C.create_string_buffer(16384)
    k = _lib().plaac_format_hmm_dot(C.addressof(P), buf, 16384)
    return buf.raw[:k].decode()
