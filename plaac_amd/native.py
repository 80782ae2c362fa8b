"""ctypes binding of the C ABI in include/plaac_native.h (libplaac_native.so).

This is the host-side mirror used by tests, bench.py and the Python tooling. It contains no
arithmetic of its own: every number comes out of the HIP kernels behind the C ABI. There is no
CPU fallback; if the shared library is missing or no gfx950 device is usable the calls raise.
"""
import ctypes as C
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PLAAC_NATIVE_LIB: another build of the same library (tools/kb_probe.sh uses the cost-breakdown build)
LIB_PATH = os.environ.get("PLAAC_NATIVE_LIB") or os.path.join(_HERE, "libplaac_native.so")

NAA = 22
LUTLEN = 4001
ABI_VERSION = 2

PLAAC_OK, PLAAC_ERR_ARG, PLAAC_ERR_DEVICE, PLAAC_ERR_NOMEM, PLAAC_ERR_IO, PLAAC_ERR_UNSUPPORTED = range(6)

# Test hooks of the library (plaac_debug_set_knob; include/plaac_native.h "environment and test hooks"): the library itself
# reads eight documented environment variables and nothing else; THIS binding turns PLAAC_<KEY> of its own process
# environment into hook calls whenever a Context / Node is created, so that tests and tools/ keep their switches.
HOOK_KEYS = ("KB_LANE", "KB_LANE_MIN_GROUPS", "MIXED", "MIXED_GROUPS", "MIXED_MIN_REST", "TRACK_CONSEC", "KB_PER_PROTEIN",
             "GENERIC_TRACKS", "CORE_PAR", "CORE_LIST", "FI_INT", "LSE_CLAMP", "SWEEP_LATENCY", "SWEEP_CORE_ASIDE", "KB_PRIO",
             "KB_SIDE", "SEGMENT_MIN_ROWS", "RF_GRID", "LAT_UNIT", "POLL_PLAN", "STREAM_PROBE")
# forms that only the diagnostic build (`make DIAG=1` -> libplaac_native_diag.so, PLAAC_NATIVE_LIB) compiles
DIAG_KEYS = ("SWEEP_REST_ASIDE", "SWEEP_CHAINS_FIRST", "CORE_LONG_LIST", "PIPE_SEGMENTS", "TRACK_SEGMENTS", "TRACK_FUSED",
             "TRACK_VIT_MIXED", "TRACK_CKPT", "FWD_DIRECT", "TRACK_KB_LDS", "TRACK_POST_FORM", "TRACK_POST_OCC", "TRACK_ONE_PASS",
             "TRACK_KB_LATE", "TRACK_VIT_EARLY", "FINISH_KERNEL", "VIT_STOP", "DEBUG_SKIP", "DEBUG_SKIP_FROM", "DEBUG_COUNTER")


class DiagKnob(RuntimeError):
    """the environment asks for a form that only the diagnostic build of the library has (tests: skipped, see conftest.py)"""


def apply_env_knobs(L=None):
    L = L or load()
    if not hasattr(L, "plaac_debug_set_knob"):  # (an older build: it reads its switches from the environment itself)
        return
    for key in HOOK_KEYS + DIAG_KEYS:
        v = os.environ.get("PLAAC_" + key)
        st = L.plaac_debug_set_knob(key.encode(), None if v is None else v.encode())
        if st == PLAAC_ERR_UNSUPPORTED:
            if v is not None:
                raise DiagKnob("PLAAC_%s needs the diagnostic build of the library (make DIAG=1; PLAAC_NATIVE_LIB)" % key)
        elif st != PLAAC_OK:
            raise PlaacError(st, "plaac_debug_set_knob(%s) failed" % key)


class PlaacError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("plaac_native status %d: %s" % (status, msg))
        self.status = status


class Hmm(C.Structure):
    _fields_ = [("lt", (C.c_double * 2) * 2), ("li", C.c_double * 2), ("le", (C.c_double * NAA) * 2),
                ("lf", C.c_double * 2)]


class Params(C.Structure):
    """plaac_params"""
    _fields_ = [("corelength", C.c_int32), ("ww1", C.c_int32), ("ww2", C.c_int32), ("ww3", C.c_int32),
                ("adjustprolines", C.c_int32), ("reserved_", C.c_int32), ("alpha", C.c_double),
                ("cc", C.c_double * 3), ("big_neg", C.c_double),
                ("fg", C.c_double * NAA), ("bgscer", C.c_double * NAA), ("bgthis", C.c_double * NAA),
                ("bg", C.c_double * NAA), ("llr", C.c_double * NAA), ("lodpapa", C.c_double * NAA),
                ("hydro2", C.c_double * NAA), ("charge", C.c_double * NAA), ("hmm1", Hmm), ("hmm0", Hmm),
                ("loglut", C.c_double * LUTLEN)]


ROW_DTYPE = np.dtype([
    ("llr_score", "<f8"), ("core_score", "<f8"), ("prd_score", "<f8"), ("hmm_all", "<f8"), ("hmm_vit", "<f8"),
    ("fi_meanhydro", "<f8"), ("fi_meancharge", "<f8"), ("fi_meancombo", "<f8"),
    ("papa_combo", "<f8"), ("papa_prop", "<f8"), ("papa_fi", "<f8"), ("papa_llr", "<f8"), ("papa_llr2", "<f8"),
    ("mw_score", "<i4"), ("mw_start", "<i4"), ("mw_end", "<i4"), ("llr_start", "<i4"), ("llr_end", "<i4"),
    ("vit_maxrun", "<i4"), ("core_start", "<i4"), ("core_end", "<i4"), ("prd_start", "<i4"), ("prd_end", "<i4"),
    ("prot_len", "<i4"), ("fi_numaa", "<i4"), ("fi_maxrun", "<i4"), ("papa_cen", "<i4"),
])
ROW_BYTES = 160
assert ROW_DTYPE.itemsize == ROW_BYTES

TRACK_U8 = ("vit", "map")
TRACK_F64 = ("charge", "hydro", "fi", "plaacllr", "papa", "fix2", "plaacllrx2", "papax2", "post0", "post1")
TRACK_BYTES_PER_RESIDUE = 2 + 8 * len(TRACK_F64)  # 82


class Tracks(C.Structure):
    """plaac_tracks"""
    _fields_ = [(k, C.c_void_p) for k in TRACK_U8 + TRACK_F64]


# every symbol include/plaac_native.h declares
EXPORTS = (
    "plaac_abi_version", "plaac_sizeof_params", "plaac_sizeof_row", "plaac_builtin_tables", "plaac_params_init",
    "plaac_encode", "plaac_ctx_create", "plaac_ctx_set_params", "plaac_ctx_destroy", "plaac_last_error",
    "plaac_histogram", "plaac_score", "plaac_score_device", "plaac_histogram_device", "plaac_ctx_sync",
    "plaac_last_timings", "plaac_timings_mean", "plaac_batch_upload", "plaac_batch_histogram", "plaac_batch_score",
    "plaac_batch_free", "plaac_batch_sweep", "plaac_score_sweep_device", "plaac_last_exact_fallbacks",
    "plaac_fi_integer_form", "plaac_calibration_reads", "plaac_clock_probe", "plaac_ctx_set_overlap",
    "plaac_device_count", "plaac_node_create", "plaac_node_destroy", "plaac_node_size", "plaac_node_ctx",
    "plaac_node_set_params", "plaac_node_histogram", "plaac_node_score", "plaac_node_last_error",
    "plaac_node_set_overlap", "plaac_shard_plan", "plaac_node_batch_upload", "plaac_node_batch_histogram",
    "plaac_node_batch_score", "plaac_node_batch_sweep", "plaac_node_batch_free", "plaac_node_batch_records",
    "plaac_node_batch_residues", "plaac_node_batch_last_error", "plaac_rows_to_wire", "plaac_rows_from_wire", "plaac_score_begin", "plaac_score_end", "plaac_score_begin_counting", "plaac_score_end_counts", "plaac_score_begin_text", "plaac_score_end_text", "plaac_score_end_text_table_size", "plaac_score_end_text_table", "plaac_histogram_begin_text", "plaac_histogram_end_text", "plaac_text_upload", "plaac_score_begin_uploaded", "plaac_text_batch_free", "plaac_score_tracks_table", "plaac_table_free", "plaac_debug_schedule",
    "plaac_node_text_begin", "plaac_node_text_upload", "plaac_node_text_begin_uploaded", "plaac_node_text_batch_free",
    "plaac_node_text_table_size", "plaac_node_text_table", "plaac_node_text_rows", "plaac_node_text_discard",
    "plaac_node_text_pending", "plaac_node_text_oldest_records", "plaac_node_text_reset", "plaac_node_histogram_text_begin",
    "plaac_node_histogram_text_end", "plaac_node_score_tracks_table", "plaac_debug_set_knob", "plaac_diag_build", "plaac_text_upload_error", "plaac_ctx_set_value_tolerance",
)

_lib = None


def load():
    """Load libplaac_native.so. Raises (loudly) when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "plaac_amd: %s is missing - build the HIP extension first (`make` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback." % LIB_PATH)
    if os.environ.get("PLAAC_AMD_NO_TORCH", "0") != "1":
        # PyTorch bundles its own libamdhip64 (same soname). Loading torch FIRST makes this library bind
        # to that one runtime, so device pointers and streams can be shared with torch tensors.
        try:
            import torch  # noqa: F401
        except Exception:  # torch is plumbing, not a requirement
            pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    L.plaac_abi_version.restype = C.c_int
    L.plaac_sizeof_params.restype = C.c_size_t
    L.plaac_sizeof_row.restype = C.c_size_t
    if L.plaac_abi_version() != ABI_VERSION:
        raise ImportError("plaac_amd: ABI version mismatch (%d != %d)" % (L.plaac_abi_version(), ABI_VERSION))
    if L.plaac_sizeof_params() != C.sizeof(Params) or L.plaac_sizeof_row() != ROW_BYTES:
        raise ImportError("plaac_amd: struct layout mismatch between native.py and plaac_native.h")
    L.plaac_builtin_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.plaac_builtin_tables.restype = None
    L.plaac_params_init.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int,
                                    C.c_int, C.c_int]
    L.plaac_encode.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
    L.plaac_encode.restype = None
    L.plaac_ctx_create.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    L.plaac_ctx_set_params.argtypes = [C.c_void_p, C.c_void_p]
    L.plaac_ctx_destroy.argtypes = [C.c_void_p]
    L.plaac_ctx_destroy.restype = None
    L.plaac_last_error.argtypes = [C.c_void_p]
    L.plaac_last_error.restype = C.c_char_p
    L.plaac_histogram.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.plaac_score.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.plaac_debug_schedule.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.plaac_debug_schedule.restype = C.c_long
    L.plaac_score_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    L.plaac_score_end.argtypes = [C.c_void_p, C.c_void_p]
    L.plaac_score_begin_counting.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    L.plaac_score_end_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.plaac_score_begin_text.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_int]
    L.plaac_score_end_text.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.plaac_score_end_text_table_size.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p]
    L.plaac_score_end_text_table.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.plaac_score_tracks_table.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_char_p, C.c_void_p, C.c_void_p,
                                           C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    L.plaac_table_free.argtypes = [C.c_void_p]
    L.plaac_table_free.restype = None
    L.plaac_text_upload.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
    L.plaac_score_begin_uploaded.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.plaac_text_batch_free.argtypes = [C.c_void_p]
    L.plaac_text_batch_free.restype = None
    L.plaac_histogram_begin_text.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint32]
    L.plaac_histogram_end_text.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    L.plaac_score_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p,
                                     C.c_void_p, C.c_void_p]
    L.plaac_histogram_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.plaac_ctx_sync.argtypes = [C.c_void_p]
    L.plaac_last_timings.argtypes = [C.c_void_p, C.c_void_p]
    L.plaac_timings_mean.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.plaac_last_exact_fallbacks.argtypes = [C.c_void_p, C.c_void_p]
    L.plaac_calibration_reads.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.plaac_clock_probe.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p]
    L.plaac_ctx_set_overlap.argtypes = [C.c_void_p, C.c_int]
    L.plaac_fi_integer_form.argtypes = [C.c_void_p, C.c_void_p]
    L.plaac_fi_integer_form.restype = C.c_int
    L.plaac_device_count.restype = C.c_int
    L.plaac_node_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]
    L.plaac_node_destroy.argtypes = [C.c_void_p]
    L.plaac_node_destroy.restype = None
    L.plaac_node_size.argtypes = [C.c_void_p]
    L.plaac_node_ctx.argtypes = [C.c_void_p, C.c_int]
    L.plaac_node_ctx.restype = C.c_void_p
    L.plaac_node_set_params.argtypes = [C.c_void_p, C.c_void_p]
    L.plaac_node_histogram.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.plaac_node_score.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.plaac_node_last_error.argtypes = [C.c_void_p]
    L.plaac_node_last_error.restype = C.c_char_p
    L.plaac_node_set_overlap.argtypes = [C.c_void_p, C.c_int]
    L.plaac_shard_plan.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    L.plaac_node_batch_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
    L.plaac_node_batch_histogram.argtypes = [C.c_void_p, C.c_void_p]
    L.plaac_node_batch_score.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.plaac_node_batch_sweep.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.plaac_node_batch_free.argtypes = [C.c_void_p]
    L.plaac_node_batch_free.restype = None
    L.plaac_node_batch_records.argtypes = [C.c_void_p]
    L.plaac_node_batch_records.restype = C.c_uint32
    L.plaac_node_batch_residues.argtypes = [C.c_void_p]
    L.plaac_node_batch_residues.restype = C.c_uint64
    L.plaac_rows_to_wire.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.plaac_rows_from_wire.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_void_p]
    L.plaac_node_batch_last_error.argtypes = [C.c_void_p]
    L.plaac_node_batch_last_error.restype = C.c_char_p
    L.plaac_batch_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
    L.plaac_batch_histogram.argtypes = [C.c_void_p, C.c_void_p]
    L.plaac_batch_score.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.plaac_batch_free.argtypes = [C.c_void_p]
    L.plaac_batch_free.restype = None
    L.plaac_batch_sweep.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.plaac_score_sweep_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p,
                                           C.c_uint32, C.c_void_p, C.c_void_p]
    if not hasattr(L, "plaac_debug_set_knob"):  # an older build loaded through PLAAC_NATIVE_LIB for an A/B: no round-6 entry points
        _lib = L
        return L
    L.plaac_node_text_begin.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint32, C.c_int]
    L.plaac_node_text_upload.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]
    L.plaac_node_text_begin_uploaded.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.plaac_node_text_batch_free.argtypes = [C.c_void_p]
    L.plaac_node_text_batch_free.restype = None
    L.plaac_node_text_table_size.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
    L.plaac_node_text_table.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    L.plaac_node_text_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.plaac_node_text_discard.argtypes = [C.c_void_p]
    L.plaac_node_text_pending.argtypes = [C.c_void_p]
    L.plaac_node_text_oldest_records.argtypes = [C.c_void_p]
    L.plaac_node_text_oldest_records.restype = C.c_uint32
    L.plaac_node_text_reset.argtypes = [C.c_void_p]
    L.plaac_node_text_reset.restype = None
    L.plaac_node_histogram_text_begin.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.c_void_p, C.c_uint32]
    L.plaac_node_histogram_text_end.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]
    L.plaac_node_score_tracks_table.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_char_p, C.c_void_p, C.c_void_p,
                                                C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
    L.plaac_text_upload_error.argtypes = [C.c_void_p]
    L.plaac_text_upload_error.restype = C.c_char_p
    if hasattr(L, "plaac_ctx_set_value_tolerance"):  # (absent from builds of the first half of round 6, loaded for an A/B)
        L.plaac_ctx_set_value_tolerance.argtypes = [C.c_void_p, C.c_int]
    L.plaac_debug_set_knob.argtypes = [C.c_char_p, C.c_char_p]
    L.plaac_diag_build.restype = C.c_int
    _lib = L
    return L


def builtin_tables():
    bg, f28, f04 = (np.zeros(NAA) for _ in range(3))
    load().plaac_builtin_tables(bg.ctypes.data, f28.ctypes.data, f04.ctypes.data)
    return {"bg_scer": bg, "fg28": f28, "fg04": f04}


def make_params(fg=None, bgcounts=None, alpha=1.0, corelength=60, ww1=41, ww2=41, ww3=None, adjustprolines=True):
    """plaac_params_init: the table setup of main (plaac.java:444-500). ww3 defaults to ww2 (:355)."""
    P = Params()
    fgp = None if fg is None else np.ascontiguousarray(fg, dtype=np.float64)
    bgp = None if bgcounts is None else np.ascontiguousarray(bgcounts, dtype=np.float64)
    st = load().plaac_params_init(C.addressof(P), None if fgp is None else fgp.ctypes.data,
                                  None if bgp is None else bgp.ctypes.data, float(alpha), int(corelength), int(ww1),
                                  int(ww2), int(ww2 if ww3 is None else ww3), int(bool(adjustprolines)))
    if st != PLAAC_OK:
        raise PlaacError(st, "plaac_params_init rejected its arguments")
    return P


def fi_integer_form(P):
    """plaac_fi_integer_form: (qualifies, {A2, B2, C2, SH, SC, Hmin}) - does the filter tier sign FoldIndex in integers?"""
    info = (C.c_int32 * 6)()
    ok = load().plaac_fi_integer_form(C.addressof(P), C.addressof(info))
    return bool(ok), dict(zip(("A2", "B2", "C2", "SH", "SC", "Hmin"), list(info)))


def encode(seq):
    b = seq.encode("latin-1") if isinstance(seq, str) else bytes(seq)
    out = np.zeros(len(b), dtype=np.uint8)
    load().plaac_encode(b, len(b), out.ctypes.data)
    return out


def decode(codes):
    """codes 0..21 -> residue letters (plaac.java:26 alphabet order)"""
    return np.frombuffer(b"XACDEFGHIKLMNPQRSTVWY*", dtype=np.uint8)[np.asarray(codes, dtype=np.uint8)].tobytes().decode("ascii")


def pack(seqs):
    """records (str/bytes, untrimmed) -> (codes u8[total], offsets u64[n+1])"""
    enc = [encode(s) for s in seqs]
    offsets = np.zeros(len(enc) + 1, dtype=np.uint64)
    if enc:
        offsets[1:] = np.cumsum([len(e) for e in enc], dtype=np.uint64)
    codes = np.concatenate(enc) if enc else np.zeros(0, dtype=np.uint8)
    return np.ascontiguousarray(codes, dtype=np.uint8), offsets


def alloc_tracks(total):
    tr = {k: np.zeros(total, dtype=np.uint8) for k in TRACK_U8}
    for k in TRACK_F64:
        tr[k] = np.full(total, np.nan, dtype=np.float64)
    return tr


class Context:
    """plaac_ctx: one device, one caller."""

    def __init__(self, params=None, device=0):
        self._L = load()
        self._h = C.c_void_p()
        apply_env_knobs(self._L)
        self._h = C.c_void_p()
        self.params = params if params is not None else make_params()
        st = self._L.plaac_ctx_create(C.addressof(self.params), int(device), C.byref(self._h))
        if st != PLAAC_OK:
            msg = self._L.plaac_last_error(None)
            self._h = C.c_void_p()
            raise PlaacError(st, msg.decode() if msg else "plaac_ctx_create failed")

    def _check(self, st):
        if st != PLAAC_OK:
            msg = self._L.plaac_last_error(self._h)
            raise PlaacError(st, msg.decode() if msg else "?")

    def close(self):
        if self._h:
            self._L.plaac_ctx_destroy(self._h)
            self._h = None  # (not C.c_void_p(): at interpreter shutdown the module's globals may be gone already)

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_params(self, params):
        apply_env_knobs(self._L)  # (LSE_CLAMP is read whenever tables are built)
        self._check(self._L.plaac_ctx_set_params(self._h, C.addressof(params)))
        self.params = params

    def histogram(self, codes, offsets):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        counts = np.zeros(NAA, dtype=np.int64)
        self._check(self._L.plaac_histogram(self._h, codes.ctypes.data, offsets.ctypes.data, len(offsets) - 1,
                                            counts.ctypes.data))
        return counts

    def score(self, codes, offsets, tracks=False):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        nprot = len(offsets) - 1
        rows = np.zeros(nprot, dtype=ROW_DTYPE)
        tr, tptr, T = None, None, None
        if tracks:
            tr = alloc_tracks(int(offsets[-1]) if nprot >= 0 else 0)
            T = Tracks(**{k: tr[k].ctypes.data for k in TRACK_U8 + TRACK_F64})
            tptr = C.addressof(T)
        self._check(self._L.plaac_score(self._h, codes.ctypes.data, offsets.ctypes.data, nprot, rows.ctypes.data,
                                        tptr))
        return (rows, tr) if tracks else rows

    # ---- pipelined host-buffer scoring: two batches in flight (plaac_score_begin / plaac_score_end) ----
    def score_begin(self, codes, offsets):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._check(self._L.plaac_score_begin(self._h, codes.ctypes.data, offsets.ctypes.data, len(offsets) - 1))
        return len(offsets) - 1

    def score_end(self, nprot):
        """rows of the OLDEST pending batch (`nprot` = what score_begin returned for it)"""
        rows = np.zeros(nprot, dtype=ROW_DTYPE)
        self._check(self._L.plaac_score_end(self._h, rows.ctypes.data))
        return rows

    def score_begin_counting(self, codes, offsets):
        """plaac_score_begin_counting: as score_begin, and the batch's background counts come back with the rows"""
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self._check(self._L.plaac_score_begin_counting(self._h, codes.ctypes.data, offsets.ctypes.data, len(offsets) - 1))
        return len(offsets) - 1

    def score_end_counts(self, nprot):
        """(rows, 22 int64 counts) of the OLDEST pending batch, which was begun with score_begin_counting"""
        rows = np.zeros(nprot, dtype=ROW_DTYPE)
        counts = np.zeros(NAA, dtype=np.int64)
        self._check(self._L.plaac_score_end_counts(self._h, rows.ctypes.data, counts.ctypes.data))
        return rows, counts

    def score_text(self, text, starts, counting=False, want_codes=True):
        """plaac_score_begin_text + plaac_score_end_text (K1: FASTA text parsed and encoded on the device): `text` bytes of whole
        records, starts[i] = where record i begins (starts[-1] = len(text)). Returns (rows, codes, offsets, blank_end[, counts]);
        want_codes=False: `codes` is the records' extents instead (uint32 [nrec, 2], for hostio.text_codes)."""
        starts = np.ascontiguousarray(starts, dtype=np.uint64)
        nrec = len(starts) - 1
        self._check(self._L.plaac_score_begin_text(self._h, text, len(text), starts.ctypes.data, nrec, 1 if counting else 0))
        rows = np.zeros(nrec, dtype=ROW_DTYPE)
        codes = np.zeros(max(len(text), 1), dtype=np.uint8)
        offsets = np.zeros(nrec + 1, dtype=np.uint64)
        blank = np.zeros(max(nrec, 1), dtype=np.uint8)
        counts = np.zeros(NAA, dtype=np.int64)
        ext = np.zeros((max(nrec, 1), 2), dtype=np.uint32)
        self._check(self._L.plaac_score_end_text(self._h, rows.ctypes.data, codes.ctypes.data if want_codes else None, len(codes),
                                                 offsets.ctypes.data, blank.ctypes.data, ext.ctypes.data,
                                                 counts.ctypes.data if counting else None))
        out = (rows, codes[:int(offsets[-1])] if want_codes else ext[:nrec], offsets, blank[:nrec])
        return out + (counts,) if counting else out

    def score_tracks_table(self, codes, offsets, labels):
        """plaac_score_tracks_table: the per-residue table of a batch as text from the device; labels = list of b"ORDER\tSEQid" per
        record. Returns (table bytes or None when the host's formatter is needed, rows)."""
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        blob = b"".join(labels)
        loff = np.zeros(n + 1, dtype=np.uint64)
        loff[1:] = np.cumsum([len(x) for x in labels])
        rows = np.zeros(n, dtype=ROW_DTYPE)
        tab, tlen, needs = C.c_void_p(), C.c_uint64(), C.c_int()
        self._check(self._L.plaac_score_tracks_table(self._h, codes.ctypes.data, offsets.ctypes.data, n, blob, loff.ctypes.data, rows.ctypes.data,
                                                     C.byref(tab), C.byref(tlen), C.byref(needs)))
        if needs.value or not tab.value:
            return (None if needs.value else b""), rows
        out = C.string_at(tab.value, tlen.value)
        self._L.plaac_table_free(tab)
        return out, rows

    def text_upload(self, text, starts):
        """plaac_text_upload: the batch uploaded and parsed ahead of its scoring call (may run on another thread than the scoring calls)"""
        starts = np.ascontiguousarray(starts, dtype=np.uint64)
        tb = C.c_void_p()
        self._check(self._L.plaac_text_upload(self._h, text, len(text), starts.ctypes.data, len(starts) - 1, C.byref(tb)))
        return tb

    def score_begin_uploaded(self, tb, counting=True):
        self._check(self._L.plaac_score_begin_uploaded(self._h, tb, 1 if counting else 0))

    def histogram_text(self, text, starts):
        """plaac_histogram_begin_text + _end_text: the residue counts of a text batch (parsed on the device) and its residue count"""
        starts = np.ascontiguousarray(starts, dtype=np.uint64)
        self._check(self._L.plaac_histogram_begin_text(self._h, text, len(text), starts.ctypes.data, len(starts) - 1))
        counts = np.zeros(NAA, dtype=np.int64)
        nres = C.c_uint64()
        self._check(self._L.plaac_histogram_end_text(self._h, counts.ctypes.data, C.byref(nres)))
        return counts, int(nres.value)

    def score_text_table(self, text, starts, corelength=60, ww2=41, prev_blank=1):
        """plaac_score_begin_text + plaac_score_end_text_table_size / _table: the batch's summary rows as the table's text, made on
        the device. Returns (table bytes, last_blank, counts) - or (None, last_blank, None) with the batch still pending when the
        device asks for the host's formatter (collect it with score_text_end)."""
        starts = np.ascontiguousarray(starts, dtype=np.uint64)
        nrec = len(starts) - 1
        self._check(self._L.plaac_score_begin_text(self._h, text, len(text), starts.ctypes.data, nrec, 1))
        size, needs, lastb = C.c_uint64(), C.c_int(), C.c_int()
        self._check(self._L.plaac_score_end_text_table_size(self._h, int(corelength), int(ww2), int(prev_blank), C.byref(size),
                                                            C.byref(needs), C.byref(lastb), None))
        if needs.value:
            return None, lastb.value, None
        buf = C.create_string_buffer(max(int(size.value), 1))
        counts = np.zeros(NAA, dtype=np.int64)
        self._check(self._L.plaac_score_end_text_table(self._h, buf, size.value, counts.ctypes.data))
        return buf.raw[:size.value], lastb.value, counts

    def score_text_end(self, nrec, text_len, counting=True):
        """plaac_score_end_text for a batch left pending by score_text_table"""
        rows = np.zeros(nrec, dtype=ROW_DTYPE)
        codes = np.zeros(max(text_len, 1), dtype=np.uint8)
        offsets = np.zeros(nrec + 1, dtype=np.uint64)
        blank = np.zeros(max(nrec, 1), dtype=np.uint8)
        ext = np.zeros((max(nrec, 1), 2), dtype=np.uint32)
        counts = np.zeros(NAA, dtype=np.int64)
        self._check(self._L.plaac_score_end_text(self._h, rows.ctypes.data, codes.ctypes.data, len(codes), offsets.ctypes.data,
                                                 blank.ctypes.data, ext.ctypes.data, counts.ctypes.data if counting else None))
        return rows, codes[:int(offsets[-1])], offsets, blank[:nrec], counts

    def score_stream(self, batches, counting=False):
        """scores an iterable of (codes, offsets) batches with two in flight; yields the row arrays in order
        (counting: (rows, counts) pairs - the reference's background pass folded into the scoring pass)"""
        pending = []
        begin, end = (self.score_begin_counting, self.score_end_counts) if counting else (self.score_begin, self.score_end)
        for codes, offsets in batches:
            pending.append(begin(codes, offsets))
            if len(pending) == 2:
                yield end(pending.pop(0))
        while pending:
            yield end(pending.pop(0))

    # ---- resident batch: upload once, histogram / score (under several parameter sets) many times ----
    def upload(self, codes, offsets):
        return Batch(self, codes, offsets)

    # ---- device-resident entry points: raw device pointers (e.g. torch tensors' data_ptr()) ----
    def score_device(self, d_codes, d_offsets, nprot, total, d_rows, d_tracks=None, stream=None):
        T = None
        if d_tracks is not None:
            T = Tracks(**{k: int(d_tracks[k]) for k in TRACK_U8 + TRACK_F64})
        self._check(self._L.plaac_score_device(self._h, int(d_codes), int(d_offsets), int(nprot), int(total),
                                               int(d_rows), None if T is None else C.addressof(T),
                                               None if stream is None else int(stream)))

    def score_sweep_device(self, d_codes, d_offsets, nprot, total, param_sets, d_rows_list, stream=None):
        """plaac_score_sweep_device: one planned pass for several parameter sets, one device row array each"""
        param_sets = list(param_sets)
        pts = (Params * len(param_sets))(*param_sets)
        ptrs = (C.c_void_p * len(d_rows_list))(*[int(x) for x in d_rows_list])
        self._check(self._L.plaac_score_sweep_device(self._h, int(d_codes), int(d_offsets), int(nprot), int(total),
                                                     C.addressof(pts), len(param_sets), C.addressof(ptrs),
                                                     None if stream is None else int(stream)))

    def histogram_device(self, d_codes, d_offsets, nprot, d_counts, stream=None):
        self._check(self._L.plaac_histogram_device(self._h, int(d_codes), int(d_offsets), int(nprot), int(d_counts),
                                                   None if stream is None else int(stream)))

    def calibration_reads(self, d_codes, total_residues, stream=None):
        """diagnostic: three known-size streaming reads of the residue buffer (FETCH_SIZE calibration, tools/pmc.sh)"""
        self._check(self._L.plaac_calibration_reads(self._h, int(d_codes), int(total_residues),
                                                    None if stream is None else int(stream)))

    def set_value_tolerance(self, on=True):
        """plaac_ctx_set_value_tolerance: the five floats at the PAPA centre from sliding first-level sums (< 1e-12 from the
        reference-order values; indices and every other field unchanged); off by default"""
        self._check(self._L.plaac_ctx_set_value_tolerance(self._h, 1 if on else 0))

    def set_overlap(self, on=True):
        """consecutive calls may overlap (the next call plans beside the last window kernels of this one); see the header
        for what the caller then guarantees about its buffers"""
        self._check(self._L.plaac_ctx_set_overlap(self._h, 1 if on else 0))

    def clock_probe(self, micros=20000):
        """diagnostic: shader clock (MHz) held over the next `micros` microseconds (a sleeping wave beside whatever runs)"""
        mhz = C.c_double(0.0)
        self._check(self._L.plaac_clock_probe(self._h, int(micros), C.addressof(mhz)))
        return float(mhz.value)

    def sync(self):
        self._check(self._L.plaac_ctx_sync(self._h))

    def last_exact_fallbacks(self):
        """proteins of the last scored batch that the window-track filter handed to the exact kernel"""
        n = C.c_uint32(0)
        self._check(self._L.plaac_last_exact_fallbacks(self._h, C.addressof(n)))
        return int(n.value)

    def last_timings(self, ncalls=1):
        """device ms (HIP events on each kernel's launch stream), mean over the last `ncalls` scored batches.
        The four scoring kernels overlap on separate streams, so they do not add up to `total`."""
        ms = (C.c_float * 8)()
        self._check(self._L.plaac_timings_mean(self._h, int(ncalls), C.addressof(ms)))
        return {"total": ms[0], "plan": ms[1], "vit": ms[2], "fwd": ms[3], "win": ms[4], "tracks": ms[5],
                "pack": ms[6], "bwd": ms[7]}


class Node:
    """plaac_node: one scoring context per listed device (None = every visible device), batches sharded by sequence."""

    def __init__(self, params=None, devices=None):
        self._L = load()
        self._h = C.c_void_p()
        apply_env_knobs(self._L)
        self._h = C.c_void_p()
        self.params = params if params is not None else make_params()
        ids = None if devices is None else (C.c_int * len(devices))(*[int(d) for d in devices])
        st = self._L.plaac_node_create(C.addressof(self.params), ids, 0 if devices is None else len(devices),
                                       C.byref(self._h))
        if st != PLAAC_OK:
            msg = self._L.plaac_node_last_error(None)
            self._h = C.c_void_p()
            raise PlaacError(st, msg.decode() if msg else "plaac_node_create failed")

    def _check(self, st):
        if st != PLAAC_OK:
            msg = self._L.plaac_node_last_error(self._h)
            raise PlaacError(st, msg.decode() if msg else "?")

    def __len__(self):
        return int(self._L.plaac_node_size(self._h))

    def close(self):
        # a batch belongs to its node: close the ones still alive first (the library would only detach them)
        for ref in getattr(self, "_batches", []):
            b = ref()
            if b is not None:
                b.close()
        self._batches = []
        if self._h:
            self._L.plaac_node_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_params(self, params):
        apply_env_knobs(self._L)
        self._check(self._L.plaac_node_set_params(self._h, C.addressof(params)))
        self.params = params

    def histogram(self, codes, offsets):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        counts = np.zeros(NAA, dtype=np.int64)
        self._check(self._L.plaac_node_histogram(self._h, codes.ctypes.data, offsets.ctypes.data, len(offsets) - 1,
                                                 counts.ctypes.data))
        return counts

    def score(self, codes, offsets, tracks=False):
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        nprot = len(offsets) - 1
        rows = np.zeros(nprot, dtype=ROW_DTYPE)
        tr, tptr, T = None, None, None
        if tracks:
            tr = alloc_tracks(int(offsets[-1]) if nprot >= 0 else 0)
            T = Tracks(**{k: tr[k].ctypes.data for k in TRACK_U8 + TRACK_F64})
            tptr = C.addressof(T)
        self._check(self._L.plaac_node_score(self._h, codes.ctypes.data, offsets.ctypes.data, nprot,
                                             rows.ctypes.data, tptr))
        return (rows, tr) if tracks else rows


    def set_overlap(self, on=True):
        self._check(self._L.plaac_node_set_overlap(self._h, 1 if on else 0))

    # ---- FASTA text through the node (plaac_node_text_*): batches in file order, collected oldest first ----
    def text_begin(self, text, starts, counting=False):
        starts = np.ascontiguousarray(starts, dtype=np.uint64)
        self._check(self._L.plaac_node_text_begin(self._h, text, len(text), starts.ctypes.data, len(starts) - 1, 1 if counting else 0))

    def text_upload(self, text, starts):
        """plaac_node_text_upload (may run on another thread than the scoring calls); returns the uploaded batch's handle"""
        starts = np.ascontiguousarray(starts, dtype=np.uint64)
        tb = C.c_void_p()
        st = self._L.plaac_node_text_upload(self._h, text, len(text), starts.ctypes.data, len(starts) - 1, C.byref(tb))
        if st != PLAAC_OK:
            raise PlaacError(st, "plaac_node_text_upload failed")
        return tb

    def text_begin_uploaded(self, tb, counting=False):
        self._check(self._L.plaac_node_text_begin_uploaded(self._h, tb, 1 if counting else 0))

    def text_pending(self):
        return int(self._L.plaac_node_text_pending(self._h))

    def text_reset(self):
        self._L.plaac_node_text_reset(self._h)

    def text_discard(self):
        self._check(self._L.plaac_node_text_discard(self._h))

    def text_table(self, corelength=60, ww2=41, counting=False):
        """the oldest pending batch's table text: (bytes, residues[, counts]) - or (None, residues) with the batch still pending when
        the device asks for the host's formatter (collect it with text_rows, or text_discard)"""
        size, needs, nres = C.c_uint64(), C.c_int(), C.c_uint64()
        self._check(self._L.plaac_node_text_table_size(self._h, int(corelength), int(ww2), C.byref(size), C.byref(needs), C.byref(nres)))
        if needs.value:
            return None, int(nres.value)
        buf = C.create_string_buffer(max(int(size.value), 1))
        counts = np.zeros(NAA, dtype=np.int64)
        self._check(self._L.plaac_node_text_table(self._h, buf, size.value, counts.ctypes.data if counting else None))
        out = (buf.raw[:size.value], int(nres.value))
        return out + (counts,) if counting else out

    def text_rows(self, text_len, counting=False):
        """plaac_node_text_rows: the oldest pending batch as (rows, codes, offsets, blank_end, extents[, counts])"""
        nrec = int(self._L.plaac_node_text_oldest_records(self._h))
        rows = np.zeros(nrec, dtype=ROW_DTYPE)
        codes = np.zeros(max(text_len, 1), dtype=np.uint8)
        offsets = np.zeros(nrec + 1, dtype=np.uint64)
        blank = np.zeros(max(nrec, 1), dtype=np.uint8)
        ext = np.zeros((max(nrec, 1), 2), dtype=np.uint32)
        counts = np.zeros(NAA, dtype=np.int64)
        self._check(self._L.plaac_node_text_rows(self._h, rows.ctypes.data, codes.ctypes.data, len(codes), offsets.ctypes.data,
                                                 blank.ctypes.data, ext.ctypes.data, counts.ctypes.data if counting else None))
        out = (rows, codes[:int(offsets[-1])], offsets, blank[:nrec], ext[:nrec])
        return out + (counts,) if counting else out

    def histogram_text_begin(self, text, starts):
        starts = np.ascontiguousarray(starts, dtype=np.uint64)
        self._check(self._L.plaac_node_histogram_text_begin(self._h, text, len(text), starts.ctypes.data, len(starts) - 1))

    def histogram_text_end(self, counts, residues=0):
        """adds the oldest counting batch's counts to `counts` (int64[22]); returns the residue total so far"""
        r = C.c_uint64(int(residues))
        self._check(self._L.plaac_node_histogram_text_end(self._h, counts.ctypes.data, C.byref(r)))
        return int(r.value)

    def score_tracks_table(self, codes, offsets, labels):
        """plaac_node_score_tracks_table: (table bytes or None when the host's formatter is needed, rows)"""
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        n = len(offsets) - 1
        blob = b"".join(labels)
        loff = np.zeros(n + 1, dtype=np.uint64)
        loff[1:] = np.cumsum([len(x) for x in labels])
        rows = np.zeros(n, dtype=ROW_DTYPE)
        tab, tlen, needs = C.c_void_p(), C.c_uint64(), C.c_int()
        self._check(self._L.plaac_node_score_tracks_table(self._h, codes.ctypes.data, offsets.ctypes.data, n, blob, loff.ctypes.data,
                                                          rows.ctypes.data, C.byref(tab), C.byref(tlen), C.byref(needs)))
        if needs.value or not tab.value:
            return (None if needs.value else b""), rows
        out = C.string_at(tab.value, tlen.value)
        self._L.plaac_table_free(tab)
        return out, rows

    def upload(self, codes, offsets):
        """plaac_node_batch_upload: the batch cut with plaac_shard_plan, every shard resident on its device"""
        b = NodeBatch(self, codes, offsets)
        if not hasattr(self, "_batches"):
            self._batches = []
        self._batches = [r for r in self._batches if r() is not None] + [weakref.ref(b)]
        return b


class NodeBatch:
    """plaac_node_batch: one upload for the background pass, the scoring pass(es) and parameter sweeps on all devices of a
    node; rows / tracks come back in input order."""

    def __init__(self, node, codes, offsets):
        self.node = node
        self._L = node._L
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.nprot = len(offsets) - 1
        self.total = int(offsets[-1]) if self.nprot > 0 else 0
        self._h = C.c_void_p()
        node._check(node._L.plaac_node_batch_upload(node._h, codes.ctypes.data, offsets.ctypes.data, self.nprot,
                                                    C.byref(self._h)))

    def _check(self, st):
        if st != PLAAC_OK:
            msg = self._L.plaac_node_batch_last_error(self._h)
            raise PlaacError(st, msg.decode() if msg else "?")

    def histogram(self):
        counts = np.zeros(NAA, dtype=np.int64)
        self._check(self._L.plaac_node_batch_histogram(self._h, counts.ctypes.data))
        return counts

    def score(self, tracks=False):
        rows = np.zeros(self.nprot, dtype=ROW_DTYPE)
        tr, tptr, T = None, None, None
        if tracks:
            tr = alloc_tracks(self.total)
            T = Tracks(**{k: tr[k].ctypes.data for k in TRACK_U8 + TRACK_F64})
            tptr = C.addressof(T)
        self._check(self._L.plaac_node_batch_score(self._h, rows.ctypes.data, tptr))
        return (rows, tr) if tracks else rows

    def sweep(self, param_sets):
        param_sets = list(param_sets)
        pts = (Params * len(param_sets))(*param_sets)
        rows = [np.zeros(self.nprot, dtype=ROW_DTYPE) for _ in param_sets]
        ptrs = (C.c_void_p * len(rows))(*[r.ctypes.data for r in rows])
        self._check(self._L.plaac_node_batch_sweep(self._h, C.addressof(pts), len(param_sets), C.addressof(ptrs)))
        return rows

    def close(self):
        if getattr(self, "_h", None):
            self._L.plaac_node_batch_free(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def device_count():
    return int(load().plaac_device_count())


class SchedQuery(C.Structure):
    """plaac_sched_query"""
    _fields_ = [("nprot", C.c_uint32), ("npoints", C.c_uint32), ("residues", C.c_uint64), ("total_rows", C.c_uint32),
                ("rows_first", C.c_uint32), ("long_groups", C.c_uint32), ("long_rows", C.c_uint32),
                ("run_mark", C.c_uint32 * 7), ("ngroups_sweep", C.c_uint32), ("group_members", C.c_uint32 * 11),
                ("kb_base", C.c_int32 * 11), ("lane_possible", C.c_int32), ("fast20", C.c_int32), ("wmax", C.c_int32),
                ("core_par_tables", C.c_int32), ("tracks", C.c_int32), ("overlap", C.c_int32), ("ncalls", C.c_uint64),
                ("last_chain_bound", C.c_int32), ("last_mixed", C.c_int32), ("last_single_summary", C.c_int32),
                ("old_tail", C.c_int32)]


def debug_schedule(q):
    """plaac_debug_schedule: the schedule of a described scoring call as text (host only; no device needed)"""
    apply_env_knobs()
    for cap in (1 << 18, 1 << 25):  # (a sweep of thousands of points prints thousands of launches)
        buf = C.create_string_buffer(cap)
        n = load().plaac_debug_schedule(C.addressof(q), buf, len(buf))
        if n >= 0:
            break
    if n < 0:
        raise PlaacError(PLAAC_ERR_ARG, "plaac_debug_schedule rejected the query")
    return buf.value.decode()


WIRE_ROW_BYTES = 136


def rows_to_wire(rows, offsets):
    """plaac_rows_to_wire (host): 160-byte rows -> 136-byte wire rows (uint8 array [n * 136])"""
    rows = np.ascontiguousarray(rows)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    wire = np.empty(n * WIRE_ROW_BYTES, dtype=np.uint8)
    st = load().plaac_rows_to_wire(rows.ctypes.data, offsets.ctypes.data, n, wire.ctypes.data)
    if st != PLAAC_OK:
        raise PlaacError(st, "plaac_rows_to_wire rejected its arguments")
    return wire


def rows_from_wire(wire, offsets, corelength):
    """plaac_rows_from_wire (host): the receiver's side - wire rows + the batch's offsets + the core length -> rows"""
    wire = np.ascontiguousarray(wire, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    rows = np.zeros(n, dtype=ROW_DTYPE)
    st = load().plaac_rows_from_wire(wire.ctypes.data, offsets.ctypes.data, n, int(corelength), rows.ctypes.data)
    if st != PLAAC_OK:
        raise PlaacError(st, "plaac_rows_from_wire rejected its arguments")
    return rows


def shard_plan(offsets, parts):
    """plaac_shard_plan (the one partitioner of every multi-GPU layer: sort by length, deal): a list of `parts` uint32
    index arrays, the records of each shard in ascending input order. Host-only."""
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    nprot = len(offsets) - 1
    index = np.empty(max(nprot, 0), dtype=np.uint32)
    start = np.zeros(parts + 1, dtype=np.uint32)
    st = load().plaac_shard_plan(offsets.ctypes.data, max(nprot, 0), int(parts), index.ctypes.data, start.ctypes.data)
    if st != PLAAC_OK:
        raise PlaacError(st, "plaac_shard_plan rejected its arguments")
    return [index[int(start[k]):int(start[k + 1])] for k in range(parts)]


class Batch:
    """plaac_batch: encoded residues resident in HBM (one upload for the background pass, the scoring pass
    and every point of a parameter sweep)."""

    def __init__(self, ctx, codes, offsets):
        self.ctx = ctx
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.nprot = len(offsets) - 1
        self.total = int(offsets[-1]) if self.nprot > 0 else 0
        self._h = C.c_void_p()
        ctx._check(ctx._L.plaac_batch_upload(ctx._h, codes.ctypes.data, offsets.ctypes.data, self.nprot,
                                             C.byref(self._h)))

    def histogram(self):
        counts = np.zeros(NAA, dtype=np.int64)
        self.ctx._check(self.ctx._L.plaac_batch_histogram(self._h, counts.ctypes.data))
        return counts

    def score(self, tracks=False):
        rows = np.zeros(self.nprot, dtype=ROW_DTYPE)
        tr, tptr, T = None, None, None
        if tracks:
            tr = alloc_tracks(self.total)
            T = Tracks(**{k: tr[k].ctypes.data for k in TRACK_U8 + TRACK_F64})
            tptr = C.addressof(T)
        self.ctx._check(self.ctx._L.plaac_batch_score(self._h, rows.ctypes.data, tptr))
        return (rows, tr) if tracks else rows

    def sweep(self, param_sets, naive=False):
        """score the resident batch under each plaac_params of `param_sets`; returns a list of row arrays.
        Default: plaac_batch_sweep (points differing only in the core length share the passes that do not
        depend on it). naive=True: one full pass per point (ctx.set_params + score)."""
        param_sets = list(param_sets)
        if naive:
            out = []
            for P in param_sets:
                self.ctx.set_params(P)
                out.append(self.score())
            return out
        pts = (Params * len(param_sets))(*param_sets)
        rows = [np.zeros(self.nprot, dtype=ROW_DTYPE) for _ in param_sets]
        ptrs = (C.c_void_p * len(rows))(*[r.ctypes.data for r in rows])
        self.ctx._check(self.ctx._L.plaac_batch_sweep(self._h, C.addressof(pts), len(param_sets), C.addressof(ptrs)))
        return rows

    def close(self):
        if self._h:
            self.ctx._L.plaac_batch_free(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
