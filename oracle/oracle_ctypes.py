"""ctypes binding of oracle/libplaac_oracle.so — TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (plaac_amd) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libplaac_oracle.so")

NAA = 22
LUTLEN = 4001


class OracleHmm(C.Structure):
    _fields_ = [("lt", (C.c_double * 2) * 2), ("li", C.c_double * 2), ("le", (C.c_double * NAA) * 2),
                ("lf", C.c_double * 2)]


class OracleParams(C.Structure):
    _fields_ = [("corelength", C.c_int32), ("ww1", C.c_int32), ("ww2", C.c_int32), ("ww3", C.c_int32),
                ("adjustprolines", C.c_int32), ("pad_", C.c_int32), ("alpha", C.c_double), ("cc", C.c_double * 3),
                ("fg", C.c_double * NAA), ("bgscer", C.c_double * NAA), ("bgthis", C.c_double * NAA),
                ("bg", C.c_double * NAA), ("llr", C.c_double * NAA), ("lodpapa", C.c_double * NAA),
                ("hydro2", C.c_double * NAA), ("charge", C.c_double * NAA), ("hmm1", OracleHmm), ("hmm0", OracleHmm),
                ("loglut", C.c_double * LUTLEN)]


ROW_DTYPE = np.dtype([
    ("llr_score", "<f8"), ("core_score", "<f8"), ("prd_score", "<f8"), ("hmm_all", "<f8"), ("hmm_vit", "<f8"),
    ("fi_meanhydro", "<f8"), ("fi_meancharge", "<f8"), ("fi_meancombo", "<f8"),
    ("papa_combo", "<f8"), ("papa_prop", "<f8"), ("papa_fi", "<f8"), ("papa_llr", "<f8"), ("papa_llr2", "<f8"),
    ("mw_score", "<i4"), ("mw_start", "<i4"), ("mw_end", "<i4"), ("llr_start", "<i4"), ("llr_end", "<i4"),
    ("vit_maxrun", "<i4"), ("core_start", "<i4"), ("core_end", "<i4"), ("prd_start", "<i4"), ("prd_end", "<i4"),
    ("prot_len", "<i4"), ("fi_numaa", "<i4"), ("fi_maxrun", "<i4"), ("papa_cen", "<i4"),
])
assert ROW_DTYPE.itemsize == 160

TRACK_F64 = ("charge", "hydro", "fi", "plaacllr", "papa", "fix2", "plaacllrx2", "papax2", "post0", "post1")


class OracleTracks(C.Structure):
    _fields_ = [("vit", C.c_void_p), ("map", C.c_void_p)] + [(k, C.c_void_p) for k in TRACK_F64]


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(
            os.path.join(_HERE, "plaac_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libplaac_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.oracle_sizeof_params.restype = C.c_size_t
        L.oracle_sizeof_row.restype = C.c_size_t
        assert L.oracle_sizeof_params() == C.sizeof(OracleParams), "oracle_params layout mismatch"
        assert L.oracle_sizeof_row() == ROW_DTYPE.itemsize, "oracle_row layout mismatch"
        L.oracle_logeapeb.restype = C.c_double
        L.oracle_logeapeb.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.oracle_aatoint.restype = C.c_uint8
        L.oracle_aatoint.argtypes = [C.c_char]
        L.oracle_build_params.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.c_void_p]
        L.oracle_encode.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p]
        L.oracle_histogram.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.oracle_hss2.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.oracle_hss_brute.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.oracle_score_protein.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_score_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p,
                                         C.c_int]
        L.oracle_const_tables.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def const_tables():
    bg, f28, f04 = (np.zeros(NAA) for _ in range(3))
    lib().oracle_const_tables(bg.ctypes.data, f28.ctypes.data, f04.ctypes.data)
    return {"bg_scer": bg, "fg28": f28, "fg04": f04}


def build_params(fg=None, bgcounts=None, alpha=1.0, corelength=60, ww1=41, ww2=41, ww3=None, adjustprolines=True):
    t = const_tables()
    fg = np.ascontiguousarray(t["fg28"] if fg is None else fg, dtype=np.float64)
    bgc = np.ascontiguousarray(np.zeros(NAA) if bgcounts is None else bgcounts, dtype=np.float64)
    ww3 = ww2 if ww3 is None else ww3  # plaac.java:355
    P = OracleParams()
    lib().oracle_build_params(fg.ctypes.data, bgc.ctypes.data, float(alpha), int(corelength), int(ww1), int(ww2),
                              int(ww3), int(bool(adjustprolines)), C.addressof(P))
    return P


def encode(seq):
    b = seq.encode("latin-1") if isinstance(seq, str) else bytes(seq)
    out = np.zeros(len(b), dtype=np.uint8)
    lib().oracle_encode(b, len(b), out.ctypes.data)
    return out


def pack(seqs):
    """list of str/bytes records (untrimmed) -> (codes u8, offsets u64[n+1])"""
    enc = [encode(s) for s in seqs]
    offsets = np.zeros(len(enc) + 1, dtype=np.uint64)
    if enc:
        offsets[1:] = np.cumsum([len(e) for e in enc], dtype=np.uint64)
    codes = np.concatenate(enc) if enc else np.zeros(0, dtype=np.uint8)
    return np.ascontiguousarray(codes, dtype=np.uint8), offsets


def histogram(codes, offsets):
    counts = np.zeros(NAA, dtype=np.int64)
    lib().oracle_histogram(codes.ctypes.data, offsets.ctypes.data, len(offsets) - 1, counts.ctypes.data)
    return counts


def hss2(seq, minlength, maxlength):
    seq = np.ascontiguousarray(seq, dtype=np.float64)
    out = np.zeros(3)
    lib().oracle_hss2(seq.ctypes.data, len(seq), minlength, maxlength, out.ctypes.data)
    return out


def hss_brute(seq, length):
    seq = np.ascontiguousarray(seq, dtype=np.float64)
    out = np.zeros(3)
    lib().oracle_hss_brute(seq.ctypes.data, len(seq), length, out.ctypes.data)
    return out


def logeapeb(P, a, b):
    return lib().oracle_logeapeb(C.addressof(P.loglut), a, b)


def alloc_tracks(total):
    tr = {"vit": np.zeros(total, dtype=np.uint8), "map": np.zeros(total, dtype=np.uint8)}
    for k in TRACK_F64:
        tr[k] = np.full(total, np.nan, dtype=np.float64)
    return tr


def _tracks_struct(tr):
    T = OracleTracks()
    T.vit = tr["vit"].ctypes.data
    T.map = tr["map"].ctypes.data
    for k in TRACK_F64:
        setattr(T, k, tr[k].ctypes.data)
    return T


def score_batch(P, codes, offsets, tracks=False, nthreads=1):
    nprot = len(offsets) - 1
    rows = np.zeros(nprot, dtype=ROW_DTYPE)
    tr = None
    tptr = None
    if tracks:
        tr = alloc_tracks(int(offsets[-1]))
        T = _tracks_struct(tr)
        tptr = C.addressof(T)
    lib().oracle_score_batch(C.addressof(P), codes.ctypes.data, offsets.ctypes.data, nprot, rows.ctypes.data, tptr,
                             int(nthreads))
    return (rows, tr) if tracks else rows
