"""CPU tests of the C-ABI library: it loads, exports every symbol the header declares, its host-side
helpers agree with the oracle, and it FAILS LOUDLY (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def header_functions():
    txt = open(os.path.join(ROOT, "include", "plaac_native.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(plaac_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(native):
    L = native.load()
    declared = header_functions()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(L, name), name
    assert sorted(native.EXPORTS) == declared


def test_struct_layouts(native):
    L = native.load()
    assert L.plaac_abi_version() == 2
    assert L.plaac_sizeof_row() == 160 == native.ROW_DTYPE.itemsize
    assert L.plaac_sizeof_params() == C.sizeof(native.Params)


def test_params_init_matches_oracle_bit_for_bit(native, oracle):
    rng = np.random.default_rng(0)
    cases = [dict(), dict(alpha=0.5, bgcounts=rng.integers(1, 10**6, 22).astype(float)),
             dict(alpha=0.0, bgcounts=rng.random(22)), dict(fg=oracle.const_tables()["fg04"], corelength=30),
             dict(alpha=7.0), dict(ww1=21, ww2=61, adjustprolines=False)]
    for kw in cases:
        a, b = native.make_params(**kw), oracle.build_params(**kw)
        for f in ("fg", "bgscer", "bgthis", "bg", "llr", "lodpapa", "hydro2", "charge", "loglut", "cc"):
            assert np.array(getattr(a, f)).tobytes() == np.array(getattr(b, f)).tobytes(), (kw, f)
        for h in ("hmm1", "hmm0"):
            for f in ("lt", "li", "le", "lf"):
                assert np.array(getattr(getattr(a, h), f)).tobytes() == np.array(
                    getattr(getattr(b, h), f)).tobytes(), (kw, h, f)
        assert (a.corelength, a.ww1, a.ww2, a.ww3, a.adjustprolines, a.alpha) == (
            b.corelength, b.ww1, b.ww2, b.ww3, b.adjustprolines, b.alpha)
        assert a.big_neg == -1000000.0


def test_params_init_rejects_bad_arguments(native):
    with pytest.raises(native.PlaacError):
        native.make_params(corelength=0)
    with pytest.raises(native.PlaacError):
        native.make_params(ww1=0)


def test_builtin_tables_and_encode(native, oracle):
    t, u = native.builtin_tables(), oracle.const_tables()
    for k in ("bg_scer", "fg28", "fg04"):
        assert np.array_equal(t[k], u[k])
    s = bytes(range(256))
    assert np.array_equal(native.encode(s), oracle.encode(s))
    assert native.encode("XACDEFGHIKLMNPQRSTVWY*").tolist() == list(range(22))


def test_fi_integer_form_is_derived_from_the_tables(native):
    """host-side check behind the integer form of the filter tier (plaac_fi_integer_form): the reference's tables are
    rationals over 90 and 1000; other rational tables qualify with their own denominators; irrational ones do not"""
    import numpy as np
    P = native.make_params()
    ok, info = native.fi_integer_form(P)
    # hydro2 = aahydro/9 + 0.5 -> H = 10 aahydro + 45 in 0..90 (R = -4.5 -> 0); cc = {2.785, -1, -1.151}
    assert ok and info == {"A2": 2 * 2785, "B2": 2 * (-1151 * 90), "C2": 2 * (-1000 * 90), "SH": 90, "SC": 1000, "Hmin": 0}
    Q = native.make_params()
    for k in range(22):
        Q.hydro2[k] = (3 * k - 20) / 7.0
    Q.cc[0], Q.cc[1], Q.cc[2] = 1.5, -2.0, -0.25
    ok, info = native.fi_integer_form(Q)
    assert ok and info["SH"] == 7 and info["SC"] == 4 and info["Hmin"] == -20
    assert info["A2"] == 2 * 6 and info["B2"] == 2 * (6 * -20 + -1 * 7) and info["C2"] == 2 * (-8 * 7)
    rng = np.random.default_rng(5)
    R = native.make_params()
    for k in range(22):
        R.hydro2[k] = float(rng.random())
    assert native.fi_integer_form(R) == (False, dict.fromkeys(("A2", "B2", "C2", "SH", "SC", "Hmin"), 0))
    S = native.make_params()
    S.cc[0] = float(np.pi)
    assert not native.fi_integer_form(S)[0]
    W = native.make_params()  # rational but too wide for the bit fields / int32 window sums
    for k in range(22):
        W.hydro2[k] = 1000.0 * k
    assert not native.fi_integer_form(W)[0]


def test_no_gpu_means_loud_failure_not_fallback(native):
    """On a machine without a usable gfx950 device the product must refuse to compute."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; covered by the gpu-marked tests")
    with pytest.raises(native.PlaacError) as e:
        native.Context(native.make_params())
    assert e.value.status == native.PLAAC_ERR_DEVICE
    assert "device" in str(e.value).lower()


def test_product_never_imports_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/"""
    pkg = os.path.join(ROOT, "plaac_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle_ctypes" not in src and "plaac_oracle" not in src and "libplaac_oracle" not in src, f


def test_jni_shim_compiles_and_binds_every_native_of_the_java_class():
    """jni/plaac_jni.cpp cannot be built here (no JDK): check its syntax against a declarations-only jni.h stand-in and
    that it defines exactly the natives jni/PlaacNative.java declares (JNI name mangling: Java_<class>_<method>)."""
    import subprocess
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(ROOT, "tests", "jni_stub"), os.path.join(ROOT, "jni", "plaac_jni.cpp")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    java = open(os.path.join(ROOT, "jni", "PlaacNative.java")).read()
    natives = sorted(re.findall(r"static\s+native\s+\w+(?:\[\])?\s+(\w+)\s*\(", java))
    cpp = open(os.path.join(ROOT, "jni", "plaac_jni.cpp")).read()
    defined = sorted(re.findall(r"JNICALL\s+Java_PlaacNative_(\w+)\s*\(", cpp))
    assert natives == defined and len(natives) >= 6, (natives, defined)
    # every C-ABI function the shim calls is declared by the header
    called = set(re.findall(r"\b(plaac_[a-z0-9_]+)\s*\(", cpp))
    assert called <= set(header_functions()), called - set(header_functions())


def test_cli_fails_loudly_without_a_gpu(tmp_path):
    """no CPU fallback anywhere: the command-line host must refuse to score without a gfx950 device (exit 1, message
    on stderr, nothing but the reference's own comment lines on stdout) - and still serve the flags that need no GPU"""
    import subprocess
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    exe = os.path.join(ROOT, "bin", "plaac")
    if not os.path.exists(exe):
        pytest.skip("bin/plaac not built")
    fa = os.path.join(ROOT, "tests", "golden", "four_classic_prions.fasta")
    r = subprocess.run([exe, "-i", fa], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no usable MI355X" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l and not l.startswith("#")]
    r = subprocess.run([exe, "-B", os.path.join(ROOT, "tests", "golden", "prd_freq_scer_04.txt")], capture_output=True,
                       text=True, timeout=120)  # -B without -i: print the table, no device needed (:394-403)
    assert r.returncode == 0 and r.stdout.splitlines()[0] == "0.000000 # X" and len(r.stdout.splitlines()) == 22


def test_release_library_reads_only_the_documented_environment():
    """every PLAAC_ name inside the release library is one of the environment variables include/plaac_native.h
    documents ("environment and test hooks": at most twelve, none changes a result); every other switch of earlier rounds is a
    test hook (plaac_debug_set_knob) or exists in the diagnostic build only - whose keys the release library refuses."""
    import ctypes as C
    blob = open(os.path.join(ROOT, "plaac_amd", "libplaac_native.so"), "rb").read()
    found = set(m.decode() for m in re.findall(rb"PLAAC_[A-Z][A-Z0-9_]+", blob))
    header = open(os.path.join(ROOT, "include", "plaac_native.h")).read()
    sect = header[header.index("---- environment and test hooks"):]
    documented = set(re.findall(r"^ \*   (PLAAC_[A-Z0-9_]+)", sect, flags=re.M))
    assert 1 <= len(documented) <= 12, documented
    assert found <= documented, found - documented
    from plaac_amd import native
    L = native.load()
    if L.plaac_diag_build():
        pytest.skip("PLAAC_NATIVE_LIB points at the diagnostic build")
    assert L.plaac_debug_set_knob(b"KB_LANE_MIN_GROUPS", b"1") == native.PLAAC_OK
    assert L.plaac_debug_set_knob(b"KB_LANE_MIN_GROUPS", None) == native.PLAAC_OK
    assert L.plaac_debug_set_knob(b"TRACK_ONE_PASS", b"1") == native.PLAAC_ERR_UNSUPPORTED
    assert L.plaac_debug_set_knob(b"NO_SUCH_KEY", b"1") == native.PLAAC_ERR_ARG
    assert set(native.HOOK_KEYS) & set(native.DIAG_KEYS) == set()


def test_release_library_does_not_contain_the_result_breaking_switches():
    """PLAAC_DEBUG_SKIP / _SKIP_FROM (kernels not launched, rows stale), PLAAC_VIT_STOP and PLAAC_DEBUG_COUNTER are
    compiled only into the diagnostic build (`make DIAG=1` -> libplaac_native_diag.so); a host's environment cannot switch
    them on in the library that ships."""
    blob = open(os.path.join(ROOT, "plaac_amd", "libplaac_native.so"), "rb").read()
    for name in (b"PLAAC_DEBUG_SKIP", b"PLAAC_VIT_STOP", b"PLAAC_DEBUG_COUNTER"):
        assert name not in blob, name
    assert b"PLAAC_LATENCY_MODE" in blob  # (a documented environment switch, which changes no result, is still read)
