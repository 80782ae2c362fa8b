import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_call(item):
    """A test that forces a form only the diagnostic build of the library compiles (`make DIAG=1`; forms measured slower and
    no longer selected, EXPERIMENTS.md) is SKIPPED against the release library - plaac_amd.native raises DiagKnob when the
    environment asks for such a form. Run those tests with PLAAC_NATIVE_LIB=plaac_amd/libplaac_native_diag.so."""
    outcome = yield
    if outcome.excinfo is not None:
        from plaac_amd import native as nv
        if isinstance(outcome.excinfo[1], nv.DiagKnob):
            try:
                pytest.skip(str(outcome.excinfo[1]))
            except pytest.skip.Exception:
                outcome.force_exception(sys.exc_info()[1])


def read_fasta_simple(path):
    """plain reader for the committed fixtures (no blank lines / quirks in them)"""
    recs, name, buf = [], None, []
    with open(path) as f:
        for line in f:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if name is not None:
                    recs.append((name, "".join(buf)))
                name, buf = line[1:], []
            elif name is not None:
                buf.append(line)
    if name is not None:
        recs.append((name, "".join(buf)))
    return recs


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_ctypes as oc
    oc.lib()
    return oc


@pytest.fixture(scope="session")
def native():
    from plaac_amd import native as nv
    nv.load()
    return nv


@pytest.fixture(scope="session")
def kat28():
    recs = read_fasta_simple(os.path.join(GOLDEN, "kat28.fasta"))
    rows = []
    with open(os.path.join(GOLDEN, "kat28.tsv")) as f:
        next(f)
        for line in f:
            g, orf, s, e = line.rstrip("\n").split("\t")
            rows.append((g, orf, int(s), int(e)))
    assert [r[1] for r in rows] == [n for n, _ in recs]
    return recs, rows


@pytest.fixture(scope="session")
def classic4():
    return read_fasta_simple(os.path.join(GOLDEN, "four_classic_prions.fasta"))


def runs_of_ones(v):
    """1-based inclusive [start,end] runs of a 0/1 vector"""
    out, j, n = [], 0, len(v)
    while j < n:
        if v[j]:
            k = j
            while k < n and v[k]:
                k += 1
            out.append((j + 1, k))
            j = k
        else:
            j += 1
    return out


def quirky_fasta(seed=5, nrec=1500):
    """FASTA text with everything fastareader (plaac.java:4302-4375) has a rule for: junk before the first header, \n / \r\n /
    lone \r line ends (mixed inside a record), empty lines that end a record's sequence (the rest is skipped, the NEXT name is
    trimmed), blanks / tabs / '>' / '*' / lower case / other bytes inside sequence lines, headers with trailing blanks, records
    without a sequence, a last line without a terminator"""
    import numpy as np
    rng = np.random.default_rng(seed)
    aas = np.frombuffer(b"ACDEFGHIKLMNPQRSTVWYacdXB* ->\t", dtype=np.uint8)
    out = [b"leading junk\n>first header after junk  \n"]
    for r in range(nrec):
        out.append(b">rec%d text%s" % (r, b" \t " if r % 3 == 0 else b"") + (b"\r\n" if r % 5 == 0 else b"\n"))
        for _ in range(int(rng.integers(0, 5))):
            out.append(b"A" + bytes(rng.choice(aas, int(rng.integers(1, 90)))) + (b"\r" if r % 11 == 0 else (b"\r\n" if r % 7 == 0 else b"\n")))
        if r % 4 == 0:
            out.append((b"\r\n" if r % 8 == 0 else b"\n") + b"skipped line\n")
        if r % 97 == 0:
            out.append(b">\n>only a header\n")
    out.append(b">last\nMKVLAAGIQQ*")
    return b"".join(out)
