import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


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
