#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_rows.json: the oracle's summary rows (every field, floats as C99 hex strings) for
the four classic prions and the 28 KAT proteins, under the default parameters and under the KAT parameters
(fg = prd_freq_scer_04). NOT reference output (the reference cannot run here: no JVM) - a regression anchor that pins
the oracle itself, as SURVEY.md 8(c) C4 asks; the reference-pinned facts stay the 28 Viterbi boundaries of kat28.tsv.

    python tests/golden/make_oracle_rows.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import read_fasta_simple  # noqa: E402
from oracle import oracle_ctypes as oc  # noqa: E402


def rows_of(fasta, **kw):
    recs = read_fasta_simple(os.path.join(HERE, fasta))
    enc = [oc.encode(s) for _, s in recs]
    offs = np.zeros(len(enc) + 1, dtype=np.uint64)
    offs[1:] = np.cumsum([len(e) for e in enc])
    rows = oc.score_batch(oc.build_params(**kw), np.concatenate(enc), offs, nthreads=1)
    out = []
    for (name, _), r in zip(recs, rows):
        d = {"name": name}
        for f in rows.dtype.names:
            v = r[f]
            d[f] = float(v).hex() if rows.dtype[f].kind == "f" else int(v)
        out.append(d)
    return out


def main():
    fg04 = np.loadtxt(os.path.join(HERE, "prd_freq_scer_04.txt"), usecols=0)
    doc = {
        "note": "oracle output, not reference output; floats are float.hex() strings (bit patterns, NaN as 'nan')",
        "classic4_default": rows_of("four_classic_prions.fasta"),
        "classic4_c40_alpha0": rows_of("four_classic_prions.fasta", corelength=40, alpha=0.0,
                                       bgcounts=np.arange(22, dtype=np.float64) + 5.0),
        "kat28_fg04": rows_of("kat28.fasta", fg=fg04),
    }
    with open(os.path.join(HERE, "oracle_rows.json"), "w") as f:
        json.dump(doc, f, indent=0, sort_keys=True)
    print("wrote oracle_rows.json:", {k: len(v) for k, v in doc.items() if isinstance(v, list)})


if __name__ == "__main__":
    main()
