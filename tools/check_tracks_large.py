#!/usr/bin/env python3
"""One-off large parity check of track mode (rows + all per-residue tracks) against the oracle on the GPU box:
   python tools/check_tracks_large.py [nprot]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from plaac_amd import native, synth  # noqa: E402
from oracle import oracle_ctypes as oc  # noqa: E402
import test_gpu_parity as tp  # noqa: E402

nprot = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
P = native.make_params()
codes, offs = synth.make_batch(4, nprot=nprot, fg=np.array(P.fg), bg=np.array(P.bg), seed=99, stop_fraction=0.05)
want_rows, want_tr = oc.score_batch(oc.build_params(), codes, offs, tracks=True, nthreads=16)
with native.Context(P) as ctx:
    rows, tr = ctx.score(codes, offs, tracks=True)
tp.assert_rows_equal(rows, want_rows, "large")
tp.assert_tracks_equal(tr, want_tr, codes, offs, "large")
print("tracks parity ok: %d proteins, %d residues" % (nprot, len(codes)))
