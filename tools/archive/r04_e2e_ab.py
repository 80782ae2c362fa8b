#!/usr/bin/env python3
"""tools/r04_e2e_ab.py [nprot] - same-box A/B of bin/plaac's two host schemes on one synthetic FASTA: pipelined scoring (one
context per GPU, two batches in flight, overlapping device work: PLAAC_PIPELINE=1, the default) against round 3's two
synchronous contexts per GPU (PLAAC_PIPELINE=0); byte-identical tables; PLAAC_TIMING stage clocks of every run."""
import hashlib
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plaac_amd import native, synth  # noqa: E402

nprot = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
P = native.make_params()
path = "/tmp/e2e_ab_%d.fa" % nprot
letters = np.frombuffer(b"XACDEFGHIKLMNPQRSTVWY*", dtype=np.uint8)
nres = 0
with open(path, "wb") as f:
    for ci, start in enumerate(range(0, nprot, 500_000)):
        n = min(500_000, nprot - start)
        codes, offs = synth.make_batch(4, nprot=n, seed=1000 + ci, fg=np.array(P.fg), bg=np.array(P.bg))
        text, o = letters[codes], offs.astype(np.int64)
        nres += len(codes)
        chunks = []
        for i in range(n):
            chunks.append(b">s%07d\n" % (start + i))
            chunks.append(text[o[i]:o[i + 1]].tobytes())
            chunks.append(b"\n")
        f.write(b"".join(chunks))
print("wrote %s: %d records, %d residues, %.1f MB" % (path, nprot, nres, os.path.getsize(path) / 1e6), flush=True)
digests = {}
for rep in range(2):
    for mode in ("1", "0"):
        env = dict(os.environ, PLAAC_TIMING="1", PLAAC_PIPELINE=mode)
        t = time.time()
        with open("/tmp/e2e_ab.tsv", "wb") as out:
            r = subprocess.run([os.path.join(ROOT, "bin", "plaac"), "-i", path], stdout=out, stderr=subprocess.PIPE, env=env, text=True)
        dt = time.time() - t
        h = hashlib.sha256(open("/tmp/e2e_ab.tsv", "rb").read()).hexdigest()[:16]
        digests.setdefault(mode, h)
        print("PLAAC_PIPELINE=%s run %d: rc=%d wall %.3f s = %.3g residues/s; table sha %s\n%s" % (
            mode, rep, r.returncode, dt, nres / dt, h, "".join("    " + l + "\n" for l in r.stderr.splitlines())), flush=True)
print("tables identical:", len(set(digests.values())) == 1)
os.unlink(path)
