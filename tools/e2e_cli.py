#!/usr/bin/env python3
"""End-to-end timing of bin/plaac on a synthetic UniRef50-shaped FASTA (run on the GPU box).
   python tools/e2e_cli.py [nprot]   ->  gpurun_out/e2e_cli.txt"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from plaac_amd import native, synth  # noqa: E402

nprot = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
P = native.make_params()
codes, offs = synth.make_batch(4, nprot=nprot, fg=np.array(P.fg), bg=np.array(P.bg))
letters = np.frombuffer(b"XACDEFGHIKLMNPQRSTVWY*", dtype=np.uint8)[codes]
path = "/tmp/e2e_%d.fa" % nprot
t = time.time()
with open(path, "wb") as f:
    o = offs.astype(np.int64)
    chunks = []
    for i in range(nprot):
        chunks.append(b">sp|P%07d|SYN_%d synthetic protein %d\n" % (i, i, i))
        chunks.append(letters[o[i]:o[i + 1]].tobytes())
        chunks.append(b"\n")
        if len(chunks) > 30000:
            f.write(b"".join(chunks))
            chunks = []
    f.write(b"".join(chunks))
print("wrote %s: %d records, %d residues, %.1f MB in %.1f s" % (path, nprot, len(codes), os.path.getsize(path) / 1e6,
                                                              time.time() - t))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
env = dict(os.environ, PLAAC_TIMING="1")
for rep in range(2):
    t = time.time()
    r = subprocess.run([os.path.join(ROOT, "bin", "plaac"), "-i", path], stdout=open("/tmp/e2e_out.tsv", "wb"),
                       stderr=subprocess.PIPE, env=env, text=True)
    dt = time.time() - t
    msg = "run %d: rc=%d wall %.3f s -> %.3g residues/s end to end, %.3g proteins/s; output %.1f MB\n%s" % (
        rep, r.returncode, dt, len(codes) / dt, nprot / dt, os.path.getsize("/tmp/e2e_out.tsv") / 1e6, r.stderr)
    print(msg)
    open(os.path.join(ROOT, "gpurun_out", "e2e_cli.txt"), "a").write(msg + "\n")
