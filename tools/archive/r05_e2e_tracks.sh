#!/bin/bash
# round 5: bin/plaac -p all (the per-residue table) end to end on the real yeast proteome and on 100,000 synthetic sequences
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
python3 - > $O/e2e_tracks.txt 2>&1 <<'PY'
import os, sys, subprocess, time, hashlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from plaac_amd import native, synth
dev = torch.device("cuda", 0)
P = native.make_params()
c_, o_ = synth.make_batch_torch(4, 100000, np.array(P.fg), np.array(P.bg), dev, seed=synth.SEED0 + 21)
fa2 = "/tmp/e2e_tracks.fa"
fbytes, nres = bench.write_fasta(torch, c_, o_, 100000, fa2)
for name, fa, n in (("Scer.fasta (5,880 proteins)", "tests/golden/Scer.fasta", None), ("100,000 synthetic sequences, %d residues" % nres, fa2, nres)):
    for rep in range(3):
        tsv = "/tmp/e2e_tracks.tsv"
        if os.path.exists(tsv):
            os.unlink(tsv)
        t0 = time.perf_counter()
        with open(tsv, "wb") as fh:
            r = subprocess.run(["bin/plaac", "-i", fa, "-p", "all"], stdout=fh, stderr=subprocess.PIPE, env=dict(os.environ, PLAAC_TIMING="1"))
        dt = time.perf_counter() - t0
        print("%-50s %.3f s  rc %d  %d bytes  sha256 %s" % (name, dt, r.returncode, os.path.getsize(tsv), hashlib.sha256(open(tsv, "rb").read()).hexdigest()[:16]))
        if rep == 2:
            for l in r.stderr.decode().splitlines():
                if l.startswith("plaac-timing: ") and "busy" not in l:
                    print("    " + l)
PY
cat $O/e2e_tracks.txt
