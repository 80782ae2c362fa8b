#!/bin/bash
# round 5: bin/plaac single pass vs two passes: CLI tests, then the 10 M-sequence end-to-end leg both ways (stage clocks on stderr)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
python3 - > $O/e2e_single_pass.txt 2>&1 <<'PY'
import os, sys, subprocess, time, hashlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from plaac_amd import native, synth
dev = torch.device("cuda", 0)
P = native.make_params()
pieces, offs, base = [], [torch.zeros(1, dtype=torch.int64, device=dev)], 0
for ci, start in enumerate(range(0, 10_000_000, 1_250_000)):
    c_, o_ = synth.make_batch_torch(4, 1_250_000, np.array(P.fg), np.array(P.bg), dev, seed=synth.SEED0 + 4 + 100000 * ci)
    pieces.append(c_); offs.append(o_[1:] + base); base += int(o_[-1].item())
codes = torch.cat(pieces); offsets = torch.cat(offs)
fa, tsv = "/tmp/e2e.fa", "/tmp/e2e.tsv"
fbytes, nres = bench.write_fasta(torch, codes, offsets, 10_000_000, fa)
del codes, offsets, pieces
torch.cuda.empty_cache()
print("# 10 M sequences, %d residues, %d bytes of FASTA; bin/plaac -i <fa> > <tsv>, PLAAC_TIMING=1" % (nres, fbytes))
for rep in range(3):
    for mode, huge, fast in (("0", "0", ""), ("0", "1", ""), ("1", "1", ""), ("1", "1", "1")):
        if True:
            t0 = time.perf_counter()
            with open(tsv, "wb") as fh:
                r = subprocess.run(["bin/plaac", "-i", fa], stdout=fh, stderr=subprocess.PIPE, env=dict(os.environ, PLAAC_TIMING="1", PLAAC_SINGLE_PASS=mode, PLAAC_HUGE_PAGES=huge, **({"PLAAC_FAST_EXIT": "1"} if fast else {})))
            dt = time.perf_counter() - t0
            h = hashlib.sha256(open(tsv, "rb").read()).hexdigest()[:16]
            print("PLAAC_SINGLE_PASS=%s PLAAC_HUGE_PAGES=%s%s: %.3f s  rc %d  sha256 %s  %.3g residues/s" % (mode, huge, " PLAAC_FAST_EXIT=1" if fast else "", dt, r.returncode, h, nres / dt))
            if rep == 2:
                for l in r.stderr.decode().splitlines():
                    if l.startswith("plaac-timing"):
                        print("    " + l)
PY
cat $O/e2e_single_pass.txt
