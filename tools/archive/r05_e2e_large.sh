#!/bin/bash
# round 5: an input beyond PLAAC_KEEP_BYTES (17 M sequences, ~5 GB of FASTA): two passes, the second one reads the file again
# and goes through the device's parser and formatter; against the host's (PLAAC_DEVICE_PARSE=0)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
python3 - > $O/e2e_large.txt 2>&1 <<'PY'
import os, sys, subprocess, time, hashlib
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from plaac_amd import native, synth
dev = torch.device("cuda", 0)
P = native.make_params()
N = 17_000_000
fa, tsv = "/tmp/e2e_large.fa", "/tmp/e2e_large.tsv"
fbytes = nres = 0
with open(fa, "wb") as out:  # (piece by piece: torch's repeat_interleave is not to be trusted beyond 2^32 elements)
    for ci, start in enumerate(range(0, N, 1_000_000)):
        c_, o_ = synth.make_batch_torch(4, 1_000_000, np.array(P.fg), np.array(P.bg), dev, seed=synth.SEED0 + 11 + 100000 * ci)
        b_, r_ = bench.write_fasta(torch, c_, o_, 1_000_000, fa + ".part")
        out.write(open(fa + ".part", "rb").read())
        fbytes += b_; nres += r_
        del c_, o_
os.unlink(fa + ".part")
torch.cuda.empty_cache()
print("# %d sequences, %d residues, %d bytes of FASTA; bin/plaac -i <fa> > <tsv>" % (N, nres, fbytes))
for env in ({"PLAAC_DEVICE_PARSE": "0", "PLAAC_SINGLE_PASS": "0"}, {"PLAAC_SINGLE_PASS": "0"}, {}, {"PLAAC_DEVICE_FORMAT": "0"}):
    if os.path.exists(tsv):
        os.unlink(tsv)
    t0 = time.perf_counter()
    with open(tsv, "wb") as fh:
        r = subprocess.run(["bin/plaac", "-i", fa], stdout=fh, stderr=subprocess.PIPE, env=dict(os.environ, PLAAC_TIMING="1", **env))
    dt = time.perf_counter() - t0
    h = hashlib.sha256()
    with open(tsv, "rb") as fh:
        for chunk in iter(lambda: fh.read(1 << 26), b""):
            h.update(chunk)
    print("%-32s %.3f s  rc %d  sha256 %s  %.3g residues/s  %d bytes" % (" ".join("%s=%s" % kv for kv in env.items()) or "(default)", dt, r.returncode, h.hexdigest()[:16], nres / dt, os.path.getsize(tsv)))
    err = r.stderr.decode(errors="replace").splitlines()
    other = [l for l in err if not l.startswith("plaac-timing")]
    print("    %d other lines on stderr; first: %s" % (len(other), other[:3]))
    with open(tsv, "rb") as fh:
        head = fh.read(4000).decode(errors="replace").splitlines()
    print("    table lines: %d; line 30: %s" % (sum(1 for _ in open(tsv, "rb")), head[30][:120] if len(head) > 30 else head[-1][:120]))
    for l in err:
        if l.startswith("plaac-timing: ") and "busy" not in l:
            print("    " + l)
os.unlink(fa); os.unlink(tsv)
PY
cat $O/e2e_large.txt
