#!/bin/bash
# round 5: what a hardware queue costs a HIP process (its context-save area: 181 MB of pinned host memory each on gfx950) -
# bin/plaac's start-up, exit and 10 M-sequence wall against GPU_MAX_HW_QUEUES
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O
python3 - > $O/queues_probe.txt 2>&1 <<'PY'
import os, sys, subprocess, time, hashlib
sys.path.insert(0, os.getcwd())
def run(env, args, tag, out=subprocess.DEVNULL, reps=3):
    for rep in range(reps):
        t0 = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
        r = subprocess.run(args, stdout=out, stderr=subprocess.PIPE, env=dict(os.environ, PLAAC_TIMING="1", PLAAC_TIMING_MAPS="1", PLAAC_TIMING_T0=str(t0), **env))
        t1 = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
        err = r.stderr.decode().splitlines()
        marks = [l.split(":", 1)[1].split("(")[0].strip() for l in err if "after the launch" in l]
        nq = sum(1 for l in err if "177568 kB resident" in l) // max(1, len(marks))
        ctxw = [l.split()[-2] for l in err if "wait for GPU contexts" in l]
        print("%-44s wall %7.1f ms | ctx wait %s ms | %d x 181 MB areas | %s" % (tag, (t1 - t0) * 1e-6, ctxw[0] if ctxw else "-", nq, " | ".join(marks)))
fa4 = "tests/golden/four_classic_prions.fasta"
for q in ("", "1", "2", "3", "4"):
    run({"GPU_MAX_HW_QUEUES": q} if q else {}, ["bin/plaac", "-i", fa4], "4 sequences, GPU_MAX_HW_QUEUES=%s" % (q or "(unset)"))
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
for rep in range(2):
    for q in ("", "1", "2", "3"):
        with open(tsv, "wb") as fh:
            run({"GPU_MAX_HW_QUEUES": q} if q else {}, ["bin/plaac", "-i", fa], "10 M sequences, GPU_MAX_HW_QUEUES=%s" % (q or "(unset)"), out=fh, reps=1)
        print("    sha256 %s" % hashlib.sha256(open(tsv, "rb").read()).hexdigest()[:16])
PY
cat $O/queues_probe.txt
