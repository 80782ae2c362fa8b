#!/bin/bash
# round 5, K1: bin/plaac single pass with the FASTA parsed on the device vs on the host, the table written in place vs held back:
# the 10 M-sequence leg every way (the output file is removed before each run)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5; mkdir -p $O


python3 - > $O/e2e_device_parse.txt 2>&1 <<'PY'
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
E2E_ENVS = [dict(kv.split("=") for kv in grp.split()) for grp in os.environ.get("E2E_ENVS", "PLAAC_SINGLE_PASS=0;PLAAC_PLACED_WRITE=0 PLAAC_DEVICE_PARSE=0;PLAAC_DEVICE_PARSE=0;PLAAC_PLACED_WRITE=0;;PLAAC_FAST_EXIT=1;PLAAC_DEVICES=0,0,0,0,0,0,0,0").split(";")]
for rep in range(int(os.environ.get("E2E_REPS", "3"))):
    for env in E2E_ENVS:
        if os.path.exists(tsv):
            os.unlink(tsv)  # (the shell's truncation of a stale 2 GB file is not the program's time: 0.3 s)
        t0 = time.perf_counter()
        with open(tsv, "wb") as fh:
            r = subprocess.run(["bin/plaac", "-i", fa], stdout=fh, stderr=subprocess.PIPE, env=dict(os.environ, PLAAC_TIMING="1", PLAAC_TIMING_T0=str(time.clock_gettime_ns(time.CLOCK_MONOTONIC)), **env))
        dt = time.perf_counter() - t0
        h = hashlib.sha256(open(tsv, "rb").read()).hexdigest()[:16]
        print("%-32s %.3f s  rc %d  sha256 %s  %.3g residues/s" % (" ".join("%s=%s" % kv for kv in env.items()) or "(default)", dt, r.returncode, h, nres / dt))
        if rep == int(os.environ.get("E2E_REPS", "3")) - 1:
            for l in r.stderr.decode().splitlines():
                if l.startswith("plaac-timing"):
                    print("    " + l)
PY
cat $O/e2e_device_parse.txt
# the web application's own invocation (web/lib/server.rb:152-155): background from a -B file, alpha 0.5 - one scoring pass
python3 - >> $O/e2e_device_parse.txt 2>&1 <<'PY'
import os, subprocess, time, hashlib
fa, tsv = "/tmp/e2e.fa", "/tmp/e2e.tsv"
bg = "tests/golden/bg_freqs/bg_freqs_HUMAN.txt"
print("# bin/plaac -i <fa> -a 0.5 > <tsv>   (two passes: the background is counted from the input first)")
for rep in range(2):
    for env in ({"PLAAC_DEVICE_PARSE": "0"}, {}):
        if os.path.exists(tsv):
            os.unlink(tsv)
        t0 = time.perf_counter()
        with open(tsv, "wb") as fh:
            r = subprocess.run(["bin/plaac", "-i", fa, "-a", "0.5"], stdout=fh, stderr=subprocess.PIPE, env=dict(os.environ, PLAAC_TIMING="1", **env))
        dt = time.perf_counter() - t0
        print("%-32s %.3f s  rc %d  sha256 %s" % (" ".join("%s=%s" % kv for kv in env.items()) or "(default)", dt, r.returncode, hashlib.sha256(open(tsv, "rb").read()).hexdigest()[:16]))
        if rep == 1:
            for l in r.stderr.decode().splitlines():
                if l.startswith("plaac-timing: ") and "busy" not in l and "cpu" not in l:
                    print("    " + l)
print("# bin/plaac -i <fa> -c 60 -a 0.5 -B bg_freqs_HUMAN.txt > <tsv>")
for rep in range(2):
    for env in ({"PLAAC_DEVICE_PARSE": "0"}, {"PLAAC_DEVICE_FORMAT": "0"}, {}):
        if os.path.exists(tsv):
            os.unlink(tsv)
        t0 = time.perf_counter()
        with open(tsv, "wb") as fh:
            r = subprocess.run(["bin/plaac", "-i", fa, "-c", "60", "-a", "0.5", "-B", bg], stdout=fh, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        dt = time.perf_counter() - t0
        print("%-32s %.3f s  rc %d  sha256 %s" % (" ".join("%s=%s" % kv for kv in env.items()) or "(default)", dt, r.returncode, hashlib.sha256(open(tsv, "rb").read()).hexdigest()[:16]))
PY
tail -8 $O/e2e_device_parse.txt
