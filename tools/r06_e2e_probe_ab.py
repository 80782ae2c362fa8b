#!/usr/bin/env python3
"""What the two hardware-queue probes of a context (high class at creation, normal class at the first call) cost the 10 M-sequence
end-to-end run: bin/plaac against the DIAGNOSTIC library (LD_PRELOAD; it reads PLAAC_STREAM_PROBE from the environment) with the
probes on and off, three runs each, alternating. Run on the GPU box; needs `make DIAG=1`."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from plaac_amd import native, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda:0")
P = native.make_params()
pieces, offs, base = [], [torch.zeros(1, dtype=torch.int64, device=dev)], 0
for ci, start in enumerate(range(0, n, 1_250_000)):
    c_, o_ = synth.make_batch_torch(4, min(1_250_000, n - start), np.array(P.fg), np.array(P.bg), dev, seed=synth.SEED0 + 4 + 100000 * ci)
    pieces.append(c_); offs.append(o_[1:] + base); base += int(o_[-1].item())
codes = torch.cat(pieces); offsets = torch.cat(offs)
fa, tsv = "/tmp/probe_ab.fa", "/tmp/probe_ab.tsv"
fbytes, nres = bench.write_fasta(torch, codes, offsets, n, fa, "uniref")
del codes, offsets, pieces
torch.cuda.empty_cache()
diag = os.path.join(ROOT, "plaac_amd", "libplaac_native_diag.so")
exe = os.path.join(ROOT, "bin", "plaac")
shas = set()
for rep in range(3):
    for label, env in (("probes on ", {}), ("probes off", {"PLAAC_STREAM_PROBE": "0"}), ("release   ", None)):
        e = dict(os.environ) if env is None else dict(os.environ, LD_PRELOAD=diag, **env)
        if os.path.exists(tsv):
            os.unlink(tsv)
        t0 = time.perf_counter()
        with open(tsv, "wb") as fh:
            r = subprocess.run([exe, "-i", fa], stdout=fh, stderr=subprocess.PIPE, env=e)
        dt = time.perf_counter() - t0
        assert r.returncode == 0, r.stderr.decode()[-500:]
        if rep == 0:
            shas.add(bench.file_sha256(tsv))
        print("%s %.3f s" % (label, dt), flush=True)
print("tables identical:", len(shas) == 1)
os.unlink(fa); os.unlink(tsv)
