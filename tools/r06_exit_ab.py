#!/usr/bin/env python3
"""round 6: what does leaving through _exit (PLAAC_FAST_EXIT=1) save bin/plaac on a 10 M-sequence run? UniRef-shaped text, the
whole process timed, alternating, three rounds."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from plaac_amd import native, synth
dev = torch.device("cuda", 0)
P = native.make_params()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
pieces, offs, base = [], [torch.zeros(1, dtype=torch.int64, device=dev)], 0
for ci, start in enumerate(range(0, n, bench.SYNTH_CHUNK)):
    c_, o_ = synth.make_batch_torch(4, min(bench.SYNTH_CHUNK, n - start), np.array(P.fg), np.array(P.bg), dev, seed=synth.SEED0 + 4 + 100000 * ci)
    pieces.append(c_); offs.append(o_[1:] + base); base += int(o_[-1].item())
codes, offsets = torch.cat(pieces), torch.cat(offs)
fa = "/tmp/exit_ab.fa"
bench.write_fasta(torch, codes, offsets, n, fa, "uniref")
del codes, offsets, pieces
torch.cuda.empty_cache()
exe = os.path.join(ROOT, "bin", "plaac")
for rep in range(3):
    for label, env in (("teardown in order", {}), ("PLAAC_FAST_EXIT=1", {"PLAAC_FAST_EXIT": "1"})):
        if os.path.exists("/tmp/exit_ab.tsv"):
            os.unlink("/tmp/exit_ab.tsv")
        t0 = time.perf_counter()
        with open("/tmp/exit_ab.tsv", "wb") as fh:
            r = subprocess.run([exe, "-i", fa], stdout=fh, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        print("%-20s %.3f s rc=%d" % (label, time.perf_counter() - t0, r.returncode), flush=True)
os.unlink(fa); os.unlink("/tmp/exit_ab.tsv")
