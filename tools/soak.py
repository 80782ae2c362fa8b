#!/usr/bin/env python3
"""Stability soak on the GPU box: many contexts created and destroyed, batches of changing size and mode scored on
each, device memory compared before and after (leaks), results compared with the first pass (determinism)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from plaac_amd import native, synth  # noqa: E402

P = native.make_params()
rng = np.random.default_rng(3)
batches = [synth.make_batch(4, nprot=int(n), fg=np.array(P.fg), bg=np.array(P.bg), seed=int(n)) for n in
           (50, 7000, 300, 40000, 1, 12000)]
ref = {}
free0 = None
for rep in range(12):
    with native.Context(P) as ctx:
        for k, (codes, offs) in enumerate(batches):
            tracks = (rep + k) % 3 == 0
            out = ctx.score(codes, offs, tracks=tracks)
            rows = out[0] if tracks else out
            key = k
            if key in ref:
                assert rows.tobytes() == ref[key], "rows changed between passes (batch %d, rep %d)" % (k, rep)
            else:
                ref[key] = rows.tobytes()
        with ctx.upload(*batches[3]) as b:
            b.histogram()
            b.sweep([native.make_params(alpha=a, corelength=c) for a in (1.0, 0.5) for c in (30, 60)])
    with native.Node(P, [0, 0]) as node:  # two contexts on the device, batches sharded by sequence
        for k in (1, 3):
            assert node.score(*batches[k]).tobytes() == ref[k], "node rows differ (batch %d, rep %d)" % (k, rep)
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if rep == 1:
        free0 = free
    if rep >= 1:
        print("rep %2d free %.1f MiB (delta vs rep 1: %+.1f MiB)" % (rep, free / 2**20, (free - free0) / 2**20))
assert abs(free - free0) < 64 * 2**20, "device memory drifted"
print("soak ok")
